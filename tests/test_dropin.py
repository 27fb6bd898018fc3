"""GPU tests of the drop-in boundary at the C++ level: the reference's own CPU harness compiled UNCHANGED against
include/neighlist_cpu.hpp (oracle/_ref/make_list_dropin, built where /root/reference exists), and this repository's
driver with the flow of the reference's GPU harness (tools/make_list)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_reference_cpu_harness_runs_unchanged_on_the_hip_library():
    exe = os.path.join(ROOT, "oracle", "_ref", "make_list_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/make_list_dropin not built (needs /root/reference at build time)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "# of particles 119164" in r.stdout  # make_list.cpp:158-159
    assert "TEST is passed." in r.stderr         # make_list.cpp:222


@pytest.mark.parametrize("args,n", [
    (["--lattice", "--dtype", "f64", "--loop", "20"], 119164),
    (["--lattice", "--rho", "0.5", "--dtype", "f32", "--loop", "20"], 62500),
    (["--n", "30000", "--dtype", "f32", "--loop", "10"], 30000),
    (["--n", "200000", "--rho", "0.5", "--dtype", "f64", "--loop", "5"], 200000),
])
def test_make_list_driver(args, n):
    exe = os.path.join(ROOT, "tools", "make_list")
    if not os.path.exists(exe):
        pytest.fail("tools/make_list missing: __graft_entry__.build() builds it")
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"# of particles {n} " in r.stdout
    assert "TEST is passed." in r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["n"] == n and rec["half_pairs"] > 0 and rec["ms_per_build"] > 0
