"""CPU sanitizer run (the reference has only -Wall -Wextra -Werror, Makefile:17): the host shims
(include/neighlist_cpu.hpp, include/neighlist_gpu.hpp) over a host-memory stand-in of the C ABI, the input generator and
the oracle's C restatement, all compiled with -fsanitize=address,undefined (`make asan`).  CPU only: GPU ASan is not
available on this pool, and nothing here touches a device."""
import os
import subprocess

from tests.util import ROOT


def test_shims_and_oracle_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", ROOT, "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(ROOT, "build", "sanitize_test")], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error" not in out and "LeakSanitizer" not in out, out[-4000:]
    assert "all checks passed" in out
