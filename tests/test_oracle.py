"""CPU tests: the oracle (oracle/nl_oracle.c) is pinned against the reference's outputs.

  * every golden fixture under tests/golden/ (generated from the compiled reference by oracle/gen_golden.py)
  * the compiled reference itself where oracle/_ref exists (raw array equality, WITHOUT_LOOP_FUSION order)
  * the harness's brute-force definition (make_list.cpp:79-99)
"""
import json
import os

import numpy as np
import pytest

from md_neighbor_list_amd import inputs
from oracle import pyoracle as po
from tests.util import GOLDEN, canonical_csr, golden_names, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_restatement_matches_golden(name):
    g = load_golden(name)
    h = po.build(g["q"], float(g["rc"]), tuple(g["box"]))
    assert h.npairs == int(g["npairs"])
    assert np.array_equal(h.number_of_partners, g["number_of_partners"])
    assert np.array_equal(h.key_pointer, g["key_pointer"])
    # visit order of the WITHOUT_LOOP_FUSION variant, entry by entry
    assert np.array_equal(h.sorted_list, g["sorted_list_naive_order"])
    c = h.canonical()
    assert np.array_equal(c.sorted_list, g["sorted_list"])
    assert c.hash() == int(g["hash"])
    # numpy canonicalisation (used by the GPU tests) agrees with the C one
    assert np.array_equal(canonical_csr(h.key_pointer, h.sorted_list), g["sorted_list"])


@pytest.mark.parametrize("name", golden_names(dup=False))
def test_golden_is_a_set_of_ordered_pairs(name):
    g = load_golden(name)
    kp, lst = g["key_pointer"], g["sorted_list"]
    rows = np.repeat(np.arange(len(kp) - 1), np.diff(kp))
    assert np.all(lst > rows)  # stored on min(i,j): neighlist_cpu.hpp:225-236
    pairs = rows.astype(np.int64) << 32 | lst
    assert len(np.unique(pairs)) == len(pairs)


@pytest.mark.parametrize("name", [n for n in golden_names(dup=False) if not n.startswith(("sc_", "u4096", "u3000"))])
def test_bruteforce_agrees_where_cells_are_wider_than_cutoff(name):
    """make_list.cpp:166-220: the harness checks the class against O(N^2).  (The sc_* lattices sit on L = m*rc
    exactly, where hash rounding decides cell membership; there the class, not brute force, is the contract.)"""
    g = load_golden(name)
    b = po.bruteforce(g["q"], float(g["rc"]))
    assert b.npairs == int(g["npairs"])
    assert np.array_equal(b.sorted_list, g["sorted_list"])


@pytest.mark.parametrize("variant", ["naive", "fused"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_restatement_matches_compiled_reference(variant, dtype):
    if not po.ref_available(variant):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    for seed, n, box, rc in ((1, 700, (11.0, 12.0, 13.5), 3.3), (2, 2500, (20.0, 20.0, 20.0), 3.3), (3, 900, (9.0, 30.0, 9.5), 2.9)):
        q, box = inputs.uniform_box(n, dtype=dtype, seed=seed, box=box)
        r, _, _ = po.ref_build(q, rc, box, variant)
        h = po.build(q, rc, box)
        assert np.array_equal(r.key_pointer, h.key_pointer)
        assert np.array_equal(r.number_of_partners, h.number_of_partners)
        if variant == "naive":
            assert np.array_equal(r.sorted_list, h.sorted_list)
        assert np.array_equal(r.canonical().sorted_list, h.canonical().sorted_list)


def test_known_answer_config1():
    """N=4096 rho=1.0 -> P = 243 701, hash 994f529891f2f789 (f32 == f64)."""
    for dt in (np.float32, np.float64):
        q, box = inputs.uniform_box(4096, 1.0, dt)
        h = po.build(q, 3.3, box)
        assert h.npairs == 243701 and h.hash() == 0x994F529891F2F789


def test_reference_harness_lattice_counts():
    """make_list.cpp:17-24 problem: N = 119 164 / 62 500 particles."""
    for rho, n in ((1.0, 119164), (0.5, 62500)):
        q, _ = inputs.fcc_box(rho, 50.0, np.float64)
        assert len(q) == n
        assert float(q[:, :3].max()) < 50.0


@pytest.mark.slow
def test_known_answer_lattice_pairs():
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))
    q, box = inputs.fcc_box(0.5, 50.0, np.float64)
    h = po.build(q, 3.3, box)
    assert h.npairs == ka["fcc_L50_rho05_f64"]["npairs"] == 2268138
    assert f"{h.hash():016x}" == ka["fcc_L50_rho05_f64"]["hash"]


def test_out_of_box_is_reported():
    q, box = inputs.uniform_box(100, dtype=np.float32, seed=5, box=(12.0, 12.0, 12.0))
    q[7, 1] = 40.0  # more than one box length outside: the reference would index out of bounds
    with pytest.raises(po.OracleError):
        po.build(q, 3.3, box)
    c, mesh = po.cells(q, 3.3, box)
    assert c[7] == -1 and np.all(c[np.arange(100) != 7] >= 0) and tuple(mesh) == (3, 3, 3)


def test_minimum_image_oracle_against_float64_brute_force():
    """The minimum-image definition (oracle build_pbc) against an O(N^2) float64 numpy minimum
    image on boxes of 3..5 cells per axis, with particles outside [0, L) and on the faces."""
    rng = np.random.default_rng(5)
    for case in range(6):
        rc = float(rng.uniform(1.0, 3.0))
        mesh = rng.integers(3, 6, size=3)
        box = tuple(float(m * rc * rng.uniform(1.0, 1.3)) for m in mesh)
        n = 700
        q = np.zeros((n, 4), dtype=np.float64)
        q[:, :3] = rng.uniform(-0.2, 1.2, size=(n, 3)) * np.array(box)  # some outside the box: taken at their image
        q[:20, :3] = np.round(q[:20, :3] / np.array(box)) * np.array(box)  # on the faces 0 and L
        q[:, :3] = np.clip(q[:, :3], -0.95 * np.array(box), 1.95 * np.array(box))
        got = po.build_pbc(q, rc, box)
        d = q[None, :, :3] - q[:, None, :3]
        d -= np.round(d / np.array(box)) * np.array(box)
        r2 = (d * d).sum(axis=2)
        iu = np.triu(np.ones((n, n), dtype=bool), 1)
        near = np.abs(r2 - rc * rc) < 1e-9  # ties decided by rounding: excluded from the comparison
        want = (r2 <= rc * rc) & iu
        have = np.zeros((n, n), dtype=bool)
        rows = np.repeat(np.arange(n), np.diff(got.key_pointer))
        have[rows, got.sorted_list] = True
        assert not (have & ~iu).any()
        assert np.array_equal(have[~near], want[~near]), case
        assert got.npairs > 0


def test_minimum_image_full_oracle_is_the_row_frame_list():
    """oracle build_pbc_full: row i holds every accepted j != i in the frame of i.  Its upper part (j > i) is the half
    list exactly (same frame: the smaller id's); in float64 on random positions both directions agree, so the whole
    list is the symmetrised half list and matches the float64 brute force; in float32 with partners placed within a
    few ulp of the cut-off across a periodic face the two directions differ on some pairs."""
    rng = np.random.default_rng(6)
    rc, box = 2.5, (9.0, 10.5, 8.0)
    n = 900
    q = np.zeros((n, 4), dtype=np.float64)
    q[:, :3] = rng.uniform(-0.2, 1.2, size=(n, 3)) * np.array(box)
    half, full = po.build_pbc(q, rc, box), po.build_pbc_full(q, rc, box)
    rows = np.repeat(np.arange(n), np.diff(full.key_pointer))
    up = full.sorted_list > rows
    assert np.array_equal(full.sorted_list[up], half.sorted_list)
    assert np.array_equal(np.bincount(rows[up], minlength=n), half.number_of_partners)
    have = np.zeros((n, n), dtype=bool)
    have[rows, full.sorted_list] = True
    assert np.array_equal(have, have.T) and full.npairs == 2 * half.npairs and not have.diagonal().any()
    d = q[None, :, :3] - q[:, None, :3]
    d -= np.round(d / np.array(box)) * np.array(box)
    r2 = (d * d).sum(axis=2)
    near = np.abs(r2 - rc * rc) < 1e-9
    want = (r2 <= rc * rc) & ~np.eye(n, dtype=bool)
    assert np.array_equal(have[~near], want[~near])

    # float32, partners at rc (1 +- k ulp) across the low x face
    L, nc = 30.0, 1500
    centres = rng.uniform(0.0, L, size=(nc, 3))
    centres[:, 0] = rng.uniform(0.0, 1.0, size=nc)
    dvec = rng.normal(size=(nc, 3))
    dvec[:, 0] = -np.abs(dvec[:, 0])
    dvec /= np.linalg.norm(dvec, axis=1, keepdims=True)
    scale = 1.0 + rng.integers(-6, 7, size=(nc, 1)) * 2.0 ** -23
    q32 = np.zeros((2 * nc, 4), dtype=np.float32)
    q32[:, :3] = np.concatenate([centres, np.mod(centres + dvec * rc * scale, L)]).astype(np.float32)
    q32[:, :3] = np.minimum(q32[:, :3], np.nextafter(np.float32(L), np.float32(0)))
    half, full = po.build_pbc(q32, rc, (L, L, L)), po.build_pbc_full(q32, rc, (L, L, L))
    m = 2 * nc
    rows = np.repeat(np.arange(m), np.diff(full.key_pointer))
    up = full.sorted_list > rows
    assert np.array_equal(full.sorted_list[up], half.sorted_list)
    have = np.zeros((m, m), dtype=bool)
    have[rows, full.sorted_list] = True
    assert (have != have.T).any(), "pairs accepted in one direction only are what this mode's definition is about"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_count_mode_equals_the_stored_list(dtype):
    """nl_oracle_count (no pair storage; the source of the config 4 / config 5 known answers) against the list-building
    restatement on the same box: counts, total, pair-set hash, and the per-slab split of both."""
    from md_neighbor_list_amd import slab

    q, box = inputs.uniform_box(20000, dtype=dtype, seed=21, box=(23.0, 27.0, 33.5))
    rc = 3.3
    ref = po.build(q, rc, box)
    nop, pairs, hashes, npairs = po.count(q, rc, box)
    assert npairs == ref.npairs == int(pairs[0]) and int(hashes[0]) == ref.hash()
    assert np.array_equal(nop, ref.number_of_partners)
    mz = int(box[2] / rc)
    sol = np.zeros(mz, dtype=np.int32)
    for r, (lo, hi) in enumerate(slab.split_layers(mz, 3)):
        sol[lo:hi] = r
    nop3, pairs3, hashes3, np3 = po.count(q, rc, box, sol)
    assert np3 == ref.npairs and np.array_equal(nop3, nop)
    cells, mesh = po.cells(q, rc, box)
    owner = sol[cells // (mesh[0] * mesh[1])]
    for r in range(3):
        rows = np.nonzero(owner == r)[0]
        assert int(pairs3[r]) == int(ref.number_of_partners[rows].sum())
        kp = np.concatenate([[0], np.cumsum(np.where(owner == r, ref.number_of_partners, 0))]).astype(np.int64)
        lst = np.concatenate([ref.sorted_list[ref.key_pointer[i]:ref.key_pointer[i + 1]] for i in rows] or [np.zeros(0, np.int32)])
        assert int(hashes3[r]) == po.HalfList(None, kp, lst.astype(np.int32)).hash()
    assert (int(hashes3.astype(object).sum()) & ((1 << 64) - 1)) == ref.hash()


def test_count_mode_reproduces_the_compiled_reference_at_1M():
    """Pins count mode to the compiled reference at BASELINE size: the stored answer of config 2 (written by
    gen_golden.py --big from oracle/_ref) is reproduced exactly."""
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["u1M_rho1_f32"]
    q, box = inputs.uniform_box(ka["n"], 1.0, np.float32)
    nop, _, hashes, npairs = po.count(q, ka["rc"], box)
    assert npairs == ka["npairs"] and f"{int(hashes[0]):016x}" == ka["hash"]
    nop = nop.astype(np.int64)
    assert int(nop.max()) == ka["nop_max"] and int((nop * (np.arange(len(nop)) % 1000003)).sum()) == ka["nop_weighted_sum"]
