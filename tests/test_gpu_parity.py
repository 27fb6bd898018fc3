"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  * the golden fixtures generated from the compiled reference,
  * the CPU oracle on seeded inputs (sizes the oracle finishes in seconds),
  * size-independent properties and stored known answers at the BASELINE sizes.
Integer results are compared bit-exactly after the reference's canonical sort (make_list.cpp:120-128,211-220).
"""
import json
import os

import numpy as np
import pytest

from md_neighbor_list_amd import inputs
from tests.util import GOLDEN, canonical_csr, golden_names, gpu_build, load_golden

pytestmark = pytest.mark.gpu

SWEEP_VARIANTS = [1, 3]  # NL_SWEEP_VARIANT values the library accepts: 1 = COUNT + FILL distance sweeps, 3 = hit masks + expansion


def _po():
    from oracle import pyoracle as po

    return po


def _loaded_native():
    """The HIP library must be what ran (no silent fallback exists, but say so explicitly)."""
    with open("/proc/self/maps") as f:
        return any("libnl_hip.so" in line for line in f)


@pytest.mark.parametrize("name", golden_names(dup=False))
def test_golden(name):
    g = load_golden(name)
    _, nop, kp, sl = gpu_build(g["q"], float(g["rc"]), tuple(g["box"]))
    assert _loaded_native()
    assert int(kp[-1]) == int(g["npairs"])
    assert np.array_equal(nop, g["number_of_partners"])
    assert np.array_equal(kp.astype(np.int64), g["key_pointer"])
    assert np.array_equal(canonical_csr(kp, sl), g["sorted_list"])


@pytest.mark.parametrize("name", ["u4096_rho1_f32", "sc_ties_f64", "outside_wrap_f32"])
def test_golden_async_and_stride3(name):
    """sync=False (the reference's timing loop, make_list.cu:124-127) and the 3-scalar Vec of make_list.cpp:26-32."""
    g = load_golden(name)
    q3 = np.ascontiguousarray(g["q"][:, :3])
    _, nop, kp, sl = gpu_build(q3, float(g["rc"]), tuple(g["box"]), sync=False)
    assert np.array_equal(nop, g["number_of_partners"])
    assert np.array_equal(canonical_csr(kp, sl), g["sorted_list"])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", [
    (20000, (30.0, 30.0, 30.0), 3.3, 21),    # rho 0.74
    (50000, (36.84, 36.84, 36.84), 3.3, 22),  # rho 1.0, mesh 11
    (30000, (25.0, 40.0, 33.0), 2.7, 23),    # non-cubic
    (12000, (13.2, 13.2, 40.0), 3.3, 24),    # L = 4*rc exactly in x,y (hash rounding at cell faces)
    (3000, (60.0, 60.0, 60.0), 3.3, 25),     # very sparse: mostly empty cells
    (40000, (20.0, 20.0, 20.0), 3.3, 26),    # rho 5: long rows, stencil larger than one LDS batch (f64)
])
def test_against_oracle(case, dtype):
    n, box, rc, seed = case
    q, box = inputs.uniform_box(n, dtype=dtype, seed=seed, box=box)
    ref = _po().build(q, rc, box)
    _, nop, kp, sl = gpu_build(q, rc, box)
    assert int(kp[-1]) == ref.npairs
    assert np.array_equal(nop, ref.number_of_partners)
    assert np.array_equal(kp.astype(np.int64), ref.key_pointer)
    assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)


def test_two_times_cutoff_fp64():
    """BASELINE config 5 regime (fp64, rc = 6.6: ~560 half pairs per particle) at an oracle-sized N."""
    q, box = inputs.uniform_box(32768, 1.0, np.float64, seed=31)
    ref = _po().build(q, 6.6, box)
    _, nop, kp, sl = gpu_build(q, 6.6, box)
    assert int(kp[-1]) == ref.npairs
    assert np.array_equal(nop, ref.number_of_partners)
    assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)


@pytest.mark.parametrize("rows", [0, 1])
def test_hash_and_sort_stage(rows, monkeypatch):
    """a3-a5: every particle lands in the cell the reference's GenHash gives it (neighlist_cpu.hpp:51-59); rows = 1: in
    the fine-row layout of nl_rows.hpp (the cell's particles as four quarters along z)."""
    import torch

    monkeypatch.setenv("NL_ROWS", str(rows))

    q, box = inputs.uniform_box(100000, 1.0, np.float32, seed=41)
    q[:50, :3] = np.nextafter(np.float32(box[0]), np.float32(0))  # rounds up to the box edge -> wraps to cell 0
    cells, mesh = _po().cells(q, 3.3, box)
    nl, nop, kp, sl = gpu_build(q, 3.3, box)
    assert tuple(nl.mesh_size) == tuple(mesh)
    cell_start, sorted_row = (t.cpu().numpy() for t in nl.sorted_state())
    if nl.build_info()["fine_rows"]:
        # the fine-row table keeps the four quarters of a cell side by side, [(row * mx + cx) * 4 + quarter]; the particles
        # are sorted by (row, quarter, cx): back to that order
        mx = int(mesh[0])
        t = cell_start[:-1].reshape(-1, mx, 4).transpose(0, 2, 1).reshape(-1)
        cell_start = np.append(t, cell_start[-1])
    assert cell_start[0] == 0 and cell_start[-1] == len(q) and np.all(np.diff(cell_start) >= 0)
    assert np.array_equal(np.sort(sorted_row), np.arange(len(q)))
    bin_of_slot = np.repeat(np.arange(len(cell_start) - 1), np.diff(cell_start))
    if nl.build_info()["fine_rows"]:
        # bins in the order (row * 4 + quarter) * mx + cx -- every row of x-cells as four fine rows (quarters in z)
        mx = mesh[0]
        fine_row, cx = np.divmod(bin_of_slot, mx)
        cell_of_slot = (fine_row // 4) * mx + cx
        # the quarter a particle lies in: fraction of the product the reference truncates (GenHash)
        ms = (np.asarray(box, dtype=np.float64) / np.asarray(mesh)).astype(np.float32)  # neighlist_cpu.hpp:389-391, 409-411
        ims = (1.0 / ms.astype(np.float64)).astype(np.float32)
        t = q[sorted_row, :3].astype(np.float32) * ims
        frac = t[:, 2] - np.trunc(t[:, 2])
        assert np.array_equal(fine_row % 4, np.clip(np.floor(frac * np.float32(4.0)), 0, 3).astype(np.int64))
    else:
        cell_of_slot = bin_of_slot
    assert np.array_equal(cells[sorted_row], cell_of_slot)
    assert np.array_equal(np.bincount(cells, minlength=int(np.prod(mesh))), np.bincount(cell_of_slot, minlength=int(np.prod(mesh))))
    del torch


def test_full_transposed_list_matches_gpu_harness_check():
    """The GPU class's output (neighlist_gpu.hpp:468-487) checked the way make_list.cu:157-198 does."""
    q, box = inputs.uniform_box(6000, dtype=np.float64, seed=51, box=(18.0, 18.0, 18.0))
    nl, nop, kp, sl = gpu_build(q, 3.3, box)
    ref = _po().bruteforce(q, 3.3)
    tl = nl.neigh_list().cpu().numpy()
    cnt = nl.number_of_partners().cpu().numpy()
    n = len(q)
    assert nl.number_of_pairs() == 2 * ref.npairs == int(cnt.sum())
    full = [[] for _ in range(n)]
    rows = np.repeat(np.arange(n), np.diff(ref.key_pointer))
    for i, j in zip(rows, ref.sorted_list):
        full[i].append(j)
        full[j].append(i)
    for i in range(n):
        assert cnt[i] == len(full[i])
        assert sorted(tl[: cnt[i], i].tolist()) == sorted(full[i])
        assert np.all(tl[cnt[i]:, i] == -1)
    h, nhalf = _po().hash_transposed(cnt, tl, n)
    assert nhalf == ref.npairs and h == ref.hash()


@pytest.mark.parametrize("n,box,rc", [(70001, (42.0, 40.0, 38.0), 3.3),    # rows ~150: the flat LDS conversion
                                        (30000, (20.0, 21.0, 19.0), 3.3)])   # rows ~560: blocks left to the tiled kernel
def test_transposed_list_from_the_full_csr(n, box, rc):
    """nl_get_full_transposed after a NL_LIST_FULL build -- the reference GPU class's layout list[k*N + i]
    (neighlist_gpu.hpp:468-482) -- through both conversion kernels: every column holds exactly the row of the full CSR,
    -1 beyond it, and its j > i part hashes to the oracle's half list."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    q, box = inputs.uniform_box(n, dtype=np.float32, seed=52, box=box)
    ref = _po().build(q, rc, box)
    nl = NeighListGPU(rc, *box, dtype=torch.float32, full_list=True)
    nl.Initialize(n)
    nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
    kp, lst, cnt = (t.cpu().numpy() for t in nl.full_csr())
    tl = nl.neigh_list().cpu().numpy()
    tc = nl.number_of_partners().cpu().numpy()
    assert np.array_equal(tc, cnt) and tl.shape[1] == n and tl.shape[0] >= cnt.max()
    k = np.arange(tl.shape[0])[:, None]
    assert np.all(tl[k >= cnt[None, :]] == -1)
    rows = np.repeat(np.arange(n), cnt)
    ks = np.arange(len(lst)) - np.repeat(kp[:-1].astype(np.int64), cnt)
    assert np.array_equal(tl[ks, rows], lst)  # column i, entries 0..cnt[i]) = row i of the CSR, in its order
    h, nhalf = _po().hash_transposed(cnt, tl, n)
    assert nhalf == ref.npairs and h == ref.hash()


def test_errors_are_reported_not_crashes():
    import torch

    from md_neighbor_list_amd import NeighListGPU, NLError
    from md_neighbor_list_amd import _lib

    with pytest.raises(NLError) as e:
        NeighListGPU(3.3, 9.0, 20.0, 20.0)  # 2 cells along x
    assert e.value.code == _lib.NL_ERR_MESH
    q, box = inputs.uniform_box(5000, dtype=np.float32, seed=61, box=(17.0, 17.0, 17.0))
    nl = NeighListGPU(3.3, *box)
    qd = torch.from_numpy(q).cuda()
    with pytest.raises(NLError) as e:
        nl.MakeNeighList(qd)  # before Initialize
    assert e.value.code == _lib.NL_ERR_STATE
    nl.Initialize(len(q))
    bad = q.copy()
    bad[123, 2] = 60.0
    with pytest.raises(NLError) as e:
        nl.MakeNeighList(torch.from_numpy(bad).cuda())
    assert e.value.code == _lib.NL_ERR_OUT_OF_BOX
    bad[123, 2] = np.nan
    with pytest.raises(NLError) as e:
        nl.MakeNeighList(torch.from_numpy(bad).cuda())
    assert e.value.code == _lib.NL_ERR_OUT_OF_BOX
    # capacity: an asynchronous build cannot grow the list and must say so instead of overrunning
    nl.set_capacity(1000)
    nl.MakeNeighList(qd, sync=False)
    with pytest.raises(NLError) as e:
        nl.synchronize()
    assert e.value.code == _lib.NL_ERR_CAPACITY
    # a synchronous build grows it and succeeds, and the handle is reusable after every error above
    nl.MakeNeighList(qd, sync=True)
    ref = _po().build(q, 3.3, box)
    assert nl.half_number_of_pairs() == ref.npairs
    assert np.array_equal(canonical_csr(nl.key_pointer().cpu().numpy(), nl.sorted_list().cpu().numpy()),
                          ref.canonical().sorted_list)


def test_empty_and_tiny_inputs():
    import torch

    from md_neighbor_list_amd import NeighListGPU

    nl = NeighListGPU(3.3, 12.0, 12.0, 12.0)
    nl.Initialize(16)
    nl.MakeNeighList(torch.zeros((0, 4), dtype=torch.float32, device="cuda"))
    assert nl.half_number_of_pairs() == 0 and nl.key_pointer().cpu().tolist() == [0]
    q = torch.tensor([[1.0, 1.0, 1.0, 0.0], [2.0, 1.0, 1.0, 0.0], [11.9, 11.9, 11.9, 0.0]], device="cuda")
    nl.MakeNeighList(q)
    assert nl.half_number_of_pairs() == 1
    assert nl.key_pointer().cpu().tolist() == [0, 1, 1, 1] and nl.sorted_list().cpu().tolist() == [1]


@pytest.mark.parametrize("key", ["u1M_rho1_f32", "u1M_rho05_f32", "u1M_rho1_f64", "fcc_L50_rho1_f64"])
def test_baseline_sizes_known_answers(key):
    """BASELINE configs 2 and 3 (and the README lattice) at full size: pair count, pair-set hash and count checksum
    against answers stored from the compiled reference; plus structural properties that need no oracle."""
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))[key]
    dt = np.float32 if key.endswith("f32") else np.float64
    if key.startswith("fcc"):
        q, box = inputs.fcc_box(1.0, 50.0, dt)
    else:
        q, box = inputs.uniform_box(ka["n"], 1.0 if "rho1" in key else 0.5, dt)
    assert len(q) == ka["n"]
    nl, nop, kp, sl = gpu_build(q, ka["rc"], box)
    assert int(kp[-1]) == ka["npairs"] == len(sl)
    assert int(nop.max()) == ka["nop_max"]
    assert int((nop.astype(np.int64) * (np.arange(len(nop)) % 1000003)).sum()) == ka["nop_weighted_sum"]
    assert np.array_equal(np.diff(kp), nop)
    rows = np.repeat(np.arange(len(nop), dtype=np.int32), nop)
    assert np.all(sl > rows)  # half list: stored on the smaller id
    from oracle import pyoracle as po

    h = po.HalfList(nop, kp.astype(np.int64), sl).hash()
    assert f"{h:016x}" == ka["hash"]
    # idempotence: a second build of the same positions gives the same canonical list
    nl2, nop2, kp2, sl2 = gpu_build(q, ka["rc"], box, sync=False)
    assert np.array_equal(kp, kp2)
    assert po.HalfList(nop2, kp2.astype(np.int64), sl2).hash() == h
    # spot-check rows against an O(N) scan of all particles for a few i
    rng = np.random.default_rng(3)
    rc2 = ka["rc"] * ka["rc"]
    for i in rng.integers(0, len(q), size=8):
        d = q[:, :3] - q[i, :3]
        r2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        want = np.nonzero(~(r2.astype(np.float64) > rc2) & (np.arange(len(q)) > i))[0]
        assert np.array_equal(np.sort(sl[kp[i]:kp[i + 1]]), want.astype(np.int32))


@pytest.mark.parametrize("variant,binning", [(v, b) for v in SWEEP_VARIANTS for b in (0, 1)])
def test_every_sweep_variant_and_binning_path(variant, binning, monkeypatch):
    """The non-default kernels stay correct: NL_SWEEP_VARIANT 1 (COUNT + FILL sweeps), 3 (27-cell hit masks);
    NL_BINNING=1 (atomic-rank hash/reorder)."""
    monkeypatch.setenv("NL_SWEEP_VARIANT", str(variant))
    monkeypatch.setenv("NL_BINNING", str(binning))
    for n, box, rc, seed in [(50000, (36.84, 36.84, 36.84), 3.3, 41), (9000, (25.0, 14.0, 19.0), 3.1, 42)]:
        q, box = inputs.uniform_box(n, dtype=np.float32, seed=seed, box=box)
        ref = _po().build(q, rc, box)
        nl, nop, kp, sl = gpu_build(q, rc, box)
        assert nl.build_info()["variant"] == variant
        assert int(kp[-1]) == ref.npairs
        assert np.array_equal(nop, ref.number_of_partners)
        assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)


@pytest.mark.parametrize("variant", SWEEP_VARIANTS)
def test_pairs_at_the_cutoff_fp32(variant, monkeypatch):
    """Partners placed at distance rc*(1 +- k ulp) around random centres: the cut-off decision must be the reference's
    exact fp32 expression (separately rounded products and sums, compared against the double rc*rc)."""
    monkeypatch.setenv("NL_SWEEP_VARIANT", str(variant))
    rng = np.random.default_rng(77)
    rc, L = 3.3, 40.0
    nc = 6000
    centres = rng.uniform(4.0, L - 4.0, size=(nc, 3))
    d = rng.normal(size=(nc, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    scale = 1.0 + rng.integers(-6, 7, size=(nc, 1)) * 2.0 ** -23
    q = np.concatenate([centres, centres + d * rc * scale]).astype(np.float32)
    q = q[rng.permutation(len(q))]
    box = (L, L, L)
    ref = _po().build(q, rc, box)
    _, nop, kp, sl = gpu_build(q, rc, box)
    assert int(kp[-1]) == ref.npairs
    assert np.array_equal(nop, ref.number_of_partners)
    assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)


@pytest.mark.parametrize("variant", SWEEP_VARIANTS)
def test_pairs_at_the_cutoff_fp64(variant, monkeypatch):
    """The screened fp64 search (fp32 screen + the reference's fp64 expression inside an error band, nl_kernels.hpp
    "screened fp64 search") under adversarial input: partners at rc (1 +- k 2^-52) -- the last bits of the exact decision --
    and at rc (1 +- k 2^-24), k up to 96 -- on both sides of the band's edges -- around centres spread over a 300-wide box,
    where the coordinates relative to the cell centre and hence R_i + R_j of the band formula are as large as they get
    (and the absolute coordinates have lost 8 bits to the box size).  The decision must be the reference's,
    neighlist_cpu.hpp:219-223: half and full list."""
    import torch

    monkeypatch.setenv("NL_SWEEP_VARIANT", str(variant))
    rng = np.random.default_rng(78)
    rc, L = 3.3, 300.0
    nc = 20000
    centres = rng.uniform(4.0, L - 4.0, size=(nc, 3))
    # a third of the centres in the far corners of their cells (largest L1 norm relative to the cell centre)
    ms = L / int(L / rc)
    corner = (np.floor(centres[: nc // 3] / ms) + rng.choice([0.002, 0.998], size=(nc // 3, 3))) * ms
    centres[: nc // 3] = corner
    d = rng.normal(size=(nc, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k52 = rng.integers(-8, 9, size=(nc, 1)) * 2.0 ** -52
    k24 = rng.integers(-96, 97, size=(nc, 1)) * 2.0 ** -24
    scale = 1.0 + np.where(rng.random((nc, 1)) < 0.5, k52, k24)
    q = np.concatenate([centres, centres + d * rc * scale]).astype(np.float64)
    q = np.clip(q, 0.0, np.nextafter(L, 0.0))
    q = q[rng.permutation(len(q))]
    box = (L, L, L)
    ref = _po().build(q, rc, box)
    nl, nop, kp, sl = gpu_build(q, rc, box)
    assert int(kp[-1]) == ref.npairs
    assert np.array_equal(nop, ref.number_of_partners)
    assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)
    nl.set_full_list(True)
    nl.MakeNeighList(torch.from_numpy(q).cuda(), len(q))
    want_kp, want_list, want_cnt = _full_from_half(ref)
    fkp, flst, fcnt = (t.cpu().numpy() for t in nl.full_csr())
    assert np.array_equal(fcnt, want_cnt) and np.array_equal(fkp.astype(np.int64), want_kp)
    assert np.array_equal(canonical_csr(fkp, flst), want_list)


def _full_from_half(h):
    """Symmetrised oracle list: (key_pointer, canonical list, counts) of the FULL neighbour list."""
    n = len(h.key_pointer) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(h.key_pointer))
    cols = h.sorted_list.astype(np.int64)
    a = np.concatenate([rows, cols])
    b = np.concatenate([cols, rows])
    cnt = np.bincount(a, minlength=n)
    kp = np.concatenate([[0], np.cumsum(cnt)])
    key = (a << 32) | b  # (row, partner) as one 64-bit key: ids are below 2^31
    key.sort()
    return kp, (key & 0xFFFFFFFF).astype(np.int32), cnt.astype(np.int32)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("case", [
    (50000, (36.84, 36.84, 36.84), 3.3, 51),  # hit-mask path
    (9000, (25.0, 14.0, 19.0), 3.1, 52),      # non-cubic
    (40000, (20.0, 20.0, 20.0), 3.3, 53),     # rho 5: two full sweeps / multi-batch re-search
])
def test_full_list_matches_the_symmetrised_oracle(case, dtype):
    """NL_LIST_FULL (the reference GPU kernels' contract: every j != i within the cut-off, kernel_impl.cuh:24-33):
    the full CSR, and the transposed list[k*N + i] converted from it, against the oracle's half list mirrored."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    n, box, rc, seed = case
    q, box = inputs.uniform_box(n, dtype=dtype, seed=seed, box=box)
    want_kp, want_list, want_cnt = _full_from_half(_po().build(q, rc, box))
    nl = NeighListGPU(rc, *box, dtype=torch.float32 if dtype == np.float32 else torch.float64, full_list=True)
    nl.Initialize(n)
    nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
    kp, lst, cnt = (t.cpu().numpy() for t in nl.full_csr())
    assert nl.number_of_pairs() == int(want_kp[-1]) and nl.half_number_of_pairs() * 2 == int(want_kp[-1])
    assert np.array_equal(cnt, want_cnt)
    assert np.array_equal(kp.astype(np.int64), want_kp)
    assert np.array_equal(canonical_csr(kp, lst), want_list)
    with pytest.raises(Exception):
        nl.key_pointer()  # half accessors refuse a full build
    # the GPU class's layout, converted from the full CSR
    t = nl.neigh_list().cpu().numpy()
    tc = nl.number_of_partners().cpu().numpy()
    assert np.array_equal(tc, want_cnt)
    for i in list(range(0, n, max(1, n // 997))) + [n - 1]:
        assert np.array_equal(np.sort(t[: tc[i], i]), want_list[want_kp[i]:want_kp[i + 1]])
    # and back to the half list on the same handle
    nl.set_full_list(False)
    nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
    assert nl.half_number_of_pairs() * 2 == int(want_kp[-1])
    assert int(nl.key_pointer()[-1]) * 2 == int(want_kp[-1])


def test_build_is_ordered_after_pending_work_on_the_callers_stream():
    """The positions are produced by a kernel queued on the caller's stream (here torch's default stream = HIP's null
    stream) immediately before the build: the build must run after it.  (A private non-blocking stream for
    stream == NULL read half-written positions at N = 1M: regression.)"""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    with open(os.path.join(GOLDEN, "known_answers.json")) as f:
        known = json.load(f)
    want = int(known["u1M_rho1_f32"]["npairs"])
    q, box = inputs.uniform_box(1 << 20, 1.0, np.float32)
    nl = NeighListGPU(3.3, *box, dtype=torch.float32)
    nl.Initialize(len(q))
    qd = torch.from_numpy(q).cuda()
    perm = torch.randperm(len(q), device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        qp = qd[perm].contiguous()           # gather kernel, asynchronous
        nl.MakeNeighList(qp, len(q))         # must wait for it
        assert nl.half_number_of_pairs() == want
        perm = perm.flip(0)


def test_random_small_boxes_against_oracle():
    """Sixty seeded random problems: N from 1 to 6000, non-cubic boxes of 3..12 cells per axis, cut-offs from 0.5 to
    5, both dtypes, half and full list; a tenth of the particles snapped onto cell faces / box faces, a few exact
    duplicates.  Everything bit-exact against the oracle."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    rng = np.random.default_rng(20261003)
    for case in range(60):
        dtype = np.float32 if case % 2 == 0 else np.float64
        rc = float(rng.uniform(0.5, 5.0))
        mesh = rng.integers(3, 13, size=3)
        box = tuple(float(m * rc * rng.uniform(1.0, 1.3)) for m in mesh)
        n = int(rng.integers(1, 6001))
        q = np.zeros((n, 4), dtype=dtype)
        q[:, :3] = rng.uniform(0.0, 1.0, size=(n, 3)) * np.array(box)
        snap = rng.random(n) < 0.1
        ms = np.array([b / int(b / rc) for b in box])
        q[snap, :3] = np.round(q[snap, :3] / ms) * ms  # onto cell faces (incl. 0 and L)
        q[:, :3] = np.minimum(q[:, :3], np.nextafter(np.array(box, dtype=dtype), dtype(0)))
        if n > 10:
            q[-3:] = q[:3]  # exact duplicates: distinct ids at distance 0
        ref = _po().build(q, rc, box)
        nl = NeighListGPU(rc, *box, dtype=torch.float32 if dtype == np.float32 else torch.float64,
                          full_list=(case % 3 == 2))
        nl.Initialize(n)
        nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
        if case % 3 == 2:
            kp, lst, cnt = (t.cpu().numpy() for t in nl.full_csr())
            want_kp, want_list, want_cnt = _full_from_half(ref)
            assert np.array_equal(kp.astype(np.int64), want_kp), case
            assert np.array_equal(canonical_csr(kp, lst), want_list), case
        else:
            kp, sl = nl.key_pointer().cpu().numpy(), nl.sorted_list().cpu().numpy()
            assert int(kp[-1]) == ref.npairs, case
            assert np.array_equal(nl.half_number_of_partners().cpu().numpy(), ref.number_of_partners), case
            assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list), case


@pytest.mark.parametrize("variant", [v for v in SWEEP_VARIANTS if v >= 3])
def test_random_mask_pipeline_boxes_against_oracle(variant, monkeypatch):
    """Twelve seeded random problems large enough for the hit-mask pipelines: non-cubic boxes of 5..10 cells per axis
    with 15..38 particles per cell."""
    monkeypatch.setenv("NL_SWEEP_VARIANT", str(variant))
    rng = np.random.default_rng(99 + variant)
    for case in range(12):
        rc = float(rng.uniform(1.0, 4.0))
        mesh = rng.integers(5, 11, size=3)
        box = tuple(float(m * rc * rng.uniform(1.0, 1.25)) for m in mesh)
        n = int(int(mesh[0]) * int(mesh[1]) * int(mesh[2]) * rng.uniform(15.0, 38.0))
        q, box = inputs.uniform_box(n, dtype=np.float32, seed=1000 + case, box=box)
        ref = _po().build(q, rc, box)
        nl, nop, kp, sl = gpu_build(q, rc, box)
        info = nl.build_info()
        assert info["masks"] and info["variant"] == variant, (case, info)
        assert int(kp[-1]) == ref.npairs, case
        assert np.array_equal(nop, ref.number_of_partners), case
        assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list), case


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("pbc", [False, True])
def test_lj_forces_from_the_list(dtype, full, pbc):
    """The list's consumer: Lennard-Jones forces and energies from the half list (atomics) and
    from the full list (gather) against float64 numpy on the oracle's pair list; tolerance 2e-4 (fp32) / 1e-11 (fp64)
    of the largest component.  pbc: the minimum-image list -- pairs across the periodic faces must enter the forces
    at their image (the consumer used raw coordinates and dropped them silently)."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    n, box, rc = 30000, (32.0, 32.0, 32.0), 3.0
    q, box = inputs.uniform_box(n, dtype=dtype, seed=61, box=box)
    # keep particles apart (r > 0.8 sigma) so that the reference sum is well conditioned in fp32
    ref = _po().build_pbc(q, rc, box) if pbc else _po().build(q, rc, box)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ref.key_pointer))
    cols = ref.sorted_list.astype(np.int64)
    d = q[rows, :3].astype(np.float64) - q[cols, :3].astype(np.float64)
    if pbc:
        L = np.array(box, dtype=np.float64)
        d -= L * np.round(d / L)
        assert np.any(np.abs(q[rows, :3].astype(np.float64) - q[cols, :3].astype(np.float64)).max(axis=1) > 16.0)
    r2 = (d * d).sum(axis=1)
    keep = r2 > 0.64
    rows, cols, d, r2 = rows[keep], cols[keep], d[keep], r2[keep]
    s6 = (1.0 / r2) ** 3
    fr = 24.0 * (2.0 * s6 * s6 - s6) / r2
    want = np.zeros((n, 4))
    for c in range(3):
        np.add.at(want[:, c], rows, fr * d[:, c])
        np.add.at(want[:, c], cols, -fr * d[:, c])
    pe = 4.0 * (s6 * s6 - s6)
    np.add.at(want[:, 3], rows, 0.5 * pe)
    np.add.at(want[:, 3], cols, 0.5 * pe)
    close = np.zeros(n, dtype=bool)  # particles with a partner closer than 0.8: excluded from the comparison
    rr = np.repeat(np.arange(n, dtype=np.int64), np.diff(ref.key_pointer))
    close[rr[~keep]] = True
    close[ref.sorted_list.astype(np.int64)[~keep]] = True

    nl = NeighListGPU(rc, *box, dtype=torch.float32 if dtype == np.float32 else torch.float64, full_list=full,
                      minimum_image=pbc)
    nl.Initialize(n)
    qd = torch.from_numpy(q).cuda()
    nl.MakeNeighList(qd, n)
    got = nl.lj_forces(qd, 1.0, 1.0).cpu().numpy().astype(np.float64)
    ok = ~close
    scale = np.abs(want[ok]).max(axis=0)
    tol = 2e-4 if dtype == np.float32 else 1e-11
    assert np.all(np.abs(got[ok] - want[ok]) <= tol * scale), (np.abs(got[ok] - want[ok]) / scale).max(axis=0)


def test_md_loop_with_skin_rebuilds_and_resorting(monkeypatch):
    """SURVEY section 8 f2/f3 together: a Lennard-Jones droplet integrated with a skin list that is rebuilt on
    displacement and re-sorted into cell order every few rebuilds.  The forces from the reused (rc + skin) list must
    equal the forces from a fresh list at every step checked, and the total energy must be conserved."""
    import torch

    sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util

    spec = importlib.util.spec_from_file_location("md_loop", os.path.join(sys_path_root, "tools", "md_loop.py"))
    md = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(md)
    from md_neighbor_list_amd import NeighListGPU

    monkeypatch.setattr(md, "SORT_FREQ", 3)
    q, v = md.fcc_droplet(8, 1.56, 32.0, np.float64)
    sim = md.Simulation(q, v, 32.0, skin=0.2)
    e0 = sim.energy()
    fresh = NeighListGPU(2.5, 32.0, 32.0, 32.0, dtype=torch.float64, full_list=True)
    fresh.Initialize(len(q))
    for step in range(300):
        sim.step()
        if step % 37 == 0:
            fresh.MakeNeighList(sim.q, len(q))
            want = fresh.lj_forces(sim.q, 1.0, 1.0)
            assert torch.allclose(sim.f, want, rtol=1e-10, atol=1e-10), step
    assert sim.builds >= 5 and sim.sorts >= 1, (sim.builds, sim.sorts)
    assert sorted(sim.ids.cpu().tolist()) == list(range(len(q)))
    # (the potential is truncated at rc without a shift: every pair crossing rc moves the energy by 0.016 eps, so the
    # total is conserved only to a few 1e-3; an integration or list error would show at the 1e-1 level)
    assert abs(sim.energy() - e0) < 5e-3 * abs(e0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_minimum_image_mode_against_its_oracle(dtype):
    """nl_set_periodic(1) (not in the reference): bit-exact against oracle.build_pbc, which is
    itself checked against a float64 brute force (tests/test_oracle.py).  Small and large meshes, particles outside
    [0, L) and on the faces, the mask pipeline and the two-sweep path, half and full list."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    rng = np.random.default_rng(314)
    for case in range(14):
        rc = float(rng.uniform(1.0, 3.5))
        mesh = rng.integers(3, 4 if case < 4 else 11, size=3)
        box = tuple(float(m * rc * rng.uniform(1.0, 1.3)) for m in mesh)
        ncell = int(mesh[0]) * int(mesh[1]) * int(mesh[2])
        n = int(ncell * rng.uniform(5.0, 70.0 if case % 3 == 0 else 36.0))
        q = np.zeros((n, 4), dtype=dtype)
        q[:, :3] = rng.uniform(-0.3, 1.3, size=(n, 3)) * np.array(box)
        q[:30, :3] = np.round(q[:30, :3] / np.array(box)) * np.array(box)
        q[:, :3] = np.clip(q[:, :3], -0.9 * np.array(box), 1.9 * np.array(box))
        ref = _po().build_pbc(q, rc, box)
        full = case % 4 == 3
        nl = NeighListGPU(rc, *box, dtype=torch.float32 if dtype == np.float32 else torch.float64,
                          minimum_image=True, full_list=full)
        nl.Initialize(n)
        nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
        if full:
            # row i in the frame of particle i (ghost-particle semantics), NOT the symmetrised half list: see
            # test_minimum_image_full_list_is_evaluated_in_the_row_frame
            kp, lst, cnt = (t.cpu().numpy() for t in nl.full_csr())
            want = _po().build_pbc_full(q, rc, box)
            assert np.array_equal(kp.astype(np.int64), want.key_pointer), case
            assert np.array_equal(canonical_csr(kp, lst), want.sorted_list), case
        else:
            kp, sl = nl.key_pointer().cpu().numpy(), nl.sorted_list().cpu().numpy()
            assert int(kp[-1]) == ref.npairs, (case, int(kp[-1]), ref.npairs)
            assert np.array_equal(nl.half_number_of_partners().cpu().numpy(), ref.number_of_partners), case
            assert np.array_equal(canonical_csr(kp, sl), ref.sorted_list), case
    # and the open box (the reference's semantics) is a different list on the same positions
    nl.set_periodic(False)
    nl.set_full_list(False)
    nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
    assert nl.half_number_of_pairs() < ref.npairs


def test_minimum_image_full_list_is_evaluated_in_the_row_frame():
    """fp32, partners across a periodic face at distance rc*(1 +- k ulp): (q_j + L) - q_i and (q_i - L) - q_j round
    differently, so some of these pairs are accepted in one direction only.  The FULL minimum-image list is defined row
    by row in the frame of the row's particle (oracle.build_pbc_full; what a code with ghost particles computes), the
    half list in the frame of the smaller id (oracle.build_pbc): both bit-exact, and this case must contain pairs on
    which the two directions disagree (else it tests nothing)."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    rng = np.random.default_rng(2718)
    rc, L = 3.3, 40.0
    box = (L, L, L)
    nc = 8000
    centres = rng.uniform(0.0, L, size=(nc, 3))
    centres[:, 0] = rng.uniform(0.0, 1.5, size=nc)  # near the low x face: most partners lie across it
    d = rng.normal(size=(nc, 3))
    d[:, 0] = -np.abs(d[:, 0])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    scale = 1.0 + rng.integers(-6, 7, size=(nc, 1)) * 2.0 ** -23
    partners = centres + d * rc * scale
    q = np.zeros((2 * nc, 4), dtype=np.float32)
    q[:, :3] = np.concatenate([centres, np.mod(partners, L)]).astype(np.float32)
    q[:, :3] = np.minimum(q[:, :3], np.nextafter(np.float32(L), np.float32(0)))
    q = q[rng.permutation(len(q))]
    n = len(q)
    half, full = _po().build_pbc(q, rc, box), _po().build_pbc_full(q, rc, box)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(full.key_pointer))
    directed = set(zip(rows.tolist(), full.sorted_list.tolist()))
    one_way = sum((j, i) not in directed for i, j in directed)
    assert one_way > 0, "the case must contain pairs present in one direction only"
    for want, is_full in ((half, False), (full, True)):
        nl = NeighListGPU(rc, *box, dtype=torch.float32, minimum_image=True, full_list=is_full)
        nl.Initialize(n)
        nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
        if is_full:
            kp, lst, _ = (t.cpu().numpy() for t in nl.full_csr())
        else:
            kp, lst = nl.key_pointer().cpu().numpy(), nl.sorted_list().cpu().numpy()
        assert np.array_equal(kp.astype(np.int64), want.key_pointer)
        assert np.array_equal(canonical_csr(kp, lst), want.sorted_list)


@pytest.mark.parametrize("variant", [v for v in SWEEP_VARIANTS if v >= 3])
def test_full_list_at_baseline_size(variant, monkeypatch):
    """BASELINE config 2 (N = 1 048 576) as a FULL list -- the search without the id test (own bit cleared at the
    end): twice the stored pair count, every row symmetric on a sample, and the upper part (j > i) of the list hashes
    to the stored answer of the half list."""
    import torch

    from md_neighbor_list_amd import NeighListGPU
    from oracle import pyoracle as po

    monkeypatch.setenv("NL_SWEEP_VARIANT", str(variant))
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["u1M_rho1_f32"]
    q, box = inputs.uniform_box(ka["n"], 1.0, np.float32)
    n = len(q)
    nl = NeighListGPU(ka["rc"], *box, dtype=torch.float32, full_list=True)
    nl.Initialize(n)
    nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
    assert nl.build_info()["variant"] == variant
    kp, lst, cnt = (t.cpu().numpy() for t in nl.full_csr())
    assert int(kp[-1]) == 2 * ka["npairs"] == len(lst)
    assert np.array_equal(np.diff(kp), cnt)
    rows = np.repeat(np.arange(n, dtype=np.int32), cnt)
    assert not np.any(lst == rows)  # no self pairs
    up = lst > rows
    nop = np.bincount(rows[up], minlength=n).astype(np.int32)
    hk = np.concatenate([[0], np.cumsum(nop)]).astype(np.int64)
    assert int(nop.max()) == ka["nop_max"]
    assert f"{po.HalfList(nop, hk, lst[up]).hash():016x}" == ka["hash"]
    # symmetry on a sample of rows: i in row j for every j in row i
    rng = np.random.default_rng(11)
    for i in rng.integers(0, n, size=200):
        for j in lst[kp[i]:kp[i + 1]][:8]:
            assert i in lst[kp[j]:kp[j + 1]]


def test_graph_replay_builds_the_same_lists():
    """nl_set_graph(1): asynchronous builds replayed from a captured hipGraph -- new positions in the same buffer
    (replay), another particle count and another buffer (capture again), a list that has to grow (capture again), a
    synchronous build in between; every result against the oracle."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    rng = np.random.default_rng(8)
    rc, box = 3.0, (36.0, 33.0, 30.0)
    n = 30000
    nl = NeighListGPU(rc, *box, dtype=torch.float32)
    nl.Initialize(n)
    nl.set_graph(True)
    buf = torch.empty((n, 4), dtype=torch.float32, device="cuda")

    def check(q, t, m):
        ref = _po().build(q, rc, box)
        kp, sl = nl.key_pointer().cpu().numpy(), nl.sorted_list().cpu().numpy()
        assert int(kp[-1]) == ref.npairs
        assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)

    for step in range(4):  # same buffer, same count: one capture, three replays
        q = np.zeros((n, 4), dtype=np.float32)
        q[:, :3] = rng.uniform(0.0, 1.0, size=(n, 3)) * np.array(box)
        q[:, :3] = np.minimum(q[:, :3], np.nextafter(np.array(box, dtype=np.float32), np.float32(0)))
        buf.copy_(torch.from_numpy(q))
        nl.MakeNeighList(buf, n, sync=False)
        nl.synchronize()
        check(q, buf, n)
    m = 21000  # fewer particles, another buffer
    q2 = np.zeros((m, 4), dtype=np.float32)
    q2[:, :3] = rng.uniform(0.0, 1.0, size=(m, 3)) * np.array(box)
    q2[:, :3] = np.minimum(q2[:, :3], np.nextafter(np.array(box, dtype=np.float32), np.float32(0)))
    t2 = torch.from_numpy(q2).cuda()
    nl.MakeNeighList(t2, m, sync=False)
    nl.synchronize()
    check(q2, t2, m)
    nl.MakeNeighList(t2, m, sync=True)  # synchronous build through the same graph
    check(q2, t2, m)
    # a clustered configuration overflows the list: the synchronous build grows it (new buffer: capture again)
    q3 = q2.copy()
    q3[:, :3] = q3[:, :3] * 0.45
    t3 = torch.from_numpy(q3).cuda()
    nl.MakeNeighList(t3, m, sync=True)
    check(q3, t3, m)
    nl.MakeNeighList(t3, m, sync=False)
    nl.synchronize()
    check(q3, t3, m)


@pytest.mark.parametrize("key", ["w2x1M_rho1_f32", "w8x1M_rho1_f32", "u8M_rho1_f32"])
def test_weak_scaling_boxes_on_one_gpu(key):
    """The boxes of bench.py --gpus 2 / 8 (the 1 M cube repeated along z: inputs.weak_scaling_box) and the cubic 8 M
    box, built on ONE GPU against the compiled reference's answers for them: pair count, the checksum and the maximum
    of number_of_partners -- the numbers the multi-GPU bench line is compared with (`half_pairs_reference`).  No list
    download (2.5 GB at 8 M)."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))[key]
    if key.startswith("w"):
        q, box = inputs.weak_scaling_box(int(key[1:key.index("x")]))
    else:
        q, box = inputs.uniform_box(ka["n"], 1.0, np.float32)
    assert len(q) == ka["n"] and np.allclose(box, ka["box"])
    nl = NeighListGPU(ka["rc"], *box, dtype=torch.float32)
    nl.Initialize(len(q))
    nl.MakeNeighList(torch.from_numpy(q).cuda(), len(q))
    assert nl.half_number_of_pairs() == ka["npairs"]
    nop = nl.half_number_of_partners().cpu().numpy().astype(np.int64)
    assert int(nop.max()) == ka["nop_max"]
    assert int((nop * (np.arange(len(nop)) % 1000003)).sum()) == ka["nop_weighted_sum"]
    kp = nl.key_pointer().cpu().numpy()
    assert int(kp[-1]) == ka["npairs"] and np.array_equal(np.diff(kp), nop)


# ---------------------------------------------------------------------------------------------- 64-bit list offsets
@pytest.mark.parametrize("variant", SWEEP_VARIANTS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_wide_offsets_give_the_same_lists(variant, dtype, monkeypatch):
    """nl_set_offset_width(64): the build keeps int64 list offsets (what a list beyond INT32_MAX entries needs, e.g.
    BASELINE config 4 on one device) -- on boxes the oracle finishes in seconds the lists must be the reference's, the
    int64 key_pointer its prefix sums, and the int32 accessor a faithful conversion.  Half and full list."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    monkeypatch.setenv("NL_SWEEP_VARIANT", str(variant))
    for n, box, rc, seed in [(40000, (34.2, 34.2, 34.2), 3.3, 5), (4096, (16.0, 16.0, 16.0), 3.3, 6), (12000, (20.0, 20.0, 20.0), 6.0, 7)]:
        q, box = inputs.uniform_box(n, dtype=dtype, seed=seed, box=box)
        ref = _po().build(q, rc, box)
        nl = NeighListGPU(rc, *box, dtype=torch.float32 if dtype == np.float32 else torch.float64)
        nl.set_offset_width(64)
        nl.Initialize(n)
        qd = torch.from_numpy(q).cuda()
        nl.MakeNeighList(qd, n)
        assert nl.build_info()["offset_bits"] == 64
        kp64 = nl.key_pointer64().cpu().numpy()
        assert kp64.dtype == np.int64 and np.array_equal(kp64, ref.key_pointer)
        kp32 = nl.key_pointer().cpu().numpy()
        assert kp32.dtype == np.int32 and np.array_equal(kp32.astype(np.int64), kp64)
        sl = nl.sorted_list().cpu().numpy()
        assert np.array_equal(canonical_csr(kp64, sl), ref.canonical().sorted_list)
        cs, ne = nl.list_checksum()
        assert ne == ref.npairs and cs == ref.hash()
        f = nl.lj_forces(qd)  # the consumer reads the int64 offsets
        nl.set_offset_width(32)
        nl.MakeNeighList(qd, n)
        assert nl.build_info()["offset_bits"] == 32
        assert np.array_equal(nl.key_pointer64().cpu().numpy(), ref.key_pointer)
        # same list, same kernel, other offset type: equal up to the order of the floating-point atomics (half list);
        # compared where the forces are moderate (uniform random particles can sit arbitrarily close: 1e7 and beyond)
        f2 = nl.lj_forces(qd)
        calm = f.abs().amax(dim=1) < 1e3
        assert torch.allclose(f[calm], f2[calm], rtol=1e-3, atol=1e-2)
        # full list
        nl.set_offset_width(64)
        nl.set_full_list(True)
        nl.MakeNeighList(qd, n)
        want_kp, want_list, want_cnt = _full_from_half(ref)
        kp, lst, cnt = (t.cpu().numpy() for t in nl.full_csr(64))
        assert np.array_equal(kp, want_kp) and np.array_equal(cnt, want_cnt)
        assert np.array_equal(canonical_csr(kp, lst), want_list)


def test_wide_offsets_by_capacity_and_overflow_status():
    """The width follows the capacity: a handle whose list may hold more than INT32_MAX entries builds with int64
    offsets; nl_set_offset_width(32) on a list that needs more is an error status, never a wrapped list."""
    import torch

    from md_neighbor_list_amd import NeighListGPU
    from md_neighbor_list_amd._lib import NL_ERR_INDEX_OVERFLOW, NLError

    q, box = inputs.uniform_box(36000, dtype=np.float32, seed=7, box=(34.2, 34.2, 34.2))
    ref = _po().build(q, 3.3, box)
    nl = NeighListGPU(3.3, *box, dtype=torch.float32)
    nl.Initialize(len(q))
    nl.set_capacity((1 << 31) + 4096)  # 8 GiB of list: fits a 288 GB device many times over
    nl.MakeNeighList(torch.from_numpy(q).cuda(), len(q))
    assert nl.build_info()["offset_bits"] == 64
    assert nl.half_number_of_pairs() == ref.npairs
    assert np.array_equal(canonical_csr(nl.key_pointer64().cpu().numpy(), nl.sorted_list().cpu().numpy()),
                          ref.canonical().sorted_list)


def test_list_between_2_30_and_2_31_entries():
    """A list of 1.2e9 entries (16 M particles, rho = 1) with 32-bit offsets: beyond 2^30 entries a 32-bit BYTE offset
    wraps -- every list address is formed in 64 bits.  Pair count, number_of_partners checksum and the device-side
    pair-set checksum against the restatement's count-mode answer; both pipelines (masks, two sweeps)."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["u16M_rho1_f32"]
    q, box = inputs.uniform_box(ka["n"], 1.0, np.float32)
    qd = torch.from_numpy(q).cuda()
    for variant in SWEEP_VARIANTS:
        os.environ["NL_SWEEP_VARIANT"] = str(variant)
        try:
            nl = NeighListGPU(ka["rc"], *box, dtype=torch.float32)
        finally:
            del os.environ["NL_SWEEP_VARIANT"]
        nl.set_offset_width(32)
        nl.Initialize(len(q))
        nl.MakeNeighList(qd, len(q))
        assert nl.build_info()["offset_bits"] == 32 and nl.build_info()["variant"] == variant
        assert (1 << 30) < nl.half_number_of_pairs() == ka["npairs"] < (1 << 31)
        nop = nl.half_number_of_partners().cpu().numpy().astype(np.int64)
        assert int((nop * (np.arange(len(nop)) % 1000003)).sum()) == ka["nop_weighted_sum"]
        cs, ne = nl.list_checksum()
        assert ne == ka["npairs"] and f"{cs:016x}" == ka["hash"]
        del nl


def _check_known_answer_on_device(nl, ka):
    """Pair count, count checksum, maximum and the device-side pair-set checksum of the last build against a stored answer."""
    assert nl.half_number_of_pairs() == ka["npairs"]
    nop = nl.half_number_of_partners().cpu().numpy().astype(np.int64)
    assert int(nop.max()) == ka["nop_max"]
    assert int((nop * (np.arange(len(nop)) % 1000003)).sum()) == ka["nop_weighted_sum"]
    cs, ne = nl.list_checksum()
    assert ne == ka["npairs"] and f"{cs:016x}" == ka["hash"]
    return nop


def test_config4_on_one_gpu():
    """BASELINE config 4 (N = 33 554 432, rho = 1: 2 496 524 913 pairs > INT32_MAX) on ONE device: 64-bit offsets by
    capacity, the restatement's count-mode answer (the reference itself overflows here, neighlist_cpu.hpp:15,29)."""
    import torch

    from md_neighbor_list_amd import NeighListGPU
    from md_neighbor_list_amd._lib import NL_ERR_INDEX_OVERFLOW, NLError

    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["u32M_rho1_f32"]
    q, box = inputs.uniform_box(ka["n"], 1.0, np.float32)
    nl = NeighListGPU(ka["rc"], *box, dtype=torch.float32)
    nl.Initialize(len(q))
    qd = torch.from_numpy(q).cuda()
    nl.MakeNeighList(qd, len(q))
    assert nl.build_info()["offset_bits"] == 64
    nop = _check_known_answer_on_device(nl, ka)
    kp = nl.key_pointer64().cpu().numpy()
    assert int(kp[-1]) == ka["npairs"] and np.array_equal(np.diff(kp), nop)
    with pytest.raises(NLError) as e:  # the reference's int32 key_pointer cannot address this list
        nl.key_pointer()
    assert e.value.code == NL_ERR_INDEX_OVERFLOW
    # a few rows against an O(N) scan
    sl = nl.sorted_list()
    rng = np.random.default_rng(4)
    rc2 = ka["rc"] * ka["rc"]
    for i in rng.integers(0, len(q), size=4):
        d = q[:, :3] - q[i, :3]
        r2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        want = np.nonzero(~(r2.astype(np.float64) > rc2) & (np.arange(len(q)) > i))[0]
        assert np.array_equal(np.sort(sl[kp[i]:kp[i + 1]].cpu().numpy()), want.astype(np.int32))


def test_config4_as_eight_slabs_on_one_gpu():
    """BASELINE config 4 cut into the 8 z-slabs of bench.py --gpus 8, built one after another on ONE device (owned
    particles + the two ghost layers, exactly what every rank builds): per-slab pair count, count checksum and pair-set
    checksum against the restatement's per-slab answers; their sum is the global answer."""
    import torch

    from md_neighbor_list_amd import NeighListGPU, slab

    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["u32M_rho1_f32"]
    q, box = inputs.uniform_box(ka["n"], 1.0, np.float32)
    rc = ka["rc"]
    qd = torch.from_numpy(q).cuda()
    iz = slab.z_layer(qd, box, rc)
    mz = int(box[2] / rc)
    gid = torch.arange(len(q), dtype=torch.int32, device="cuda")
    w = (torch.arange(len(q), dtype=torch.int64, device="cuda") % 1000003)
    total_pairs, total_cs = 0, 0
    nl = NeighListGPU(rc, *box, dtype=torch.float32)
    for r, (z_lo, z_hi) in enumerate(slab.split_layers(mz, 8)):
        want = ka["slabs8"][r]
        assert (z_lo, z_hi) == (want["z_lo"], want["z_hi"])
        own = torch.nonzero((iz >= z_lo) & (iz < z_hi)).flatten()
        g_lo = torch.nonzero(iz == (z_lo - 1) % mz).flatten()
        g_hi = torch.nonzero(iz == z_hi % mz).flatten()
        idx = torch.cat([own, g_lo, g_hi])
        q_all, gid_all = qd[idx].contiguous(), gid[idx].contiguous()
        n_rows = int(own.numel())
        assert n_rows == want["n_rows"]
        nl.Initialize(int(idx.numel()))
        nl.set_capacity(int(n_rows * 75 * 1.3) + 4096)
        nl.MakeNeighListSlab(q_all, gid_all, n_rows, z_lo, z_hi)
        assert nl.half_number_of_pairs() == want["npairs"]
        nop = nl.half_number_of_partners().to(torch.int64)
        assert int((nop * w[own]).sum().item()) == want["nop_weighted_sum"]
        cs, ne = nl.list_checksum()
        assert ne == want["npairs"] and f"{cs:016x}" == want["hash"], r
        total_pairs += ne
        total_cs = (total_cs + cs) & ((1 << 64) - 1)
    assert total_pairs == ka["npairs"] and f"{total_cs:016x}" == ka["hash"]


@pytest.mark.parametrize("key", ["u1M_rho1_f64_rc66", "u1M_rho1_f32_rc66"])
def test_config5_full_size(key):
    """BASELINE config 5 at its stated size: N = 1 048 576, rc = 6.6 (2 x cut-off: 311 particles per cell, 5.9e8
    pairs), fp64 (and the fp32 sibling).  The reference overruns its MAX_PARTNERS*N buffers there
    (neighlist_cpu.hpp:37,76-78); the answer is the restatement's count mode."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))[key]
    dt = np.float64 if "f64" in key else np.float32
    q, box = inputs.uniform_box(ka["n"], 1.0, dt)
    nl = NeighListGPU(ka["rc"], *box, dtype=torch.float64 if dt == np.float64 else torch.float32)
    nl.Initialize(len(q))
    nl.MakeNeighList(torch.from_numpy(q).cuda(), len(q))
    nop = _check_known_answer_on_device(nl, ka)
    kp = nl.key_pointer().cpu().numpy()
    assert np.array_equal(np.diff(kp), nop)
    sl = nl.sorted_list()
    rng = np.random.default_rng(5)
    rc2 = ka["rc"] * ka["rc"]
    for i in rng.integers(0, len(q), size=6):
        d = q[:, :3] - q[i, :3]
        r2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        want = np.nonzero(~(r2.astype(np.float64) > rc2) & (np.arange(len(q)) > i))[0]
        assert np.array_equal(np.sort(sl[kp[i]:kp[i + 1]].cpu().numpy()), want.astype(np.int32))


# ---------------------------------------------------------------------------------------------- periodic re-sorting
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_resort_then_rebuild_equals_the_oracle_on_the_permuted_input(dtype):
    """SURVEY section 8 f2 (the reference's dead SORT_FREQ / CopyGather / SortPtclData hooks, neighlist_gpu.hpp:72,144-151,
    neighlist_cpu.hpp:176-180,421) as a library feature: nl_resort permutes the caller's arrays into the last build's
    cell order; the list built afterwards must be exactly the ORACLE's list of the permuted particles, and -- mapped
    back through the carried ids -- the oracle's pair set of the original particles.  Repeated, as an MD loop would."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    n, box, rc = 40000, (34.2, 30.0, 41.0), 3.3
    q, box = inputs.uniform_box(n, dtype=dtype, seed=31, box=box)
    ref0 = _po().build(q, rc, box)
    kp0 = ref0.key_pointer
    lo0 = np.repeat(np.arange(n, dtype=np.int64), np.diff(kp0))
    pairs0 = np.unique(lo0 * n + ref0.sorted_list.astype(np.int64))
    nl = NeighListGPU(rc, *box, dtype=torch.float32 if dtype == np.float32 else torch.float64)
    nl.Initialize(n)
    qd = torch.from_numpy(q).cuda()
    ids = torch.arange(n, dtype=torch.int32, device="cuda")
    vel = torch.from_numpy(np.random.default_rng(1).normal(size=(n, 3)).astype(dtype)).cuda()  # a 12- / 24-byte array
    vel0 = vel.clone()
    nl.MakeNeighList(qd, n)
    for rnd in range(2):
        order = nl.cell_order().cpu().numpy()
        assert np.array_equal(np.sort(order), np.arange(n))  # a permutation
        cell, _ = _po().cells(q, rc, box)
        q_host_before = qd.cpu().numpy()
        nl.resort(qd, ids, vel)
        assert np.array_equal(qd.cpu().numpy(), q_host_before[order])  # array[s] <- array[order[s]]
        nl.MakeNeighList(qd, n)
        qp = qd.cpu().numpy()
        idp = ids.cpu().numpy().astype(np.int64)
        assert np.array_equal(qp, q[idp]) and np.array_equal(vel.cpu().numpy(), vel0.cpu().numpy()[idp])
        # cell order: the cells of the stored particles are non-decreasing
        cellp, _ = _po().cells(qp, rc, box)
        assert np.all(np.diff(cellp) >= 0)
        ref = _po().build(qp, rc, box)  # the oracle on the permuted input
        kp, sl = nl.key_pointer().cpu().numpy(), nl.sorted_list().cpu().numpy()
        assert np.array_equal(kp.astype(np.int64), ref.key_pointer)
        assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)
        # ids mapped back: the same set of unordered pairs as the original list
        lo = np.repeat(np.arange(n, dtype=np.int64), np.diff(kp))
        a, b = idp[lo], idp[sl.astype(np.int64)]
        pairs = np.unique(np.minimum(a, b) * n + np.maximum(a, b))
        assert np.array_equal(pairs, pairs0)
        # a second round re-sorts an already sorted system: the order is the identity up to ties inside a cell


def test_cells_handed_to_the_batched_search():
    """Cells whose stencil stream does not fit one LDS batch are put on a device-side list by the COUNT sweep and searched
    by the batched kernels (k_sweep_list_f32 / k_fill_list): a box whose cluster produces such cells among one-batch
    neighbours, half and full list, next to plain boxes."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    rng = np.random.default_rng(5)
    cases = [(50000, (36.84, 36.84, 36.84), 3.3, 0.0), (9000, (25.0, 14.0, 19.0), 3.1, 0.0), (4096, (16.0, 16.0, 16.0), 3.3, 0.0),
             (60000, (40.0, 40.0, 40.0), 3.3, 0.12), (120000, (50.0, 60.0, 45.0), 3.0, 0.02),
             # sparse boxes: the small instances of the COUNT sweep and the expansion (half the LDS buffer), plain and with
             # a cluster whose cells exceed half the buffer (hand-over with masks) and the whole buffer (re-search)
             (20000, (50.0, 50.0, 50.0), 3.3, 0.0), (40000, (50.0, 60.0, 45.0), 3.0, 0.03), (60000, (50.0, 60.0, 45.0), 3.0, 0.008)]
    for n, box, rc, clustered in cases:
        q, box = inputs.uniform_box(n, dtype=np.float32, seed=int(rng.integers(1 << 30)), box=box)
        if clustered:  # part of the particles into two cells' worth of volume: streams of several LDS batches there
            k = int(clustered * n)
            q[:k, :3] = (np.array(box) * 0.5 + rng.uniform(-0.9 * rc, 0.9 * rc, size=(k, 3))).astype(np.float32)
        ref = _po().build(q, rc, box)
        nl, nop, kp, sl = gpu_build(q, rc, box)
        assert nl.build_info()["small_cells"] == (1 if n / (int(box[0] / rc) * int(box[1] / rc) * int(box[2] / rc)) < 15 else 0), nl.build_info()
        assert int(kp[-1]) == ref.npairs, (n, clustered)
        assert np.array_equal(nop, ref.number_of_partners), (n, clustered)
        assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list), (n, clustered)
        assert nl.list_checksum() == (ref.hash(), ref.npairs)
        # the full list on the same handle
        nl.set_full_list(True)
        nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
        want_kp, want_list, want_cnt = _full_from_half(ref)
        fkp, flst, fcnt = (t.cpu().numpy() for t in nl.full_csr())
        assert np.array_equal(fcnt, want_cnt) and np.array_equal(fkp.astype(np.int64), want_kp), (n, clustered)
        assert np.array_equal(canonical_csr(fkp, flst), want_list), (n, clustered)


@pytest.mark.parametrize("rows", [4, 1, 2, 3, 0])
def test_fine_row_search_configurations(rows, monkeypatch):
    """NL_ROWS: the fine-row search of nl_rows.hpp (default -1: in dense boxes; 4: wherever a build qualifies, RowsCfg by
    density) forced to each of its three configurations (1: 16-bit hit words; 2, 3: 32-bit words, larger LDS streams)
    and switched off (0: the 27-cell path) -- same lists.  Boxes: cubic and not, a mesh of 3 along x and y (every cell at the periodic
    wrap in x: windows of two pieces), sparse, a cluster (cells whose stream exceeds the LDS buffer: k_rows_overflow),
    particles outside the box (the reference files them by the truncated, wrapped cell index), half and full list."""
    import torch

    monkeypatch.setenv("NL_ROWS", str(rows))
    rng = np.random.default_rng(70 + rows)
    cases = [(50000, (36.84, 36.84, 36.84), 3.3, ""), (9000, (25.0, 14.0, 19.0), 3.1, ""), (4096, (16.0, 16.0, 16.0), 3.3, ""),
             (9000, (10.5, 10.5, 40.0), 3.3, ""), (3000, (60.0, 60.0, 60.0), 3.3, ""), (60000, (40.0, 40.0, 40.0), 3.3, "cluster"),
             (30000, (30.0, 33.5, 36.5), 3.3, "outside"), (120000, (50.0, 61.0, 46.0), 3.0, "cluster2")]
    for n, box, rc, kind in cases:
        q, box = inputs.uniform_box(n, dtype=np.float32, seed=int(rng.integers(1 << 30)), box=box)
        if kind.startswith("cluster"):
            k = int((0.12 if kind == "cluster" else 0.02) * n)
            q[:k, :3] = (np.array(box) * 0.5 + rng.uniform(-0.9 * rc, 0.9 * rc, size=(k, 3))).astype(np.float32)
        if kind == "outside":  # up to 0.9 box lengths outside, on every side
            k = n // 20
            q[:k, :3] += (rng.choice([-1.0, 1.0], size=(k, 3)) * rng.uniform(0.0, 0.9, size=(k, 3)) * np.array(box)).astype(np.float32)
        ref = _po().build(q, rc, box)
        nl, nop, kp, sl = gpu_build(q, rc, box)
        info = nl.build_info()
        if rows <= 3:
            assert info["fine_rows"] == rows, (kind, info)
        assert int(kp[-1]) == ref.npairs, (n, kind, info)
        assert np.array_equal(nop, ref.number_of_partners), (n, kind, info)
        assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list), (n, kind, info)
        assert nl.list_checksum() == (ref.hash(), ref.npairs)
        nl.set_full_list(True)
        nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
        want_kp, want_list, want_cnt = _full_from_half(ref)
        fkp, flst, fcnt = (t.cpu().numpy() for t in nl.full_csr())
        assert np.array_equal(fcnt, want_cnt) and np.array_equal(fkp.astype(np.int64), want_kp), (n, kind, info)
        assert np.array_equal(canonical_csr(fkp, flst), want_list), (n, kind, info)


def test_dense_boxes_take_the_fine_row_search_by_default():
    """From 40.3 particles per cell on the 27-cell streams no longer fit one LDS batch and the default path is the
    fine-row search (RowsCfg by density): 45 and 75 particles per cell, half and full list, against the oracle."""
    import torch

    for n, box, want in ((45000, (35.9, 35.9, 35.9), 2), (75000, (35.9, 35.9, 35.9), 3)):
        q, box = inputs.uniform_box(n, dtype=np.float32, seed=61, box=box)
        ref = _po().build(q, 3.3, box)
        nl, nop, kp, sl = gpu_build(q, 3.3, box)
        assert nl.build_info()["fine_rows"] == want, nl.build_info()
        assert np.array_equal(nop, ref.number_of_partners)
        assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)
        nl.set_full_list(True)
        nl.MakeNeighList(torch.from_numpy(q).cuda(), n)
        want_kp, want_list, want_cnt = _full_from_half(ref)
        fkp, flst, fcnt = (t.cpu().numpy() for t in nl.full_csr())
        assert np.array_equal(fcnt, want_cnt) and np.array_equal(canonical_csr(fkp, flst), want_list)


def test_fine_row_search_is_not_taken_without_margin(monkeypatch):
    """A box whose cell edge equals the cut-off along z (Lz / rc an integer) keeps the 27-cell search: two particles five
    quarter-planes apart may then be within the cut-off after rounding (rows_margin_ok in nl_api.hip)."""
    monkeypatch.setenv("NL_ROWS", "4")
    q, box = inputs.uniform_box(40000, dtype=np.float32, seed=5, box=(33.0, 33.0, 33.0))
    ref = _po().build(q, 3.3, box)
    nl, nop, kp, sl = gpu_build(q, 3.3, box)
    assert nl.build_info()["fine_rows"] == 0 and nl.build_info()["masks"]
    assert np.array_equal(canonical_csr(kp, sl), ref.canonical().sorted_list)
    q, box = inputs.uniform_box(40000, dtype=np.float32, seed=5, box=(33.0, 33.0, 33.9))  # the margin matters along z only
    nl, nop, kp, sl = gpu_build(q, 3.3, box)
    assert nl.build_info()["fine_rows"] == 1
    assert np.array_equal(canonical_csr(kp, sl), _po().build(q, 3.3, box).canonical().sorted_list)
