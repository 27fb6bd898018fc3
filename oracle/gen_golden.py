#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref, compiled from /root/reference).

Run here (where /root/reference exists):   python oracle/gen_golden.py
Each fixture holds inputs (q, rc, box) and the reference's outputs: number_of_partners, key_pointer and the
canonical (per-particle ascending, make_list.cpp:120-128) sorted_list, plus npairs and the pair-set hash.
A fixture is data only; no reference source text is stored.

The scalar variants of the reference must agree before a fixture is written: -DWITHOUT_LOOP_FUSION and
-DLOOP_FUSION always; -DLOOP_FUSION_SWP only on the dense cases (SWP_OK), because its software-pipelined loop
(neighlist_cpu.hpp:311-356) reads one slot past the stencil list and registers a garbage pair when a particle is the
last entry of its cell's list, which happens as soon as a cell has 13 empty neighbour cells.  Cases with a mesh below 3 per axis are written with ``dup=1``: there the reference
visits some cell pairs more than once and emits duplicate pairs (oracle restatement reproduces that; the HIP path
rejects such boxes with NL_ERR_MESH).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from md_neighbor_list_amd import inputs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SWP_OK = ("u4096_rho1", "fcc_L12", "sc_ties")


def lattice(n_side, a, dtype, origin=0.0):
    g = np.arange(n_side, dtype=np.float64) * a + origin
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    q = np.zeros((n_side**3, 4), dtype=dtype)
    q[:, 0], q[:, 1], q[:, 2] = x.ravel(), y.ravel(), z.ravel()
    rng = np.random.default_rng(7)
    return q[rng.permutation(len(q))]  # index order must not be spatial


def cases():
    for dt in (np.float32, np.float64):
        s = "f32" if dt == np.float32 else "f64"
        yield f"u64_rc1_{s}", *inputs.uniform_box(64, dtype=dt, seed=11, box=(4.0, 4.0, 4.0)), 1.0
        yield f"u512_m3_{s}", *inputs.uniform_box(512, dtype=dt, seed=12, box=(10.0, 10.0, 10.0)), 3.3
        yield f"u4096_rho1_{s}", *inputs.uniform_box(4096, 1.0, dt, seed=12345), 3.3  # BASELINE config 1
        yield f"u4096_rho05_{s}", *inputs.uniform_box(4096, 0.5, dt, seed=12345), 3.3
        yield f"u3000_noncubic_{s}", *inputs.uniform_box(3000, dtype=dt, seed=13, box=(12.0, 17.5, 23.0)), 2.5
        q, box = inputs.fcc_box(1.0, 12.0, dt)
        yield f"fcc_L12_{s}", q, box, 3.3
        # simple cubic lattice, spacing 1.1, rc = 3.3 = 3 spacings: many pairs sit exactly on r2 ~ rc2, and
        # lattice planes coincide with cell faces (L = 13.2 = 4 cells of 3.3): hash-rounding and tie stress.
        yield f"sc_ties_{s}", lattice(12, 1.1, dt), (13.2, 13.2, 13.2), 3.3
        # same lattice pushed against the upper faces: coordinates that round up to the box edge wrap to cell 0
        q = lattice(12, 1.1, dt, origin=1.1)
        q[:, :3] = np.minimum(q[:, :3], np.nextafter(dt(13.2), dt(0)))
        yield f"sc_upper_edge_{s}", q, (13.2, 13.2, 13.2), 3.3
        # half-empty box: whole slabs of empty cells
        q, box = inputs.uniform_box(2000, dtype=dt, seed=14, box=(20.0, 20.0, 20.0))
        q[:, 0] *= 0.5
        yield f"half_empty_{s}", q, box, 3.3
        # slightly outside the box: negative and >= L coordinates wrap by one period (ApplyPBC)
        q, box = inputs.uniform_box(1500, dtype=dt, seed=15, box=(14.0, 14.0, 14.0))
        q[:, :3] = q[:, :3] * dt(1.2) - dt(1.4)
        yield f"outside_wrap_{s}", q, box, 3.3
        # meshes below 3: reference emits duplicates (dup fixtures pin the restatement only); N is tiny because
        # the reference's unchecked 100*N pair buffers (neighlist_cpu.hpp:76-78) must hold the duplicates
        yield f"dup_m1_{s}", *inputs.uniform_box(6, dtype=dt, seed=16, box=(4.0, 4.0, 4.0)), 3.3
        yield f"dup_m2_{s}", *inputs.uniform_box(40, dtype=dt, seed=17, box=(7.0, 7.0, 7.0)), 3.3


def main():
    os.makedirs(OUT, exist_ok=True)
    for v in ("naive", "fused", "swp"):
        if not po.ref_available(v):
            sys.exit(f"oracle/_ref/libnl_ref_{v}.so missing: run `make -C oracle` where /root/reference exists")
    for name, q, box, rc in cases():
        print(f"{name:24s}", end=" ", flush=True)
        variants = ("naive", "fused", "swp") if name.startswith(SWP_OK) else ("naive", "fused")
        outs = {v: po.ref_build(q, rc, box, v)[0] for v in variants}
        can = {v: h.canonical() for v, h in outs.items()}
        for v in variants[1:]:
            assert np.array_equal(can[v].sorted_list, can["naive"].sorted_list), (name, v)
            assert np.array_equal(outs[v].key_pointer, outs["naive"].key_pointer), (name, v)
        h = can["fused"]
        mesh = [int(b / rc) for b in box]
        dup = int(min(mesh) < 3)
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            q=q, rc=np.float64(rc), box=np.array(box, dtype=np.float64), mesh=np.array(mesh, dtype=np.int32),
            number_of_partners=h.number_of_partners, key_pointer=h.key_pointer.astype(np.int64),
            sorted_list=h.sorted_list, npairs=np.int64(h.npairs), hash=np.uint64(h.hash()), dup=np.int32(dup),
            # the raw (visit-order) list of the WITHOUT_LOOP_FUSION variant pins the restatement's order too
            sorted_list_naive_order=outs["naive"].sorted_list,
        )
        print(f"N={len(q):5d} mesh={mesh} P={h.npairs:7d} hash={h.hash():016x} dup={dup}")


def big(only=None):
    """Known answers at the BASELINE sizes (too big to store as lists): pair count, pair-set hash, a checksum of
    number_of_partners, its maximum.  From the compiled reference (-DLOOP_FUSION build with pinned flags)."""
    import json

    out = {}
    for name, gen, rc in (
        ("u1M_rho1_f32", lambda: inputs.uniform_box(1 << 20, 1.0, np.float32), 3.3),   # BASELINE config 2
        ("u1M_rho1_f64", lambda: inputs.uniform_box(1 << 20, 1.0, np.float64), 3.3),
        ("u1M_rho05_f32", lambda: inputs.uniform_box(1 << 20, 0.5, np.float32), 3.3),  # BASELINE config 3
        ("u1M_rho05_f64", lambda: inputs.uniform_box(1 << 20, 0.5, np.float64), 3.3),
        # cubic boxes of 2 / 4 / 8 M particles at rho = 1
        ("u2M_rho1_f32", lambda: inputs.uniform_box(2 << 20, 1.0, np.float32), 3.3),
        ("u4M_rho1_f32", lambda: inputs.uniform_box(4 << 20, 1.0, np.float32), 3.3),
        ("u8M_rho1_f32", lambda: inputs.uniform_box(8 << 20, 1.0, np.float32), 3.3),
        # the weak-scaling boxes of bench.py --gpus 2 / 4 / 8: the 1 M cube repeated along z (inputs.weak_scaling_box)
        ("w2x1M_rho1_f32", lambda: inputs.weak_scaling_box(2), 3.3),
        ("w4x1M_rho1_f32", lambda: inputs.weak_scaling_box(4), 3.3),
        ("w8x1M_rho1_f32", lambda: inputs.weak_scaling_box(8), 3.3),
        ("fcc_L50_rho1_f64", lambda: inputs.fcc_box(1.0, 50.0, np.float64), 3.3),       # the README point
        ("fcc_L50_rho05_f64", lambda: inputs.fcc_box(0.5, 50.0, np.float64), 3.3),
        ("fcc_L50_rho1_f32", lambda: inputs.fcc_box(1.0, 50.0, np.float32), 3.3),
    ):
        if only and name not in only:
            continue
        q, box = gen()
        h = po.ref_build(q, rc, box, "fused")[0]
        nop = h.number_of_partners.astype(np.int64)
        out[name] = {
            "n": int(len(q)), "rc": rc, "box": list(box), "npairs": h.npairs, "hash": f"{h.hash():016x}",
            "nop_max": int(nop.max()), "nop_weighted_sum": int((nop * (np.arange(len(nop)) % 1000003)).sum()),
        }
        print(name, out[name], flush=True)
    path = os.path.join(OUT, "known_answers.json")
    if only and os.path.exists(path):  # merge a subset into the stored answers
        old = json.load(open(path))
        old.update(out)
        out = old
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def big_count(only=None):
    """Known answers for the boxes beyond the reference's own limits, from the restatement's count / hash mode
    (oracle/nl_oracle_impl.h nl_oracle_count; pinned to the compiled reference on the 1 M boxes by
    tests/test_oracle.py): BASELINE config 4 (33 554 432 particles, 2.5e9 pairs > INT32_MAX) with the per-slab
    answers of the 2-, 4- and 8-way z-slab decompositions, and config 5 (1 M particles, fp64, rc = 6.6)."""
    import json

    from md_neighbor_list_amd import slab

    path = os.path.join(OUT, "known_answers.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name, gen, rc, worlds in (
        ("u32M_rho1_f32", lambda: inputs.uniform_box(1 << 25, 1.0, np.float32), 3.3, (2, 4, 8)),  # BASELINE config 4
        ("u16M_rho1_f32", lambda: inputs.uniform_box(1 << 24, 1.0, np.float32), 3.3, ()),  # 1.2e9 pairs: in [2^30, 2^31)
        ("u1M_rho1_f64_rc66", lambda: inputs.uniform_box(1 << 20, 1.0, np.float64), 6.6, ()),    # BASELINE config 5
        ("u1M_rho1_f32_rc66", lambda: inputs.uniform_box(1 << 20, 1.0, np.float32), 6.6, ()),
    ):
        if only and name not in only:
            continue
        q, box = gen()
        nop, _, hashes, npairs = po.count(q, rc, box)
        nop64 = nop.astype(np.int64)
        w = np.arange(len(nop64)) % 1000003
        out[name] = {"n": int(len(q)), "rc": rc, "box": list(box), "npairs": npairs, "hash": f"{int(hashes[0]):016x}",
                     "nop_max": int(nop64.max()), "nop_weighted_sum": int((nop64 * w).sum()),
                     "source": "restatement, count/hash mode"}
        mz = int(box[2] / rc)
        cells, _ = po.cells(q, rc, box)
        layer_of = cells // (int(box[0] / rc) * int(box[1] / rc))
        for world in worlds:
            sol = np.zeros(mz, dtype=np.int32)
            for r, (lo, hi) in enumerate(slab.split_layers(mz, world)):
                sol[lo:hi] = r
            _, pairs, hs, np2 = po.count(q, rc, box, sol)
            assert np2 == npairs
            owner = sol[layer_of]
            out[name][f"slabs{world}"] = [
                {"z_lo": lo, "z_hi": hi, "n_rows": int((owner == r).sum()), "npairs": int(pairs[r]), "hash": f"{int(hs[r]):016x}",
                 "nop_weighted_sum": int((nop64 * w)[owner == r].sum())}
                for r, (lo, hi) in enumerate(slab.split_layers(mz, world))]
        print(name, {k: v for k, v in out[name].items() if not k.startswith("slabs")}, flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if "--big-count" in sys.argv:
        k = sys.argv.index("--big-count")
        big_count(set(sys.argv[k + 1].split(",")) if len(sys.argv) > k + 1 else None)
    elif "--big" in sys.argv:  # optionally: --big name,name  (only those entries, merged into the stored file)
        k = sys.argv.index("--big")
        big(set(sys.argv[k + 1].split(",")) if len(sys.argv) > k + 1 else None)
    else:
        main()
