#!/usr/bin/env python3
"""Soak: many seeded random problems (both sweep variants, the fine-row search forced / allowed / off, both binning
paths, both dtypes, both offset widths, half and full lists, open box and minimum image, uniform and clustered particles
-- dense cells among sparse ones exercise the hand-over lists, the dense-mask pipeline and the fine-row overflow
kernel) against the oracle.
usage: tools/soak_parity.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests.util import canonical_csr  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
bad = 0
paths = {}
for case in range(cases):
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    os.environ["NL_SWEEP_VARIANT"] = os.environ.get("SOAK_VARIANT") or str(rng.choice([1, 3, 3]))
    os.environ["NL_OFFSET_WIDTH"] = str(rng.choice([0, 0, 64]))
    os.environ["NL_BINNING"] = str(rng.integers(0, 2))
    rows = int(rng.choice([-1, -1, 0, 4, 4, 1, 2, 3]))  # -1: the library's own choice
    os.environ.pop("NL_ROWS", None)
    if rows >= 0:
        os.environ["NL_ROWS"] = str(rows)
    rc = float(rng.uniform(0.5, 5.0))
    mesh = rng.integers(3, 14, size=3)
    box = tuple(float(m * rc * rng.uniform(1.0, 1.3)) for m in mesh)
    ncell = int(mesh[0]) * int(mesh[1]) * int(mesh[2])
    n = int(min(120000, max(1, ncell * rng.uniform(*((10.0, 45.0) if os.environ.get("SOAK_VARIANT") else (0.05, 60.0))))))
    q = np.zeros((n, 4), dtype=dtype)
    q[:, :3] = rng.uniform(0.0, 1.0, size=(n, 3)) * np.array(box)
    if rng.random() < 0.25:  # a cluster: a third of the particles in a few cells
        k = n // 3
        centre = rng.uniform(0.2, 0.8, size=3) * np.array(box)
        q[:k, :3] = centre + rng.uniform(-1.0, 1.0, size=(k, 3)) * rc * rng.uniform(0.6, 2.0)
        q[:k, :3] = np.clip(q[:k, :3], 0.0, None)
    q[:, :3] = np.minimum(q[:, :3], np.nextafter(np.array(box, dtype=dtype), dtype(0)))
    full = rng.random() < 0.3
    pbc = rng.random() < 0.3
    if pbc:  # minimum-image mode: also particles outside the box
        q[:, :3] = np.clip(rng.uniform(-0.3, 1.3, size=(n, 3)) * np.array(box), -0.9 * np.array(box), 1.9 * np.array(box))
    ref = (po.build_pbc(q, rc, box) if pbc else po.build(q, rc, box)).canonical()
    nl = NeighListGPU(rc, *box, dtype=torch.float32 if dtype == np.float32 else torch.float64, full_list=full,
                      minimum_image=pbc)
    nl.Initialize(n)
    qd = torch.from_numpy(q).cuda()
    for rep in range(2):  # twice on the same handle
        nl.MakeNeighList(qd, n, sync=(rep == 0))
        nl.synchronize()
        if full and pbc:  # minimum image: row i in the frame of particle i (not the symmetrised half list)
            kp, lst, _ = (t.cpu().numpy() for t in nl.full_csr())
            want = po.build_pbc_full(q, rc, box)
            ok = np.array_equal(kp.astype(np.int64), want.key_pointer) and np.array_equal(canonical_csr(kp, lst), want.sorted_list)
        elif full:
            kp, lst, _ = (t.cpu().numpy() for t in nl.full_csr())
            ok = int(kp[-1]) == 2 * ref.npairs
            if ok:
                rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(kp))
                half = lst.astype(np.int64) > rows
                hk = np.concatenate([[0], np.cumsum(np.bincount(rows[half], minlength=n))])
                ok = np.array_equal(canonical_csr(hk, lst[half]), ref.sorted_list)
        else:
            kp, sl = nl.key_pointer64().cpu().numpy(), nl.sorted_list().cpu().numpy()
            ok = int(kp[-1]) == ref.npairs and np.array_equal(canonical_csr(kp, sl), ref.sorted_list)
            ok = ok and nl.list_checksum() == (ref.hash(), ref.npairs)
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} rep {rep}: n={n} box={box} rc={rc} dtype={dtype.__name__} full={full} "
                  f"pbc={pbc} variant={os.environ['NL_SWEEP_VARIANT']} width={os.environ['NL_OFFSET_WIDTH']} binning={os.environ['NL_BINNING']} rows={rows} info={nl.build_info()}", flush=True)
    info = nl.build_info()
    key = f"rows{info['fine_rows']}" if info["fine_rows"] else ("masks" + ("x" if info["mask_rows"] > 1 else "") if info["masks"] else "two sweeps")
    paths[key] = paths.get(key, 0) + 1
    if case % 50 == 49:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"soak done: {cases} cases, {bad} mismatches; search paths taken: {dict(sorted(paths.items()))}")
sys.exit(1 if bad else 0)
