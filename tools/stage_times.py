#!/usr/bin/env python3
"""Per-stage device times (HIP events) of the build for the BASELINE single-GPU configs. usage: stage_times.py [cfg...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs, _lib  # noqa: E402

if os.environ.get("NL_LIB"):  # development only: A/B another (possibly older) build of the library in the same session
    import ctypes

    _lib.LIB_PATH = os.path.abspath(os.environ["NL_LIB"])
    _old = ctypes.CDLL(_lib.LIB_PATH)
    for _name in list(_lib.PROTOTYPES):
        if not hasattr(_old, _name):
            del _lib.PROTOTYPES[_name]

CFGS = {
    "cfg2": (1 << 20, 1.0, 3.3, np.float32),
    "cfg3": (1 << 20, 0.5, 3.3, np.float32),
    "cfg5": (1 << 20, 1.0, 6.6, np.float64),
    "cfg2_f64": (1 << 20, 1.0, 3.3, np.float64),
    "fcc": None,
}

_pre = None
if os.environ.get("PREALLOC_MB"):  # experiment: what the allocations that come before the handle do to the timings
    _pre = torch.empty(int(os.environ["PREALLOC_MB"]) << 20, dtype=torch.uint8, device="cuda")

for name in sys.argv[1:] or ["cfg2", "cfg3"]:
    if name == "fcc":
        q, box = inputs.fcc_box(1.0, 50.0, np.float64)
        rc = 3.3
    else:
        n, rho, rc, dt = CFGS[name]
        q, box = inputs.uniform_box(n, rho, dt)
    nl = NeighListGPU(rc, *box, dtype=torch.float32 if q.dtype == np.float32 else torch.float64)
    nl.Initialize(len(q))
    qd = torch.from_numpy(q).cuda()
    st = nl.profile_stages(qd, reps=20)
    p = nl.half_number_of_pairs()
    print(f"{name:9s} N={len(q)} P={p} " + " ".join(f"{k}={v*1e3:.1f}us" for k, v in st.items()), flush=True)
    flags = int(os.environ.get("NL_DEBUG_FLAGS", "0"))
    if flags & 12:
        buf = np.zeros(64 + 4 * 4096, dtype=np.uint64)
        nl._lib.nl_debug_read(nl._h, buf.ctypes.data, len(buf), 1)
        if flags & 4:
            v = [int(x) for x in buf[:5]]
            tot = sum(v[:4]) or 1
            print("   wave cycles: barrier %.1f%%  lookahead %.1f%%  search %.1f%%  other %.1f%%  (waves*launches=%d, per wave-launch %.0f cycles)"
                  % (100 * v[0] / tot, 100 * v[1] / tot, 100 * v[2] / tot, 100 * v[3] / tot, v[4], tot / max(v[4], 1)))
        if flags & 8:  # per-workgroup records of the LAST launch (the fill sweep)
            rec = buf[64:].reshape(-1, 4)
            rec = rec[rec[:, 1] > 0]
            t0 = rec[:, 0].min()
            start = (rec[:, 0] - t0).astype(np.float64) / 100.0  # us
            end = (rec[:, 1] - t0).astype(np.float64) / 100.0
            xcc = (rec[:, 2] >> np.uint64(32)).astype(np.int64) & 0xF
            hwid = (rec[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
            cu = (hwid >> 8) & 0xF
            se = (hwid >> 13) & 0x7
            print(f"   {len(rec)} workgroups: start us min/med/max {start.min():.1f}/{np.median(start):.1f}/{start.max():.1f}  "
                  f"end us min/med/max {end.min():.1f}/{np.median(end):.1f}/{end.max():.1f}  dur med {np.median(end-start):.1f}")
            dur = end - start
            print("   median duration by XCC:", " ".join(f"{x}:{np.median(dur[xcc == x]):.0f}" for x in range(8)))
            bidx = np.arange(len(rec))
            j = bidx >> 3
            print("   median duration by position of the cell run inside its XCD slab (16 bins):",
                  " ".join(f"{np.median(dur[(j * 16 // (j.max() + 1)) == b]):.0f}" for b in range(16)))
            print("   median duration by CU id:", " ".join(f"{np.median(dur[cu == c]):.0f}" for c in range(16)))
            print("   median duration by SE id:", " ".join(f"{np.median(dur[se == c]):.0f}" for c in range(8) if (se == c).any()))
            ncd = rec[:, 3].astype(np.int64)
            print(f"   cells per workgroup min/med/max {ncd.min()}/{int(np.median(ncd))}/{ncd.max()}; workgroups with 0 cells: {(ncd == 0).sum()}")
            late = start > 5.0
            print(f"   workgroups starting later than 5 us: {late.sum()}")
            key = xcc * 1000 + se * 16 + cu
            uniq, cnt = np.unique(key[~late], return_counts=True)
            print(f"   distinct (xcc,se,cu) among the early ones: {len(uniq)}; workgroups per CU histogram: {np.bincount(cnt)}")
