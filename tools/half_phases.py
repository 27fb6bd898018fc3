#!/usr/bin/env python3
"""Per-phase wave cycles of k_sweep_half (NL_DEBUG_FLAGS=4): where a workgroup's time goes.  usage: half_phases.py [cfg2|cfg3]"""
import os
import sys

os.environ["NL_DEBUG_FLAGS"] = "4"
os.environ.setdefault("NL_SWEEP_VARIANT", "6")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs  # noqa: E402

for name in sys.argv[1:] or ["cfg2"]:
    rho = 0.5 if name == "cfg3" else 1.0
    q, box = inputs.uniform_box(1 << 20, rho, np.float32)
    nl = NeighListGPU(3.3, *box, dtype=torch.float32)
    nl.Initialize(len(q))
    qd = torch.from_numpy(q).cuda()
    nl.MakeNeighList(qd, len(q))
    buf = np.zeros(64, dtype=np.uint64)
    nl._lib.nl_debug_read(nl._h, buf.ctypes.data, len(buf), 1)  # reset
    reps = 5
    for _ in range(reps):
        nl.MakeNeighList(qd, len(q))
    nl._lib.nl_debug_read(nl._h, buf.ctypes.data, len(buf), 1)
    v = buf[8:12].astype(np.float64)
    nw = float(buf[15]) or 1.0
    names = ["table", "staging", "barrier", "search + barrier + R write-out"]
    print(f"{name}: cycles per wave and cell: " + "  ".join(f"{n} {x / nw:.0f}" for n, x in zip(names, v)) + f"  total {v.sum() / nw:.0f}  ({int(nw)} wave-cells)")
