#!/usr/bin/env python3
"""Times nl_get_full_transposed (the reference GPU class's output format, derived from the half CSR) at cfg 2."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs  # noqa: E402

q, box = inputs.uniform_box(1 << 20, 1.0, np.float32)
qd = torch.from_numpy(q).cuda()
for full in (False, True):
    nl = NeighListGPU(3.3, *box, dtype=torch.float32, full_list=full)
    nl.Initialize(len(q))
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nl.MakeNeighList(qd, len(q), sync=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lst = nl.neigh_list()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{'full' if full else 'half'}-list build, rep {rep}: build {1e3 * (t1 - t0):.3f} ms, transposed list "
              f"{'converted' if full else 'derived'} in {1e3 * (t2 - t1):.3f} ms; shape {tuple(lst.shape)}", flush=True)
    if full:
        print("stages (full list):", {k: round(v * 1e3, 1) for k, v in nl.profile_stages(qd, reps=10).items()})
