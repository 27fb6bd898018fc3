#!/usr/bin/env python3
"""Does the build get faster as the GPU stays busy (clock ramp), on ONE handle?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs
q, box = inputs.uniform_box(1 << 20, 1.0, np.float32)
qd = torch.from_numpy(q).cuda()
nl = NeighListGPU(3.3, *box, dtype=torch.float32)
nl.Initialize(len(q))
t0 = time.time()
for trial in range(14):
    st = nl.profile_stages(qd, reps=30)
    print(f"t={time.time() - t0:5.2f}s  count {st['count'] * 1e3:.1f}  fill {st['fill'] * 1e3:.1f}  total {st['total'] * 1e3:.1f} us", flush=True)
print("-- second handle in the same process")
nl2 = NeighListGPU(3.3, *box, dtype=torch.float32)
nl2.Initialize(len(q))
for trial in range(4):
    st = nl2.profile_stages(qd, reps=30)
    print(f"t={time.time() - t0:5.2f}s  count {st['count'] * 1e3:.1f}  fill {st['fill'] * 1e3:.1f}  total {st['total'] * 1e3:.1f} us", flush=True)
print("-- first handle again")
for trial in range(3):
    st = nl.profile_stages(qd, reps=30)
    print(f"t={time.time() - t0:5.2f}s  count {st['count'] * 1e3:.1f}  fill {st['fill'] * 1e3:.1f}  total {st['total'] * 1e3:.1f} us", flush=True)
