#!/usr/bin/env python3
"""The placement kernel takes 210 or 232 us at cfg 2 depending on the process.  Does it depend on the allocations
(new handle in the same process) or on the process?"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs  # noqa: E402

q, box = inputs.uniform_box(1 << 20, 1.0, np.float32)
qd = torch.from_numpy(q).cuda()
keep = []
for trial in range(8):
    nl = NeighListGPU(3.3, *box, dtype=torch.float32)
    nl.Initialize(len(q))
    st = nl.profile_stages(qd, reps=10)
    kp = nl.key_pointer()
    print(f"handle {trial}: fill {st['fill'] * 1e3:.1f} us count {st['count'] * 1e3:.1f} us  list@{nl.sorted_list().data_ptr():#x} kp@{kp.data_ptr():#x}", flush=True)
    if trial % 2 == 0:
        keep.append(nl)  # keep some handles alive so that later ones land at other addresses
    pad = torch.empty((trial + 1) * 1234567, dtype=torch.uint8, device="cuda")
    keep.append(pad)

print("-- one handle, the list re-allocated at other addresses (nl_set_capacity)")
nl = NeighListGPU(3.3, *box, dtype=torch.float32)
nl.Initialize(len(q))
for trial in range(10):
    nl.set_capacity(100_000_000 + trial * 3_000_017)
    st = nl.profile_stages(qd, reps=10)
    a = nl.sorted_list().data_ptr()
    print(f"capacity trial {trial}: fill {st['fill'] * 1e3:.1f} us count {st['count'] * 1e3:.1f} us  list@{a:#x} (list mod 1GiB = {a % (1 << 30):#x})", flush=True)
    keep.append(torch.empty((trial + 1) * 7654321, dtype=torch.uint8, device="cuda"))
