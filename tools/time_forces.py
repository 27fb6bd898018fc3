#!/usr/bin/env python3
"""Which list kind serves a force loop better?  cfg 2: build + nl_lj_forces for the half list
(pair once + atomics) and the full list (gather, no atomics)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs  # noqa: E402

q, box = inputs.uniform_box(1 << 20, 1.0, np.float32)
qd0 = torch.from_numpy(q).cuda()
nl0 = NeighListGPU(3.3, *box, dtype=torch.float32)
nl0.Initialize(len(q))
nl0.MakeNeighList(qd0, len(q))
qs = qd0[nl0.sorted_state()[1].long()].contiguous()  # the same particles in cell order (SORT_FREQ, section 8 f2)
torch.cuda.synchronize()
for order, qd, full in (("random", qd0, False), ("random", qd0, True), ("cell", qs, False), ("cell", qs, True)):
    nl = NeighListGPU(3.3, *box, dtype=torch.float32, full_list=full)
    nl.Initialize(len(q))
    nl.MakeNeighList(qd, len(q))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tb, tf = [], []
    for rep in range(6):
        ev[0].record()
        nl.MakeNeighList(qd, len(q), sync=False)
        ev[1].record()
        f = nl.lj_forces(qd, 1.0, 1.0, rc_force=3.0)
        ev[2].record()
        torch.cuda.synchronize()
        tb.append(ev[0].elapsed_time(ev[1])), tf.append(ev[1].elapsed_time(ev[2]))
    print(f"{order:6s} order, {'full' if full else 'half'} list: build {min(tb[1:]):.3f} ms, forces {min(tf[1:]):.3f} ms, "
          f"sum {min(tb[1:]) + min(tf[1:]):.3f} ms; |f| checksum {float(f[:, :3].abs().sum()):.6e}", flush=True)
