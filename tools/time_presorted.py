#!/usr/bin/env python3
"""How much a cell-ordered input helps (the reference's SORT_FREQ idea, neighlist_gpu.hpp:72): build once, permute the
particles into the build's cell order (nl_get_sorted), build again."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs  # noqa: E402

q, box = inputs.uniform_box(1 << 20, 1.0, np.float32)
nl = NeighListGPU(3.3, *box, dtype=torch.float32)
nl.Initialize(len(q))
qd = torch.from_numpy(q).cuda()
st = nl.profile_stages(qd, reps=20)
print("random order :", " ".join(f"{k}={v * 1e3:.1f}us" for k, v in st.items()), flush=True)
_, sorted_row = nl.sorted_state()
qs = qd[sorted_row.long()].contiguous()
st = nl.profile_stages(qs, reps=20)
print("cell order   :", " ".join(f"{k}={v * 1e3:.1f}us" for k, v in st.items()), "pairs", nl.half_number_of_pairs(), flush=True)
