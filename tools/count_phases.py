#!/usr/bin/env python3
"""Where a wave of the COUNT sweep spends its cycles (library built with -DNL_STAMP=1; NL_HIP_LIB selects it).
usage: NL_HIP_LIB=build/ab/stamp.so python tools/count_phases.py [cfg2|cfg3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
rho = {"cfg2": 1.0, "cfg3": 0.5}[cfg]
q, box = inputs.uniform_box(1 << 20, rho, np.float32)
qd = torch.from_numpy(q).cuda()
nl = NeighListGPU(3.3, *box, dtype=torch.float32)
nl.Initialize(len(q))
for _ in range(3):
    nl.MakeNeighList(qd, len(q))
nl.synchronize()
buf = np.zeros(64 + 4 * 4096, dtype=np.uint64)
nl._lib.nl_debug_read(nl._h, buf.ctypes.data, len(buf), 1)  # reset
reps = 10
for _ in range(reps):
    nl.MakeNeighList(qd, len(q))
nl.synchronize()
nl._lib.nl_debug_read(nl._h, buf.ctypes.data, len(buf), 1)
v = [int(x) for x in buf[64:].reshape(1024, 16).sum(axis=0)[:10]]
names = ["setup (cell + segment table)", "staging (loads -> LDS)", "barrier", "group prologue", "search_group", "count store"]
tot = sum(v[:6])
waves = v[9]
print(f"{cfg}: {waves // reps} waves per build, {tot / waves:.0f} cycles per wave, {v[8] / reps:.0f} wave-tests per build")
for n, c in zip(names, v[:6]):
    print(f"  {n:32s} {100 * c / tot:5.1f} %   {c / waves:8.0f} cycles per wave")
print(f"  search_group: {v[4] / v[8]:.1f} cycles per wave-test per wave")
