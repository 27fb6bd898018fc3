#!/usr/bin/env python3
"""Where a wave of the fine-row COUNT sweep / expansion spends its cycles (library built with -DNL_STAMP=1 or
-DNL_STAMP_FILL=1; NL_HIP_LIB selects it).  usage: NL_HIP_LIB=build/ab/stamp.so python tools/rows_phases.py count|fill [cfg2|cfg3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "count"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
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
names = {"count": ["cell + window table + scan", "DMA issue", "wait + barrier", "group set-up", "rows_group (readlanes, tiles, words)", "stores"],
         "fill": ["cell + window table", "row loads + id DMA issue", "wait + barrier", "words, popcounts, scans", "bit loops", "read back + stores"]}[which]
tot = sum(v[:6])
waves = max(v[9], 1)
print(f"{which} {cfg} {nl.build_info()}: {waves // reps} waves per build, {tot / waves:.0f} cycles per wave" +
      (f", {v[8] / reps:.0f} wave-tests per build, {v[4] / max(v[8], 1):.1f} cycles per wave-test in rows_group" if which == "count" else ""))
for n, c in zip(names, v[:6]):
    print(f"  {n:40s} {100 * c / tot:5.1f} %   {c / waves:8.0f} cycles per wave")
