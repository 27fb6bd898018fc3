#!/usr/bin/env python3
"""A/B of builds on alternating fresh handles in one process (placement of the buffers moves kernels by +-10 %).
FILL_AB_ENV="NAME=v1,v2,..." sweeps one environment variable read at handle creation (default NL_DEBUG_FLAGS=0)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs, _lib
if os.environ.get("NL_LIB"):  # A/B another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["NL_LIB"])
q, box = inputs.uniform_box(int(os.environ.get('FILL_AB_N', 1 << 20)), float(os.environ.get('FILL_AB_RHO', '1.0')), np.float32)
qd = torch.from_numpy(q).cuda()
name, vals = os.environ.get("FILL_AB_ENV", "NL_DEBUG_FLAGS=0").split("=")
for full in (False, True)[:int(os.environ.get('FILL_AB_KINDS', '2'))]:
    for trial in range(int(os.environ.get("FILL_AB_TRIALS", "3"))):
        for v in vals.split(","):
            os.environ[name] = v
            nl = NeighListGPU(3.3, *box, dtype=torch.float32, full_list=full)
            nl.Initialize(len(q))
            nl.profile_stages(qd, reps=20)
            st = nl.profile_stages(qd, reps=40)
            print(f"full={int(full)} {name}={v:>5}  count {st['count'] * 1e3:6.1f}  fill {st['fill'] * 1e3:6.1f}  total {st['total'] * 1e3:6.1f} us", flush=True)
            del nl
