#!/usr/bin/env python3
"""All stage times of the default build on alternating fresh handles (NL_LIB selects another build of the library)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs, _lib
if os.environ.get("NL_LIB"):
    import ctypes
    _lib.LIB_PATH = os.path.abspath(os.environ["NL_LIB"])
    _old = ctypes.CDLL(_lib.LIB_PATH)
    for _name in list(_lib.PROTOTYPES):  # an older build lacks the newer entry points
        if not hasattr(_old, _name):
            del _lib.PROTOTYPES[_name]
q, box = inputs.uniform_box(1 << 20, 1.0, np.float32)
qd = torch.from_numpy(q).cuda()
for trial in range(4):
    nl = NeighListGPU(3.3, *box, dtype=torch.float32)
    nl.Initialize(len(q))
    nl.profile_stages(qd, reps=20)
    st = nl.profile_stages(qd, reps=40)
    print(" ".join(f"{k}={v * 1e3:.1f}" for k, v in st.items()), flush=True)
    del nl
