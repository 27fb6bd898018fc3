#!/usr/bin/env python3
"""tools/debug_parity.py -- HIP build vs oracle on a few inputs; prints the first differing rows with distances."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import inputs, _lib  # noqa: E402

if os.environ.get("NL_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["NL_LIB"])
from oracle import pyoracle as po  # noqa: E402
from tests.util import canonical_csr, gpu_build  # noqa: E402


def check(name, q, rc, box):
    ref = po.build(q, rc, box).canonical()
    _, nop, kp, sl = gpu_build(q, rc, box)
    sl = canonical_csr(kp, sl)
    bad = np.nonzero(nop != ref.number_of_partners)[0]
    print(f"{name}: N={len(q)} pairs hip={int(kp[-1])} ref={ref.npairs} rows differing={len(bad)}", flush=True)
    ms = [b / int(b / rc) for b in box]
    for i in bad[:4]:
        mine = set(sl[kp[i]:kp[i + 1]].tolist())
        theirs = set(ref.sorted_list[ref.key_pointer[i]:ref.key_pointer[i + 1]].tolist())
        print(f"  row {i} pos {q[i, :3]} cell {[int(q[i, d] / ms[d]) for d in range(3)]} hip {len(mine)} ref {len(theirs)}")
        for j in sorted(mine - theirs)[:8]:
            d = q[j, :3].astype(np.float64) - q[i, :3].astype(np.float64)
            print(f"     extra j={j} pos {q[j, :3]} cell {[int(q[j, d_] / ms[d_]) for d_ in range(3)]} r2={np.dot(d, d):.6f} (rc2={rc * rc:.6f})")
        for j in sorted(theirs - mine)[:8]:
            d = q[j, :3].astype(np.float64) - q[i, :3].astype(np.float64)
            print(f"     missing j={j} r2={np.dot(d, d):.6f}")


rc = 3.3
q, box = inputs.fcc_box(0.5, 16.0, np.float32)
check("fcc rho0.5 L16", q, rc, box)
q, box = inputs.uniform_box(6000, dtype=np.float32, seed=3, box=(20.0, 20.0, 20.0))
check("uniform rho0.75 L20", q, rc, box)
q, box = inputs.uniform_box(20000, dtype=np.float32, seed=4, box=(27.0, 27.0, 27.0))
check("uniform rho1 L27", q, rc, box)

if int(os.environ.get("NL_DEBUG_FLAGS", "0")) & (64 | 4):
    import torch
    from md_neighbor_list_amd import NeighListGPU
    nl = NeighListGPU(rc, *box, dtype=torch.float32)
    nl.Initialize(len(q))
    buf = np.zeros(8, dtype=np.uint64)
    nl._lib.nl_debug_read(nl._h, buf.ctypes.data, len(buf), 1)  # reset
    nl.MakeNeighList(torch.from_numpy(q).cuda(), len(q))
    nl._lib.nl_debug_read(nl._h, buf.ctypes.data, len(buf), 1)
    err = np.array([buf[1]], dtype=np.uint64).astype(np.uint32).view(np.float32)[0]
    print(f"mfma diagnostics: uncertain re-tests {buf[0]}  max |acc - (r2-rc2)| (r2 < 4 rc2) {err:.3e}  decisive-but-wrong {buf[2]}  elements {buf[3]}"
          f"  delta {nl.build_info()}")
