#!/usr/bin/env python3
"""ms per build and Gpairs/s at N = 1 M, rc = 3.3 over a range of densities: the 27-cell path
(NL_ROWS=0), the fine-row search (NL_ROWS=4: wherever it qualifies, RowsCfg by density) and the default (the fine-row
search from 40.3 particles per cell on) side by side, with build_info per point.
usage: tools/density_sweep.py [rho ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs  # noqa: E402

rhos = [float(x) for x in sys.argv[1:]] or [0.5, 0.65, 0.8, 0.9, 1.0, 1.05, 1.1, 1.2, 1.35, 1.5, 1.75, 2.0]
n = 1 << 20
print(f"N = {n}, rc = 3.3, fp32, half list; median of 7 batches of 20 builds (HIP events)")
for rho in rhos:
    q, box = inputs.uniform_box(n, rho, np.float32)
    qd = torch.from_numpy(q).cuda()
    line = [f"rho {rho:4.2f}  mesh {int(box[0] / 3.3):3d}  per cell {n / int(box[0] / 3.3) ** 3:5.1f}"]
    ref = None
    for rows in ("0", "4", "default"):
        os.environ.pop("NL_ROWS", None)
        if rows != "default":
            os.environ["NL_ROWS"] = rows
        nl = NeighListGPU(3.3, *box, dtype=torch.float32)
        nl.Initialize(n)
        nl.MakeNeighList(qd, n)  # synchronous: grows the list if the estimate was short
        pairs = nl.half_number_of_pairs()
        chk = nl.list_checksum()
        if ref is None:
            ref = chk
        assert chk == ref, (rho, rows, chk, ref)
        for _ in range(3):
            nl.MakeNeighList(qd, n, sync=False)
        nl.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                nl.MakeNeighList(qd, n, sync=False)
            e1.record()
            nl.synchronize()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        ms = float(np.median(ts))
        info = nl.build_info()
        tag = f"rows{info['fine_rows']}" if info["fine_rows"] else ("masks" + (f"x{info['mask_rows']}" if info["mask_rows"] > 1 else "") if info["masks"] else "two sweeps")
        line.append(f"NL_ROWS={rows:>7s}: {ms:6.3f} ms {pairs / ms / 1e6:6.1f} Gpairs/s [{tag}]")
        del nl
    print(" | ".join(line) + f" | pairs {pairs}", flush=True)
