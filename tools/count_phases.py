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
if os.environ.get("NL_PIPE", "4") != "0":  # k_sweep_pipe_f32: per wave over its whole run of cells; v[8] = cells walked
    names = ["loop overhead", "wait for the stream + barrier", "deferred stores", "tables (finish next, issue next + 1)", "DMA issue", "search"]
tot = sum(v[:6])
waves = v[9]
print(f"{cfg}: {waves // reps} waves per build, {tot / waves:.0f} cycles per wave, {v[8] / reps:.0f} wave-tests per build")
for n, c in zip(names, v[:6]):
    print(f"  {n:32s} {100 * c / tot:5.1f} %   {c / waves:8.0f} cycles per wave")
if os.environ.get("NL_PIPE", "4") != "0":
    print(f"  {v[8] / waves:.1f} cells per wave, {tot / v[8]:.0f} cycles per cell")
    slots = buf[64:].reshape(1024, 16).astype(np.float64)
    used = slots[:, 9] > 0
    per_wave = slots[used, :6].sum(axis=1) / slots[used, 9]  # cycles per wave of that workgroup slot (per build)
    print(f"  cycles per wave by workgroup: min {per_wave.min():.0f}  median {np.median(per_wave):.0f}  p90 {np.percentile(per_wave, 90):.0f}  max {per_wave.max():.0f}")
    for ph, n in enumerate(names):
        x = slots[used, ph] / slots[used, 9]
        print(f"    {n:40s} min {x.min():8.0f} median {np.median(x):8.0f} max {x.max():8.0f}")
    raw = buf[64:].reshape(1024, 16)[used]
    t0 = raw[:, 10].astype(np.float64) / 100.0
    t1 = raw[:, 11].astype(np.float64) / 100.0
    base = t0.min()
    start, end = t0 - base, t1 - base
    hw = (raw[:, 12] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = (raw[:, 12] >> np.uint64(32)).astype(np.int64) & 0xF
    cu, se = (hw >> 8) & 0xF, (hw >> 13) & 0x7
    print(f"  last launch: start us min/median/max {start.min():.1f}/{np.median(start):.1f}/{start.max():.1f}   end us min/median/max {end.min():.1f}/{np.median(end):.1f}/{end.max():.1f}")
    dur = end - start
    print(f"  duration us min/median/p90/max {dur.min():.1f}/{np.median(dur):.1f}/{np.percentile(dur, 90):.1f}/{dur.max():.1f}")
    print("  median duration by XCC:", " ".join(f"{x}:{np.median(dur[xcc == x]):.0f}(n={int((xcc == x).sum())})" for x in range(8)))
    key = xcc * 10000 + se * 100 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"  distinct (xcc, se, cu): {len(uniq)}; workgroups per CU: " + " ".join(f"{c}:{int((cnt == c).sum())}" for c in sorted(set(cnt))))
    for c in sorted(set(cnt)):
        sel = np.isin(key, uniq[cnt == c])
        print(f"    CUs with {c} workgroups: median duration {np.median(dur[sel]):.1f} us, max {dur[sel].max():.1f}")
    st = nl.profile_stages(qd, reps=20)
    print("  stage times:", " ".join(f"{k}={v_ * 1e3:.1f}us" for k, v_ in st.items()))
else:
    print(f"  search_group: {v[4] / v[8]:.1f} cycles per wave-test per wave")
