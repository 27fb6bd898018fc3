#!/usr/bin/env python3
"""BASELINE config 4 (33 554 432 particles, rho = 1.0, rc = 3.3, fp32) cut into 2, 4 and 8 z-slabs, every slab built on
ONE GPU one after another and timed (HIP events; median of 5 batches of 3 builds): what each rank of bench.py --gpus N
computes between two halo exchanges, and the projection t(N = 1) / max over ranks t(slab).  No exchange is timed here:
the ghost layers are in place.  usage: tools/cfg4_slab_costs.py [worlds ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs, slab  # noqa: E402

RC = 3.3
worlds = [int(x) for x in sys.argv[1:]] or [2, 4, 8]


def timed(fn, sync, batches=5, reps=3):
    fn()
    sync()
    ts = []
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        sync()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


n = 1 << 25
q, box = inputs.uniform_box(n, 1.0, np.float32)
mz = int(box[2] / RC)
per = (2.0 / 3.0) * np.pi * RC ** 3
iz = slab.z_layer(torch.from_numpy(q), box, RC).numpy()
print(f"config 4: N = {n}, box {box[0]:.2f}^3, mesh {mz}^3, {n / mz:.0f} particles per z layer", flush=True)
nl = NeighListGPU(RC, *box, dtype=torch.float32)
nl.Initialize(n)
qd = torch.from_numpy(q).cuda()
nl.MakeNeighList(qd, n)
t1 = timed(lambda: nl.MakeNeighList(qd, n, sync=False), nl.synchronize)
p1 = nl.half_number_of_pairs()
print(f"N = 1 (whole box on one device, {nl.build_info()['offset_bits']}-bit offsets): {t1:.3f} ms, {p1} pairs, {p1 / t1 / 1e6:.1f} Gpairs/s", flush=True)
del nl, qd
torch.cuda.empty_cache()
order_z = np.argsort(iz, kind="stable")
starts = np.searchsorted(iz[order_z], np.arange(mz + 1))
for world in worlds:
    worst, total_pairs = 0.0, 0
    for rank, (z_lo, z_hi) in enumerate(slab.split_layers(mz, world)):
        own = order_z[starts[z_lo]:starts[z_hi]]
        glo = order_z[starts[(z_lo - 1) % mz]:starts[(z_lo - 1) % mz + 1]]
        ghi = order_z[starts[z_hi % mz]:starts[z_hi % mz + 1]]
        idx = np.concatenate([own, glo, ghi])
        qa = torch.from_numpy(q[idx]).cuda()
        qa[:, 3] = torch.from_numpy(idx.astype(np.int32)).cuda().view(torch.float32)
        nl = NeighListGPU(RC, *box, dtype=torch.float32)
        nl.Initialize(len(idx))
        nl.set_capacity(int(len(own) * per * 1.3) + 64 * len(own) + 4096)
        one = lambda: nl.MakeNeighListSlab(qa, nl.GID_IN_W, len(own), z_lo, z_hi, sync=False)  # noqa: E731

        def two():
            nl.MakeNeighListSlabBegin(qa, nl.GID_IN_W, len(own), len(glo), z_lo, z_hi)
            nl.MakeNeighListSlabFinish(sync=False)

        def begin_only():
            nl.MakeNeighListSlabBegin(qa, nl.GID_IN_W, len(own), len(glo), z_lo, z_hi)

        t_one = timed(one, nl.synchronize)
        pairs = nl.half_number_of_pairs()
        t_two = timed(two, nl.synchronize)
        begin_only()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            begin_only()
        e1.record()
        torch.cuda.synchronize()
        t_begin = e0.elapsed_time(e1) / 3
        nl.MakeNeighListSlabFinish(sync=True)
        worst = max(worst, t_two)
        total_pairs += pairs
        print(f"  world {world} rank {rank}: layers [{z_lo:3d},{z_hi:3d})  {len(own):8d} owned + {len(glo):6d} + {len(ghi):6d} ghosts ({16e-6 * (len(glo) + len(ghi)):.1f} MB in, "
              f"the same out)  one call {t_one:6.3f} ms  begin + finish {t_two:6.3f} ms  (begin {t_begin:5.3f})  {pairs} pairs [{nl.build_info()}]", flush=True)
        del nl, qa
        torch.cuda.empty_cache()
    assert total_pairs == p1, (total_pairs, p1)
    print(f"world {world}: slowest slab {worst:.3f} ms -> projection {t1 / worst:.2f} x of {world} (exchange not included; the union of the slabs' lists "
          f"has the {p1} pairs of the whole box)", flush=True)
