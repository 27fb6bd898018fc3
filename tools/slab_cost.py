#!/usr/bin/env python3
"""What one rank of the 8-GPU weak-scaling run computes, on one GPU and without any exchange: the slab of rank 3 of the
8 M-particle weak-scaling box (SLAB_COST_BOX=cube: the cubic 8 M box) (owned layers + two ghost layers), built as slab.build does (begin + finish) and in one call, against
the plain 1 M-particle build."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs, slab

RC = 3.3
def timed(fn, sync, reps=50):
    for _ in range(5): fn()
    sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

q1, box1 = inputs.uniform_box(1 << 20, 1.0, np.float32)
nl1 = NeighListGPU(RC, *box1, dtype=torch.float32); nl1.Initialize(len(q1))
qd1 = torch.from_numpy(q1).cuda()
print(f"plain build, N = 1 M: {timed(lambda: nl1.MakeNeighList(qd1, len(q1), sync=False), nl1.synchronize):.1f} us", flush=True)

world, rank = 8, 3
q, box = inputs.weak_scaling_box(8) if os.environ.get("SLAB_COST_BOX", "weak") == "weak" else inputs.uniform_box(8 << 20, 1.0, np.float32)
mz = int(box[2] / RC)
z_lo, z_hi = slab.split_layers(mz, world)[rank]
iz = slab.z_layer(torch.from_numpy(q), box, RC).numpy()
own = np.nonzero((iz >= z_lo) & (iz < z_hi))[0]
glo = np.nonzero(iz == z_lo - 1)[0]
ghi = np.nonzero(iz == z_hi)[0]
order = np.concatenate([own, glo, ghi])
qa = torch.from_numpy(q[order]).cuda()
qa[:, 3] = torch.from_numpy(order.astype(np.int32)).cuda().view(torch.float32)
nl = NeighListGPU(RC, *box, dtype=torch.float32); nl.Initialize(len(order))
per = (2.0 / 3.0) * np.pi * RC ** 3
nl.set_capacity(int(len(own) * per * 1.3) + 64 * len(own) + 4096)
print(f"slab of rank {rank}/{world} of the 8 M box: layers [{z_lo},{z_hi}) of {mz}, {len(own)} owned + {len(glo)} + {len(ghi)} ghosts")
one = lambda: nl.MakeNeighListSlab(qa, nl.GID_IN_W, len(own), z_lo, z_hi, sync=False)
def two():
    nl.MakeNeighListSlabBegin(qa, nl.GID_IN_W, len(own), len(glo), z_lo, z_hi)
    nl.MakeNeighListSlabFinish(sync=False)
print(f"  one call        : {timed(one, nl.synchronize):.1f} us   pairs {nl.half_number_of_pairs()}", flush=True)
print(f"  begin + finish  : {timed(two, nl.synchronize):.1f} us   pairs {nl.half_number_of_pairs()}", flush=True)
nl.MakeNeighListSlabBegin(qa, nl.GID_IN_W, len(own), len(glo), z_lo, z_hi)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    nl.MakeNeighListSlabBegin(qa, nl.GID_IN_W, len(own), len(glo), z_lo, z_hi)
torch.cuda.synchronize()
print(f"  begin alone     : {(time.perf_counter() - t0) / 50 * 1e6:.1f} us (what a halo exchange can hide behind)", flush=True)
nl.MakeNeighListSlabFinish(sync=True)
