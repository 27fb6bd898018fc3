#!/usr/bin/env python3
"""An MD loop around the list: velocity Verlet for a Lennard-Jones droplet with
  * a Verlet list built with cut-off rc + skin, reused until some particle has moved more than skin / 2
    (max displacement since the last build), then rebuilt;
  * every SORT_FREQ rebuilds (the reference declares SORT_FREQ = 50 and never uses it, neighlist_gpu.hpp:72) the
    particle arrays are permuted into the build's cell order (nl_resort), which speeds up both the next builds
    and the force gathers (profiles/r01_force_consumer_timing.txt);
  * forces from nl_lj_forces on the full list (one gather per row, no atomics).
The list has no minimum image (neither has the reference): the droplet sits in the middle of an open box.

usage: tools/md_loop.py [--cells 12] [--steps 400] [--dtype f64]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU  # noqa: E402

SORT_FREQ = 50


def fcc_droplet(cells, a, box, dtype, seed=1):
    base = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    g = np.stack(np.meshgrid(*[np.arange(cells)] * 3, indexing="ij"), -1).reshape(-1, 1, 3)
    pos = ((g + base).reshape(-1, 3) * a).astype(np.float64)
    pos += 0.5 * (box - cells * a)
    rng = np.random.default_rng(seed)
    vel = rng.normal(0.0, 0.3, size=pos.shape)
    vel -= vel.mean(axis=0)
    q = np.zeros((len(pos), 4), dtype=dtype)
    q[:, :3] = pos
    return q, vel.astype(dtype)


class Simulation:
    def __init__(self, q, v, box, rc=2.5, skin=0.4, dt=0.004, device="cuda"):
        self.tdt = torch.float32 if q.dtype == np.float32 else torch.float64
        self.q = torch.from_numpy(q).to(device)
        self.v = torch.from_numpy(v).to(device)
        self.ids = torch.arange(len(q), device=device)  # original identity of every slot (changes when re-sorted)
        self.rc, self.skin, self.dt, self.box = rc, skin, dt, box
        self.nl = NeighListGPU(rc + skin, box, box, box, dtype=self.tdt, full_list=True)
        self.nl.Initialize(len(q))
        self.builds = self.sorts = 0
        self.q_built = None
        self.rebuild()
        self.f = self.nl.lj_forces(self.q, 1.0, 1.0, rc_force=self.rc)

    def rebuild(self):
        if self.builds and self.builds % SORT_FREQ == 0:
            # re-sort: the previous build's cell order becomes the storage order (nl_resort, in place)
            self.nl.resort(self.q, self.v, self.ids)
            self.sorts += 1
        self.nl.MakeNeighList(self.q, len(self.q))
        self.q_built = self.q.clone()
        self.builds += 1

    def step(self):
        dt = self.dt
        self.v += 0.5 * dt * self.f[:, :3]
        self.q[:, :3] += dt * self.v
        moved = (self.q[:, :3] - self.q_built[:, :3]).square().sum(dim=1).max()
        if float(moved) > (0.5 * self.skin) ** 2:  # (one host sync per step: the rebuild decision)
            self.rebuild()
        self.f = self.nl.lj_forces(self.q, 1.0, 1.0, rc_force=self.rc)
        self.v += 0.5 * dt * self.f[:, :3]

    def energy(self):
        return float(self.f[:, 3].sum() + 0.5 * self.v.square().sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=12)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--dtype", default="f64", choices=["f32", "f64"])
    args = ap.parse_args()
    a, box = 1.56, 4.0 * args.cells
    q, v = fcc_droplet(args.cells, a, box, np.float32 if args.dtype == "f32" else np.float64)
    sim = Simulation(q, v, box)
    e0 = sim.energy()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e1 = sim.energy()
    print(f"N={len(q)} steps={args.steps} builds={sim.builds} re-sorts={sim.sorts} E0={e0:.6f} E1={e1:.6f} "
          f"drift={(e1 - e0) / abs(e0):.2e} {1e3 * dt / args.steps:.3f} ms/step")


if __name__ == "__main__":
    main()
