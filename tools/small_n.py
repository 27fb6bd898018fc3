#!/usr/bin/env python3
"""Build time against system size: back-to-back asynchronous builds (what an MD loop pays) and the sum of the stage
times from HIP events (what the kernels take): the difference is launch overhead."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_neighbor_list_amd import NeighListGPU, inputs, _lib
if os.environ.get("NL_LIB"):  # A/B another build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["NL_LIB"])
for n in (4096, 32768, 119164, 262144, 1 << 20):
    q, box = inputs.uniform_box(n, 1.0, np.float32)
    qd = torch.from_numpy(q).cuda()
    nl = NeighListGPU(3.3, *box, dtype=torch.float32)
    nl.Initialize(len(q))
    for _ in range(10):
        nl.MakeNeighList(qd, len(q), sync=False)
    nl.synchronize()
    reps = 200
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        nl.MakeNeighList(qd, len(q), sync=False)
    t_enq = time.perf_counter() - t0
    nl.synchronize(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(50):
        nl.MakeNeighList(qd, len(q), sync=True)
    t_sync = (time.perf_counter() - t0) / 50
    st = nl.profile_stages(qd, reps=30)
    nl.set_graph(True)
    for _ in range(5):
        nl.MakeNeighList(qd, len(q), sync=False)
    nl.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        nl.MakeNeighList(qd, len(q), sync=False)
    t_genq = time.perf_counter() - t0
    nl.synchronize(); torch.cuda.synchronize()
    t_graph = time.perf_counter() - t0
    pairs_graph = nl.half_number_of_pairs()
    nl.set_graph(False)
    print(f"N={len(q):8d}  async {t_all / reps * 1e6:7.1f} us/build (host enqueue {t_enq / reps * 1e6:6.1f})  sync {t_sync * 1e6:7.1f}  "
          f"stage sum {st['total'] * 1e3:7.1f} us  | hipGraph replay {t_graph / reps * 1e6:7.1f} us/build (host {t_genq / reps * 1e6:5.1f})  pairs {pairs_graph}", flush=True)
