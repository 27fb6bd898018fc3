"""Shared helpers of the test-suite."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_names(dup=None):
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    if dup is None:
        return names
    return [n for n in names if n.startswith("dup_") == bool(dup)]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def canonical_csr(key_pointer, sorted_list):
    """Per-particle ascending partners (make_list.cpp:120-128) with numpy only."""
    kp = np.asarray(key_pointer, dtype=np.int64)
    lst = np.asarray(sorted_list, dtype=np.int64)
    n = len(kp) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(kp))
    if len(lst) and int(lst.min()) >= 0 and int(lst.max()) < 2**31 and n < 2**31:
        key = (rows << 32) | lst  # one 64-bit key per entry: a plain sort instead of a two-key lexsort (4 x faster)
        key.sort()
        return (key & 0xFFFFFFFF).astype(np.int32)
    order = np.lexsort((lst, rows))
    return lst[order].astype(np.int32)


def gpu_build(q, rc, box, sync=True):
    """Runs the HIP path through the C ABI; returns (number_of_partners, key_pointer, sorted_list) on the host."""
    import torch

    from md_neighbor_list_amd import NeighListGPU

    dt = torch.float32 if q.dtype == np.float32 else torch.float64
    nl = NeighListGPU(rc, box[0], box[1], box[2], dtype=dt)
    nl.Initialize(len(q))
    qd = torch.from_numpy(np.ascontiguousarray(q)).cuda()
    nl.MakeNeighList(qd, len(q), sync=sync)
    if not sync:
        nl.synchronize()
    nop = nl.half_number_of_partners().cpu().numpy()
    kp = nl.key_pointer().cpu().numpy()
    sl = nl.sorted_list().cpu().numpy()
    return nl, nop, kp, sl
