"""Multi-process tests of the slab decomposition: world_size 2, 3 and 8 over gloo.
CPU: decomposition + ghost exchange + ownership rule, per-rank build emulated by the oracle.
GPU: the same with the real nl_make_list_slab on every rank (ranks share the one GPU of the test box)."""
import numpy as np
import pytest

from md_neighbor_list_amd import slab
from tests.slab_worker import run


def test_split_layers():
    assert slab.split_layers(30, 8) == [(0, 4), (4, 8), (8, 12), (12, 16), (16, 20), (20, 24), (24, 27), (27, 30)]
    assert slab.split_layers(97, 8)[0] == (0, 13) and slab.split_layers(97, 8)[-1] == (85, 97)
    assert slab.split_layers(5, 1) == [(0, 5)]
    with pytest.raises(ValueError):
        slab.split_layers(3, 2)  # a rank with 2 of 3 layers would see the same layer as both ghosts
    with pytest.raises(ValueError):
        slab.split_layers(4, 8)


def test_z_layer_matches_oracle_hash():
    import torch

    from md_neighbor_list_amd import inputs
    from oracle import pyoracle as po

    for dt in (np.float32, np.float64):
        q, box = inputs.uniform_box(20000, dtype=dt, seed=9, box=(13.2, 13.2, 26.4))
        q[:100, 2] = np.nextafter(dt(26.4), dt(0))
        q[100:200, 2] = dt(-0.5)
        cells, mesh = po.cells(q, 3.3, box)
        iz = slab.z_layer(torch.from_numpy(q), box, 3.3).numpy()
        assert np.array_equal(iz, cells // (mesh[0] * mesh[1]))


@pytest.mark.parametrize("world,case", [
    (2, (6000, (14.0, 14.0, 20.0), 3.3, "float32", 71)),   # 6 layers: 3 + 3
    (3, (9000, (13.5, 15.0, 30.0), 3.3, "float64", 72)),   # 9 layers: 3 + 3 + 3
    (2, (4000, (12.0, 12.0, 17.0), 3.3, "float32", 73)),   # 5 layers: 3 + 2
    (2, (6000, (14.0, 14.0, 20.0), 3.3, "float32", 74, [2, 3])),  # layers 2 and 3 empty: zero-size ghost messages
    (3, (9000, (13.5, 15.0, 30.0), 3.3, "float32", 75, [0, 5])),  # empty layers at a slab top and at the box bottom
    (8, (20000, (12.0, 12.0, 66.5), 3.3, "float32", 181)),  # the world size of the scaling run: 20 layers, 3+3+3+3+2+2+2+2
])
def test_slab_union_equals_global_list_cpu(world, case):
    res = run(world, "oracle", case)
    assert res[0] == "ok"


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [
    (2, (60000, (30.0, 30.0, 66.0), 3.3, "float32", 81)),
    (3, (50000, (25.0, 25.0, 80.0), 3.3, "float64", 82)),
])
def test_slab_union_equals_global_list_gpu(world, case):
    res = run(world, "hip", case)
    assert res[0] == "ok"


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [
    (2, (60000, (30.0, 30.0, 66.0), 3.3, "float32", 83)),
    (3, (50000, (25.0, 25.0, 80.0), 3.3, "float64", 84)),
    (3, (40000, (25.0, 25.0, 80.0), 3.3, "float32", 85, [7, 8])),  # two empty layers: a rank with nothing to send upwards
    (5, (90000, (25.0, 25.0, 120.0), 3.3, "float32", 301)),  # interior ranks with four distinct neighbours' messages in flight
])
def test_distributed_build_behind_the_c_abi_with_moving_particles(world, case):
    """nl_comm_create_callbacks + nl_make_list_distributed: the pack kernel, the count and halo
    exchange and the slab build inside the library, the transport being gloo; a second pair of builds after every
    particle has moved (ghost counts change, particles change layer and owner)."""
    res = run(world, "cabi", case)
    assert res[0] == "ok"


@pytest.mark.gpu
def test_distributed_build_over_rccl_with_one_rank():
    """nl_comm_unique_id + nl_comm_create (RCCL resolved at run time, ncclCommInitRank) with a world of one -- all a
    one-GPU box can host: the communicator comes up and the build is the whole-box build.  The N > 1 RCCL exchange
    itself (grouped ncclSend / ncclRecv) has not run on hardware: DESIGN.md section 6."""
    res = run(1, "cabi_rccl1", (30000, (25.0, 25.0, 40.0), 3.3, "float32", 86))
    assert res[0] == "ok"


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [
    (2, (30000, (20.0, 22.0, 40.0), 3.3, "float32", 91)),   # 12 layers: the end ranks' ghosts are periodic images
    (3, (30000, (21.0, 20.0, 31.0), 3.3, "float64", 92)),   # 9 layers, 3 + 3 + 3
])
def test_slab_union_equals_global_minimum_image_list_gpu(world, case):
    res = run(world, "hip_pbc", case)
    assert res[0] == "ok"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,pbc", [("float32", False), ("float64", False), ("float32", True)])
def test_split_slab_build_equals_the_single_call(dtype, pbc):
    """nl_make_list_slab_begin + _finish (owned layers binned first, ghosts later: what slab.build does so that the
    halo exchange overlaps the first part) against nl_make_list_slab on the same slab of a box, one process: identical
    rows; a wrong n_ghost_lo is reported, not used."""
    import torch

    from md_neighbor_list_amd import NeighListGPU, inputs
    from md_neighbor_list_amd._lib import NLError
    from tests.util import canonical_csr

    dt = np.float32 if dtype == "float32" else np.float64
    rc, box = 3.3, (27.0, 24.0, 40.0)  # 12 layers
    q, box = inputs.uniform_box(42000, dtype=dt, seed=77, box=box)
    mz = int(box[2] / rc)
    iz = slab.z_layer(torch.from_numpy(q), box, rc).numpy()
    for z_lo, z_hi in ((4, 8), (0, 5), (9, 12)):
        own = np.nonzero((iz >= z_lo) & (iz < z_hi))[0]
        glo = np.nonzero(iz == (z_lo - 1) % mz)[0]
        ghi = np.nonzero(iz == z_hi % mz)[0]
        order = np.concatenate([own, glo, ghi])
        qa = torch.from_numpy(q[order]).cuda()
        gid = torch.from_numpy(order.astype(np.int32)).cuda()
        tdt = torch.float32 if dt == np.float32 else torch.float64
        res = []
        for split in (False, True):
            nl = NeighListGPU(rc, *box, dtype=tdt, minimum_image=pbc)
            nl.Initialize(len(order))
            if split:
                nl.MakeNeighListSlabBegin(qa, gid, len(own), len(glo), z_lo, z_hi)
                nl.MakeNeighListSlabFinish(sync=True)
            else:
                nl.MakeNeighListSlab(qa, gid, len(own), z_lo, z_hi, sync=True)
            kp, sl = nl.key_pointer().cpu().numpy(), nl.sorted_list().cpu().numpy()
            res.append((kp, canonical_csr(kp, sl)))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert int(res[0][0][-1]) > 0
        if len(glo) > 3:
            nl = NeighListGPU(rc, *box, dtype=tdt, minimum_image=pbc)
            nl.Initialize(len(order))
            nl.MakeNeighListSlabBegin(qa, gid, len(own), len(glo) - 3, z_lo, z_hi)
            with pytest.raises(NLError):
                nl.MakeNeighListSlabFinish(sync=True)
