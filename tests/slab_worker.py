"""Worker of the multi-process slab tests (gloo, world_size >= 2).  mode "oracle": the per-rank build is emulated
with the CPU oracle (CPU-only machines; checks decomposition, ghost exchange and the ownership rule).  mode "hip":
every rank runs the real nl_make_list_slab on the (shared) GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pairs_of(kp, lst, row_gid=None):
    rows = np.repeat(np.arange(len(kp) - 1, dtype=np.int64), np.diff(kp))
    if row_gid is not None:
        rows = row_gid.astype(np.int64)[rows]
    return (rows << 32) | lst.astype(np.int64)


def worker(rank, world, port, mode, case, ret):
    import torch
    import torch.distributed as dist

    from md_neighbor_list_amd import inputs, slab
    from oracle import pyoracle as po

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, box, rc, dtype, seed = case[:5]
        q, box = inputs.uniform_box(n, dtype=np.dtype(dtype).type, seed=seed, box=box)
        if len(case) > 5:  # vacate whole cell layers: some rank then has an EMPTY boundary layer to send
            iz = slab.z_layer(torch.from_numpy(q), box, rc).numpy()
            q = np.ascontiguousarray(q[~np.isin(iz, case[5])])
            n = len(q)
        pbc = mode == "hip_pbc"  # minimum-image mode (nl_set_periodic): the end ranks' ghost layers are images
        if mode in ("cabi", "cabi_rccl1"):
            return worker_cabi(rank, world, mode, q, box, rc, ret)
        dev = "cuda" if mode in ("hip", "hip_pbc") else "cpu"
        qt = torch.from_numpy(q).to(dev)
        st = slab.setup(qt, None, box, rc)
        if mode == "oracle":
            slab.exchange_ghosts(st)
            lq = st.q_all.numpy()
            # ids travel in the w component as integer bit patterns (slab.setup); ghosts have them only there
            gid = (lq[:, 3].view(np.int32) if lq.dtype == np.float32 else lq[:, 3].view(np.int64).astype(np.int32)).copy()
            assert np.array_equal(gid[: st.n_rows], st.gid_all.numpy()[: st.n_rows])
            # every ghost must lie in one of my two neighbour layers, every owned particle in my slab
            iz = slab.z_layer(st.q_all, box, rc).numpy()
            mz = int(box[2] / rc)
            assert np.all((iz[: st.n_rows] >= st.z_lo) & (iz[: st.n_rows] < st.z_hi))
            n1 = st.n_rows + st.n_ghost_lo
            assert np.all(iz[st.n_rows:n1] == (st.z_lo - 1) % mz) and np.all(iz[n1:] == st.z_hi % mz)
            h = po.build(lq, rc, box)
            a = np.repeat(np.arange(len(lq), dtype=np.int64), np.diff(h.key_pointer))
            b = h.sorted_list.astype(np.int64)
            ga, gb = gid[a].astype(np.int64), gid[b].astype(np.int64)
            lo_is_a = ga < gb
            keep = np.where(lo_is_a, a, b) < st.n_rows  # the particle with the smaller global id is mine
            mine = (np.minimum(ga, gb)[keep] << 32) | np.maximum(ga, gb)[keep]
        else:
            from md_neighbor_list_amd import NeighListGPU

            tdt = torch.float32 if q.dtype == np.float32 else torch.float64
            nl = NeighListGPU(rc, *box, dtype=tdt, minimum_image=pbc)
            nl.Initialize(st.q_all.shape[0])
            slab.build(nl, st, sync=True)
            kp = nl.key_pointer().cpu().numpy()
            sl = nl.sorted_list().cpu().numpy()
            assert len(kp) == st.n_rows + 1
            mine = pairs_of(kp, sl, st.gid_all[: st.n_rows].cpu().numpy())
            # second build, asynchronous, must agree
            slab.build(nl, st, sync=False)
            nl.synchronize()
            again = pairs_of(nl.key_pointer().cpu().numpy(), nl.sorted_list().cpu().numpy(),
                             st.gid_all[: st.n_rows].cpu().numpy())
            assert np.array_equal(np.sort(mine), np.sort(again))
        gathered = [None] * world
        dist.all_gather_object(gathered, (mine, st.n_rows, st.n_ghost_lo, st.n_ghost_hi))
        if rank == 0:
            ref = po.build_pbc(q, rc, box) if pbc else po.build(q, rc, box)
            want = np.sort(pairs_of(ref.key_pointer, ref.sorted_list))
            got = np.sort(np.concatenate([g[0] for g in gathered]))
            assert sum(g[1] for g in gathered) == n
            assert len(got) == len(want) == ref.npairs, (len(got), len(want))
            assert np.array_equal(got, want)
            assert len(np.unique(got)) == len(got)  # every pair exactly once over all ranks
            ret.put(("ok", ref.npairs, [g[1:] for g in gathered]))
    except Exception as e:  # pragma: no cover
        import traceback

        ret.put(("fail", rank, traceback.format_exc()))
        raise e
    finally:
        dist.destroy_process_group()


def worker_cabi(rank, world, mode, q, box, rc, ret):
    """nl_make_list_distributed (pack kernel, count + halo exchange and slab build inside libnl_hip.so) over the host
    transport (gloo), twice: the second time every particle has moved -- boundary-layer populations change, particles
    change layer and owner (the test migrates them between ranks, as an MD code would) -- with the SAME handle and
    communicator.  The union of the ranks' rows must be the oracle's global list both times."""
    import torch
    import torch.distributed as dist

    from md_neighbor_list_amd import NeighListGPU
    from md_neighbor_list_amd.dist import DistributedNeighList
    from oracle import pyoracle as po

    n = len(q)
    tdt = torch.float32 if q.dtype == np.float32 else torch.float64
    nl = NeighListGPU(rc, *box, dtype=tdt)
    nl.Initialize(int(2.2 * n / world) + 8192)
    dn = DistributedNeighList(nl, rank, world, transport="rccl" if mode == "cabi_rccl1" else "host")
    rng = np.random.default_rng(5)
    ghosts_seen = []
    for rnd in range(3):
        if rnd == 1:  # everybody moves by up to 0.45 cells; coordinates stay inside the box
            q = q.copy()
            q[:, :3] += rng.uniform(-1.5, 1.5, size=(n, 3)).astype(q.dtype)
            q[:, :3] = np.mod(q[:, :3], np.array(box, dtype=q.dtype))
            q[:, :3] = np.minimum(q[:, :3], np.nextafter(np.array(box, dtype=q.dtype), q.dtype.type(0)))
        if rnd == 2:  # a ninth of the particles gathers in the two z layers at the first cut: those boundary layers grow
            # beyond the capacity their messages were negotiated for (count + 25 % + 1024)
            q = q.copy()
            mz = int(box[2] / rc)
            cut = (mz // world + (1 if mz % world else 0)) * (box[2] / mz)
            k = n // 9
            q[:k, 2] = (cut + rng.uniform(-0.95, 0.95, size=k) * (box[2] / mz)).astype(q.dtype)
        dn.scatter(torch.from_numpy(q).cuda(), box, rc)
        for sync in ((True, False) if rnd < 2 else (False, True, False)):  # (the synchronous build also grows the pair list for the crowd)
            dn.build(sync=sync)
            if rnd == 2 and world > 1 and sync is False and not ghosts_seen[-1][1:] == ("renegotiated",):
                # the first asynchronous build after the crowd formed: its messages were too small, and it says so at
                # its synchronisation; the next build negotiates new capacities
                try:
                    nl.synchronize()
                    overflowed = False
                except Exception as e:  # NL_ERR_CAPACITY
                    overflowed = "capacity" in str(e).lower()
                flags = [None] * world
                dist.all_gather_object(flags, overflowed)
                assert any(flags), "no rank saw a boundary layer outgrow its message"
                ghosts_seen.append((ghosts_seen[-1][0], "renegotiated"))
                continue
            nl.synchronize()
            dn.ghosts()
            kp = nl.key_pointer().cpu().numpy()
            sl = nl.sorted_list().cpu().numpy()
            assert len(kp) == dn.n_owned + 1
            mine = pairs_of(kp, sl, dn.gid_owned.cpu().numpy())
            gathered = [None] * world
            dist.all_gather_object(gathered, (mine, dn.n_owned, dn.n_ghost_lo, dn.n_ghost_hi))
            if rank == 0:
                ref = po.build(q, rc, box)
                want = np.sort(pairs_of(ref.key_pointer, ref.sorted_list))
                got = np.sort(np.concatenate([g[0] for g in gathered]))
                assert sum(g[1] for g in gathered) == n
                assert len(got) == len(want) == ref.npairs, (rnd, len(got), len(want))
                assert np.array_equal(got, want), rnd
        ghosts_seen.append((dn.n_ghost_lo, dn.n_ghost_hi))
    g2 = [None] * world
    dist.all_gather_object(g2, [g for g in ghosts_seen if g[1:] != ("renegotiated",)])
    if rank == 0:
        if world > 1:
            assert any(a[0] != a[1] for a in g2), g2  # the ghost counts did change between the builds
        ret.put(("ok", 0, g2))


def run(world, mode, case, timeout=600):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, mode, case, ret)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = ret.get(timeout=timeout)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert res[0] == "ok", res
    return res
