"""Multi-GPU domain decomposition: z-slabs of the cell mesh + one-cell ghost-layer exchange.

The reference is single-GPU; this is new functionality.  The box is cut along z into ``world`` slabs of whole cell
layers of the GLOBAL mesh (mesh_z = int(Lz / rc), exactly the reference's mesh, neighlist_cpu.hpp:384-386).  Rank r
owns the particles whose reference cell lies in its layers [z_lo, z_hi) and receives, every build, the particles of
the two periodic neighbour layers z_lo-1 and z_hi (mod mesh_z) from ranks r-1 and r+1: positions unshifted (the
reference wraps cell indices but never applies a minimum image, neighlist_cpu.hpp:219-223) plus global ids.  It
then builds rows for its owned particles only; a pair (i, j) lands in the row of min(i, j) on the rank that owns
that particle, so the union over ranks is exactly the global half list.

Communication is point-to-point only (``torch.distributed`` send/recv: RCCL over xGMI with the ``nccl`` backend,
``gloo`` on CPU for tests); there is no collective on the data path.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist


def mesh_of(box, rc):
    """mesh_size[d] = (int)(L_d / rc), neighlist_cpu.hpp:384-386."""
    return tuple(int(b / rc) for b in box)


def split_layers(mesh_z: int, world: int):
    """[z_lo, z_hi) per rank: contiguous, as even as possible.  Every rank needs >= 1 layer and the two ghost layers
    of a rank must be distinct layers, i.e. mesh_z - owned >= 2."""
    if world < 1 or mesh_z < 3:
        raise ValueError("need world >= 1 and at least 3 cell layers")
    base, rem = divmod(mesh_z, world)
    if world > 1 and (base < 1 or mesh_z - (base + (1 if rem else 0)) < 2):
        raise ValueError(f"cannot cut {mesh_z} cell layers into {world} slabs with distinct ghost layers")
    out, z = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((z, z + n))
        z += n
    return out


def z_layer(q: torch.Tensor, box, rc) -> torch.Tensor:
    """The reference's z cell index of every particle: (int32)(z * ims_z), one periodic wrap
    (GenHash + ApplyPBC, neighlist_cpu.hpp:51-66), with ims_z rounded to the position type as the class stores it
    (neighlist_cpu.hpp:12,409-411).  The library re-derives this on the device and flags any disagreement
    (NL_ERR_DOMAIN), so this copy of the rule cannot silently drift."""
    mz = int(box[2] / rc)
    if q.dtype == torch.float32:
        ims = np.float32(1.0 / np.float64(np.float32(box[2] / mz)))
        t = q[:, 2] * torch.tensor(ims, dtype=torch.float32, device=q.device)
    else:
        ims = 1.0 / (box[2] / mz)
        t = q[:, 2] * torch.tensor(ims, dtype=torch.float64, device=q.device)
    iz = t.to(torch.int32)  # truncation toward zero
    iz = torch.where(iz < 0, iz + mz, iz)
    iz = torch.where(iz >= mz, iz - mz, iz)
    return iz


@dataclass
class SlabState:
    rank: int
    world: int
    z_lo: int
    z_hi: int
    n_rows: int                # owned particles
    q_all: torch.Tensor        # [n_rows + ghosts_max, 3|4] owned first, then ghost_low, ghost_high
    gid_all: torch.Tensor      # int32 global ids, same order
    send_lo_idx: torch.Tensor  # owned rows lying in layer z_lo      (go to rank-1 as its upper ghosts)
    send_hi_idx: torch.Tensor  # owned rows lying in layer z_hi - 1  (go to rank+1 as its lower ghosts)
    n_ghost_lo: int
    n_ghost_hi: int

    @property
    def n_total(self):
        return self.n_rows + self.n_ghost_lo + self.n_ghost_hi


def _p2p(ops_spec, staging_cpu: bool, defer: bool = False):
    """ops_spec: list of ("send"|"recv", tensor, peer).  One grouped batch of point-to-point transfers.
    defer: start the transfers and return the function that completes them (None when there is nothing to wait for)."""
    # An empty boundary layer sends nothing: both sides know the count (setup exchanged it), so both skip the message.
    ops_spec = [(kind, t, peer) for kind, t, peer in ops_spec if t.numel() > 0]
    if not ops_spec:
        return None
    bufs, ops = [], []
    for kind, t, peer in ops_spec:
        if staging_cpu:
            c = t.detach().to("cpu").contiguous() if kind == "send" else torch.empty(t.shape, dtype=t.dtype)
            bufs.append((kind, t, c))
            ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, c, peer))
        else:
            ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer))
    works = dist.batch_isend_irecv(ops)
    keep_alive = (ops, ops_spec)  # the send buffers must outlive the transfers when completion is deferred

    def complete():
        _ = keep_alive
        for w in works:
            w.wait()  # (RCCL: makes the current stream wait; gloo: blocks the host)
        for kind, t, c in bufs:
            if kind == "recv":
                t.copy_(c)

    if not defer:
        complete()
        return None
    return complete


def _staging():
    return dist.get_backend() != "nccl"


def setup(q_global: torch.Tensor, gid_global: torch.Tensor | None, box, rc, rank=None, world=None) -> SlabState:
    """Initial scatter, outside the timed build (the reference copies positions to the device once before its
    timing loop, make_list.cu:119,124): every rank looks at the whole (synthetic) box and keeps its slab.
    Also exchanges the ghost counts once so that the per-build exchange needs no size negotiation."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    mz = int(box[2] / rc)
    z_lo, z_hi = split_layers(mz, world)[rank]
    iz = z_layer(q_global, box, rc)
    if gid_global is None:
        gid_global = torch.arange(q_global.shape[0], dtype=torch.int32, device=q_global.device)
    own = (iz >= z_lo) & (iz < z_hi)
    q_own = q_global[own].contiguous()
    gid_own = gid_global[own].contiguous()
    iz_own = iz[own]
    send_lo = torch.nonzero(iz_own == z_lo).flatten()
    send_hi = torch.nonzero(iz_own == z_hi - 1).flatten()
    n_rows = int(q_own.shape[0])
    n_glo = n_ghi = 0
    if world > 1:
        lo_peer, hi_peer = (rank - 1) % world, (rank + 1) % world
        cnt_send = torch.tensor([send_lo.numel(), send_hi.numel()], dtype=torch.int64, device=q_global.device)
        cnt_from_lo = torch.zeros(1, dtype=torch.int64, device=q_global.device)
        cnt_from_hi = torch.zeros(1, dtype=torch.int64, device=q_global.device)
        # a peer's bottom-layer count is MY upper-ghost count; with world == 2 both messages come from the same
        # rank in the order [its bottom, its top], so receive the upper-ghost count first (see exchange_ghosts)
        recvs = [("recv", cnt_from_hi, hi_peer), ("recv", cnt_from_lo, lo_peer)]
        if lo_peer != hi_peer:
            recvs.reverse()
        _p2p([("send", cnt_send[0:1], lo_peer), ("send", cnt_send[1:2], hi_peer)] + recvs, _staging())
        n_glo, n_ghi = int(cnt_from_lo.item()), int(cnt_from_hi.item())
    q_all = torch.empty((n_rows + n_glo + n_ghi, q_global.shape[1]), dtype=q_global.dtype, device=q_global.device)
    gid_all = torch.empty(n_rows + n_glo + n_ghi, dtype=torch.int32, device=q_global.device)
    q_all[:n_rows] = q_own
    gid_all[:n_rows] = gid_own
    if q_all.shape[1] == 4:
        # ids ride in the w component (bit pattern of an int32 / int64): one message per neighbour instead of two
        if q_all.dtype == torch.float32:
            q_all[:n_rows, 3].copy_(gid_own.view(torch.float32))
        else:
            q_all[:n_rows, 3].copy_(gid_own.to(torch.int64).view(torch.float64))
    return SlabState(rank, world, z_lo, z_hi, n_rows, q_all, gid_all, send_lo, send_hi, n_glo, n_ghi)


def exchange_ghosts(st: SlabState, defer: bool = False):
    """The per-build halo exchange: my bottom layer goes down, my top layer goes up; the neighbours' layers land
    directly in the ghost region of q_all / gid_all.  With world == 2 both neighbours are the same rank: the send
    of my bottom layer pairs with its receive of 'upper ghosts' by message order inside the batch."""
    if st.world == 1:
        return None
    lo_peer, hi_peer = (st.rank - 1) % st.world, (st.rank + 1) % st.world
    n0, n1 = st.n_rows, st.n_rows + st.n_ghost_lo
    in_w = st.q_all.shape[1] == 4  # ids travel inside the positions
    # one gather for both layers (one small kernel per build instead of two): [bottom layer; top layer]
    n_lo = st.send_lo_idx.numel()
    if not hasattr(st, "_send_idx"):
        st._send_idx = torch.cat([st.send_lo_idx, st.send_hi_idx])
    packed = st.q_all[:n0].index_select(0, st._send_idx)
    q_lo, q_hi = packed[:n_lo], packed[n_lo:]
    sends = [("send", q_lo, lo_peer), ("send", q_hi, hi_peer)]
    recv_lo = [("recv", st.q_all[n0:n1], lo_peer)]
    recv_hi = [("recv", st.q_all[n1:n1 + st.n_ghost_hi], hi_peer)]
    if not in_w:
        g_packed = st.gid_all[:n0].index_select(0, st._send_idx)
        g_lo, g_hi = g_packed[:n_lo], g_packed[n_lo:]
        sends = [("send", q_lo, lo_peer), ("send", g_lo, lo_peer), ("send", q_hi, hi_peer), ("send", g_hi, hi_peer)]
        recv_lo.append(("recv", st.gid_all[n0:n1], lo_peer))
        recv_hi.append(("recv", st.gid_all[n1:n1 + st.n_ghost_hi], hi_peer))
    # Order matters when lo_peer == hi_peer (world 2): the peer posts [its bottom layer, its top layer]; its bottom
    # layer is MY upper ghost layer and its top layer my lower one, so the upper-ghost receive is posted first.
    recvs = recv_hi + recv_lo if lo_peer == hi_peer else recv_lo + recv_hi
    return _p2p(sends + recvs, _staging(), defer)


def build(nl, st: SlabState, sync=True, overlap=None) -> None:
    """One domain-decomposed build on this rank: halo exchange, then the slab build on owned + ghost particles.
    overlap: run the binning of the owned layers under the transfer (nl_make_list_slab_begin / _finish).  Default: on
    over gloo, where every test runs it; OFF over nccl, whose deferred completion has never run on hardware
    -- NL_SLAB_OVERLAP=1 turns it on there."""
    import os

    if overlap is None:
        overlap = dist.get_backend() != "nccl" or os.environ.get("NL_SLAB_OVERLAP") == "1" if st.world > 1 else False
    gid = nl.GID_IN_W if st.q_all.shape[1] == 4 else st.gid_all
    if st.world == 1:
        nl.MakeNeighListSlab(st.q_all, gid, st.n_rows, 0, nl.mesh_size[2], sync=sync)
        return
    if not overlap:
        exchange_ghosts(st)
        nl.MakeNeighListSlab(st.q_all, gid, st.n_rows, st.z_lo, st.z_hi, sync=sync)
        return
    # The exchange is started, the binning of the OWNED layers is enqueued behind the pack kernel on the current stream
    # (it reads q_all[:n_rows] only and fills the owned region of the cell-sorted array), then the current stream is
    # made to wait for the ghosts and the rest of the build follows: the transfer runs under the owned binning.
    complete = exchange_ghosts(st, defer=True)
    nl.MakeNeighListSlabBegin(st.q_all, gid, st.n_rows, st.n_ghost_lo, st.z_lo, st.z_hi)
    if complete is not None:
        complete()
    nl.MakeNeighListSlabFinish(sync=sync)
