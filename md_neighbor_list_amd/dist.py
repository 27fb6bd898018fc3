"""The decomposed build behind the C ABI (nl_comm_create*, nl_make_list_distributed): pack
kernel, count exchange, halo exchange and the slab build all inside libnl_hip.so.  This module only creates the
communicator -- over RCCL (the unique id travels through ``torch.distributed``), or over a host transport made of
``torch.distributed`` point-to-point calls (gloo: tests, rehearsals) -- and holds the caller's position buffer.

``md_neighbor_list_amd.slab`` is the older, pure-``torch.distributed`` form of the same exchange (static ghost counts).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, slab
from ._lib import check


def _gloo_transport():
    """nl_sendrecv_fn over torch.distributed point-to-point calls on host memory (blocking)."""

    def fn(user, peer_to, send, send_bytes, peer_from, recv, recv_bytes):
        try:
            ops = []
            keep = []
            if send_bytes:
                a = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(send_bytes,))
                t = torch.from_numpy(a.copy())
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer_to))
            r = None
            if recv_bytes:
                r = torch.empty(recv_bytes, dtype=torch.uint8)
                ops.append(dist.P2POp(dist.irecv, r, peer_from))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if r is not None:
                C.memmove(recv, r.numpy().ctypes.data, recv_bytes)
            return 0
        except Exception:  # pragma: no cover  (reported as NL_ERR_COMM)
            import traceback

            traceback.print_exc()
            return 1

    return _lib.SENDRECV_FN(fn)


class DistributedNeighList:
    """One rank of the decomposed build: ``nl`` is this rank's NeighListGPU (created for the GLOBAL box)."""

    def __init__(self, nl, rank=None, world=None, transport="auto"):
        self.nl, self._lib = nl, _lib.load()
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        if transport == "auto":
            transport = "rccl" if self.world > 1 and dist.get_backend() == "nccl" else "host"
        self.transport = transport
        self._comm = C.c_void_p()
        self._cb = None
        dev = nl.device.index or 0
        if transport == "rccl":
            uid = torch.zeros(_lib.NL_UNIQUE_ID_BYTES, dtype=torch.uint8)
            if self.rank == 0:
                buf = (C.c_uint8 * _lib.NL_UNIQUE_ID_BYTES)()
                check(self._lib.nl_comm_unique_id(buf), "nl_comm_unique_id")
                uid = torch.from_numpy(np.frombuffer(buf, dtype=np.uint8).copy())
            if self.world > 1:
                on_dev = dist.get_backend() == "nccl"
                t = uid.to(nl.device) if on_dev else uid
                dist.broadcast(t, 0)
                uid = t.cpu()
            raw = uid.numpy().tobytes()
            check(self._lib.nl_comm_create(C.byref(self._comm), self.rank, self.world, raw, dev), "nl_comm_create")
        else:
            self._cb = _gloo_transport()
            check(self._lib.nl_comm_create_callbacks(C.byref(self._comm), self.rank, self.world, self._cb, None, dev),
                  "nl_comm_create_callbacks")
        z_lo, z_hi = C.c_int32(), C.c_int32()
        check(self._lib.nl_comm_layers(nl._h, self._comm, C.byref(z_lo), C.byref(z_hi)), "nl_comm_layers")
        self.z_lo, self.z_hi = int(z_lo.value), int(z_hi.value)
        self.q = None
        self.n_owned = 0

    def __del__(self):
        c, self._comm = getattr(self, "_comm", None), None
        if c:
            try:
                self._lib.nl_comm_destroy(c)
            except Exception:  # pragma: no cover
                pass

    def scatter(self, q_global: torch.Tensor, box, rc, gid_global=None, slack=1.5):
        """Keeps this rank's slab of a (synthetic) global box: owned particles first, the global id in the w component,
        room for the ghosts behind them.  Outside the timed build, like slab.setup."""
        iz = slab.z_layer(q_global, box, rc)
        own = (iz >= self.z_lo) & (iz < self.z_hi)
        if gid_global is None:
            gid_global = torch.arange(q_global.shape[0], dtype=torch.int32, device=q_global.device)
        q_own, gid_own = q_global[own], gid_global[own]
        n = int(q_own.shape[0])
        layers = max(self.z_hi - self.z_lo, 1)
        cap = n + (0 if self.world == 1 else int(2 * slack * n / layers) + 4096)
        self.q = torch.zeros((cap, 4), dtype=q_global.dtype, device=self.nl.device)
        self.q[:n, :3] = q_own[:, :3].to(self.nl.device)
        g = gid_own.to(self.nl.device)
        self.q[:n, 3] = g.view(torch.float32) if self.q.dtype == torch.float32 else g.to(torch.int64).view(torch.float64)
        self.n_owned = n
        self.gid_owned = g
        return n

    def build(self, sync=True):
        """nl_make_list_distributed on the held buffer: pack, halo messages, unpack, slab build -- all in the library, and
        with sync=False without a single wait on the host (the ghost counts stay on the device: ``ghosts()``)."""
        nl = self.nl
        stream = torch.cuda.current_stream(nl.device).cuda_stream
        nl._q = self.q
        check(self._lib.nl_make_list_distributed(nl._h, self._comm, self.q.data_ptr(), self.q.shape[0], self.n_owned, stream,
                                                 1 if sync else 0), "nl_make_list_distributed")
        nl._n_rows = self.n_owned
        if sync:
            self.ghosts()
        else:
            nl._n = self.n_owned + getattr(self, "n_ghost_lo", 0) + getattr(self, "n_ghost_hi", 0)

    def ghosts(self):
        """(ghosts from below, from above) of the last build; waits for it."""
        lo, hi = C.c_int32(), C.c_int32()
        check(self._lib.nl_distributed_ghosts(self._comm, C.byref(lo), C.byref(hi)))
        self.n_ghost_lo, self.n_ghost_hi = int(lo.value), int(hi.value)
        self.nl._n = self.n_owned + self.n_ghost_lo + self.n_ghost_hi
        return self.n_ghost_lo, self.n_ghost_hi
