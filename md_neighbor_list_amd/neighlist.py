"""Python mirror of the reference's builder classes, over the C ABI of libnl_hip.so.

``NeighListGPU`` keeps the call surface of the reference's ``NeighListGPU<Vec,Dtype>`` (neighlist_gpu.hpp:43-488:
ctor, Initialize, MakeNeighList, neigh_list, number_of_partners, number_of_pairs) and adds the accessors of the
scalar CPU class ``NeighList<Vec>`` (neighlist_cpu.hpp:437-463: key_pointer, sorted_list, half counts), because the
CPU class's half CSR is the native output of the HIP path and the contract it is checked against.

torch is plumbing only here: device memory for positions/results and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import NLError, check  # noqa: F401


class _DevView:
    """A borrowed device buffer exposed through __cuda_array_interface__ (zero copy into torch)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape),
            "typestr": typestr,
            "data": (int(ptr), False),
            "version": 2,
            "strides": None,
        }
        self._owner = owner  # keeps the handle alive


def _as_tensor(ptr, shape, typestr, owner, device):
    n = 1
    for s in shape:
        n *= int(s)
    if n == 0 or not ptr:
        dt = {"<i4": torch.int32, "<i8": torch.int64, "<f4": torch.float32, "<f8": torch.float64}[typestr]
        return torch.empty(tuple(int(s) for s in shape), dtype=dt, device=device)
    return torch.as_tensor(_DevView(ptr, shape, typestr, owner), device=device)


class NeighListGPU:
    """Verlet neighbour-list builder on one MI355X.

    Parameters follow neighlist_gpu.hpp:236-255: ``search_length`` (cut-off rc) and the box edges.  ``dtype``
    plays the role of the reference's compile-time ``Dtype``/``Vec`` choice (make_list.cu:6-12).
    """

    def __init__(self, search_length, Lx, Ly, Lz, dtype=torch.float32, device=None, full_list=False,
                 minimum_image=False):
        if dtype not in (torch.float32, torch.float64):
            raise TypeError("dtype must be torch.float32 or torch.float64")
        self._lib = _lib.load()  # raises when the HIP extension is missing
        if not torch.cuda.is_available():
            raise RuntimeError("NeighListGPU needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dtype = dtype
        self.search_length = float(search_length)
        self._h = C.c_void_p()
        check(
            self._lib.nl_create(C.byref(self._h), _lib.NL_F32 if dtype == torch.float32 else _lib.NL_F64,
                                float(search_length), float(Lx), float(Ly), float(Lz), self.device.index or 0),
            "nl_create",
        )
        mesh = (C.c_int32 * 3)()
        ncell = C.c_int64()
        check(self._lib.nl_get_mesh(self._h, C.byref(mesh), C.byref(ncell)))
        self.mesh_size = tuple(mesh)
        self.number_of_mesh = int(ncell.value)
        self._n = 0
        self._n_rows = 0
        self.full_list = False
        self.minimum_image = False
        if full_list:
            self.set_full_list(True)
        if minimum_image:
            self.set_periodic(True)
        self._q = None  # keeps the positions of an asynchronous build alive

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.nl_destroy(h)
            except Exception:  # pragma: no cover
                pass

    # ------------------------------------------------------------------ reference surface
    def Initialize(self, particle_number):
        """neighlist_gpu.hpp:268-287 / neighlist_cpu.hpp:408-415."""
        check(self._lib.nl_initialize(self._h, int(particle_number)), "nl_initialize")

    def set_capacity(self, max_pairs):
        check(self._lib.nl_set_capacity(self._h, int(max_pairs)), "nl_set_capacity")

    def set_offset_width(self, bits=0):
        """Width of the list offsets (nl_set_offset_width): 0 = 64-bit as soon as the list capacity exceeds INT32_MAX
        entries, 32 / 64 = forced.  The reference's int32 offsets wrap beyond 2^31 pairs (neighlist_cpu.hpp:15,29)."""
        check(self._lib.nl_set_offset_width(self._h, int(bits)), "nl_set_offset_width")

    def set_graph(self, on: bool = True):
        """Replay asynchronous builds from a captured hipGraph (nl_set_graph): saves launch overhead on small systems."""
        check(self._lib.nl_set_graph(self._h, 1 if on else 0), "nl_set_graph")

    def set_periodic(self, minimum_image=True):
        """Minimum-image distances across the periodic faces (nl_set_periodic).  The reference, and the default here,
        wrap the cell stencil but measure distances in an open box."""
        check(self._lib.nl_set_periodic(self._h, 1 if minimum_image else 0), "nl_set_periodic")
        self.minimum_image = bool(minimum_image)

    def set_full_list(self, full=True):
        """Builds produce the FULL list (every pair in both rows: the reference GPU kernels' contract,
        kernel_impl.cuh:24-33) instead of the scalar CPU class's half list; ``neigh_list()`` is then one coalesced
        conversion pass away.  The half-list accessors raise after a full build and vice versa."""
        check(self._lib.nl_set_list_kind(self._h, 1 if full else 0), "nl_set_list_kind")
        self.full_list = bool(full)

    def _check_q(self, q, n):
        if not isinstance(q, torch.Tensor) or q.device.type != "cuda":
            raise TypeError("q must be a torch tensor on the HIP device")
        if q.dtype != self.dtype or q.dim() != 2 or q.shape[1] not in (3, 4) or not q.is_contiguous():
            raise TypeError(f"q must be a contiguous (N, 3|4) tensor of {self.dtype}")
        n = q.shape[0] if n is None else int(n)
        if n > q.shape[0]:
            raise ValueError("particle_number exceeds the buffer")
        return n

    def MakeNeighList(self, q, particle_number=None, sync=True, tblock_size=128, smem_hei=7):
        """neighlist_gpu.hpp:289-466.  ``tblock_size``/``smem_hei`` select among the reference's CUDA variants
        and are accepted for source compatibility only."""
        n = self._check_q(q, particle_number)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._q = q
        self._n = self._n_rows = n
        check(self._lib.nl_make_list(self._h, q.data_ptr(), q.shape[1], n, stream, 1 if sync else 0), "nl_make_list")

    GID_IN_W = "w"  # MakeNeighListSlab(gid=GID_IN_W): ids are stored in q[:, 3] as integer bit patterns (NL_GID_IN_W)

    def MakeNeighListSlab(self, q, gid, n_rows, z_lo, z_hi, sync=True):
        """Domain-decomposed build: rows for the first ``n_rows`` (owned) particles, the rest
        are ghosts of the two neighbouring cell layers; ``gid`` are global ids: an int32 device tensor, None
        (identity) or ``GID_IN_W`` (taken from the w component of the positions)."""
        n = self._check_q(q, None)
        if isinstance(gid, str):
            if gid != self.GID_IN_W or q.shape[1] != 4:
                raise TypeError("gid='w' needs 4-component positions")
            gid_ptr = 1
        elif gid is not None:
            if gid.device.type != "cuda" or gid.dtype != torch.int32 or gid.numel() != n or not gid.is_contiguous():
                raise TypeError("gid must be a contiguous int32 device tensor with one id per particle")
            gid_ptr = gid.data_ptr()
        else:
            gid_ptr = None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._q = (q, gid)
        self._n, self._n_rows = n, int(n_rows)
        check(
            self._lib.nl_make_list_slab(self._h, q.data_ptr(), q.shape[1], gid_ptr, int(n_rows), n, int(z_lo), int(z_hi),
                                        stream, 1 if sync else 0),
            "nl_make_list_slab",
        )

    def MakeNeighListSlabBegin(self, q, gid, n_rows, n_ghost_lo, z_lo, z_hi):
        """nl_make_list_slab_begin: the part of a slab build that needs only the owned particles q[:n_rows]; the ghost
        rows of q may still be in flight.  Follow with MakeNeighListSlabFinish once the current stream waits for them."""
        n = self._check_q(q, None)
        if isinstance(gid, str):
            if gid != self.GID_IN_W or q.shape[1] != 4:
                raise TypeError("gid='w' needs 4-component positions")
            gid_ptr = 1
        elif gid is not None:
            if gid.device.type != "cuda" or gid.dtype != torch.int32 or gid.numel() != n or not gid.is_contiguous():
                raise TypeError("gid must be a contiguous int32 device tensor with one id per particle")
            gid_ptr = gid.data_ptr()
        else:
            gid_ptr = None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._q = (q, gid)
        self._n, self._n_rows = n, int(n_rows)
        check(
            self._lib.nl_make_list_slab_begin(self._h, q.data_ptr(), q.shape[1], gid_ptr, int(n_rows), n, int(n_ghost_lo),
                                              int(z_lo), int(z_hi), stream),
            "nl_make_list_slab_begin",
        )

    def MakeNeighListSlabFinish(self, sync=True):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self._lib.nl_make_list_slab_finish(self._h, stream, 1 if sync else 0), "nl_make_list_slab_finish")

    def synchronize(self):
        check(self._lib.nl_synchronize(self._h), "nl_synchronize")

    def neigh_list(self):
        """neighlist_gpu.hpp:468-474: full list, transposed, ``[k, i]`` = k-th neighbour of i, -1 padded."""
        lst, cnt, stride, mx = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int32()
        check(self._lib.nl_get_full_transposed(self._h, C.byref(lst), C.byref(cnt), C.byref(stride), C.byref(mx)),
              "nl_get_full_transposed")
        return _as_tensor(lst.value, (max(mx.value, 1), stride.value), "<i4", self, self.device)

    def number_of_partners(self):
        """neighlist_gpu.hpp:476-482: full counts."""
        lst, cnt, stride, mx = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int32()
        check(self._lib.nl_get_full_transposed(self._h, C.byref(lst), C.byref(cnt), C.byref(stride), C.byref(mx)),
              "nl_get_full_transposed")
        return _as_tensor(cnt.value, (stride.value,), "<i4", self, self.device)

    def number_of_pairs(self):
        """neighlist_gpu.hpp:484-487: the sum of the FULL counts (= 2 x half pairs)."""
        return 2 * self.half_number_of_pairs()

    # ------------------------------------------------------------------ CPU-class surface (half CSR)
    def _half(self, width=32):
        """(key_pointer, sorted_list, counts, npairs) pointers; width = 32 | 64 (key_pointer type) | 0 (no key_pointer)."""
        kp, sl, nop, npairs = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        f = self._lib.nl_get_half_csr if width == 32 else self._lib.nl_get_half_csr64
        check(f(self._h, C.byref(kp) if width else None, C.byref(sl), C.byref(nop), C.byref(npairs)), "nl_get_half_csr")
        return kp.value, sl.value, nop.value, int(npairs.value)

    def _full(self, width=32):
        kp, sl, nop, ne = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        f = self._lib.nl_get_full_csr if width == 32 else self._lib.nl_get_full_csr64
        check(f(self._h, C.byref(kp) if width else None, C.byref(sl), C.byref(nop), C.byref(ne)), "nl_get_full_csr")
        return kp.value, sl.value, nop.value, int(ne.value)

    def full_csr(self, width=32):
        """(key_pointer[N+1], list[2P], counts[N]) of a full-list build, as views valid until the next build."""
        kp, sl, nop, ne = self._full(width)
        return (_as_tensor(kp, (self._n_rows + 1,), "<i4" if width == 32 else "<i8", self, self.device),
                _as_tensor(sl, (ne,), "<i4", self, self.device),
                _as_tensor(nop, (self._n_rows,), "<i4", self, self.device))

    def half_number_of_pairs(self):
        """neighlist_cpu.hpp:437-439."""
        npairs = C.c_int64()
        check(self._lib.nl_number_of_pairs(self._h, C.byref(npairs)), "nl_number_of_pairs")
        return int(npairs.value)

    def key_pointer(self):
        """neighlist_cpu.hpp:449-455 (view, valid until the next build).  int32 like the reference's: raises
        NL_ERR_INDEX_OVERFLOW for a list of more than INT32_MAX entries (use key_pointer64)."""
        kp, _, _, _ = self._half()
        return _as_tensor(kp, (self._n_rows + 1,), "<i4", self, self.device)

    def key_pointer64(self):
        """The same offsets as int64 (nl_get_half_csr64): for lists beyond the reference's int32 limit."""
        kp, _, _, _ = self._half(64)
        return _as_tensor(kp, (self._n_rows + 1,), "<i8", self, self.device)

    def sorted_list(self):
        """neighlist_cpu.hpp:441-447 (view, valid until the next build)."""
        _, sl, _, npairs = self._half(0 if npairs_exceeds_int32(self) else 32)
        return _as_tensor(sl, (npairs,), "<i4", self, self.device)

    def half_number_of_partners(self):
        """neighlist_cpu.hpp:457-463 (view, valid until the next build)."""
        _, _, nop, _ = self._half(0 if npairs_exceeds_int32(self) else 32)
        return _as_tensor(nop, (self._n_rows,), "<i4", self, self.device)

    def list_checksum(self):
        """Order-independent checksum of the last list, computed on the device (nl_list_checksum): the pair-set hash of
        the known answers.  Returns (checksum, entries)."""
        cs, ne = C.c_uint64(), C.c_int64()
        check(self._lib.nl_list_checksum(self._h, C.byref(cs), C.byref(ne)), "nl_list_checksum")
        return int(cs.value), int(ne.value)

    # ------------------------------------------------------------------ a consumer of the list
    def lj_forces(self, q, epsilon=1.0, sigma=1.0, rc_force=None):
        """Truncated Lennard-Jones forces and per-particle energies ``(n, 4) = {fx, fy, fz, pe_i}`` from the list of
        the last build (nl_lj_forces): gather per row after a full-list build, pair-once with atomics after a half
        build."""
        n = self._check_q(q, None)
        if n != self._n:
            raise ValueError("q must hold the particles the list was built from")
        f = torch.empty((n, 4), dtype=self.dtype, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self._lib.nl_lj_forces(self._h, q.data_ptr(), q.shape[1], float(epsilon), float(sigma),
                                     float(self.search_length if rc_force is None else rc_force), f.data_ptr(), stream),
              "nl_lj_forces")
        return f

    # ------------------------------------------------------------------ periodic re-sorting (SORT_FREQ)
    def cell_order(self):
        """order[s] = input index of the particle at cell-ordered slot s of the last build (nl_get_cell_order; the
        reference's ptcl_id_in_mesh).  View, valid until the next build."""
        ptr, n = C.c_void_p(), C.c_int32()
        check(self._lib.nl_get_cell_order(self._h, C.byref(ptr), C.byref(n)), "nl_get_cell_order")
        return _as_tensor(ptr.value, (n.value,), "<i4", self, self.device)

    def resort(self, *arrays):
        """Permutes per-particle device arrays in place into the cell order of the last build (nl_resort): the re-sort
        the reference declares (SORT_FREQ, CopyGather, SortPtclData) and never calls.  Rebuild afterwards."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for t in arrays:
            if not isinstance(t, torch.Tensor) or t.device.type != "cuda" or not t.is_contiguous() or t.shape[0] != self._n:
                raise TypeError("resort() takes contiguous device tensors with one leading entry per particle")
            check(self._lib.nl_resort(self._h, t.data_ptr(), t.element_size() * (t.numel() // max(t.shape[0], 1)), stream), "nl_resort")

    # ------------------------------------------------------------------ introspection
    def sorted_state(self):
        """(cell_start, sorted_row) of the last build -- for tests of the hash/sort stage.  In a fine-row build
        (build_info()['fine_rows'] > 0) the table has 4 entries per cell: [(row * mx + cx) * 4 + quarter], the four
        fine rows (quarters of the cell along z) of every row of x-cells."""
        cs, sp, sr, ncl = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self._lib.nl_get_sorted(self._h, C.byref(cs), C.byref(sp), C.byref(sr), C.byref(ncl)), "nl_get_sorted")
        entries = ncl.value * (4 if self.build_info()["fine_rows"] else 1) + 1
        return (_as_tensor(cs.value, (entries,), "<i4", self, self.device),
                _as_tensor(sr.value, (self._n,), "<i4", self, self.device))

    def profile_stages(self, q, reps=10):
        """Average device milliseconds per pipeline stage (HIP events on the launch stream)."""
        n = self._check_q(q, None)
        ms = (C.c_double * _lib.NL_NUM_STAGES)()
        check(self._lib.nl_profile_stages(self._h, q.data_ptr(), q.shape[1], n, int(reps), C.byref(ms)),
              "nl_profile_stages")
        self._n = self._n_rows = n
        return dict(zip(_lib.STAGE_NAMES, (float(v) for v in ms)))


    def build_info(self):
        """{'masks': bool, 'variant': int, 'lds_batch': int, 'cus': int, 'offset_bits': 32 | 64, 'mask_rows': int,
        'fine_rows': 0 | 1 + RowsCfg} of the last build (masks: the list was expanded from hit masks; mask_rows > 1: dense
        build; fine_rows: the fine-row search of nl_rows.hpp)."""
        info = (C.c_int32 * 8)()
        check(self._lib.nl_get_build_info(self._h, C.byref(info)), "nl_get_build_info")
        return {"masks": bool(info[0]), "variant": int(info[1]), "lds_batch": int(info[2]), "cus": int(info[3]),
                "offset_bits": int(info[4]), "mask_rows": int(info[5]), "fine_rows": int(info[6]),
                "small_cells": int(info[7])}

    def profile_last_build(self, reps=10):
        """Same for the last build (also a slab build); its position/id tensors are kept alive by this object."""
        ms = (C.c_double * _lib.NL_NUM_STAGES)()
        check(self._lib.nl_profile_last_build(self._h, int(reps), C.byref(ms)), "nl_profile_last_build")
        return dict(zip(_lib.STAGE_NAMES, (float(v) for v in ms)))


def npairs_exceeds_int32(nl) -> bool:
    """True when the last list is too long for int32 offsets (then the int32 key_pointer accessor is not asked for)."""
    n = C.c_int64()
    check(nl._lib.nl_number_of_pairs(nl._h, C.byref(n)), "nl_number_of_pairs")
    return (2 if nl.full_list else 1) * int(n.value) > 2147483647


def device_count() -> int:
    c = C.c_int()
    check(_lib.load().nl_device_count(C.byref(c)))
    return int(c.value)
