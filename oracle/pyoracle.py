"""ctypes front end of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this module; the
product package ``md_neighbor_list_amd`` never does (tests/test_boundary.py greps for that).

Two back ends:
  * ``build`` / ``bruteforce`` / ``cells``  -- the C restatement, oracle/liboracle.so (always available)
  * ``ref_build``                            -- the REFERENCE's own class compiled from /root/reference into
                                               oracle/_ref/ (present where oracle/Makefile could build it)
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE = os.path.join(HERE, "liboracle.so")
_REF_DIR = os.path.join(HERE, "_ref")

ERRORS = {1: "bad argument", 2: "out of memory", 3: "particle outside the box (reference: out-of-bounds hash)", 4: "list buffer too small"}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code}: {ERRORS.get(code, '?')}")
        self.code = code


@dataclass
class HalfList:
    """Half neighbour list in the reference's CSR form (neighlist_cpu.hpp:437-463)."""

    number_of_partners: np.ndarray  # int32 [N]
    key_pointer: np.ndarray  # int64 [N+1]
    sorted_list: np.ndarray  # int32 [P]

    @property
    def npairs(self) -> int:
        return int(self.key_pointer[-1])

    def canonical(self) -> "HalfList":
        lst = self.sorted_list.copy()
        kp = np.ascontiguousarray(self.key_pointer, dtype=np.int64)
        _lib().nl_oracle_canonicalize(len(kp) - 1, kp.ctypes.data, lst.ctypes.data)
        return HalfList(self.number_of_partners, kp, lst)

    def hash(self) -> int:
        kp = np.ascontiguousarray(self.key_pointer, dtype=np.int64)
        lst = np.ascontiguousarray(self.sorted_list, dtype=np.int32)
        return int(_lib().nl_oracle_hash(len(kp) - 1, kp.ctypes.data, lst.ctypes.data))


_cache = {}


def _lib():
    if "o" not in _cache:
        if not os.path.exists(_ORACLE):
            raise RuntimeError(f"{_ORACLE} missing: run `make -C oracle`")
        lib = C.CDLL(_ORACLE)
        lib.nl_oracle_hash.restype = C.c_uint64
        lib.nl_oracle_hash.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
        lib.nl_oracle_hash_transposed.restype = C.c_uint64
        lib.nl_oracle_hash_transposed.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.nl_oracle_canonicalize.restype = None
        lib.nl_oracle_canonicalize.argtypes = [C.c_int64, C.c_void_p, C.c_void_p]
        lib.nl_oracle_free.restype = None
        lib.nl_oracle_free.argtypes = [C.c_void_p]
        for s in ("f32", "f64"):
            f = getattr(lib, "nl_oracle_build_" + s)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                          C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
            f = getattr(lib, "nl_oracle_build_pbc_" + s)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                          C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
            f = getattr(lib, "nl_oracle_build_pbc_full_" + s)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                          C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
            f = getattr(lib, "nl_oracle_bruteforce_" + s)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_int,
                          C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
            f = getattr(lib, "nl_oracle_count_" + s)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                          C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
            f = getattr(lib, "nl_oracle_cells_" + s)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                          C.c_void_p, C.c_void_p]
        _cache["o"] = lib
    return _cache["o"]


def _prep(q):
    q = np.ascontiguousarray(q)
    if q.ndim != 2 or q.shape[1] < 3 or q.dtype not in (np.float32, np.float64):
        raise TypeError("q must be (N, >=3) float32/float64")
    return q, ("f32" if q.dtype == np.float32 else "f64")


def _take(ptr, n):
    out = np.empty(n, dtype=np.int32)
    if n:
        C.memmove(out.ctypes.data, ptr.value, 4 * n)
    _lib().nl_oracle_free(ptr)
    return out


def build(q, rc, box) -> HalfList:
    """The restated NeighList<Vec>::MakeNeighList (neighlist_cpu.hpp:417-435); visit order preserved."""
    q, s = _prep(q)
    n = q.shape[0]
    nop = np.zeros(n, dtype=np.int32)
    kp = np.zeros(n + 1, dtype=np.int64)
    ptr, npairs = C.c_void_p(), C.c_int64()
    rc_ = getattr(_lib(), "nl_oracle_build_" + s)(q.ctypes.data, q.shape[1], n, rc, box[0], box[1], box[2],
                                                  nop.ctypes.data, kp.ctypes.data, C.byref(ptr), C.byref(npairs))
    if rc_:
        raise OracleError(rc_)
    return HalfList(nop, kp, _take(ptr, npairs.value))


def count(q, rc, box, slab_of_layer=None):
    """Count / hash mode of the restated build (nl_oracle_count): nothing is stored but number_of_partners, and per
    slab of z cell layers the pair count and the pair-set hash of the rows it owns.  For boxes beyond the reference's
    own limits (BASELINE configs 4 and 5).  Returns (number_of_partners[N], pairs_per_slab, hash_per_slab, npairs)."""
    q, s = _prep(q)
    n = q.shape[0]
    nop = np.zeros(n, dtype=np.int32)
    if slab_of_layer is None:
        nslab, sol_ptr = 1, None
    else:
        sol = np.ascontiguousarray(slab_of_layer, dtype=np.int32)
        nslab, sol_ptr = int(sol.max()) + 1, sol.ctypes.data
    hashes = np.zeros(nslab, dtype=np.uint64)
    pairs = np.zeros(nslab, dtype=np.int64)
    npairs = C.c_int64()
    rc_ = getattr(_lib(), "nl_oracle_count_" + s)(q.ctypes.data, q.shape[1], n, rc, box[0], box[1], box[2],
                                                  nop.ctypes.data, nslab, sol_ptr, hashes.ctypes.data, pairs.ctypes.data,
                                                  C.byref(npairs))
    if rc_:
        raise OracleError(rc_)
    return nop, pairs, hashes, int(npairs.value)


def build_pbc(q, rc, box) -> HalfList:
    """Minimum-image half list: the definition nl_set_periodic(1) is tested against; canonical
    order.  Not a restatement of the reference, which has no such mode."""
    q, s = _prep(q)
    n = q.shape[0]
    nop = np.zeros(n, dtype=np.int32)
    kp = np.zeros(n + 1, dtype=np.int64)
    ptr, npairs = C.c_void_p(), C.c_int64()
    rc_ = getattr(_lib(), "nl_oracle_build_pbc_" + s)(q.ctypes.data, q.shape[1], n, rc, box[0], box[1], box[2],
                                                      nop.ctypes.data, kp.ctypes.data, C.byref(ptr), C.byref(npairs))
    if rc_:
        raise OracleError(rc_)
    return HalfList(nop, kp, _take(ptr, npairs.value))


def build_pbc_full(q, rc, box) -> HalfList:
    """Minimum-image FULL list, row i evaluated in the frame of particle i (the ghost-particle semantics; see
    nl_oracle_impl.h): a directed CSR, canonical order.  Not the symmetrised half list."""
    q, s = _prep(q)
    n = q.shape[0]
    nop = np.zeros(n, dtype=np.int32)
    kp = np.zeros(n + 1, dtype=np.int64)
    ptr, npairs = C.c_void_p(), C.c_int64()
    rc_ = getattr(_lib(), "nl_oracle_build_pbc_full_" + s)(q.ctypes.data, q.shape[1], n, rc, box[0], box[1], box[2],
                                                           nop.ctypes.data, kp.ctypes.data, C.byref(ptr), C.byref(npairs))
    if rc_:
        raise OracleError(rc_)
    return HalfList(nop, kp, _take(ptr, npairs.value))


def bruteforce(q, rc, rc2_in_position_type=False) -> HalfList:
    """make_neighlist_bruteforce + make_sorted_list (make_list.cpp:79-118). O(N^2): keep N small."""
    q, s = _prep(q)
    n = q.shape[0]
    nop = np.zeros(n, dtype=np.int32)
    kp = np.zeros(n + 1, dtype=np.int64)
    ptr, npairs = C.c_void_p(), C.c_int64()
    rc_ = getattr(_lib(), "nl_oracle_bruteforce_" + s)(q.ctypes.data, q.shape[1], n, rc, int(rc2_in_position_type),
                                                       nop.ctypes.data, kp.ctypes.data, C.byref(ptr), C.byref(npairs))
    if rc_:
        raise OracleError(rc_)
    return HalfList(nop, kp, _take(ptr, npairs.value))


def cells(q, rc, box):
    """Cell id per particle (GenHash, neighlist_cpu.hpp:51-59) and the mesh; -1 = reference out of bounds."""
    q, s = _prep(q)
    n = q.shape[0]
    cell = np.zeros(n, dtype=np.int32)
    mesh = np.zeros(3, dtype=np.int32)
    rc_ = getattr(_lib(), "nl_oracle_cells_" + s)(q.ctypes.data, q.shape[1], n, rc, box[0], box[1], box[2],
                                                  cell.ctypes.data, mesh.ctypes.data)
    if rc_:
        raise OracleError(rc_)
    return cell, mesh


def hash_transposed(count, lst, row_stride):
    """Pair-set hash and half-pair count of a GPU-style transposed full list (make_list.cu:178-198 layout)."""
    count = np.ascontiguousarray(count, dtype=np.int32)
    lst = np.ascontiguousarray(lst, dtype=np.int32)
    nhalf = C.c_int64()
    h = _lib().nl_oracle_hash_transposed(len(count), row_stride, count.ctypes.data, lst.ctypes.data, C.byref(nhalf))
    return int(h), int(nhalf.value)


# ---------------------------------------------------------------- compiled reference (oracle/_ref)
REF_VARIANTS = ("naive", "fused", "swp", "fused_native", "avx2_4x1", "avx512_8x1")


def ref_available(variant="fused") -> bool:
    return os.path.exists(os.path.join(_REF_DIR, f"libnl_ref_{variant}.so"))


def _cpu_has(flag) -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def ref_runnable(variant) -> bool:
    """The SIMD/native builds need the ISA they were compiled for (else SIGILL)."""
    if not ref_available(variant):
        return False
    if variant == "avx512_8x1":
        return _cpu_has("avx512f") and _cpu_has("avx512dq") and _cpu_has("avx512vl") and _cpu_has("avx512bw")
    if variant in ("avx2_4x1", "fused_native"):
        return _cpu_has("avx2") and _cpu_has("fma")
    return True


def _ref(variant):
    key = "ref_" + variant
    if key not in _cache:
        lib = C.CDLL(os.path.join(_REF_DIR, f"libnl_ref_{variant}.so"))
        for s in ("f32", "f64"):
            if not hasattr(lib, "nl_ref_build_" + s):
                continue
            f = getattr(lib, "nl_ref_build_" + s)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32,
                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        _cache[key] = lib
    return _cache[key]


def ref_build(q, rc, box, variant="fused", loops=1, want_list=True):
    """Runs the reference class itself.  Returns (HalfList | None, seconds for `loops` builds, npairs)."""
    q, s = _prep(q)
    lib = _ref(variant)
    if not hasattr(lib, "nl_ref_build_" + s):
        raise TypeError(f"reference variant {variant} has no {s} build (SIMD classes are fp64-only)")
    n = q.shape[0]
    nop = np.zeros(n, dtype=np.int32)
    kp = np.zeros(n + 1, dtype=np.int32)
    cap = 100 * max(n, 1)  # the reference's own buffer size (neighlist_cpu.hpp:37,76-78)
    lst = np.zeros(cap if want_list else 1, dtype=np.int32)
    npairs, secs = C.c_int32(), C.c_double()
    rc_ = getattr(lib, "nl_ref_build_" + s)(q.ctypes.data, q.shape[1], n, rc, box[0], box[1], box[2], loops,
                                            nop.ctypes.data, kp.ctypes.data, lst.ctypes.data if want_list else None,
                                            cap, C.byref(npairs), C.byref(secs))
    if rc_:
        raise OracleError(rc_)
    hl = HalfList(nop, kp.astype(np.int64), lst[: npairs.value].copy()) if want_list else None
    return hl, float(secs.value), int(npairs.value)
