"""Deterministic synthetic particle boxes (host side), via libnl_inputs.so.

``uniform_box``  -- the benchmark workload of SURVEY.md section 8(d): uniform random positions in
                    [0,L)^3 from std::mt19937_64(seed), index = generation order.
``fcc_box``      -- the reference harness's jittered FCC lattice (make_list.cpp:34-77).

Positions are returned as ``(N, 4)`` arrays ``x, y, z, 0`` -- the float4/double4 ``Vec`` layout the
reference GPU harness uses (make_list.cu:6-12).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libnl_inputs.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` or `make`"
            )
        lib = C.CDLL(_LIB_PATH)
        lib.nl_box_length.restype = C.c_double
        lib.nl_box_length.argtypes = [C.c_int64, C.c_double]
        for name in ("nl_gen_uniform_f32", "nl_gen_uniform_f64"):
            f = getattr(lib, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_uint64]
        for name in ("nl_gen_fcc_f32", "nl_gen_fcc_f64"):
            f = getattr(lib, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_double]
        _lib = lib
    return _lib


def _suffix(dtype) -> str:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"positions must be float32 or float64, not {dtype}")


def box_length(n: int, density: float) -> float:
    """L = cbrt(N / rho) in double, as the benchmark configs define it."""
    return float(_load().nl_box_length(int(n), float(density)))


def uniform_box(n: int, density: float = 1.0, dtype=np.float32, seed: int = 12345, box=None):
    """Returns ``(q, (Lx, Ly, Lz))`` with ``q`` of shape (n, 4)."""
    lib = _load()
    if box is None:
        L = box_length(n, density)
        box = (L, L, L)
    q = np.zeros((n, 4), dtype=dtype)
    getattr(lib, "nl_gen_uniform_" + _suffix(dtype))(
        q.ctypes.data, 4, n, float(box[0]), float(box[1]), float(box[2]), int(seed)
    )
    return q, tuple(float(b) for b in box)


def weak_scaling_box(n_ranks: int, n_per_rank: int = 1 << 20, density: float = 1.0, dtype=np.float32, seed: int = 12345):
    """The box of a weak-scaling run over ``n_ranks`` z-slabs: the single-GPU cube (side cbrt(n_per_rank / rho))
    repeated ``n_ranks`` times along z, filled with n_ranks * n_per_rank uniform random particles -- every rank's slab
    is the single-GPU problem plus its two ghost layers, whatever the number of ranks.  n_ranks = 1 is uniform_box."""
    L = box_length(n_per_rank, density)
    return uniform_box(n_ranks * n_per_rank, density, dtype, seed, box=(L, L, n_ranks * L))


def fcc_box(density: float = 1.0, L: float = 50.0, dtype=np.float64):
    """Returns ``(q, (L, L, L))``; N = 4 * int(L / s)**3 (119 164 at rho=1, 62 500 at rho=0.5)."""
    lib = _load()
    gen = getattr(lib, "nl_gen_fcc_" + _suffix(dtype))
    n = int(gen(None, 4, 0, float(density), float(L)))
    q = np.zeros((n, 4), dtype=dtype)
    gen(q.ctypes.data, 4, n, float(density), float(L))
    return q, (float(L), float(L), float(L))
