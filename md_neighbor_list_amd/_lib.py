"""ctypes binding of libnl_hip.so (include/nl_hip.h).  There is no fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NL_HIP_LIB: another build of the same library (kernel A/B runs, tools/ab_libs.sh); never a different implementation
LIB_PATH = os.environ.get("NL_HIP_LIB") or os.path.join(_HERE, "lib", "libnl_hip.so")

NL_F32, NL_F64 = 0, 1
NL_OK = 0
(NL_ERR_ARG, NL_ERR_NOMEM, NL_ERR_OUT_OF_BOX, NL_ERR_CAPACITY, NL_ERR_HIP, NL_ERR_STATE, NL_ERR_MESH,
 NL_ERR_INDEX_OVERFLOW, NL_ERR_NO_DEVICE, NL_ERR_DOMAIN, NL_ERR_COMM) = range(1, 12)
NL_UNIQUE_ID_BYTES = 128
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t)
_P, _I32, _I64, _SZ, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_double
NL_NUM_STAGES = 7
STAGE_NAMES = ("hash", "cell_scan", "reorder", "count", "row_scan", "fill", "total")

# every symbol include/nl_hip.h declares: (name, restype, argtypes)
PROTOTYPES = {
    "nl_status_string": (C.c_char_p, [C.c_int]),
    "nl_create": (C.c_int, [C.POINTER(_P), C.c_int, _D, _D, _D, _D, C.c_int]),
    "nl_initialize": (C.c_int, [_P, _I32]),
    "nl_set_capacity": (C.c_int, [_P, _I64]),
    "nl_set_list_kind": (C.c_int, [_P, C.c_int]),
    "nl_set_periodic": (C.c_int, [_P, C.c_int]),
    "nl_set_graph": (C.c_int, [_P, C.c_int]),
    "nl_destroy": (C.c_int, [_P]),
    "nl_make_list": (C.c_int, [_P, _P, _I32, _I32, _P, C.c_int]),
    "nl_make_list_slab": (C.c_int, [_P, _P, _I32, _P, _I32, _I32, _I32, _I32, _P, C.c_int]),
    "nl_make_list_slab_begin": (C.c_int, [_P, _P, _I32, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "nl_make_list_slab_finish": (C.c_int, [_P, _P, C.c_int]),
    "nl_synchronize": (C.c_int, [_P]),
    "nl_comm_unique_id": (C.c_int, [_P]),
    "nl_comm_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, _P, C.c_int]),
    "nl_comm_create_callbacks": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, SENDRECV_FN, _P, C.c_int]),
    "nl_comm_destroy": (C.c_int, [_P]),
    "nl_comm_layers": (C.c_int, [_P, _P, C.POINTER(_I32), C.POINTER(_I32)]),
    "nl_make_list_distributed": (C.c_int, [_P, _P, _P, _I32, _I32, _P, C.c_int]),
    "nl_distributed_ghosts": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "nl_get_half_csr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    "nl_get_full_csr": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    "nl_get_half_csr64": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    "nl_get_full_csr64": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    "nl_list_checksum": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(_I64)]),
    "nl_set_offset_width": (C.c_int, [_P, C.c_int]),
    "nl_get_cell_order": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I32)]),
    "nl_resort": (C.c_int, [_P, _P, _SZ, _P]),
    "nl_lj_forces": (C.c_int, [_P, _P, _I32, _D, _D, _D, _P, _P]),
    "nl_get_full_transposed": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64), C.POINTER(_I32)]),
    "nl_number_of_pairs": (C.c_int, [_P, C.POINTER(_I64)]),
    "nl_get_mesh": (C.c_int, [_P, C.POINTER(_I32 * 3), C.POINTER(_I64)]),
    "nl_get_sorted": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64)]),
    "nl_debug_read": (C.c_int, [_P, C.c_void_p, _I32, C.c_int]),
    "nl_debug_occupancy": (C.c_int, [C.POINTER(_I32 * 8)]),
    "nl_get_build_info": (C.c_int, [_P, C.POINTER(_I32 * 8)]),
    "nl_last_error": (C.c_int, [_P]),
    "nl_last_hip_error": (C.c_int, [_P]),
    "nl_profile_stages": (C.c_int, [_P, _P, _I32, _I32, _I32, C.POINTER(_D * NL_NUM_STAGES)]),
    "nl_profile_last_build": (C.c_int, [_P, _I32, C.POINTER(_D * NL_NUM_STAGES)]),
    "nl_buf_alloc": (C.c_int, [C.POINTER(_P), C.POINTER(_P), _SZ]),
    "nl_buf_free": (C.c_int, [_P, _P]),
    "nl_buf_h2d": (C.c_int, [_P, _P, _SZ]),
    "nl_buf_d2h": (C.c_int, [_P, _P, _SZ]),
    "nl_buf_fill32": (C.c_int, [_P, C.c_uint32, _SZ]),
    "nl_buf_fill64": (C.c_int, [_P, C.c_uint64, _SZ]),
    "nl_device_synchronize": (C.c_int, []),
    "nl_device_count": (C.c_int, [C.POINTER(C.c_int)]),
}


class NLError(RuntimeError):
    """A non-zero nl_status from the C ABI."""

    def __init__(self, code, what=""):
        self.code = int(code)
        try:
            msg = load().nl_status_string(self.code).decode()
        except Exception:  # pragma: no cover
            msg = "?"
        super().__init__(f"libnl_hip: status {self.code} ({msg}){' in ' + what if what else ''}")


_lib = None


def load():
    """Loads libnl_hip.so and checks every declared symbol.  Raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built (run `make lib` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(lib, name)  # AttributeError = symbol missing = broken build
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


def check(code, what=""):
    if code != NL_OK:
        raise NLError(code, what)
