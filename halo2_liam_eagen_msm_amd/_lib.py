"""ctypes loader for the C-ABI library (include/lemsm.h).

The product path has NO CPU fallback: if liblemsm.so is missing or no gfx950
device is usable, loading / context creation raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "liblemsm.so")
CSRC = os.path.join(_HERE, "csrc")

LEMSM_OK = 0
LEMSM_ERR_LEN_MISMATCH = 1
LEMSM_ERR_SCALAR_OUT_OF_RANGE = 2
LEMSM_ERR_BAD_BASE = 3
LEMSM_ERR_BAD_CURVE = 4
LEMSM_ERR_HIP = 5
LEMSM_ERR_BAD_ARG = 6
LEMSM_ERR_NOMEM = 7
LEMSM_ERR_TOO_MANY_DIGITS = 8
LEMSM_ERR_RCCL = 9
LEMSM_ERR_INDEX_OUT_OF_BOUNDS = 10
LEMSM_ERR_ARITH_OVERFLOW = 11
LEMSM_ERR_SUM_NOT_IDENTITY = 12
LEMSM_ERR_WOULD_NOT_TERMINATE = 13
LEMSM_ERR_DIVISION_BY_ZERO = 14
LEMSM_COMM_ID_BYTES = 128

BN254_G1 = 0
GRUMPKIN = 1

# Every symbol include/lemsm.h declares (tests/test_abi.py checks the header against this list
# and that the built library exports each of them).
SYMBOLS = [
    "lemsm_create", "lemsm_destroy", "lemsm_strerror", "lemsm_last_error", "lemsm_last_bad_index", "lemsm_last_truncated_count", "lemsm_set_option",
    "lemsm_last_timing", "lemsm_last_accum_clock_mhz", "lemsm_debug_last_merge_counts", "lemsm_debug_divisor_last_reuse_levels",
    "lemsm_msm", "lemsm_msm_bn254_g1", "lemsm_msm_grumpkin", "lemsm_msm_device", "lemsm_msm_batch_device",
    "lemsm_msm_plan", "lemsm_msm_partial_device", "lemsm_msm_combine",
    "lemsm_num_digits", "lemsm_negbase_decompose_batch",
    "lemsm_lhs_msm", "lemsm_lhs_msm_grumpkin", "lemsm_lhs_msm_bn254_g1", "lemsm_lhs_msm_device",
    "lemsm_lhs_plan", "lemsm_lhs_partial_device", "lemsm_lhs_combine",
    "lemsm_precompute_multiplicities", "lemsm_precompute_multiplicities_affine",
    "lemsm_jacobian_to_canonical", "lemsm_jacobian_sum",
    "lemsm_device_alloc", "lemsm_device_free", "lemsm_device_upload", "lemsm_device_download",
    "lemsm_device_gen_walk",
    "lemsm_debug_montmul", "lemsm_debug_fieldop", "lemsm_debug_pointop",
    "lemsm_comm_unique_id", "lemsm_comm_init", "lemsm_comm_destroy", "lemsm_comm_info",
    "lemsm_msm_sharded_device", "lemsm_lhs_msm_sharded_device",
    "lemsm_node_create", "lemsm_node_destroy", "lemsm_node_size", "lemsm_node_ctx", "lemsm_node_last_error",
    "lemsm_node_set_bases", "lemsm_node_msm", "lemsm_node_lhs_msm",
    "lemsm_bases_upload", "lemsm_bases_free", "lemsm_bases_device_ptr", "lemsm_msm_with_bases", "lemsm_msm_batch_with_bases",
    "lemsm_debug_msm_sharded_sim", "lemsm_debug_lhs_sharded_sim",
    "lemsm_prepare_scalar_witness_batch", "lemsm_table_entries",
    "lemsm_divisor_witness", "lemsm_divisor_witness_device", "lemsm_divisor_witness_batch", "lemsm_divisor_last_ntt", "lemsm_lhs_witness", "lemsm_lhs_witness_device", "lemsm_lhs_witness_device_range", "lemsm_lhs_witness_last_phases", "lemsm_debug_ntt",
    "lemsm_to_curve_x", "lemsm_y_from_x", "lemsm_slope",
]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "lemsm.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs if os.path.isfile(s)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def load() -> ctypes.CDLL:
    """Load liblemsm.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MSM path has no CPU fallback)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, u8p, u64p = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p
    i, u32p, szp = ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_size_t)
    sig = {
        "lemsm_create": (i, [i, ctypes.POINTER(vp)]),
        "lemsm_destroy": (None, [vp]),
        "lemsm_strerror": (ctypes.c_char_p, [i]),
        "lemsm_last_error": (ctypes.c_char_p, [vp]),
        "lemsm_last_bad_index": (ctypes.c_size_t, [vp]),
        "lemsm_last_truncated_count": (ctypes.c_size_t, [vp]),
        "lemsm_set_option": (i, [vp, ctypes.c_char_p, ctypes.c_long]),
        "lemsm_last_timing": (i, [vp, ctypes.POINTER(ctypes.c_double)]),
        "lemsm_last_accum_clock_mhz": (i, [vp, ctypes.POINTER(ctypes.c_double)]),
        "lemsm_debug_last_merge_counts": (i, [vp, ctypes.POINTER(ctypes.c_uint64)]),
        "lemsm_debug_divisor_last_reuse_levels": (i, [vp, ctypes.POINTER(ctypes.c_uint32)]),
        "lemsm_msm": (i, [vp, i, u8p, u64p, sz, u64p]),
        "lemsm_msm_bn254_g1": (i, [vp, u8p, u64p, sz, u64p]),
        "lemsm_msm_grumpkin": (i, [vp, u8p, u64p, sz, u64p]),
        "lemsm_msm_device": (i, [vp, i, vp, vp, sz, u64p]),
        "lemsm_msm_batch_device": (i, [vp, i, ctypes.POINTER(ctypes.c_void_p), sz, vp, sz, u64p]),
        "lemsm_msm_plan": (i, [vp, i, sz, u32p, szp]),
        "lemsm_msm_partial_device": (i, [vp, i, vp, vp, sz, ctypes.c_uint32, ctypes.c_uint32, u8p]),
        "lemsm_msm_combine": (i, [vp, i, sz, u8p, u64p]),
        "lemsm_num_digits": (i, [i, ctypes.c_uint8, u32p]),
        "lemsm_negbase_decompose_batch": (i, [vp, u8p, sz, ctypes.c_uint8, ctypes.c_uint32, u8p]),
        "lemsm_lhs_msm": (i, [vp, i, u8p, u64p, sz, ctypes.c_uint8, u64p, u64p, szp]),
        "lemsm_lhs_msm_grumpkin": (i, [vp, u8p, u64p, sz, ctypes.c_uint8, u64p, u64p, szp]),
        "lemsm_lhs_msm_bn254_g1": (i, [vp, u8p, u64p, sz, ctypes.c_uint8, u64p, u64p, szp]),
        "lemsm_lhs_msm_device": (i, [vp, i, vp, vp, sz, ctypes.c_uint8, u64p, u64p, szp]),
        "lemsm_lhs_plan": (i, [i, ctypes.c_uint8, u32p, szp]),
        "lemsm_lhs_partial_device": (i, [vp, i, vp, vp, sz, ctypes.c_uint8, ctypes.c_uint32, ctypes.c_uint32, u8p, szp]),
        "lemsm_lhs_combine": (i, [i, ctypes.c_uint8, u8p, u64p, u64p]),
        "lemsm_precompute_multiplicities": (i, [vp, i, u64p, sz, ctypes.c_uint8, u64p]),
        "lemsm_precompute_multiplicities_affine": (i, [vp, i, u64p, sz, ctypes.c_uint8, u64p]),
        "lemsm_jacobian_to_canonical": (i, [i, u64p, u8p]),
        "lemsm_jacobian_sum": (i, [i, u64p, sz, u64p]),
        "lemsm_device_alloc": (i, [vp, sz, ctypes.POINTER(vp)]),
        "lemsm_device_free": (i, [vp, vp]),
        "lemsm_device_upload": (i, [vp, vp, vp, sz]),
        "lemsm_device_download": (i, [vp, vp, vp, sz]),
        "lemsm_device_gen_walk": (i, [vp, i, u64p, sz, vp]),
        "lemsm_debug_montmul": (i, [vp, i, u64p, u64p, u64p, sz]),
        "lemsm_debug_fieldop": (i, [vp, i, i, u64p, u64p, u64p, sz]),
        "lemsm_debug_pointop": (i, [vp, i, i, u64p, u64p, u64p, sz]),
        "lemsm_comm_unique_id": (i, [u8p]),
        "lemsm_comm_init": (i, [vp, u8p, i, i]),
        "lemsm_comm_destroy": (i, [vp]),
        "lemsm_comm_info": (i, [vp, ctypes.POINTER(i), ctypes.POINTER(i)]),
        "lemsm_msm_sharded_device": (i, [vp, i, vp, vp, sz, u64p]),
        "lemsm_lhs_msm_sharded_device": (i, [vp, i, vp, vp, sz, ctypes.c_uint8, u64p, u64p, szp]),
        "lemsm_node_create": (i, [ctypes.POINTER(i), i, ctypes.POINTER(vp)]),
        "lemsm_node_destroy": (None, [vp]),
        "lemsm_node_size": (i, [vp]),
        "lemsm_node_ctx": (vp, [vp, i]),
        "lemsm_node_last_error": (ctypes.c_char_p, [vp]),
        "lemsm_node_set_bases": (i, [vp, i, u64p, sz]),
        "lemsm_node_msm": (i, [vp, u8p, sz, u64p]),
        "lemsm_node_lhs_msm": (i, [vp, u8p, sz, ctypes.c_uint8, u64p, u64p, szp]),
        "lemsm_bases_upload": (i, [vp, i, u64p, sz, ctypes.POINTER(vp)]),
        "lemsm_bases_free": (None, [vp]),
        "lemsm_bases_device_ptr": (vp, [vp]),
        "lemsm_msm_with_bases": (i, [vp, vp, u8p, sz, u64p]),
        "lemsm_msm_batch_with_bases": (i, [vp, vp, vp, sz, sz, u64p]),
        "lemsm_debug_msm_sharded_sim": (i, [vp, i, vp, vp, sz, i, u64p]),
        "lemsm_debug_lhs_sharded_sim": (i, [vp, i, vp, vp, sz, ctypes.c_uint8, i, u64p, u64p, szp]),
        "lemsm_prepare_scalar_witness_batch": (i, [vp, u8p, u8p, sz, ctypes.c_uint8, ctypes.c_uint32, ctypes.c_uint32, u8p, szp]),
        "lemsm_table_entries": (i, [vp, i, ctypes.c_uint8, ctypes.c_uint64, sz, u64p]),
        "lemsm_divisor_witness": (i, [vp, i, u64p, sz, i, i, u64p, sz, szp, u64p, sz, szp, u64p]),
        "lemsm_divisor_witness_device": (i, [vp, i, vp, sz, i, i, u64p, sz, szp, u64p, sz, szp, u64p]),
        "lemsm_divisor_witness_batch": (i, [vp, i, u64p, szp, sz, i, i, u64p, sz, szp, u64p]),
        "lemsm_lhs_witness_last_phases": (i, [vp, ctypes.POINTER(ctypes.c_double)]),
        "lemsm_divisor_last_ntt": (i, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
        "lemsm_lhs_witness": (i, [vp, i, u8p, u64p, sz, ctypes.c_uint8, u64p, u64p, sz, szp, i, szp]),
        "lemsm_lhs_witness_device": (i, [vp, i, vp, vp, sz, ctypes.c_uint8, u64p, vp, sz, szp, i, szp]),
        "lemsm_lhs_witness_device_range": (i, [vp, i, vp, vp, sz, ctypes.c_uint8, ctypes.c_uint32, ctypes.c_uint32, u64p, vp, sz, szp, i, szp]),
        "lemsm_debug_ntt": (i, [vp, u64p, u64p, sz, ctypes.c_uint32, i]),
        "lemsm_to_curve_x": (i, [i, u64p, u64p]),
        "lemsm_y_from_x": (i, [i, u64p, u64p, ctypes.POINTER(i)]),
        "lemsm_slope": (i, [i, u64p, u64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
