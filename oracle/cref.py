"""ctypes binding of the C oracle (oracle/c/lemsm_oracle.c).

TEST INFRASTRUCTURE ONLY - see the header of the C file.  Used by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LEMSM_ORACLE_LIB: an alternative build of the same source, e.g. `make -C oracle asan` run under LD_PRELOAD=libasan (tests/test_host_logic.py)
_LIB_PATH = os.environ.get("LEMSM_ORACLE_LIB") or os.path.join(_HERE, "_build", "liblemsm_oracle.so")

ORC_OK, ORC_LEN_MISMATCH, ORC_SCALAR_OUT_OF_RANGE, ORC_BAD_BASE, ORC_BAD_CURVE = 0, 1, 2, 3, 4

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "c", "lemsm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p8(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


def _p64(a: np.ndarray):
    return a.ctypes.data_as(_u64p)


def _check(rc: int, what: str):
    if rc != ORC_OK:
        raise ValueError(f"{what}: oracle error {rc}")


def field_consts(cid: int):
    mod = np.zeros(4, np.uint64); R = np.zeros(4, np.uint64); R2 = np.zeros(4, np.uint64)
    inv = ctypes.c_uint64()
    _check(lib().orc_field_consts(cid, _p64(mod), ctypes.byref(inv), _p64(R), _p64(R2)), "field_consts")
    return mod, inv.value, R, R2


def montmul(cid: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros(4, np.uint64)
    a = np.ascontiguousarray(a, np.uint64); b = np.ascontiguousarray(b, np.uint64)
    _check(lib().orc_montmul(cid, _p64(a), _p64(b), _p64(out)), "montmul")
    return out


def generator(cid: int) -> np.ndarray:
    out = np.zeros(8, np.uint64)
    _check(lib().orc_generator(cid, _p64(out)), "generator")
    return out


def is_on_curve(cid: int, pt_aff: np.ndarray) -> bool:
    pt_aff = np.ascontiguousarray(pt_aff, np.uint64)
    return bool(lib().orc_is_on_curve_aff(cid, _p64(pt_aff)))


def jac_to_canonical(cid: int, jac: np.ndarray) -> bytes:
    jac = np.ascontiguousarray(jac, np.uint64)
    assert jac.size == 12
    out = np.zeros(64, np.uint8)
    _check(lib().orc_jac_to_canonical(cid, _p64(jac), _p8(out)), "jac_to_canonical")
    return out.tobytes()


def jac_to_aff_raw(cid: int, jac: np.ndarray) -> np.ndarray:
    jac = np.ascontiguousarray(jac, np.uint64)
    out = np.zeros(8, np.uint64)
    _check(lib().orc_jac_to_aff_raw(cid, _p64(jac), _p64(out)), "jac_to_aff_raw")
    return out


def jac_add(cid: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros(12, np.uint64)
    a = np.ascontiguousarray(a, np.uint64); b = np.ascontiguousarray(b, np.uint64)
    _check(lib().orc_jac_add(cid, _p64(a), _p64(b), _p64(out)), "jac_add")
    return out


def jac_eq(cid: int, a: np.ndarray, b: np.ndarray) -> bool:
    a = np.ascontiguousarray(a, np.uint64); b = np.ascontiguousarray(b, np.uint64)
    return bool(lib().orc_jac_eq(cid, _p64(a), _p64(b)))


def scalar_mul(cid: int, k: int, pt_aff: np.ndarray) -> np.ndarray:
    kb = np.frombuffer(int(k).to_bytes(32, "little"), np.uint8).copy()
    pt_aff = np.ascontiguousarray(pt_aff, np.uint64)
    out = np.zeros(12, np.uint64)
    _check(lib().orc_scalar_mul(cid, _p8(kb), _p64(pt_aff), _p64(out)), "scalar_mul")
    return out


def gen_points(cid: int, seed: int, n: int) -> np.ndarray:
    """n affine points k_i*G (raw Montgomery, shape (n, 8) uint64)."""
    out = np.zeros((max(n, 1), 8), np.uint64)
    _check(lib().orc_gen_points(cid, ctypes.c_uint64(seed), ctypes.c_size_t(n), _p64(out)), "gen_points")
    return out[:n]


def gen_walk(cid: int, q_aff: np.ndarray, n: int) -> np.ndarray:
    """P_i = (i+1)*Q, affine raw Montgomery (n, 8)."""
    q_aff = np.ascontiguousarray(q_aff, np.uint64)
    out = np.zeros((max(n, 1), 8), np.uint64)
    _check(lib().orc_gen_walk(cid, _p64(q_aff), ctypes.c_size_t(n), _p64(out)), "gen_walk")
    return out[:n]


def walk_dot(cid: int, scalars: np.ndarray) -> int:
    """sum_i s_i*(i+1) mod order."""
    scalars = np.ascontiguousarray(scalars, np.uint8).reshape(-1, 32)
    out = np.zeros(32, np.uint8)
    _check(lib().orc_walk_dot(cid, _p8(scalars), ctypes.c_size_t(scalars.shape[0]), _p8(out)), "walk_dot")
    return int.from_bytes(out.tobytes(), "little")


def _limbs4(x: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(32, "little"), np.uint64).copy()


def order_of(cid: int) -> int:
    from . import pyref
    return {0: pyref.BN254_G1.order, 1: pyref.GRUMPKIN.order}[cid]


def gen_scalars(cid: int, seed: int, n: int, half: bool = False) -> np.ndarray:
    """(n, 32) uint8 canonical LE scalars; full = mod order, half = mod isqrt(order)."""
    order = order_of(cid)
    modulus = math.isqrt(order) if half else order
    out = np.zeros((max(n, 1), 32), np.uint8)
    m = _limbs4(modulus)
    _check(lib().orc_gen_scalars(cid, ctypes.c_uint64(seed), ctypes.c_size_t(n), _p64(m), _p8(out)), "gen_scalars")
    return out[:n]


def negbase_decompose(x: int, base: int):
    xb = np.frombuffer(int(x).to_bytes(32, "little"), np.uint8).copy()
    digits = np.zeros(300, np.uint8)
    ln = ctypes.c_size_t()
    _check(lib().orc_negbase_decompose(_p8(xb), ctypes.c_uint8(base), _p8(digits), ctypes.c_size_t(300), ctypes.byref(ln)), "negbase")
    return digits[: ln.value].tolist()


def num_digits(cid: int, base: int) -> Tuple[int, int]:
    d = ctypes.c_uint32(); bound = np.zeros(4, np.uint64)
    _check(lib().orc_num_digits(cid, ctypes.c_uint8(base), ctypes.byref(d), _p64(bound)), "num_digits")
    return d.value, int.from_bytes(bound.tobytes(), "little")


def negbase_decompose_batch(scalars: np.ndarray, base: int, d: int) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, np.uint8).reshape(-1, 32)
    n = scalars.shape[0]
    out = np.zeros((max(n, 1), d), np.uint8)
    _check(lib().orc_negbase_decompose_batch(_p8(scalars), ctypes.c_size_t(n), ctypes.c_uint8(base), ctypes.c_uint32(d), _p8(out)), "negbase_batch")
    return out[:n]


def precompute_multiplicities(cid: int, pt_jac: np.ndarray, base: int) -> np.ndarray:
    pt_jac = np.ascontiguousarray(pt_jac, np.uint64)
    out = np.zeros((base - 1, 12), np.uint64)
    _check(lib().orc_precompute_multiplicities(cid, _p64(pt_jac), ctypes.c_uint8(base), _p64(out)), "precompute_multiplicities")
    return out


def lhs_msm(cid: int, scalars: np.ndarray, pts_jac: np.ndarray, base: int, want_carries: bool = True):
    """Returns (carry[12], carries[d,12] | None).  Raises ValueError like the reference's asserts."""
    scalars = np.ascontiguousarray(scalars, np.uint8).reshape(-1, 32)
    pts_jac = np.ascontiguousarray(pts_jac, np.uint64).reshape(-1, 12)
    if scalars.shape[0] != pts_jac.shape[0]:
        raise ValueError("incompatible amount of coefficients")
    n = scalars.shape[0]
    d, _ = num_digits(cid, base)
    carry = np.zeros(12, np.uint64)
    carries = np.zeros((d, 12), np.uint64) if want_carries else None
    bad = ctypes.c_size_t()
    rc = lib().orc_lhs_msm(cid, _p8(scalars), _p64(pts_jac), ctypes.c_size_t(n), ctypes.c_uint8(base),
                           _p64(carry), _p64(carries) if want_carries else None, ctypes.byref(bad))
    if rc == ORC_SCALAR_OUT_OF_RANGE:
        raise ValueError(f"scalar {bad.value} out of range")
    _check(rc, "lhs_msm")
    return carry, carries


def best_multiexp(cid: int, scalars: np.ndarray, bases_aff: np.ndarray, threads: int = 1) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, np.uint8).reshape(-1, 32)
    bases_aff = np.ascontiguousarray(bases_aff, np.uint64).reshape(-1, 8)
    if scalars.shape[0] != bases_aff.shape[0]:
        raise ValueError("length mismatch")
    out = np.zeros(12, np.uint64)
    _check(lib().orc_best_multiexp(cid, _p8(scalars), _p64(bases_aff), ctypes.c_size_t(scalars.shape[0]),
                                   ctypes.c_int(threads), _p64(out)), "best_multiexp")
    return out


def msm_naive(cid: int, scalars: np.ndarray, bases_aff: np.ndarray) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, np.uint8).reshape(-1, 32)
    bases_aff = np.ascontiguousarray(bases_aff, np.uint64).reshape(-1, 8)
    out = np.zeros(12, np.uint64)
    _check(lib().orc_msm_naive(cid, _p8(scalars), _p64(bases_aff), ctypes.c_size_t(scalars.shape[0]), _p64(out)), "msm_naive")
    return out


def aff_to_jac(cid: int, pts_aff: np.ndarray) -> np.ndarray:
    """(n,8) affine raw -> (n,12) Jacobian raw with Z = R (Montgomery one); identity -> zeros."""
    pts_aff = np.ascontiguousarray(pts_aff, np.uint64).reshape(-1, 8)
    _, _, R, _ = field_consts(cid)
    out = np.zeros((pts_aff.shape[0], 12), np.uint64)
    out[:, :8] = pts_aff
    nonid = np.any(pts_aff != 0, axis=1)
    out[nonid, 8:12] = R
    return out


# ---- compiled, threaded restatement of the divisor-witness path (oracle/c/witness_oracle.inc) ----------------------
_R_INV = {}


def _from_mont_list(raw: np.ndarray, p: int):
    """(k, 4) raw Montgomery limbs -> list of canonical Python ints"""
    if p not in _R_INV:
        _R_INV[p] = pow(1 << 256, -1, p)
    ri = _R_INV[p]
    b = np.ascontiguousarray(raw, np.uint64).tobytes()
    return [int.from_bytes(b[32 * i:32 * i + 32], "little") * ri % p for i in range(len(b) // 32)]


def divisor_witness(pts_jac: np.ndarray, omega0_mont: np.ndarray, threads: int = 1):
    """compute_divisor_witness over Grumpkin (regular_functions_utils.rs:476-480) -> (status, a, b): status 0 ok,
    1 = the points do not sum to the identity, 2 = the reference's usize underflow; a, b = coefficient lists (canonical ints)"""
    pts = np.ascontiguousarray(pts_jac, np.uint64).reshape(-1, 12)
    n = pts.shape[0]
    cap = 2 * n + 16
    coeffs = np.zeros((cap, 4), np.uint64)
    la = ctypes.c_size_t(); lb = ctypes.c_size_t()
    fn = lib().orc_divisor_witness
    fn.argtypes = [_u64p, ctypes.c_size_t, _u64p, ctypes.c_int, _u64p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    st = fn(_p64(pts), n, _p64(np.ascontiguousarray(omega0_mont, np.uint64)), threads, _p64(coeffs), cap, ctypes.byref(la), ctypes.byref(lb))
    if st == 3:
        raise ValueError("divisor_witness: coefficient buffer too small")
    p = int.from_bytes(field_consts(1)[0].tobytes(), "little")
    vals = _from_mont_list(coeffs[:la.value + lb.value], p)
    return st, vals[:la.value], vals[la.value:]


def lhs_witness(scalars: np.ndarray, pts_jac: np.ndarray, base: int, omega0_mont: np.ndarray, threads: int = 1, decode: bool = True):
    """compute_lhs_witness over Grumpkin (argument_witness_calc.rs:87-136) -> (status, carry (12 limbs), [(a, b)] per returned
    function); decode=False skips the conversion of the coefficients to Python ints (timing runs) and returns the lengths"""
    sc = np.ascontiguousarray(scalars, np.uint8).reshape(-1, 32)
    pts = np.ascontiguousarray(pts_jac, np.uint64).reshape(-1, 12)
    n = sc.shape[0]
    d, _ = num_digits(1, base)
    cap = d * (n + base + 8) + 64
    coeffs = np.zeros((cap, 4), np.uint64)
    lens = np.zeros(2 * d, np.uint64)
    carry = np.zeros(12, np.uint64)
    fn = lib().orc_lhs_witness
    fn.argtypes = [_u8p, _u64p, ctypes.c_size_t, ctypes.c_uint8, _u64p, ctypes.c_int, _u64p, _u64p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    st = fn(_p8(sc), _p64(pts), n, base, _p64(np.ascontiguousarray(omega0_mont, np.uint64)), threads, _p64(carry), _p64(coeffs), cap,
            lens.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)))
    if st == 3:
        raise ValueError("lhs_witness: coefficient buffer too small")
    if not decode:
        return st, carry, [(int(lens[2 * f]), int(lens[2 * f + 1])) for f in range(d)]
    p = int.from_bytes(field_consts(1)[0].tobytes(), "little")
    out = []; off = 0
    for f in range(d):
        la, lb = int(lens[2 * f]), int(lens[2 * f + 1])
        vals = _from_mont_list(coeffs[off:off + la + lb], p)
        out.append((vals[:la], vals[la:])); off += la + lb
    return st, carry, out
