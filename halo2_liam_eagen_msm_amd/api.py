"""Host-side mirror of the reference's interface for the MSM witness path.

Names, argument meaning and error behaviour follow the reference crate
(paths relative to /root/reference):

  best_multiexp(coeffs, bases)              halo2::arithmetic::best_multiexp, imported
                                            src/argument_witness_calc.rs:20, called :144
  compute_lhs_witness(scalars, pts, base)   src/argument_witness_calc.rs:87-136 (MSM core; returns the
                                            per-digit carries from which the Rust side builds the
                                            divisor witnesses of :129)
  negbase_decompose(x, base)                src/negbase_utils.rs:20-36
  precompute_multiplicities(pt, base)       src/argument_witness_calc.rs:43-51
  logb_ceil, order, num_digits              src/argument_witness_calc.rs:32-40,54-56,89-91
  id_by_digit, digit_by_id                  src/negbase_utils.rs:46-56

Everything that touches points or batches of scalars runs on the GPU through the C ABI
(include/lemsm.h); there is no CPU fallback.  Data formats are the C ABI's: scalars (n,32)
uint8 canonical little-endian, field elements 4 x uint64 Montgomery limbs, affine (n,8),
Jacobian (n,12).
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import BN254_G1, GRUMPKIN

ORDER = {
    BN254_G1: 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    GRUMPKIN: 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
}
CURVE_IDS = {"bn254_g1": BN254_G1, "grumpkin": GRUMPKIN}


class LemsmError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"lemsm status {status}: {msg}")
        self.status = status


class LengthMismatch(LemsmError, AssertionError):
    """reference: assert!(scalars.len() == pts.len(), "incompatible amount of coefficients") (:88)"""


class ScalarOutOfRange(LemsmError, AssertionError):
    """reference: assert!(&x < &sq_p) (:97)"""

    def __init__(self, status, msg, index):
        super().__init__(status, msg)
        self.index = index


class BadBase(LemsmError, ValueError):
    pass


class TooManyDigits(LemsmError, AssertionError):
    """reference: assert!(digits.len() <= num_digits) (src/negbase_utils.rs:81)"""

    def __init__(self, status, msg, index):
        super().__init__(status, msg)
        self.index = index


class RefIndexOutOfBounds(LemsmError, IndexError):
    """reference: slice index out of bounds at src/negbase_utils.rs:98-101 (i % logtable + 1 > num_limbs)"""

    def __init__(self, status, msg, index):
        super().__init__(status, msg)
        self.index = index


class SumNotIdentity(LemsmError, AssertionError):
    """reference: `if tmp.1 != C::identity() {panic!()}` (src/regular_functions_utils.rs:478)"""


class RefArithmeticOverflow(LemsmError, OverflowError):
    """reference (debug build): `attempt to multiply / add with overflow` in pow or += at src/negbase_utils.rs:97-101"""

    def __init__(self, status, msg, index):
        super().__init__(status, msg)
        self.index = index


# one Entry of prepare_scalar_witness (src/negbase_utils.rs:39-43) as the C ABI lays it out: 24 bytes
ENTRY_DTYPE = np.dtype([("lo", "<u8"), ("hi", "<i8"), ("mask", "<u4"), ("kind", "<u4")])
ENTRY_KINDS = ("Scalar", "Bucket", "Limb")


def _curve_id(curve) -> int:
    if isinstance(curve, str):
        return CURVE_IDS[curve]
    return int(curve)


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _scalars(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.size % 32:
        raise ValueError("scalars must be n x 32 bytes")
    return a.reshape(-1, 32)


def _limbs(a, width: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.size % width:
        raise ValueError(f"points must be n x {width} uint64 limbs")
    return a.reshape(-1, width)


class DeviceBuffer:
    """GPU memory owned by a Context (for inputs that stay resident across calls)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        ctx._check(ctx.lib.lemsm_device_alloc(ctx.h, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.lemsm_device_upload(self.ctx.h, self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, dtype=np.uint8, count: Optional[int] = None) -> np.ndarray:
        nbytes = self.nbytes if count is None else count
        out = np.empty(nbytes, np.uint8)
        self.ctx._check(self.ctx.lib.lemsm_device_download(self.ctx.h, _ptr(out), self.ptr, nbytes))
        return out.view(dtype)

    def free(self):
        if self.ptr:
            self.ctx.lib.lemsm_device_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream + workspace (lemsm_ctx)."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self.lib.lemsm_create(int(device), ctypes.byref(h))
        if rc != _lib.LEMSM_OK:
            raise LemsmError(rc, "lemsm_create failed: no usable gfx950 device (the MSM path has no CPU fallback)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.lemsm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, bad_index: Optional[int] = None):
        if rc == _lib.LEMSM_OK:
            return
        msg = self.lib.lemsm_last_error(self.h).decode() or self.lib.lemsm_strerror(rc).decode()
        if rc == _lib.LEMSM_ERR_LEN_MISMATCH:
            raise LengthMismatch(rc, "incompatible amount of coefficients")
        if rc == _lib.LEMSM_ERR_SCALAR_OUT_OF_RANGE:
            if bad_index is None:
                bad_index = int(self.lib.lemsm_last_bad_index(self.h))
            raise ScalarOutOfRange(rc, msg, bad_index)
        if rc == _lib.LEMSM_ERR_BAD_BASE:
            raise BadBase(rc, msg)
        if rc == _lib.LEMSM_ERR_SUM_NOT_IDENTITY:
            raise SumNotIdentity(rc, msg)
        if rc in (_lib.LEMSM_ERR_TOO_MANY_DIGITS, _lib.LEMSM_ERR_INDEX_OUT_OF_BOUNDS, _lib.LEMSM_ERR_ARITH_OVERFLOW):
            if bad_index is None:
                bad_index = int(self.lib.lemsm_last_bad_index(self.h))
            cls = {_lib.LEMSM_ERR_TOO_MANY_DIGITS: TooManyDigits, _lib.LEMSM_ERR_INDEX_OUT_OF_BOUNDS: RefIndexOutOfBounds,
                   _lib.LEMSM_ERR_ARITH_OVERFLOW: RefArithmeticOverflow}[rc]
            raise cls(rc, msg, bad_index)
        raise LemsmError(rc, msg)

    def set_option(self, name: str, value: int):
        self._check(self.lib.lemsm_set_option(self.h, name.encode(), int(value)))

    def last_timing(self) -> Tuple[float, float, int]:
        out = (ctypes.c_double * 3)()
        self._check(self.lib.lemsm_last_timing(self.h, out))
        return out[0], out[1], int(out[2])

    def divisor_last_reuse_levels(self) -> int:
        """levels of the last divisor-witness forest that transformed onto the odd half of their domain only (test aid)"""
        out = ctypes.c_uint32()
        self._check(self.lib.lemsm_debug_divisor_last_reuse_levels(self.h, ctypes.byref(out)))
        return out.value

    def last_merge_counts(self) -> Tuple[int, int, int, int]:
        """edge-record merge of the last MSM call: buckets queued as short (3..8 pieces), medium (9..32), slices of
        long ones, multi-slice buckets (test / profiling aid)"""
        out = (ctypes.c_uint64 * 4)()
        self._check(self.lib.lemsm_debug_last_merge_counts(self.h, out))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def last_accum_clock_mhz(self) -> float:
        """shader clock (MHz) the accumulate kernel of the last MSM call sustained (in-kernel stamps)"""
        out = ctypes.c_double()
        self._check(self.lib.lemsm_last_accum_clock_mhz(self.h, ctypes.byref(out)))
        return out.value

    def last_truncated_count(self) -> int:
        """scalars of the last negabase pass whose expansion did not fit d digits (silently
        truncated like the reference's `.take(d)`, src/argument_witness_calc.rs:99)"""
        return int(self.lib.lemsm_last_truncated_count(self.h))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr: np.ndarray) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 16)).upload(arr)

    # ---- best_multiexp -------------------------------------------------------------
    def msm(self, curve, scalars, points_affine) -> np.ndarray:
        cid = _curve_id(curve)
        s = _scalars(scalars)
        p = _limbs(points_affine, 8)
        if s.shape[0] != p.shape[0]:
            raise LengthMismatch(_lib.LEMSM_ERR_LEN_MISMATCH, "incompatible amount of coefficients")
        out = np.zeros(12, np.uint64)
        self._check(self.lib.lemsm_msm(self.h, cid, _ptr(s), _ptr(p), s.shape[0], _ptr(out)))
        return out

    def msm_device(self, curve, d_scalars: int, d_points: int, n: int) -> np.ndarray:
        out = np.zeros(12, np.uint64)
        self._check(self.lib.lemsm_msm_device(self.h, _curve_id(curve), d_scalars, d_points, n, _ptr(out)))
        return out

    def msm_batch_device(self, curve, d_scalars_list, d_points: int, n: int) -> np.ndarray:
        """len(d_scalars_list) MSMs over the same resident points, pipelined inside the library (lemsm_msm_batch_device);
        returns (K, 12) Jacobian results"""
        K = len(d_scalars_list)
        out = np.zeros((max(K, 1), 12), np.uint64)
        ptrs = (ctypes.c_void_p * max(K, 1))(*[int(p) for p in d_scalars_list])
        self._check(self.lib.lemsm_msm_batch_device(self.h, _curve_id(curve), ptrs, K, d_points, n, _ptr(out)))
        return out[:K]

    def msm_plan(self, curve, n: int) -> Tuple[int, int]:
        w = ctypes.c_uint32()
        b = ctypes.c_size_t()
        self._check(self.lib.lemsm_msm_plan(self.h, _curve_id(curve), n, ctypes.byref(w), ctypes.byref(b)))
        return w.value, b.value

    def msm_partial_device(self, curve, d_scalars: int, d_points: int, n: int, win_begin: int, win_end: int) -> np.ndarray:
        _, rec = self.msm_plan(curve, n)
        out = np.zeros((win_end - win_begin) * rec, np.uint8)
        self._check(self.lib.lemsm_msm_partial_device(self.h, _curve_id(curve), d_scalars, d_points, n, win_begin, win_end,
                                                      _ptr(out) if out.size else None))
        return out

    def msm_combine(self, curve, n: int, partials: np.ndarray) -> np.ndarray:
        partials = np.ascontiguousarray(partials, np.uint8)
        out = np.zeros(12, np.uint64)
        self._check(self.lib.lemsm_msm_combine(self.h, _curve_id(curve), n, _ptr(partials), _ptr(out)))
        return out

    # ---- multi-GPU (RCCL behind the C ABI) ---------------------------------------------
    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        """collective: every rank calls it with rank 0's comm_unique_id() (lemsm_comm_init = ncclCommInitRank)"""
        buf = np.frombuffer(bytes(unique_id), np.uint8).copy()
        assert buf.size == _lib.LEMSM_COMM_ID_BYTES
        self._check(self.lib.lemsm_comm_init(self.h, _ptr(buf), int(nranks), int(rank)))

    def comm_destroy(self):
        self._check(self.lib.lemsm_comm_destroy(self.h))

    def comm_info(self) -> Tuple[int, int]:
        n = ctypes.c_int(); r = ctypes.c_int()
        self._check(self.lib.lemsm_comm_info(self.h, ctypes.byref(n), ctypes.byref(r)))
        return n.value, r.value

    def msm_sharded_device(self, curve, d_scalars: int, d_points: int, n: int) -> np.ndarray:
        """collective over the context's communicator: this rank's Pippenger windows, one ncclAllGather of the raw
        device records, the combine; every rank returns the same Jacobian point"""
        out = np.zeros(12, np.uint64)
        self._check(self.lib.lemsm_msm_sharded_device(self.h, _curve_id(curve), d_scalars, d_points, n, _ptr(out)))
        return out

    def lhs_msm_sharded_device(self, curve, d_scalars: int, d_points_affine: int, n: int, base: int, want_carries: bool = True):
        cid = _curve_id(curve)
        d = num_digits(cid, base)
        carry = np.zeros(12, np.uint64)
        carries = np.zeros((d, 12), np.uint64) if want_carries else None
        bad = ctypes.c_size_t(0)
        rc = self.lib.lemsm_lhs_msm_sharded_device(self.h, cid, d_scalars, d_points_affine, n, base, _ptr(carry),
                                                   _ptr(carries) if want_carries else None, ctypes.byref(bad))
        self._check(rc, bad.value)
        return carry, carries

    def debug_msm_sharded_sim(self, curve, d_scalars: int, d_points: int, n: int, world: int) -> np.ndarray:
        out = np.zeros(12, np.uint64)
        self._check(self.lib.lemsm_debug_msm_sharded_sim(self.h, _curve_id(curve), d_scalars, d_points, n, world, _ptr(out)))
        return out

    def debug_lhs_sharded_sim(self, curve, d_scalars: int, d_points_affine: int, n: int, base: int, world: int):
        cid = _curve_id(curve)
        d = num_digits(cid, base)
        carry = np.zeros(12, np.uint64); carries = np.zeros((d, 12), np.uint64)
        bad = ctypes.c_size_t(0)
        rc = self.lib.lemsm_debug_lhs_sharded_sim(self.h, cid, d_scalars, d_points_affine, n, base, world, _ptr(carry), _ptr(carries),
                                                  ctypes.byref(bad))
        self._check(rc, bad.value)
        return carry, carries

    # ---- resident bases ------------------------------------------------------------------
    def bases_upload(self, curve, points_affine) -> "Bases":
        return Bases(self, curve, points_affine)

    def msm_with_bases(self, bases: "Bases", scalars) -> np.ndarray:
        s = _scalars(scalars)
        out = np.zeros(12, np.uint64)
        self._check(self.lib.lemsm_msm_with_bases(self.h, bases.h, _ptr(s), s.shape[0], _ptr(out)))
        return out

    def msm_batch_with_bases(self, bases: "Bases", scalars_list) -> np.ndarray:
        """len(scalars_list) MSMs over the same resident bases, scalar vectors in host memory, uploads pipelined with the
        compute inside the library (lemsm_msm_batch_with_bases); returns (K, 12) Jacobian results"""
        ss = [_scalars(x) for x in scalars_list]
        K = len(ss)
        n = ss[0].shape[0] if K else 0
        if any(x.shape[0] != n for x in ss):
            raise ValueError("all scalar vectors of a batch have the same length")
        out = np.zeros((max(K, 1), 12), np.uint64)
        ptrs = (ctypes.c_void_p * max(K, 1))(*[x.ctypes.data for x in ss])
        self._check(self.lib.lemsm_msm_batch_with_bases(self.h, bases.h, ptrs, K, n, _ptr(out)))
        return out[:K]

    # ---- negabase ------------------------------------------------------------------
    def negbase_decompose_batch(self, scalars, base: int, d: int) -> np.ndarray:
        s = _scalars(scalars)
        out = np.zeros((s.shape[0], d), np.uint8)
        self._check(self.lib.lemsm_negbase_decompose_batch(self.h, _ptr(s), s.shape[0], base, d, _ptr(out)))
        return out

    # ---- divisor witness ------------------------------------------------------------------
    def divisor_witness(self, curve, points_affine, require_zero_sum: bool = True, normalise: bool = True):
        """compute_divisor_witness{,_partial} (src/regular_functions_utils.rs:453-480) on the GPU.
        Returns (a, b, out_point): coefficient arrays (len, 4) of raw Montgomery limbs -- lengths exactly the reference's --
        and the tree's output point (affine, zeros = identity)."""
        p = _limbs(points_affine, 8)
        n = p.shape[0]
        cap = n + 4
        a = np.zeros((cap, 4), np.uint64); b = np.zeros((cap, 4), np.uint64)
        la = ctypes.c_size_t(); lb = ctypes.c_size_t()
        outp = np.zeros(8, np.uint64)
        self._check(self.lib.lemsm_divisor_witness(self.h, _curve_id(curve), _ptr(p) if n else None, n, int(require_zero_sum), int(normalise),
                                                   _ptr(a), cap, ctypes.byref(la), _ptr(b), cap, ctypes.byref(lb), _ptr(outp)))
        return a[: la.value].copy(), b[: lb.value].copy(), outp

    def divisor_witness_batch(self, curve, lists, require_zero_sum: bool = True, normalise: bool = True):
        """several point lists in one batch (lemsm_divisor_witness_batch); returns [(a, b, out_point)] per list"""
        arrs = [_limbs(l, 8) if len(l) else np.zeros((0, 8), np.uint64) for l in lists]
        T = len(arrs)
        counts = np.array([a.shape[0] for a in arrs], np.uintp)
        pts = np.concatenate(arrs) if T and counts.sum() else np.zeros((0, 8), np.uint64)
        cap = 2 * int(counts.sum()) + 4 * T + 4
        coeffs = np.zeros((cap, 4), np.uint64); index = np.zeros((max(T, 1), 4), np.uintp); outp = np.zeros((max(T, 1), 8), np.uint64)
        szp = ctypes.POINTER(ctypes.c_size_t)
        self._check(self.lib.lemsm_divisor_witness_batch(self.h, _curve_id(curve), _ptr(pts) if pts.size else None, counts.ctypes.data_as(szp), T,
                                                         int(require_zero_sum), int(normalise), _ptr(coeffs), cap, index.ctypes.data_as(szp), _ptr(outp)))
        out = []
        for t in range(T):
            oa, la, ob, lb = (int(v) for v in index[t])
            out.append((coeffs[oa: oa + la].copy(), coeffs[ob: ob + lb].copy(), outp[t].copy()))
        return out

    def divisor_last_ntt(self) -> Tuple[float, int, int]:
        """(device ms, algorithmic bytes, butterflies) of the transforms of the last divisor-witness call"""
        ms = ctypes.c_double(); by = ctypes.c_uint64(); bf = ctypes.c_uint64()
        self._check(self.lib.lemsm_divisor_last_ntt(self.h, ctypes.byref(ms), ctypes.byref(by), ctypes.byref(bf)))
        return ms.value, by.value, bf.value

    def lhs_witness_last_phases(self) -> Tuple[float, float, float, float]:
        """host-clock ms of the last lhs_witness call: (MSM core, point lists, merge forest, coefficient download)"""
        out = (ctypes.c_double * 4)()
        self._check(self.lib.lemsm_lhs_witness_last_phases(self.h, out))
        return tuple(out)

    def lhs_witness(self, curve, scalars, pts_jacobian, base: int, normalise: bool = True):
        """compute_lhs_witness in full (src/argument_witness_calc.rs:87-136): (carry, [(a, b)] * d) with the functions in the
        reference's (reversed) order."""
        cid = _curve_id(curve)
        s = _scalars(scalars)
        p = _limbs(pts_jacobian, 12)
        if s.shape[0] != p.shape[0]:
            raise LengthMismatch(_lib.LEMSM_ERR_LEN_MISMATCH, "incompatible amount of coefficients")
        if not (3 <= base <= 255):
            raise BadBase(_lib.LEMSM_ERR_BAD_BASE, "base must be in 3..=255")
        n = s.shape[0]
        d = num_digits(cid, base)
        cap = 2 * d * (n + base + 3)
        coeffs = np.empty((cap, 4), np.uint64)     # untouched pages cost nothing; the functions below are views into it
        index = np.zeros((d, 4), np.uintp)
        carry = np.zeros(12, np.uint64)
        bad = ctypes.c_size_t(0)
        rc = self.lib.lemsm_lhs_witness(self.h, cid, _ptr(s) if n else None, _ptr(p) if n else None, n, base, _ptr(carry), _ptr(coeffs), cap,
                                        index.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)), int(normalise), ctypes.byref(bad))
        self._check(rc, bad.value)
        fns = []
        for f in range(d):
            oa, la, ob, lb = (int(v) for v in index[f])
            fns.append((coeffs[oa: oa + la], coeffs[ob: ob + lb]))
        return carry, fns

    def lhs_witness_device(self, curve, d_scalars: int, d_points_affine: int, n: int, base: int, normalise: bool = True,
                           out: Optional[DeviceBuffer] = None, f_range: Optional[Tuple[int, int]] = None):
        """compute_lhs_witness in full with scalars / affine points resident in HBM and the coefficients left there:
        (carry, index[d, 4] = {offset_a, len_a, offset_b, len_b} in 32-byte elements, DeviceBuffer of the coefficients).
        `out`: a buffer of at least 2 d (n + base + 3) elements to reuse across calls.  `f_range` = (begin, end): only
        those functions (a rank's share; the others' index rows read length 0)."""
        cid = _curve_id(curve)
        if not (3 <= base <= 255):
            raise BadBase(_lib.LEMSM_ERR_BAD_BASE, "base must be in 3..=255")
        d = num_digits(cid, base)
        cap = 2 * d * (n + base + 3)
        if out is None:
            out = DeviceBuffer(self, cap * 32)
        assert out.nbytes >= cap * 32
        index = np.zeros((d, 4), np.uintp)
        carry = np.zeros(12, np.uint64)
        bad = ctypes.c_size_t(0)
        if f_range is None:
            rc = self.lib.lemsm_lhs_witness_device(self.h, cid, d_scalars, d_points_affine, n, base, _ptr(carry), out.ptr, cap,
                                                   index.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)), int(normalise), ctypes.byref(bad))
        else:
            rc = self.lib.lemsm_lhs_witness_device_range(self.h, cid, d_scalars, d_points_affine, n, base, int(f_range[0]), int(f_range[1]), _ptr(carry),
                                                         out.ptr, cap, index.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t)), int(normalise), ctypes.byref(bad))
        self._check(rc, bad.value)
        return carry, index, out

    def debug_ntt(self, data, logn: int, inverse: bool = False) -> np.ndarray:
        a = _limbs(data, 4)
        assert a.shape[0] % (1 << logn) == 0
        out = np.zeros_like(a)
        self._check(self.lib.lemsm_debug_ntt(self.h, _ptr(a), _ptr(out), a.shape[0] >> logn, logn, int(inverse)))
        return out

    # ---- prepare_scalar_witness / table_entry_by_id ---------------------------------------
    def prepare_scalar_witness_batch(self, scalars, negative, base: int, num_digits: int, logtable: int) -> np.ndarray:
        """(n, base, num_limbs+1) array of ENTRY_DTYPE: prepare_scalar_witness (src/negbase_utils.rs:79-124) of every
        scalar; scalars are (n, 32) little-endian magnitudes, `negative` optional (n,) flags"""
        s = _scalars(scalars)
        n = s.shape[0]
        if logtable <= 0:
            raise LemsmError(_lib.LEMSM_ERR_BAD_ARG, "logtable == 0: the reference divides by it (:82)")
        cols = (num_digits + logtable - 1) // logtable + 1
        out = np.zeros((n, base, cols), ENTRY_DTYPE)
        neg = None if negative is None else np.ascontiguousarray(negative, np.uint8).reshape(n)
        bad = ctypes.c_size_t(0)
        rc = self.lib.lemsm_prepare_scalar_witness_batch(self.h, _ptr(s), _ptr(neg) if neg is not None else None, n, base, num_digits,
                                                         logtable, _ptr(out) if out.size else None, ctypes.byref(bad))
        self._check(rc, bad.value)
        return out

    def table_entries(self, curve, base: int, id_begin: int, count: int) -> np.ndarray:
        """(count, 4) raw Montgomery limbs of table_entry_by_id(base, id) (src/negbase_utils.rs:58-77), id_begin <= id <
        id_begin + count, in the BASE field of `curve` (the circuit's native field)"""
        out = np.zeros((count, 4), np.uint64)
        self._check(self.lib.lemsm_table_entries(self.h, _curve_id(curve), base, id_begin, count, _ptr(out) if count else None))
        return out

    # ---- compute_lhs_witness MSM core -------------------------------------------------
    def lhs_msm(self, curve, scalars, pts_jacobian, base: int, want_carries: bool = True):
        cid = _curve_id(curve)
        s = _scalars(scalars)
        p = _limbs(pts_jacobian, 12)
        if s.shape[0] != p.shape[0]:
            raise LengthMismatch(_lib.LEMSM_ERR_LEN_MISMATCH, "incompatible amount of coefficients")
        if not (3 <= base <= 255):
            raise BadBase(_lib.LEMSM_ERR_BAD_BASE, "base must be in 3..=255")
        d = num_digits(cid, base)
        carry = np.zeros(12, np.uint64)
        carries = np.zeros((d, 12), np.uint64) if want_carries else None
        bad = ctypes.c_size_t(0)
        rc = self.lib.lemsm_lhs_msm(self.h, cid, _ptr(s), _ptr(p), s.shape[0], base, _ptr(carry),
                                    _ptr(carries) if want_carries else None, ctypes.byref(bad))
        self._check(rc, bad.value)
        return carry, carries

    def lhs_msm_device(self, curve, d_scalars: int, d_points_affine: int, n: int, base: int, want_carries: bool = True):
        cid = _curve_id(curve)
        d = num_digits(cid, base)
        carry = np.zeros(12, np.uint64)
        carries = np.zeros((d, 12), np.uint64) if want_carries else None
        bad = ctypes.c_size_t(0)
        rc = self.lib.lemsm_lhs_msm_device(self.h, cid, d_scalars, d_points_affine, n, base, _ptr(carry),
                                           _ptr(carries) if want_carries else None, ctypes.byref(bad))
        self._check(rc, bad.value)
        return carry, carries

    def lhs_plan(self, curve, base: int) -> Tuple[int, int]:
        d = ctypes.c_uint32()
        b = ctypes.c_size_t()
        self._check(self.lib.lemsm_lhs_plan(_curve_id(curve), base, ctypes.byref(d), ctypes.byref(b)))
        return d.value, b.value

    def lhs_partial_device(self, curve, d_scalars: int, d_points_affine: int, n: int, base: int, pos_begin: int, pos_end: int) -> np.ndarray:
        _, rec = self.lhs_plan(curve, base)
        out = np.zeros(max(pos_end - pos_begin, 0) * rec, np.uint8)
        bad = ctypes.c_size_t(0)
        rc = self.lib.lemsm_lhs_partial_device(self.h, _curve_id(curve), d_scalars, d_points_affine, n, base, pos_begin, pos_end,
                                               _ptr(out) if out.size else None, ctypes.byref(bad))
        self._check(rc, bad.value)
        return out

    def lhs_combine(self, curve, base: int, partials: np.ndarray, want_carries: bool = True):
        cid = _curve_id(curve)
        d = num_digits(cid, base)
        partials = np.ascontiguousarray(partials, np.uint8)
        carry = np.zeros(12, np.uint64)
        carries = np.zeros((d, 12), np.uint64) if want_carries else None
        self._check(self.lib.lemsm_lhs_combine(cid, base, _ptr(partials), _ptr(carry), _ptr(carries) if want_carries else None))
        return carry, carries

    def precompute_multiplicities(self, curve, pts_jacobian, base: int) -> np.ndarray:
        p = _limbs(pts_jacobian, 12)
        out = np.zeros((p.shape[0], max(base - 1, 0), 12), np.uint64)
        self._check(self.lib.lemsm_precompute_multiplicities(self.h, _curve_id(curve), _ptr(p), p.shape[0], base, _ptr(out)))
        return out

    def precompute_multiplicities_affine(self, curve, pts_jacobian, base: int) -> np.ndarray:
        """(n, base-1, 8): affine k*P_j, the table src/config.rs:542-560 fills."""
        p = _limbs(pts_jacobian, 12)
        out = np.zeros((p.shape[0], max(base - 1, 0), 8), np.uint64)
        self._check(self.lib.lemsm_precompute_multiplicities_affine(self.h, _curve_id(curve), _ptr(p), p.shape[0], base, _ptr(out)))
        return out

    def gen_walk(self, curve, q_affine: np.ndarray, n: int) -> DeviceBuffer:
        q = np.ascontiguousarray(q_affine, np.uint64).reshape(8)
        buf = self.alloc(max(n * 64, 16))
        self._check(self.lib.lemsm_device_gen_walk(self.h, _curve_id(curve), _ptr(q), n, buf.ptr))
        return buf

    # ---- debug hooks -----------------------------------------------------------------
    def debug_montmul(self, curve, a, b) -> np.ndarray:
        a = _limbs(a, 4); b = _limbs(b, 4)
        out = np.zeros_like(a)
        self._check(self.lib.lemsm_debug_montmul(self.h, _curve_id(curve), _ptr(a), _ptr(b), _ptr(out), a.shape[0]))
        return out

    def debug_fieldop(self, curve, op: int, a, b) -> np.ndarray:
        a = _limbs(a, 4); b = _limbs(b, 4)
        out = np.zeros_like(a)
        self._check(self.lib.lemsm_debug_fieldop(self.h, _curve_id(curve), op, _ptr(a), _ptr(b), _ptr(out), a.shape[0]))
        return out

    def debug_pointop(self, curve, op: int, acc_xyzz, q) -> np.ndarray:
        acc = _limbs(acc_xyzz, 16)
        q = _limbs(q, 16 if op == 1 else 8)
        out = np.zeros_like(acc)
        self._check(self.lib.lemsm_debug_pointop(self.h, _curve_id(curve), op, _ptr(acc), _ptr(q), _ptr(out), acc.shape[0]))
        return out


class Bases:
    """affine bases resident on one GPU (lemsm_bases_upload): later MSMs upload only their scalars"""

    def __init__(self, ctx: Context, curve, points_affine):
        p = _limbs(points_affine, 8)
        h = ctypes.c_void_p()
        ctx._check(ctx.lib.lemsm_bases_upload(ctx.h, _curve_id(curve), _ptr(p), p.shape[0], ctypes.byref(h)))
        self.ctx, self.h, self.n = ctx, h, p.shape[0]

    @property
    def ptr(self) -> int:
        return self.ctx.lib.lemsm_bases_device_ptr(self.h)

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.lemsm_bases_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def comm_unique_id() -> bytes:
    """rank 0: the 128-byte RCCL unique id every rank passes to Context.comm_init"""
    buf = np.zeros(_lib.LEMSM_COMM_ID_BYTES, np.uint8)
    rc = _lib.load().lemsm_comm_unique_id(_ptr(buf))
    if rc:
        raise LemsmError(rc, "lemsm_comm_unique_id (is librccl loadable?)")
    return buf.tobytes()


class Node:
    """All GPUs of one node from one process (lemsm_node_*): what a Rust host binds."""

    def __init__(self, devices: Optional[Sequence[int]] = None, ndev: Optional[int] = None):
        self.lib = _lib.load()
        if devices is not None:
            arr = (ctypes.c_int * len(devices))(*devices); nd = len(devices)
        else:
            arr = None; nd = int(ndev or 1)
        h = ctypes.c_void_p()
        rc = self.lib.lemsm_node_create(arr, nd, ctypes.byref(h))
        if rc:
            raise LemsmError(rc, "lemsm_node_create failed")
        self.h = h

    def _check(self, rc: int, bad_index: Optional[int] = None):
        if rc == _lib.LEMSM_OK:
            return
        msg = self.lib.lemsm_node_last_error(self.h).decode() or self.lib.lemsm_strerror(rc).decode()
        if rc == _lib.LEMSM_ERR_LEN_MISMATCH:
            raise LengthMismatch(rc, "incompatible amount of coefficients")
        if rc == _lib.LEMSM_ERR_SCALAR_OUT_OF_RANGE:
            raise ScalarOutOfRange(rc, msg, bad_index if bad_index is not None else 0)
        if rc == _lib.LEMSM_ERR_BAD_BASE:
            raise BadBase(rc, msg)
        raise LemsmError(rc, msg)

    @property
    def size(self) -> int:
        return self.lib.lemsm_node_size(self.h)

    def set_bases(self, curve, points_affine):
        p = _limbs(points_affine, 8)
        self._check(self.lib.lemsm_node_set_bases(self.h, _curve_id(curve), _ptr(p), p.shape[0]))
        self.curve = _curve_id(curve)

    def msm(self, scalars) -> np.ndarray:
        s = _scalars(scalars)
        out = np.zeros(12, np.uint64)
        self._check(self.lib.lemsm_node_msm(self.h, _ptr(s), s.shape[0], _ptr(out)))
        return out

    def lhs_msm(self, scalars, base: int, want_carries: bool = True):
        s = _scalars(scalars)
        if not (3 <= base <= 255):
            raise BadBase(_lib.LEMSM_ERR_BAD_BASE, "base must be in 3..=255")
        d = num_digits(self.curve, base)
        carry = np.zeros(12, np.uint64)
        carries = np.zeros((d, 12), np.uint64) if want_carries else None
        bad = ctypes.c_size_t(0)
        rc = self.lib.lemsm_node_lhs_msm(self.h, _ptr(s), s.shape[0], base, _ptr(carry), _ptr(carries) if want_carries else None,
                                         ctypes.byref(bad))
        self._check(rc, bad.value)
        return carry, carries

    def close(self):
        if getattr(self, "h", None):
            self.lib.lemsm_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def jacobian_to_canonical(curve, jac) -> bytes:
    jac = np.ascontiguousarray(jac, np.uint64).reshape(12)
    out = np.zeros(64, np.uint8)
    rc = _lib.load().lemsm_jacobian_to_canonical(_curve_id(curve), _ptr(jac), _ptr(out))
    if rc:
        raise LemsmError(rc, "jacobian_to_canonical")
    return out.tobytes()


def jacobian_sum(curve, jacs) -> np.ndarray:
    """Host-side sum of Jacobian points (the combine step of a point-sharded MSM; the fold over
    per-thread results inside halo2's best_multiexp)."""
    jacs = np.ascontiguousarray(jacs, np.uint64).reshape(-1, 12)
    out = np.zeros(12, np.uint64)
    rc = _lib.load().lemsm_jacobian_sum(_curve_id(curve), _ptr(jacs), jacs.shape[0], _ptr(out))
    if rc:
        raise LemsmError(rc, "jacobian_sum")
    return out


# ---- scalar helpers of the reference (pure integer logic, no points) ------------------
def order(curve) -> int:                                   # src/argument_witness_calc.rs:54-56
    return ORDER[_curve_id(curve)]


def logb_ceil(x: int, base: int) -> int:                   # src/argument_witness_calc.rs:32-40
    i = 0
    while x > 0:
        x //= base
        i += 1
    return i


def num_digits(curve, base: int) -> int:                   # src/argument_witness_calc.rs:89-91
    d = ctypes.c_uint32()
    rc = _lib.load().lemsm_num_digits(_curve_id(curve), base, ctypes.byref(d))
    if rc:
        raise BadBase(rc, "bad base")
    return d.value


def id_by_digit(digit: int) -> Optional[int]:              # src/negbase_utils.rs:46-51
    return None if digit == 0 else digit - 1


def digit_by_id(idx: int) -> int:                          # src/negbase_utils.rs:54-56
    return idx + 1


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---- reference-named entry points --------------------------------------------------------
def best_multiexp(coeffs, bases, curve="bn254_g1", ctx: Optional[Context] = None) -> np.ndarray:
    """sum_i coeffs[i]*bases[i]; returns a Jacobian point (12 limbs)."""
    return (ctx or default_context()).msm(curve, coeffs, bases)


def compute_lhs_witness(scalars, pts, base: int, curve="grumpkin", ctx: Optional[Context] = None, normalise: bool = True):
    """The reference's return value (C, Vec<RegularFunction<C>>) (src/argument_witness_calc.rs:87, :134): the carry
    sum_j scalars[j] * pts[j] as a Jacobian point and the d divisor witnesses of :129 as (a, b) coefficient arrays
    (raw Montgomery limbs), in the reference's reversed order (:132).  A RegularFunction is defined up to a scalar
    (linefunc takes whatever projective representative a point has); normalise=True returns the representative whose
    coefficient of highest pole order is 1."""
    return (ctx or default_context()).lhs_witness(curve, scalars, pts, base, normalise)


def compute_divisor_witness(pts_affine, curve="grumpkin", ctx: Optional[Context] = None, normalise: bool = True):
    """src/regular_functions_utils.rs:476-480: (a, b) of the regular function vanishing on the points; raises
    SumNotIdentity where the reference panics (:478)."""
    a, b, _ = (ctx or default_context()).divisor_witness(curve, pts_affine, True, normalise)
    return a, b


def compute_divisor_witness_partial(pts_affine, curve="grumpkin", ctx: Optional[Context] = None, normalise: bool = True):
    """src/regular_functions_utils.rs:453-467: ((a, b), output point as affine raw limbs, zeros = identity)"""
    a, b, out = (ctx or default_context()).divisor_witness(curve, pts_affine, False, normalise)
    return (a, b), out


def _neg_affine_raw(curve_id: int, pts: np.ndarray) -> np.ndarray:
    """-(x, y) = (x, p - y) on raw Montgomery limbs (negation commutes with the Montgomery factor);
    the identity (0, 0) stays (0, 0)."""
    fp = ORDER[GRUMPKIN] if curve_id == BN254_G1 else ORDER[BN254_G1]   # base field of one = scalar field of the other
    out = np.array(pts, np.uint64).reshape(-1, 8).copy()
    for r in out:
        y = int.from_bytes(r[4:].tobytes(), "little")
        if y:
            r[4:] = np.frombuffer(((fp - y) % fp).to_bytes(32, "little"), np.uint64)
    return out


def compute_lhs_witness_inputs(scalars, pts, base: int, curve="grumpkin", ctx: Optional[Context] = None):
    """(carry, tmp_lists): the point lists the reference collects in `tmp` and hands to
    compute_divisor_witness, one per iteration of its digit loop, MSB first
    (src/argument_witness_calc.rs:108-127; the reference returns the resulting vector reversed, :129):

        tmp_i = [-carry_{i-1}] * base   (only if carry_{i-1} is not the identity, :112-116)
              + [digit_{j,i} * P_j for every j whose digit is non-zero, in order of j]   (:120-124)
              + [-carry_i]                                                               (:127)

    as affine raw-Montgomery limbs, shape (len, 8), identity = (0, 0).  Every group operation behind
    them runs on the GPU -- the per-digit carries (lemsm_lhs_msm), the table of affine multiples
    (lemsm_precompute_multiplicities_affine, which also turns the carries affine) and the digit
    matrix (lemsm_negbase_decompose_batch); what is left here is indexing, so the Rust side can
    build `tmp` without a single serial EC addition.  The polynomial step
    compute_divisor_witness itself stays on the Rust side (out of scope)."""
    c = ctx or default_context()
    cid = _curve_id(curve)
    carry, carries = c.lhs_msm(cid, scalars, pts, base, True)
    d = carries.shape[0]
    digits = c.negbase_decompose_batch(scalars, base, d)                  # (n, d), LSB first
    table = c.precompute_multiplicities_affine(cid, pts, base)            # (n, base-1, 8)
    neg_carries = _neg_affine_raw(cid, c.precompute_multiplicities_affine(cid, carries, 2).reshape(d, 8))
    tmp_lists = []
    for i in range(d):
        dig = digits[:, d - 1 - i].astype(np.int64)                       # the reference reverses the digits (:101)
        nz = np.nonzero(dig)[0]
        parts = []
        if i > 0 and neg_carries[i - 1].any():
            parts.append(np.repeat(neg_carries[i - 1][None, :], base, axis=0))
        parts.append(table[nz, dig[nz] - 1])                              # precomputed_points[j][id_by_digit(digit)]
        parts.append(neg_carries[i][None, :])
        tmp_lists.append(np.concatenate(parts, axis=0))
    return carry, tmp_lists


def negbase_decompose(x: int, base: int, ctx: Optional[Context] = None) -> List[int]:
    """Digits LSB first, no padding, [] for 0 (src/negbase_utils.rs:20-36).  x >= 0 here: the
    reference's callers only pass non-negative scalars (src/argument_witness_calc.rs:99)."""
    if x < 0 or x >= 1 << 256:
        raise ValueError("x must be in [0, 2^256)")
    if base < 2:
        raise BadBase(_lib.LEMSM_ERR_BAD_BASE, "base must be >= 2")
    d = logb_ceil(max(x, 1), base) + 2
    s = np.frombuffer(int(x).to_bytes(32, "little"), np.uint8).reshape(1, 32)
    digs = (ctx or default_context()).negbase_decompose_batch(s, base, d)[0].tolist()
    while digs and digs[-1] == 0:
        digs.pop()
    return digs


def prepare_scalar_witness(sc: int, base: int, num_digits: int, logtable: int, ctx: Optional[Context] = None):
    """src/negbase_utils.rs:79-124: Vec<Vec<Entry>> as a list (base rows) of lists (num_limbs+1) of
    ("Scalar", sc) | ("Bucket", i128) | ("Limb", i128, u32); raises where the reference panics."""
    mag = abs(int(sc))
    if mag >= 1 << 256:
        raise ValueError("|sc| must be < 2^256")
    s = np.frombuffer(mag.to_bytes(32, "little"), np.uint8).reshape(1, 32)
    arr = (ctx or default_context()).prepare_scalar_witness_batch(s, np.array([1 if sc < 0 else 0], np.uint8), base, num_digits, logtable)[0]
    out = []
    for i in range(arr.shape[0]):
        row = []
        for j in range(arr.shape[1]):
            e = arr[i, j]
            v = (int(e["hi"]) << 64) | int(e["lo"])
            kind = ENTRY_KINDS[int(e["kind"])]
            row.append(("Scalar", int(sc)) if kind == "Scalar" else ("Bucket", v) if kind == "Bucket" else ("Limb", v, int(e["mask"])))
        out.append(row)
    return out


def table_entry_by_id(base: int, idx: int, curve="grumpkin", ctx: Optional[Context] = None) -> np.ndarray:
    """src/negbase_utils.rs:58-77 in the base field of `curve`: 4 raw Montgomery limbs"""
    return (ctx or default_context()).table_entries(curve, base, idx, 1)[0]


class WouldNotTerminate(LemsmError):
    """reference: to_curve_x's loop never changes x (src/config.rs:170-173): a non-residue spins forever"""


def to_curve_x(c, curve="grumpkin") -> np.ndarray:
    """src/config.rs:166-175: c (one field element, 4 raw Montgomery limbs of the curve's base field) itself when c^3 + b is a
    square; WouldNotTerminate where the reference would loop forever"""
    v = np.ascontiguousarray(c, np.uint64).reshape(4)
    out = np.zeros(4, np.uint64)
    rc = _lib.load().lemsm_to_curve_x(_curve_id(curve), _ptr(v), _ptr(out))
    if rc == _lib.LEMSM_ERR_WOULD_NOT_TERMINATE:
        raise WouldNotTerminate(rc, "to_curve_x: x^3 + b is not a square and the reference's loop never changes x")
    if rc:
        raise LemsmError(rc, "to_curve_x")
    return out


def y_from_x(x, curve="grumpkin") -> Tuple[np.ndarray, bool]:
    """src/config.rs:177-182: (y, is_square) = sqrt_alt(x^3 + b) in the curve's base field (see include/lemsm.h for the root chosen)"""
    v = np.ascontiguousarray(x, np.uint64).reshape(4)
    out = np.zeros(4, np.uint64)
    flag = ctypes.c_int(0)
    rc = _lib.load().lemsm_y_from_x(_curve_id(curve), _ptr(v), _ptr(out), ctypes.byref(flag))
    if rc:
        raise LemsmError(rc, "y_from_x")
    return out, bool(flag.value)


def slope(xy, curve="grumpkin") -> np.ndarray:
    """src/config.rs:184-187: (3 x^2 + a) / (2 y) of an affine point (8 limbs); ZeroDivisionError where the reference's
    invert().unwrap() panics"""
    v = np.ascontiguousarray(xy, np.uint64).reshape(8)
    out = np.zeros(4, np.uint64)
    rc = _lib.load().lemsm_slope(_curve_id(curve), _ptr(v), _ptr(out))
    if rc == _lib.LEMSM_ERR_DIVISION_BY_ZERO:
        raise ZeroDivisionError("slope: y == 0")
    if rc:
        raise LemsmError(rc, "slope")
    return out


def precompute_multiplicities(pt_jacobian, base: int, curve="grumpkin", ctx: Optional[Context] = None) -> np.ndarray:
    """[1*P, .., (base-1)*P] as Jacobian points, shape (base-1, 12)."""
    return (ctx or default_context()).precompute_multiplicities(curve, np.asarray(pt_jacobian).reshape(1, 12), base)[0]
