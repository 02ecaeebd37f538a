"""Test-side construction of the per-window partial records the GPU pipeline produces
(total, U_0..U_{L-1} as XYZZ points), built with the big-int oracle.  Lets the host tail
(lemsm_msm_combine / lemsm_lhs_combine) and the multi-rank exchange be tested without a GPU."""
import ctypes

import numpy as np

from halo2_liam_eagen_msm_amd import _lib
from oracle import pyref


def msm_plan(curve, n):
    lib = _lib.load()
    w = ctypes.c_uint32(); b = ctypes.c_size_t()
    assert lib.lemsm_msm_plan(None, curve.cid, n, ctypes.byref(w), ctypes.byref(b)) == 0
    assert b.value == 128
    # window bits c: the library's rule (choose_c in csrc/lemsm.hip): clamp(floor(log2 n) - 3, 3, 16) below 2^24 pairs
    lg = n.bit_length() - 1
    c = 17 if lg >= 24 else max(3, min(16, lg - 3))
    return w.value, c - 1, b.value   # windows, L (c = L + 1), record bytes


def xyzz_record(curve, pt):
    """affine oracle point -> 128-byte XYZZ record with zz = zzz = 1 (raw Montgomery, R = 2^256)"""
    if pt is None:
        return bytes(128)
    one = curve.to_mont(1).to_bytes(32, "little")
    return curve.to_mont(pt[0]).to_bytes(32, "little") + curve.to_mont(pt[1]).to_bytes(32, "little") + one + one


def window_record(curve, buckets, L):
    """buckets[j] (weight j+1) -> the window sum S = sum_j (j+1) * B_j as one XYZZ record (running sum)"""
    running = None
    acc = None
    for bkt in reversed(buckets):
        running = curve.add(running, bkt)
        acc = curve.add(acc, running)
    return xyzz_record(curve, acc)


def msm_records(curve, scalars, pts, n_for_plan=None):
    """all windows' records for the signed-window Pippenger plan the library uses for n points"""
    n = len(scalars) if n_for_plan is None else n_for_plan
    W, L, rec = msm_plan(curve, n)
    c = L + 1
    nb = 1 << (c - 1)
    K = sum((1 << (c - 1)) << (c * w) for w in range(W - 1))
    out = []
    for w in range(W):
        buckets = [None] * nb
        for s, p in zip(scalars, pts):
            raw = ((s + K) >> (c * w)) & ((1 << c) - 1)
            d = raw - (1 << (c - 1)) if w < W - 1 else raw
            if d == 0:
                continue
            q = p if d > 0 else curve.neg(p)
            buckets[abs(d) - 1] = curve.add(buckets[abs(d) - 1], q)
        out.append(window_record(curve, buckets, L))
    return W, rec, out


def lhs_plan(curve, base):
    lib = _lib.load()
    d = ctypes.c_uint32(); b = ctypes.c_size_t()
    assert lib.lemsm_lhs_plan(curve.cid, base, ctypes.byref(d), ctypes.byref(b)) == 0
    assert b.value == 128
    return d.value, 0, b.value


def lhs_records(curve, scalars, pts, base):
    d, L, rec = lhs_plan(curve, base)
    out = []
    digs = [pyref.negbase_digits_padded(s, base, d) for s in scalars]
    for pos in range(d):
        buckets = [None] * (base - 1)
        for dg, p in zip(digs, pts):
            if dg[pos]:
                buckets[dg[pos] - 1] = curve.add(buckets[dg[pos] - 1], p)
        out.append(window_record(curve, buckets, L))
    return d, rec, out


def msm_combine(curve, n, records_bytes):
    lib = _lib.load()
    buf = np.frombuffer(records_bytes, np.uint8).copy()
    out = np.zeros(12, np.uint64)
    assert lib.lemsm_msm_combine(None, curve.cid, n, buf.ctypes.data, out.ctypes.data) == 0
    return out


def lhs_combine(curve, base, records_bytes):
    lib = _lib.load()
    d, _, _ = lhs_plan(curve, base)
    buf = np.frombuffer(records_bytes, np.uint8).copy()
    carry = np.zeros(12, np.uint64); carries = np.zeros((d, 12), np.uint64)
    assert lib.lemsm_lhs_combine(curve.cid, base, buf.ctypes.data, carry.ctypes.data, carries.ctypes.data) == 0
    return carry, carries
