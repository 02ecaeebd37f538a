#!/usr/bin/env python3
"""Timing of the GPU divisor witness (lemsm_divisor_witness_device) and of compute_lhs_witness in full.
usage: divisor_timing.py [LOGN ...]     prints per size: wall ms, NTT-stage device ms, NTT algorithmic GB/s"""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context, api, _lib
from bench import gen_scalars, ORDER
import math

logns = [int(x) for x in sys.argv[1:]] or [10, 14, 16, 18, 20]
ctx = Context(0)
r = ORDER["bn254_g1"]       # Grumpkin's base field
gx, gy = 1, 0x2CF135E7506A45D632D270D45F1181294833FC48D823F272C
q = np.zeros(8, np.uint64)
q[:4] = np.frombuffer(((gx << 256) % r).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((gy << 256) % r).to_bytes(32, "little"), np.uint64)
for logn in logns:
    n = 1 << logn
    dp = ctx.gen_walk(1, q, n + 3)           # (k Q): partial witness of the first n points (no zero-sum requirement)
    a = np.zeros((n + 4, 4), np.uint64); b = np.zeros((n + 4, 4), np.uint64)
    la = ctypes.c_size_t(); lb = ctypes.c_size_t(); outp = np.zeros(8, np.uint64)
    def run():
        ctx._check(ctx.lib.lemsm_divisor_witness_device(ctx.h, 1, dp.ptr, n, 0, 1, a.ctypes.data, n + 4, ctypes.byref(la), b.ctypes.data, n + 4, ctypes.byref(lb), outp.ctypes.data))
    run()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
    ms, by, bf = ctx.divisor_last_ntt()
    print("divisor witness 2^%d points: %.2f ms wall (%.1f ns/point), lengths (%d, %d); transforms %.2f ms device: %.2f GB algorithmic -> %.0f GB/s (%.3f of 8 TB/s), %.3g butterflies -> %.1f G/s"
          % (logn, best * 1e3, best * 1e9 / n, la.value, lb.value, ms, by / 1e9, by / ms / 1e6 if ms else 0, by / ms / 1e6 / 8000 if ms else 0, bf, bf / ms / 1e6 if ms else 0), flush=True)
    dp.free()
# compute_lhs_witness in full at 2^14 / 2^16 (host-pointer entry: Jacobian points in, d functions out)
for logn in [x for x in logns if x <= 18][-2:]:
    n = 1 << logn
    order = ORDER["grumpkin"]
    sc = gen_scalars(n, math.isqrt(order), 77 + logn)
    dp = ctx.gen_walk(1, q, n)
    aff = dp.download(np.uint64).reshape(-1, 8)
    jac = np.zeros((n, 12), np.uint64); jac[:, :8] = aff; jac[:, 8:] = np.frombuffer(((1 << 256) % r).to_bytes(32, "little"), np.uint64)
    for base in (16,):
        t0 = time.perf_counter(); carry, fns = ctx.lhs_witness(1, sc, jac, base, True); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); carry, fns = ctx.lhs_witness(1, sc, jac, base, True); dt = min(dt, time.perf_counter() - t0)
        ms, by, bf = ctx.divisor_last_ntt()
        print("compute_lhs_witness 2^%d points base %d: %.1f ms wall for the carry + %d divisor witnesses (%d coefficients); transforms %.1f ms, %.0f GB/s algorithmic, %.1f G butterflies/s"
              % (logn, base, dt * 1e3, len(fns), sum(a.shape[0] + b.shape[0] for a, b in fns), ms, by / ms / 1e6 if ms else 0, bf / ms / 1e6 if ms else 0), flush=True)
    dp.free()
