#!/usr/bin/env python3
"""compute_lhs_witness in full at 2^LOGN points, base 16, twice (for rocprofv3 --kernel-trace --stats). usage: lhs_witness_profile.py LOGN"""
import os, sys, math, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context
from bench import gen_scalars, ORDER
logn = int(sys.argv[1]); n = 1 << logn
ctx = Context(0)
r = ORDER["bn254_g1"]
gx, gy = 1, 0x2CF135E7506A45D632D270D45F1181294833FC48D823F272C
q = np.zeros(8, np.uint64)
q[:4] = np.frombuffer(((gx << 256) % r).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((gy << 256) % r).to_bytes(32, "little"), np.uint64)
sc = gen_scalars(n, math.isqrt(ORDER["grumpkin"]), 77 + logn)
dp = ctx.gen_walk(1, q, n)
aff = dp.download(np.uint64).reshape(-1, 8)
jac = np.zeros((n, 12), np.uint64); jac[:, :8] = aff; jac[:, 8:] = np.frombuffer(((1 << 256) % r).to_bytes(32, "little"), np.uint64)
for _ in range(2):
    t0 = time.perf_counter(); carry, fns = ctx.lhs_witness(1, sc, jac, 16, True); dt = time.perf_counter() - t0
print("2^%d: %.1f ms, %d functions, %d coefficients, ntt %s" % (logn, dt * 1e3, len(fns), sum(a.shape[0] + b.shape[0] for a, b in fns), ctx.divisor_last_ntt()), "phases ms (msm, lists, forest, download)", ["%.1f" % v for v in ctx.lhs_witness_last_phases()])
