#!/usr/bin/env python3
"""Large-size check (test infrastructure: uses the oracle as checker): 2^LOGN-point MSM on
device-generated walk points P_i = (i+1)Q, verified through sum s_i P_i == (sum s_i (i+1)) Q."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context, jacobian_to_canonical
from bench import gen_scalars, ORDER
from oracle import cref

logn = int(sys.argv[1]); curve = sys.argv[2] if len(sys.argv) > 2 else "bn254_g1"
cid = {"bn254_g1": 0, "grumpkin": 1}[curve]
n = 1 << logn
ctx = Context(0)
t0 = time.time(); sc = gen_scalars(n, ORDER[curve], 77 + logn); print("scalars %.1fs" % (time.time() - t0), flush=True)
ds = ctx.to_device(sc)
q = cref.gen_points(cid, 5, 1)[0]
t0 = time.time(); dp = ctx.gen_walk(cid, q, n); print("walk %.2fs" % (time.time() - t0), flush=True)
for it in range(3):
    t0 = time.perf_counter(); out = ctx.msm_device(cid, ds.ptr, dp.ptr, n); dt = time.perf_counter() - t0
    print("msm 2^%d %s: %.2f ms  (%.1f Mpairs/s)  device %.2f ms accum %.2f ms x%d" % ((logn, curve, dt * 1e3, n / dt / 1e6) + ctx.last_timing()), flush=True)
t0 = time.time(); dot = cref.walk_dot(cid, sc); exp = cref.scalar_mul(cid, dot, q); print("oracle dot %.1fs" % (time.time() - t0), flush=True)
ok = cref.jac_to_canonical(cid, out) == cref.jac_to_canonical(cid, exp)
print("PARITY", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
