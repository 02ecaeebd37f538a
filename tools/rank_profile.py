#!/usr/bin/env python3
"""One window-sharded rank's call (windows [w0, w1) of a 2^LOGN MSM at 16-bit windows), repeated, for
rocprofv3 --kernel-trace --stats.  usage: rank_profile.py LOGN W0 W1"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context
from bench import gen_scalars, ORDER
logn, w0, w1 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); n = 1 << logn
ctx = Context(0)
sc = gen_scalars(n, ORDER["bn254_g1"], 5)
q = np.zeros(8, np.uint64); fp = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
q[:4] = np.frombuffer(((1 << 256) % fp).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((2 << 256) % fp).to_bytes(32, "little"), np.uint64)
dp = ctx.gen_walk(0, q, n); ds = ctx.to_device(sc)
ctx.set_option("window_bits", 16)
for _ in range(5):
    ctx.msm_partial_device(0, ds.ptr, dp.ptr, n, w0, w1)
print(ctx.last_timing())
