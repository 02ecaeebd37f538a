#!/usr/bin/env python3
"""Per-rank device work of the two multi-GPU partitions of one 2^LOGN MSM, rehearsed on ONE GPU:
  window sharding (north star / dist.sharded_msm): all points, W/G windows per rank
  point sharding (SURVEY 8e alternative):          n/G points, all windows per rank
Prints the wall time of one rank's call for G = 1, 2, 4, 8 (exchange + combine not included: a
2 KiB all-gather and the same host Horner in both).   usage: shard_rehearsal.py [LOGN]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context
from bench import gen_scalars, ORDER

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
ctx = Context(0)
sc = gen_scalars(n, ORDER["bn254_g1"], 5)
q = np.zeros(8, np.uint64); fp = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
q[:4] = np.frombuffer(((1 << 256) % fp).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((2 << 256) % fp).to_bytes(32, "little"), np.uint64)
dp = ctx.gen_walk(0, q, n)
ds = ctx.to_device(sc)
ctx.set_option("window_bits", 16)   # what bench.py pins for window sharding (16 windows split evenly)
W, rec = ctx.msm_plan(0, n)
ctx.set_option("window_bits", 0)

def best(fn, reps=5):
    fn(); b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return b * 1e3

for G in (1, 2, 4, 8):
    w1 = W // G
    ctx.set_option("window_bits", 16)
    tw = best(lambda: ctx.msm_partial_device(0, ds.ptr, dp.ptr, n, 0, w1))
    ctx.set_option("window_bits", 0)
    m = n // G
    tp = best(lambda: ctx.msm_partial_device(0, ds.ptr, dp.ptr, m, 0, ctx.msm_plan(0, m)[0]))
    print("2^%d over %d ranks: window-sharded rank %.2f ms (%d of %d windows, all points) | point-sharded rank %.2f ms (2^%d points, %d windows)"
          % (logn, G, tw, w1, W, tp, logn - int(np.log2(G)), ctx.msm_plan(0, m)[0]), flush=True)
