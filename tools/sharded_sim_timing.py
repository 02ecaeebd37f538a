#!/usr/bin/env python3
"""One-GPU rehearsal of the C ABI's sharded MSM entry (lemsm_debug_msm_sharded_sim): the G ranks' pipelines of one
2^LOGN MSM run one after the other; prints wall time per call and per rank, and the device time per rank.
usage: [SIM_OPTIONS=name=value,...] sharded_sim_timing.py [LOGN] [G...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context
from bench import gen_scalars, ORDER
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
Gs = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
n = 1 << logn
ctx = Context(0)
for kv in filter(None, os.environ.get("SIM_OPTIONS", "").split(",")):      # e.g. SIM_OPTIONS=slab_tail=2
    k, v = kv.split("="); ctx.set_option(k, int(v))
sc = gen_scalars(n, ORDER["bn254_g1"], 5)
q = np.zeros(8, np.uint64); fp = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
q[:4] = np.frombuffer(((1 << 256) % fp).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((2 << 256) % fp).to_bytes(32, "little"), np.uint64)
dp = ctx.gen_walk(0, q, n); ds = ctx.to_device(sc)
ref = ctx.msm_device(0, ds.ptr, dp.ptr, n)
for G in Gs:
    out = ctx.debug_msm_sharded_sim(0, ds.ptr, dp.ptr, n, G)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); out = ctx.debug_msm_sharded_sim(0, ds.ptr, dp.ptr, n, G); best = min(best, time.perf_counter() - t0)
    tt, ta, nl = ctx.last_timing()
    print("2^%d, %d simulated ranks: %.2f ms per call = %.2f ms per rank wall; device %.2f ms per rank of which k_accum1 %.2f (%d launches)"
          % (logn, G, best * 1e3, best * 1e3 / G, tt / G, ta / G, nl), flush=True)
