#!/usr/bin/env python3
"""Interleaved A/B timing of pipeline variants in ONE process (cdna guide rule 24).
usage: ab_bench.py LOGN 'opt=val,opt=val' 'opt=val' ...   (each arg = one variant)"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context
from bench import gen_scalars, ORDER

logn = int(sys.argv[1]); n = 1 << logn
variants = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[2:]]
workload = os.environ.get("AB_WORKLOAD", "msm")
ctx = Context(0)
import math
order = ORDER["bn254_g1"]
sc = gen_scalars(n, order if workload == "msm" else math.isqrt(order), 1234)
ds = ctx.to_device(sc)
q = np.zeros(8, np.uint64); fp = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
q[:4] = np.frombuffer(((1 << 256) % fp).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((2 << 256) % fp).to_bytes(32, "little"), np.uint64)
dp = ctx.gen_walk(0, q, n)
ALL = ["window_bits", "chunk", "tile", "field", "accum_waves", "merge_slice", "merge_wave_th", "abi_points", "stage2x", "xcd_windows", "entry_ring", "binsort", "scatter_lean"]
res = {i: [] for i in range(len(variants))}
ref = None
for rnd in range(int(os.environ.get("AB_ROUNDS", "4"))):
    for i, v in enumerate(variants):
        for k in ALL: ctx.set_option(k, v.get(k, 0))
        os.environ["LEMSM_ABLATE"] = str(v.get("ablate", 0))
        t0 = time.perf_counter()
        out = ctx.msm_device(0, ds.ptr, dp.ptr, n) if workload == "msm" else ctx.lhs_msm_device(0, ds.ptr, dp.ptr, n, 16)[0]
        wall = (time.perf_counter() - t0) * 1e3
        tt, ta, nl = ctx.last_timing()
        from halo2_liam_eagen_msm_amd import jacobian_to_canonical
        c = jacobian_to_canonical(0, out)
        if ref is None: ref = c
        assert os.environ.get("AB_NOCHECK") or c == ref, "variant %d result differs" % i
        if rnd: res[i].append((wall, tt, ta))
for i, v in enumerate(variants):
    a = np.array(res[i])
    print("variant %-40s wall_ms med %.3f min %.3f | device_ms med %.3f | accum_ms med %.3f min %.3f" % (v, np.median(a[:,0]), a[:,0].min(), np.median(a[:,1]), np.median(a[:,2]), a[:,2].min()))
