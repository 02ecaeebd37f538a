#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-pointer entry lemsm_msm (what a Rust caller with host buffers
sees), next to the device-resident entry.  usage: host_path_timing.py LOGN"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context, jacobian_to_canonical
from bench import gen_scalars, ORDER

logn = int(sys.argv[1]); n = 1 << logn
ctx = Context(0)
sc = gen_scalars(n, ORDER["bn254_g1"], 99)
q = np.zeros(8, np.uint64); fp = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
q[:4] = np.frombuffer(((1 << 256) % fp).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((2 << 256) % fp).to_bytes(32, "little"), np.uint64)
dp = ctx.gen_walk(0, q, n)
pts = dp.download(np.uint64).reshape(-1, 8).copy()
ds = ctx.to_device(sc)
ref = jacobian_to_canonical(0, ctx.msm_device(0, ds.ptr, dp.ptr, n))
for bits in (24, 21, 20, 19, 0):
    ctx.set_option("host_slab_bits", bits)
    best = 1e9
    for it in range(4):
        t0 = time.perf_counter(); out = ctx.msm(0, sc, pts); dt = time.perf_counter() - t0
        assert jacobian_to_canonical(0, out) == ref
        best = min(best, dt)
    t1 = time.perf_counter(); ctx.msm_device(0, ds.ptr, dp.ptr, n); dd = time.perf_counter() - t1
    print("2^%d host_slab_bits=%d: host-pointer entry %.1f ms (%.0f Mpairs/s, %.1f GB/s of input) | device-resident entry %.1f ms"
          % (logn, bits, best * 1e3, n / best / 1e6, n * 96 / best / 1e9, dd * 1e3), flush=True)

# resident bases: the points uploaded once (lemsm_bases_upload), only the scalars cross PCIe per call
ctx.set_option("host_slab_bits", 0)
bases = ctx.bases_upload(0, pts)
for bits in (0, 22, 21, 20):
    ctx.set_option("host_slab_bits", bits)
    best = 1e9
    for it in range(4):
        t0 = time.perf_counter(); out = ctx.msm_with_bases(bases, sc); dt = time.perf_counter() - t0
        assert jacobian_to_canonical(0, out) == ref
        best = min(best, dt)
    print("2^%d host_slab_bits=%d: lemsm_msm_with_bases (scalars from host, bases resident) %.1f ms (%.0f Mpairs/s)" % (logn, bits, best * 1e3, n / best / 1e6), flush=True)

# the same as a batch: K scalar vectors from host memory, call k's upload beside call k - 1's compute (lemsm_msm_batch_with_bases)
ctx.set_option("host_slab_bits", 0)
distinct = [sc] + [gen_scalars(n, ORDER["bn254_g1"], 100 + k) for k in range(1, 3)]
for K in (4, 12):
    scs = [distinct[k % 3] for k in range(K)]
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter(); outs = ctx.msm_batch_with_bases(bases, scs); dt = time.perf_counter() - t0
        assert jacobian_to_canonical(0, outs[0]) == ref and jacobian_to_canonical(0, outs[3]) == ref
        best = min(best, dt)
    print("2^%d lemsm_msm_batch_with_bases, K = %d scalar vectors from (pageable) host memory: %.1f ms per call = %.1f ms per MSM (%.0f Mpairs/s)"
          % (logn, K, best * 1e3, best * 1e3 / K, n * K / best / 1e6), flush=True)
