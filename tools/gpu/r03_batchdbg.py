#!/usr/bin/env python3
"""where a call of lemsm_msm_batch_with_bases spends its time (LEMSM_DEBUG_STAMPS=1), pageable against pinned scalar vectors"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from halo2_liam_eagen_msm_amd import Context, jacobian_to_canonical
from bench import gen_scalars, ORDER
logn = 24; n = 1 << logn
ctx = Context(0)
q = np.zeros(8, np.uint64); fp = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
q[:4] = np.frombuffer(((1 << 256) % fp).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((2 << 256) % fp).to_bytes(32, "little"), np.uint64)
dp = ctx.gen_walk(0, q, n)
pts = dp.download(np.uint64).reshape(-1, 8).copy()
bases = ctx.bases_upload(0, pts)
K = 4
scs = [gen_scalars(n, ORDER["bn254_g1"], 100 + k) for k in range(K)]
for kind in ("pageable", "pinned"):
    if kind == "pinned":
        pins = [torch.empty((n, 32), dtype=torch.uint8).pin_memory() for _ in range(K)]
        for p, s in zip(pins, scs): p.numpy()[:] = s
        scs = [p.numpy() for p in pins]
    for it in range(2):
        t0 = time.perf_counter(); outs = ctx.msm_batch_with_bases(bases, scs); dt = time.perf_counter() - t0
        print("%s scalars: %.1f ms per call of K = %d -> %.1f ms per MSM" % (kind, dt * 1e3, K, dt * 1e3 / K), flush=True)
    t0 = time.perf_counter(); ctx.msm_with_bases(bases, scs[0]); print("%s single lemsm_msm_with_bases %.1f ms" % (kind, (time.perf_counter() - t0) * 1e3), flush=True)
