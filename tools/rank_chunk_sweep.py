import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from halo2_liam_eagen_msm_amd import Context
from bench import gen_scalars, ORDER
n = 1 << 24
ctx = Context(0)
sc = gen_scalars(n, ORDER["bn254_g1"], 5)
q = np.zeros(8, np.uint64); fp = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
q[:4] = np.frombuffer(((1 << 256) % fp).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((2 << 256) % fp).to_bytes(32, "little"), np.uint64)
dp = ctx.gen_walk(0, q, n); ds = ctx.to_device(sc)
ctx.set_option("window_bits", 16)
res = {}
variants = [(2, 0), (2, 96), (2, 128), (2, 176), (2, 256), (4, 0), (4, 176), (4, 256), (8, 0), (8, 256)]
for rnd in range(5):
    for (nw, ch) in variants:
        ctx.set_option("chunk", ch)
        t0 = time.perf_counter(); ctx.msm_partial_device(0, ds.ptr, dp.ptr, n, 0, nw); dt = (time.perf_counter() - t0) * 1e3
        tt, ta, nl = ctx.last_timing()
        if rnd: res.setdefault((nw, ch), []).append((dt, tt, ta))
for k, v in res.items():
    a = np.array(v); print("windows %d chunk %3d: wall %.3f device %.3f accum %.3f" % (k[0], k[1], np.median(a[:,0]), np.median(a[:,1]), np.median(a[:,2])))
