import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from halo2_liam_eagen_msm_amd import Context
from bench import gen_scalars, ORDER
ctx = Context(0)
r = ORDER["bn254_g1"]
gx, gy = 1, 0x2CF135E7506A45D632D270D45F1181294833FC48D823F272C
q = np.zeros(8, np.uint64)
q[:4] = np.frombuffer(((gx << 256) % r).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((gy << 256) % r).to_bytes(32, "little"), np.uint64)
for logn in (10, 14, 16):
    n = 1 << logn
    sc = gen_scalars(n, math.isqrt(ORDER["grumpkin"]), 77 + logn)
    dp = ctx.gen_walk(1, q, n); ds = ctx.to_device(sc)
    carry, index, out = ctx.lhs_witness_device(1, ds.ptr, dp.ptr, n, 16, True)
    print("lhs_witness 2^%d: reuse levels" % logn, ctx.divisor_last_reuse_levels(), "ntt", ctx.divisor_last_ntt())
    rows = dp.download(np.uint64).reshape(-1, 8)
    a, b, o = ctx.divisor_witness(1, rows, False, True)
    print("single witness 2^%d: reuse levels" % logn, ctx.divisor_last_reuse_levels())
