#!/usr/bin/env python3
"""One divisor witness of 2^LOGN walk points, twice (for rocprofv3 --kernel-trace --stats). usage: divisor_profile.py LOGN"""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from halo2_liam_eagen_msm_amd import Context
logn = int(sys.argv[1]); n = 1 << logn
ctx = Context(0)
r = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
gx, gy = 1, 0x2CF135E7506A45D632D270D45F1181294833FC48D823F272C
q = np.zeros(8, np.uint64)
q[:4] = np.frombuffer(((gx << 256) % r).to_bytes(32, "little"), np.uint64); q[4:] = np.frombuffer(((gy << 256) % r).to_bytes(32, "little"), np.uint64)
dp = ctx.gen_walk(1, q, n)
a = np.zeros((n + 4, 4), np.uint64); b = np.zeros((n + 4, 4), np.uint64)
la = ctypes.c_size_t(); lb = ctypes.c_size_t(); outp = np.zeros(8, np.uint64)
for _ in range(2):
    ctx._check(ctx.lib.lemsm_divisor_witness_device(ctx.h, 1, dp.ptr, n, 0, 1, a.ctypes.data, n + 4, ctypes.byref(la), b.ctypes.data, n + 4, ctypes.byref(lb), outp.ctypes.data))
print(la.value, lb.value, ctx.divisor_last_ntt())
