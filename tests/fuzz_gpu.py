#!/usr/bin/env python3
"""Randomised parity run of the HIP path against the oracle (test infrastructure; uses oracle/).

Not collected by pytest (no test_ prefix): a longer soak than the suite, run by hand on the GPU box:
    python tests/fuzz_gpu.py SECONDS [SEED]
Each case draws a size, a curve, an input shape and a set of tuning options (window width, chunk,
tile, arithmetic, point-domain form, slab size, edge-record fan-in), runs best_multiexp or the lhs
MSM through the C ABI and compares canonical affine bytes with the C oracle.  Input shapes aim at
the rare branches: P next to -P with equal scalars (cancellation), repeated points (doubling),
identity points, all-equal and tiny scalars, scalars = order-1."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from halo2_liam_eagen_msm_amd import api   # noqa: E402
from oracle import cref, pyref             # noqa: E402

CURVES = [pyref.BN254_G1, pyref.GRUMPKIN]


def neg_points(curve, pts):
    out = pts.copy()
    for r in out:
        y = int.from_bytes(r[4:].tobytes(), "little")
        if y:
            r[4:] = np.frombuffer(((curve.fp - y) % curve.fp).to_bytes(32, "little"), np.uint64)
    return out


def make_case(rng, curve):
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 3000), rng.integers(3000, 70000), rng.integers(3000, 70000),
                        rng.integers(70000, 1 << 20)]))
    base_pts = cref.gen_points(curve.cid, int(rng.integers(1, 1 << 30)), min(n, int(rng.integers(1, 400))))
    shape = rng.choice(["uniform", "few_points", "pairs_cancel", "identities", "equal_scalars", "tiny_scalars", "top_scalars"])
    idx = rng.integers(0, base_pts.shape[0], n)
    pts = base_pts[idx].copy()
    sc = cref.gen_scalars(curve.cid, int(rng.integers(1, 1 << 30)), n)
    if shape == "few_points":
        pts = base_pts[idx % max(1, min(3, base_pts.shape[0]))].copy()
    elif shape == "pairs_cancel":
        half = n // 2
        pts[half:2 * half] = neg_points(curve, pts[:half]); sc[half:2 * half] = sc[:half]
    elif shape == "identities":
        pts[rng.random(n) < 0.3] = 0
    elif shape == "equal_scalars":
        sc[:] = sc[0]
    elif shape == "tiny_scalars":
        sc[:] = 0; sc[:, 0] = rng.integers(0, 4, n)
    elif shape == "top_scalars":
        sc[:] = np.frombuffer(int(curve.order - 1).to_bytes(32, "little"), np.uint8)
        sc[::3] = cref.gen_scalars(curve.cid, 7, (n + 2) // 3)
    return n, shape, sc, pts


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    ctx = api.Context(0)
    t0 = time.time(); cases = 0
    names = ["window_bits", "chunk", "tile", "field", "abi_points", "slab_bits", "seg_records", "accum_waves", "host_slab_bits", "groups", "entry_ring", "xcd_windows"]
    while time.time() - t0 < secs:
        curve = CURVES[int(rng.integers(0, 2))]
        opts = {"window_bits": int(rng.choice([0, 0, 2, 3, 5, 8, 11, 13, 16, 17])), "chunk": int(rng.choice([0, 0, 1, 3, 17, 64, 300])),
                "tile": int(rng.choice([0, 0, 256, 1000])), "field": int(rng.choice([0, 0, 1])), "abi_points": int(rng.integers(0, 3)),
                "slab_bits": int(rng.choice([0, 0, 12, 14])), "seg_records": int(rng.choice([0, 2, 5, 8, 16])),
                "accum_waves": int(rng.choice([0, 0, 2, 4])), "host_slab_bits": int(rng.choice([0, 12, 13, 16])),
                "groups": 0, "entry_ring": int(rng.integers(0, 2)), "xcd_windows": int(rng.integers(0, 2))}
        host_entry = rng.random() < 0.5
        if not host_entry:
            opts["groups"] = int(rng.choice([0, 0, 2, 3]))     # pipelined window groups: device-pointer entries only
        for k in names:
            ctx.set_option(k, opts[k])
        if rng.random() < 0.75:
            n, shape, sc, pts = make_case(rng, curve)
            try:
                got = ctx.msm(curve.cid, sc, pts) if host_entry else None
                if got is None:
                    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
                    got = ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)
            except Exception as ex:
                print("EXCEPTION %r seed=%d case=%d %s n=%d shape=%s host_entry=%s opts=%s" % (ex, seed, cases, curve.name, n, shape, host_entry, opts), flush=True)
                raise
            exp = cref.best_multiexp(curve.cid, sc, pts, 8)
            what = "msm"
        else:
            n = int(rng.integers(1, 3000)); shape = "lhs"; base = int(rng.choice([3, 4, 5, 16, 17, 255]))
            pts = cref.gen_points(curve.cid, int(rng.integers(1, 1 << 30)), min(n, 50))[rng.integers(0, min(n, 50), n)]
            sc = cref.gen_scalars(curve.cid, int(rng.integers(1, 1 << 30)), n, half=True)
            if rng.random() < 0.3:
                sc[:] = sc[0]
            pj = cref.aff_to_jac(curve.cid, pts)
            got, carries = ctx.lhs_msm(curve.cid, sc, pj, base, True)
            exp, ecar = cref.lhs_msm(curve.cid, sc, pj, base, True)
            for i in range(carries.shape[0]):
                assert cref.jac_to_canonical(curve.cid, carries[i]) == cref.jac_to_canonical(curve.cid, ecar[i]), ("carry", i, seed, cases, opts)
            what = "lhs base %d" % base
        ok = cref.jac_to_canonical(curve.cid, np.ascontiguousarray(got, np.uint64)) == cref.jac_to_canonical(curve.cid, exp)
        if not ok:
            print("MISMATCH seed=%d case=%d %s %s n=%d shape=%s opts=%s" % (seed, cases, what, curve.name, n, shape, opts), flush=True)
            sys.exit(1)
        cases += 1
        if cases % 50 == 0:
            print("%d cases ok (%.0f s)" % (cases, time.time() - t0), flush=True)
    print("fuzz ok: %d cases, seed %d" % (cases, seed))


if __name__ == "__main__":
    main()
