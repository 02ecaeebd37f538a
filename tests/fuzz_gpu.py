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
from oracle import divisor as dv           # noqa: E402
import json                                # noqa: E402

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


def _from_mont(arr, p):
    rinv = pow(1 << 256, -1, p)
    a = np.ascontiguousarray(arr, np.uint64).reshape(-1, 4)
    return [int.from_bytes(a[i].tobytes(), "little") * rinv % p for i in range(a.shape[0])]


def witness_case(rng, ctx, O, seed, cases):
    """a forest of random point lists (identities, repeated points, P / -P neighbours) through lemsm_divisor_witness_batch
    against the restatement of regular_functions_utils.rs: lengths exactly, coefficients normalised, outputs as points;
    lists the reference panics on (two empty polynomials meeting) must be reported as such"""
    g = pyref.GRUMPKIN; p = g.fp
    pool = [g.raw_to_affine(r.tobytes()) for r in cref.gen_points(g.cid, int(rng.integers(1, 1 << 30)), 12)]
    lists = []
    for _ in range(int(rng.integers(1, 6))):
        n = int(rng.choice([0, 1, 2, 3, 5, 8, 17, 40, 130]))
        l = []
        for _ in range(n):
            r = rng.random()
            if r < 0.12: l.append(None)
            elif r < 0.25 and l and l[-1] is not None: l.append(g.neg(l[-1]))
            elif r < 0.35 and l: l.append(l[-1])
            else: l.append(pool[int(rng.integers(0, len(pool)))])
        lists.append(l)
    exp = []
    panics = False
    for l in lists:
        try:
            w, out = O.compute_divisor_witness_partial([O.from_affine(q) for q in l])
            exp.append((O.normalise(w), O.to_affine(out)))
        except pyref.RefPanic:
            panics = True
    rows = [np.frombuffer(b"".join(g.affine_to_raw(q) for q in l), np.uint64).reshape(-1, 8) if l else np.zeros((0, 8), np.uint64) for l in lists]
    if panics:
        try:
            ctx.divisor_witness_batch(api.GRUMPKIN, rows, False, True)
        except api.RefArithmeticOverflow:
            return "witness(panic)"
        print("MISMATCH seed=%d case=%d witness: the reference panics, the library did not" % (seed, cases), flush=True); sys.exit(1)
    res = ctx.divisor_witness_batch(api.GRUMPKIN, rows, False, True)
    for t, ((a, b, outp), (w, out)) in enumerate(zip(res, exp)):
        if (_from_mont(a, p), _from_mont(b, p)) != w or g.raw_to_affine(outp.tobytes()) != out:
            print("MISMATCH seed=%d case=%d witness list %d of lengths %s" % (seed, cases, t, [len(l) for l in lists]), flush=True); sys.exit(1)
    return "witness"


def scalar_witness_case(rng, ctx, seed, cases):
    base = int(rng.choice([3, 5, 16, 17, 255])); nd = int(rng.integers(1, 60)); lt = int(rng.integers(1, 20))
    vals = [int(rng.integers(0, 1 << 62)) << int(rng.integers(0, 66)) for _ in range(40)]
    vals = [v if rng.random() < 0.7 else -v for v in vals]
    if rng.random() < 0.55:
        # Half of the cases are drawn where the reference does NOT panic (uniform parameters panic almost always:
        # `i % logtable + 1` leaves the row as soon as logtable^2 exceeds num_digits, and (-base)^i leaves i128 from
        # i ~ 127 / log2(base)): logtable^2 <= num_digits, magnitudes of at most min(num_digits, i128 range) digits
        import math
        for _ in range(20):
            lt = int(rng.integers(1, 6)); nd = int(rng.integers(lt * lt, 60))
            maxdig = max(1, min(nd, int(126 / math.log2(base))) - 1)
            vals = [int(rng.integers(0, 1 << 62)) % (base ** int(rng.integers(1, maxdig + 1))) for _ in range(40)]
            vals = [v if rng.random() < 0.7 else -v for v in vals]
            try:
                for v in vals:
                    pyref.prepare_scalar_witness(v, base, nd, lt)
                break
            except pyref.RefPanic:
                continue
    sc = np.frombuffer(b"".join(abs(v).to_bytes(32, "little") for v in vals), np.uint8).reshape(-1, 32)
    neg = np.array([1 if v < 0 else 0 for v in vals], np.uint8)
    first = None; kinds = []
    for j, v in enumerate(vals):
        try:
            kinds.append(pyref.prepare_scalar_witness(v, base, nd, lt))
        except pyref.RefPanic as e:
            kinds.append(e.kind)
            if first is None: first = j
    exc = {"too_many_digits": api.TooManyDigits, "index": api.RefIndexOutOfBounds, "overflow": api.RefArithmeticOverflow}
    try:
        arr = ctx.prepare_scalar_witness_batch(sc, neg, base, nd, lt)
        if first is not None:
            print("MISMATCH seed=%d case=%d scalar witness: expected %s at %d" % (seed, cases, kinds[first], first), flush=True); sys.exit(1)
    except (api.TooManyDigits, api.RefIndexOutOfBounds, api.RefArithmeticOverflow) as e:
        if first is None or not isinstance(e, exc[kinds[first]]) or e.index != first:
            print("MISMATCH seed=%d case=%d scalar witness error %r, expected %s at %s" % (seed, cases, e, None if first is None else kinds[first], first), flush=True); sys.exit(1)
        return "scalar_witness(panic)"
    for j, v in enumerate(vals):
        for r in range(base):
            for c in range(arr.shape[2]):
                e = arr[j, r, c]; val = (int(e["hi"]) << 64) | int(e["lo"]); want = kinds[j][r][c]
                ok = api.ENTRY_KINDS[int(e["kind"])] == want[0] and (want[0] == "Scalar" or (val == want[1] and (want[0] == "Bucket" or int(e["mask"]) == want[2])))
                if not ok:
                    print("MISMATCH seed=%d case=%d scalar witness value base=%d nd=%d lt=%d v=%d cell %d,%d" % (seed, cases, base, nd, lt, v, r, c), flush=True); sys.exit(1)
    return "scalar_witness"


def main(secs=None, seed=None):
    if secs is None:
        secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    if seed is None:
        seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    ctx = api.Context(0)
    chains = json.load(open(os.path.join(ROOT, "tests", "golden", "fr_mont_chains.json")))
    head = int.from_bytes(bytes.fromhex(chains["omega_pow"]["head"]), "little")
    O = dv.DivisorOracle(pyref.GRUMPKIN, dv.FrFft(pyref.GRUMPKIN.fp, head * pow(1 << 256, -1, pyref.GRUMPKIN.fp) % pyref.GRUMPKIN.fp))
    t0 = time.time(); cases = 0; kinds_seen = {}
    names = ["window_bits", "chunk", "tile", "field", "abi_points", "slab_bits", "merge_slice", "merge_wave_th", "accum_waves", "host_slab_bits", "groups", "entry_ring", "xcd_windows", "ws_canary", "pyr_fuse", "pyr_first2", "binsort", "dw_wrap", "dw_fuse", "dw_reuse", "dw_pw_lazy", "slab_tail", "dw_ntt_lazy", "dw_halves", "pyr_quad", "scatter_lean"]
    while time.time() - t0 < secs:
        curve = CURVES[int(rng.integers(0, 2))]
        opts = {"window_bits": int(rng.choice([0, 0, 2, 3, 5, 8, 11, 13, 16, 17])), "chunk": int(rng.choice([0, 0, 1, 3, 17, 64, 300])),
                "tile": int(rng.choice([0, 0, 256, 1000])), "field": int(rng.choice([0, 0, 1])), "abi_points": int(rng.integers(0, 3)),
                "slab_bits": int(rng.choice([0, 0, 12, 14])), "merge_slice": int(rng.choice([0, 0, 33, 64, 100])), "merge_wave_th": int(rng.choice([0, 0, 1, 3])),
                "accum_waves": int(rng.choice([0, 0, 2, 4])), "host_slab_bits": int(rng.choice([0, 12, 13, 16])),
                "groups": 0, "entry_ring": int(rng.integers(0, 2)), "xcd_windows": int(rng.integers(0, 2)),
                "ws_canary": int(rng.random() < 0.3), "pyr_fuse": int(rng.choice([0, 0, 1, 2])), "pyr_first2": int(rng.random() < 0.3),
                "binsort": int(rng.choice([0, 0, 2, 3, 40, 700])), "dw_wrap": int(rng.choice([0, 0, 2])), "dw_fuse": int(rng.choice([0, 0, 2])), "dw_reuse": int(rng.choice([0, 0, 2])), "dw_pw_lazy": int(rng.choice([0, 0, 1])),
                "slab_tail": int(rng.choice([0, 0, 2])), "dw_ntt_lazy": int(rng.choice([0, 0, 1])), "dw_halves": int(rng.choice([0, 0, 1])), "pyr_quad": int(rng.choice([0, 0, 2])), "scatter_lean": int(rng.choice([0, 1]))}     # 2: tiled pass 2 only; > 2: a bin capacity that splits the bins between both paths
        host_entry = rng.random() < 0.5
        if not host_entry:
            opts["groups"] = int(rng.choice([0, 0, 2, 3]))     # pipelined window groups: device-pointer entries only
        for k in names:
            ctx.set_option(k, opts[k])
        pick = rng.random()
        if pick < 0.12:
            k = witness_case(rng, ctx, O, seed, cases); kinds_seen[k] = kinds_seen.get(k, 0) + 1; cases += 1
            continue
        if pick < 0.18:
            k = scalar_witness_case(rng, ctx, seed, cases); kinds_seen[k] = kinds_seen.get(k, 0) + 1; cases += 1
            continue
        sharded = (not host_entry) and opts["groups"] == 0 and rng.random() < 0.2      # the C ABI's multi-GPU entries, ranks simulated on this GPU
        world = int(rng.choice([2, 3, 5, 8])) if sharded else 1
        if rng.random() < 0.75:
            n, shape, sc, pts = make_case(rng, curve)
            try:
                got = ctx.msm(curve.cid, sc, pts) if host_entry else None
                if got is None:
                    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
                    got = ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, world) if sharded else ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)
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
            if sharded:
                ds, dp = ctx.to_device(sc), ctx.to_device(pts)
                got, carries = ctx.debug_lhs_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, base, world)
            else:
                got, carries = ctx.lhs_msm(curve.cid, sc, pj, base, True)
            exp, ecar = cref.lhs_msm(curve.cid, sc, pj, base, True)
            for i in range(carries.shape[0]):
                assert cref.jac_to_canonical(curve.cid, carries[i]) == cref.jac_to_canonical(curve.cid, ecar[i]), ("carry", i, seed, cases, opts)
            what = "lhs base %d" % base
        ok = cref.jac_to_canonical(curve.cid, np.ascontiguousarray(got, np.uint64)) == cref.jac_to_canonical(curve.cid, exp)
        if not ok:
            print("MISMATCH seed=%d case=%d %s %s n=%d shape=%s host_entry=%s sharded=%s world=%d opts=%s" % (seed, cases, what, curve.name, n, shape, host_entry, sharded, world, opts), flush=True)
            if what == "msm" and os.environ.get("FUZZ_BISECT"):
                # which single option, put back to its default, makes the same inputs come out right?
                def run():
                    if host_entry:
                        return ctx.msm(curve.cid, sc, pts)
                    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
                    return ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, world) if sharded else ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)
                want = cref.jac_to_canonical(curve.cid, exp)
                for k in names:
                    if not opts[k]:
                        continue
                    ctx.set_option(k, 0)
                    try:
                        good = cref.jac_to_canonical(curve.cid, np.ascontiguousarray(run(), np.uint64)) == want
                    except Exception as ex:
                        good = "exception %r" % (ex,)
                    print("  with %s = 0 (was %d): %s" % (k, opts[k], good), flush=True)
                    ctx.set_option(k, opts[k])
            sys.exit(1)
        cases += 1
        if cases % 50 == 0:
            print("%d cases ok (%.0f s)" % (cases, time.time() - t0), flush=True)
    for k in names:
        ctx.set_option(k, 0)
    ctx.close()
    print("fuzz ok: %d cases, seed %d, of which %s" % (cases, seed, kinds_seen))
    return cases


if __name__ == "__main__":
    main()
