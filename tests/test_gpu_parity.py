"""Parity of the HIP path (through the C ABI) against the oracle, on a real MI355X.

Bit-exact bar: results are compared as canonical affine bytes (x||y 32-byte LE each,
identity = zeros); digits and error indices are compared exactly.
"""
import hashlib
import math

import numpy as np
import pytest

from helpers import (CURVES, canon, golden_points_raw, golden_scalars, int_of, jacobian_with_random_z, limbs4,
                     load_json, regenerate_chain_digests)
from halo2_liam_eagen_msm_amd import api
from oracle import cref, pyref

pytestmark = pytest.mark.gpu

# (field, abi_points): field 0 = lazy radix-2^29 arithmetic (default hot path), 1 = strict 32-bit limbs;
# abi_points 1 = points converted by a pass before the accumulation, 2 = accumulation consumes the C ABI's
# form directly (scaled accumulators); the library picks between the two by size, the tests force each
# entry_ring 1 = every lane loads its entries itself, 0 = through the per-wave LDS ring (only with abi_points form)
VARIANTS = [(0, 1, 0), (0, 2, 0), (0, 2, 1), (1, 0, 0)]


@pytest.fixture(params=VARIANTS, ids=["lazy29-convert-pass", "lazy29-abi-points-entry-ring", "lazy29-abi-points-no-ring", "strict32"])
def fctx(ctx, request):
    """context with the arithmetic of the hot kernels selected"""
    ctx.set_option("field", request.param[0]); ctx.set_option("abi_points", request.param[1]); ctx.set_option("entry_ring", request.param[2])
    yield ctx
    ctx.set_option("field", 0); ctx.set_option("abi_points", 0); ctx.set_option("entry_ring", 0)


# ------------------------------------------------------------------ field / group KATs
@pytest.fixture(params=[0, 1], ids=["lazy29", "strict32"])
def kctx(ctx, request):
    """the known-answer entries run the arithmetic option `field` selects: 0 = the hot kernels' lazy radix-2^29 field
    (Field29 / XYZZ29, the default), 1 = the strict 32-bit-limb field"""
    ctx.set_option("field", request.param)
    ctx.kat_field = request.param
    yield ctx
    ctx.set_option("field", 0)


def test_device_montmul_reference_fr_chains(kctx):
    """the device r-modulus multiplier -- Field29<Fr29Params>::mul of the hot kernels as well as the strict one --
    regenerates all 192 constants of the reference's src/precomputed_fft_data.rs (via digests in
    tests/golden/fr_mont_chains.json)"""
    ctx = kctx
    chains = load_json("fr_mont_chains.json")
    def mm(a, b):
        return ctx.debug_montmul(api.GRUMPKIN, np.frombuffer(a, np.uint64), np.frombuffer(b, np.uint64)).tobytes()
    got = regenerate_chain_digests(mm, chains)
    for name in ("omega_pow", "omega_pow_inv", "half_pow"):
        assert got[name] == chains[name]["sha256"], name


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_device_field_ops_random_and_edges(kctx, curve):
    ctx = kctx
    p = curve.fp
    rng = pyref.SplitMix64(0xF1E1D + curve.cid)
    vals = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 255) % p, (1 << 256) % p, (1 << 253)]
    vals += [rng.next256() % p for _ in range(500)]
    a = np.array([limbs4(v) for v in vals], np.uint64)
    b = np.array([limbs4(v) for v in reversed(vals)], np.uint64)
    rinv = pow(1 << 256, -1, p)
    got = ctx.debug_montmul(curve.cid, a, b)
    for i, (x, y) in enumerate(zip(vals, reversed(vals))):
        assert int_of(got[i]) == x * y * rinv % p, ("mul", i)
    ops = [(0, lambda x, y: (x + y) % p), (1, lambda x, y: (x - y) % p), (2, lambda x, y: (-x) % p), (4, lambda x, y: x * x * rinv % p)]
    if ctx.kat_field == 0:   # the fused products of the lazy field (XYZZ29's X3 and Y3) and its cheap times-32
        ops += [(5, lambda x, y: 2 * x * y * rinv % p), (6, lambda x, y: (x * x * rinv + y) % p), (7, lambda x, y: (x * y * rinv + y) % p),
                (9, lambda x, y: x)]
    for op, fn in ops:
        got = ctx.debug_fieldop(curve.cid, op, a, b)
        for i, (x, y) in enumerate(zip(vals, reversed(vals))):
            assert int_of(got[i]) == fn(x, y), (op, i)
    nz = a[1:40]
    got = ctx.debug_fieldop(curve.cid, 3, nz, nz)   # Montgomery inverse (field 0: inv_lazy)
    prod = ctx.debug_montmul(curve.cid, got, nz)
    R = (1 << 256) % p
    assert all(int_of(prod[i]) == R for i in range(len(nz)))
    assert all(int_of(got[i]) == pow(vals[1 + i], -1, p) * R * R % p for i in range(len(nz)))


def _xyzz_from_affine(curve, pt, z=1):
    """XYZZ raw record (16 limbs) of an affine point with arbitrary scaling z"""
    if pt is None:
        return np.zeros(16, np.uint64)
    p = curve.fp
    zz, zzz = z * z % p, z * z * z % p
    vals = (pt[0] * zz % p, pt[1] * zzz % p, zz, zzz)
    return np.concatenate([limbs4(curve.to_mont(v)) for v in vals])


def _affine_from_xyzz(curve, rec):
    x, y, zz, zzz = (curve.from_mont(int_of(rec[4 * i:4 * i + 4])) for i in range(4))
    if zz == 0:
        return None
    p = curve.fp
    return (x * pow(zz, -1, p) % p, y * pow(zzz, -1, p) % p)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_device_point_ops_complete(kctx, curve):
    """mixed add (both forms of the accumulate kernel: madd and madd_abi) and full add incl. P+P, P+(-P), identity
    operands (the reference's own test shape drives every bucket through the doubling branch)"""
    ctx = kctx
    rng = pyref.SplitMix64(0xADD + curve.cid)
    base = pyref.gen_points(curve, rng, 12)
    cases = []
    for i in range(10):
        cases.append((base[i], base[i + 1]))
    cases += [(base[0], base[0]), (base[1], curve.neg(base[1])), (None, base[2]), (base[3], None), (None, None),
              (curve.mul(2, base[4]), base[4]), (curve.mul(2, base[5]), curve.neg(curve.mul(2, base[5])))]
    zs = [1 + rng.next256() % (curve.fp - 1) for _ in cases]
    acc = np.array([_xyzz_from_affine(curve, a, z) for (a, _), z in zip(cases, zs)], np.uint64)
    q_aff = np.array([np.frombuffer(curve.affine_to_raw(b), np.uint64) for _, b in cases], np.uint64)
    for op, name in ((0, "madd"), (2, "madd_abi")):
        got = ctx.debug_pointop(curve.cid, op, acc, q_aff)
        for i, (a, b) in enumerate(cases):
            assert _affine_from_xyzz(curve, got[i]) == curve.add(a, b), (name, i)
    q_x = np.array([_xyzz_from_affine(curve, b, 1 + (z * 7) % (curve.fp - 1)) for (_, b), z in zip(cases, zs)], np.uint64)
    got = ctx.debug_pointop(curve.cid, 1, acc, q_x)
    for i, (a, b) in enumerate(cases):
        assert _affine_from_xyzz(curve, got[i]) == curve.add(a, b), ("add", i)


# ------------------------------------------------------------------ golden vectors
def test_golden_msm_vectors(fctx):
    ctx = fctx
    for v in load_json("msm_vectors.json")["msm"]:
        c = pyref.CURVES[v["curve"]]
        out = ctx.msm(c.cid, golden_scalars(v["scalars"]), golden_points_raw(c, v["points"]))
        assert api.jacobian_to_canonical(c.cid, out) == bytes.fromhex(v["expected"]), (v["curve"], v["n"])
        assert canon(c, out) == bytes.fromhex(v["expected"])


def test_golden_lhs_vectors(fctx):
    ctx = fctx
    for v in load_json("msm_vectors.json")["lhs"]:
        c = pyref.CURVES[v["curve"]]
        sc = golden_scalars(v["scalars"])
        pts = jacobian_with_random_z(c, golden_points_raw(c, v["points"]), v["seed"])
        carry, carries = ctx.lhs_msm(c.cid, sc, pts, v["base"])
        assert canon(c, carry) == bytes.fromhex(v["expected_carry"]), (v["curve"], v["n"], v["base"])
        for i, h in enumerate(v["expected_carries_msb_first"]):
            assert canon(c, carries[i]) == bytes.fromhex(h), (v["curve"], v["n"], v["base"], i)
        d = api.num_digits(c.cid, v["base"])
        assert ctx.negbase_decompose_batch(sc, v["base"], d).tolist() == v["digits_lsb_first"]


# ------------------------------------------------------------------ MSM vs oracle, seeded
@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 32, 33, 255, 1000, 4097])
def test_msm_matches_oracle(fctx, curve, n):
    ctx = fctx
    pts = cref.gen_points(curve.cid, 1000 + n, n)
    sc = cref.gen_scalars(curve.cid, 2000 + n, n)
    out = ctx.msm(curve.cid, sc, pts)
    exp = cref.best_multiexp(curve.cid, sc, pts, 8) if n else np.zeros(12, np.uint64)
    assert canon(curve, out) == canon(curve, exp)


@pytest.mark.parametrize("c_bits", [2, 3, 5, 8, 11, 13, 16, 17])
@pytest.mark.parametrize("chunk", [1, 7, 64])
def test_msm_window_and_chunk_sweep(fctx, c_bits, chunk):
    ctx = fctx
    """every window width (bins with LB 0..7) and ragged chunk lengths give the same group element"""
    curve = pyref.BN254_G1
    n = 700
    pts = cref.gen_points(curve.cid, 31, n)
    sc = cref.gen_scalars(curve.cid, 32, n)
    exp = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    ctx.set_option("window_bits", c_bits); ctx.set_option("chunk", chunk); ctx.set_option("tile", 256)
    try:
        assert canon(curve, ctx.msm(curve.cid, sc, pts)) == exp
    finally:
        ctx.set_option("window_bits", 0); ctx.set_option("chunk", 0); ctx.set_option("tile", 0)


@pytest.mark.parametrize("knobs", [{"scatter_lean": 1}], ids=str)
@pytest.mark.parametrize("n,c_bits", [(700, 0), (5000, 9), (70000, 17)])
def test_msm_scatter_form_knob_equals_default(ctx, knobs, n, c_bits):
    """A/B knob of r03: k_scatter1 with two ADJACENT bins per thread (8-byte loads of the counts / claims / bin starts, one block
    scan, unrolled store loop) gives the same group element as the default form (bins tid and tid + 256, two block scans)"""
    curve = pyref.BN254_G1
    pts = cref.gen_points(curve.cid, 77, n)
    sc = cref.gen_scalars(curve.cid, 78, n)
    exp = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    ctx.set_option("window_bits", c_bits)
    try:
        assert canon(curve, ctx.msm(curve.cid, sc, pts)) == exp
        for k, v in knobs.items(): ctx.set_option(k, v)
        assert canon(curve, ctx.msm(curve.cid, sc, pts)) == exp
    finally:
        ctx.set_option("window_bits", 0)
        for k in knobs: ctx.set_option(k, 0)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_adversarial_shapes(fctx, curve):
    ctx = fctx
    n = 600
    base_pts = cref.gen_points(curve.cid, 9, 4)
    sc1 = cref.gen_scalars(curve.cid, 10, 1)
    # (i) the reference's own shape: one scalar, one point, replicated
    pts = np.repeat(base_pts[:1], n, axis=0); sc = np.repeat(sc1, n, axis=0)
    k = int_of(np.frombuffer(sc1.tobytes(), np.uint64)) * n % curve.order
    exp = curve.canonical(curve.mul(k, curve.raw_to_affine(base_pts[0].tobytes())))
    assert canon(curve, ctx.msm(curve.cid, sc, pts)) == exp
    # (ii) P / -P pairs with equal scalars cancel; identity points and zero scalars are ignored
    pa = curve.raw_to_affine(base_pts[1].tobytes())
    neg = np.frombuffer(curve.affine_to_raw(curve.neg(pa)), np.uint64)
    pts = np.zeros((n, 8), np.uint64); sc = cref.gen_scalars(curve.cid, 11, n)
    for i in range(0, n - 2, 2):
        pts[i] = base_pts[1]; pts[i + 1] = neg; sc[i + 1] = sc[i]
    pts[n - 2] = 0                                     # identity point with a random scalar
    pts[n - 1] = base_pts[2]; sc[n - 1] = 0            # zero scalar
    assert canon(curve, ctx.msm(curve.cid, sc, pts)) == bytes(64)
    # (iii) extreme scalars: 1, order-1, 2^k, all with distinct points
    specials = [1, curve.order - 1, 2, 1 << 15, 1 << 16, (1 << 16) - 1, 1 << 253, curve.order - 2, (1 << 128) - 1, 0x8000]
    pts = cref.gen_points(curve.cid, 12, len(specials))
    sc = np.array([np.frombuffer(int(s).to_bytes(32, "little"), np.uint8) for s in specials])
    assert canon(curve, ctx.msm(curve.cid, sc, pts)) == canon(curve, cref.msm_naive(curve.cid, sc, pts))


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("c_bits", [0, 4, 8, 11])
def test_msm_pyramid_quad_additions_special_pairs(ctx, curve, c_bits):
    """the four-lanes-per-addition form of the narrow pyramid steps (XYZZ29::add4_mem, option pyr_quad) hands every pair it
    cannot do -- an identity operand, equal points (a doubling), opposite points -- to the complete addition: buckets built so
    that neighbouring bucket sums ARE equal (the same point under scalars 2k and 2k+1 in every window), opposite (P and -P) or
    empty, against the oracle and against one lane per addition"""
    n = 512
    g = cref.gen_points(curve.cid, 5001, 3)
    pa = curve.raw_to_affine(g[1].tobytes())
    neg = np.frombuffer(curve.affine_to_raw(curve.neg(pa)), np.uint64)
    pts = np.zeros((n, 8), np.uint64); sc = np.zeros((n, 32), np.uint8)
    width = c_bits if c_bits else 6
    for i in range(n):
        k = i % 64
        if i < 256:    # equal bucket sums: the same point in buckets 2j and 2j + 1 of every window
            pts[i] = g[0]; digit = (2 * (k // 2) + (k & 1)) % (1 << width) or 1
        elif i < 448:  # opposite bucket sums
            pts[i] = g[1] if (k & 1) == 0 else neg; digit = (2 * (k // 2) + (k & 1)) % (1 << width) or 1
        else:          # sparse: most buckets of the upper windows stay empty
            pts[i] = g[2]; digit = 1 + (k % 3)
        v = 0
        for w in range(0, 250 // width):
            v |= digit << (w * width)
        sc[i] = np.frombuffer(int(v % curve.order).to_bytes(32, "little"), np.uint8)
    exp = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 4))
    res = []
    ctx.set_option("window_bits", c_bits)
    try:
        for mode in (0, 2):
            ctx.set_option("pyr_quad", mode)
            res.append(canon(curve, ctx.msm(curve.cid, sc, pts)))
    finally:
        ctx.set_option("window_bits", 0); ctx.set_option("pyr_quad", 0)
    assert res[0] == exp and res[1] == exp


def test_msm_skewed_buckets_all_windows_equal(fctx):
    ctx = fctx
    """all scalars equal with c=16 windows: every window has a single bucket holding all points"""
    curve = pyref.BN254_G1
    n = 3000
    pts = cref.gen_points(curve.cid, 77, 64)
    pts = np.tile(pts, (n // 64 + 1, 1))[:n]
    sc = np.repeat(cref.gen_scalars(curve.cid, 78, 1), n, axis=0)
    ctx.set_option("window_bits", 16)
    try:
        out = ctx.msm(curve.cid, sc, pts)
    finally:
        ctx.set_option("window_bits", 0)
    assert canon(curve, out) == canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))


@pytest.mark.parametrize("logn", [12, 18])
def test_msm_extreme_skew_at_scale(fctx, logn):
    """every scalar equal (one bucket per window holds all n entries: the edge-record levels do all
    the merging), then additionally every point equal -- the reference's lhs_test shape, where every
    addition of the first pass is a doubling (src/argument_witness_calc.rs:141-142)"""
    ctx = fctx
    curve = pyref.BN254_G1
    n = 1 << logn
    q = cref.gen_points(curve.cid, 500 + logn, 1)[0]
    dp = ctx.gen_walk(curve.cid, q, n)
    s1 = cref.gen_scalars(curve.cid, 501, 1)
    sc = np.repeat(s1, n, axis=0)
    ds = ctx.to_device(sc)
    s_int = int.from_bytes(s1[0].tobytes(), "little")
    k = s_int * (n * (n + 1) // 2) % curve.order                 # sum_i s * (i+1)
    assert canon(curve, ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)) == canon(curve, cref.scalar_mul(curve.cid, k, q))
    same = ctx.to_device(np.repeat(q.reshape(1, 8), n, axis=0))
    k2 = s_int * n % curve.order
    assert canon(curve, ctx.msm_device(curve.cid, ds.ptr, same.ptr, n)) == canon(curve, cref.scalar_mul(curve.cid, k2, q))


@pytest.mark.parametrize("chunk,slice_,wave_th,digits", [(8, 0, 0, 40), (8, 0, 1, 40), (8, 33, 0, 2), (8, 33, 0, 3), (1, 64, 0, 300), (3, 0, 0, 1000), (17, 100, 3, 20)])
def test_msm_edge_record_merge_every_class(fctx, chunk, slice_, wave_th, digits):
    """the merge of the pieces a bucket leaves in the accumulate chunks it straddles (kernels_ec.cuh): scalars drawn from a
    small alphabet give buckets of n / digits entries, so that with short chunks every size class is met -- two pieces,
    3..8 (serial), 9..32 (one wave each, or serial when option merge_wave_th says so), slices of long buckets and the
    final pass over a multi-slice bucket's partial sums; the queue counters say which of them ran"""
    ctx = fctx
    curve = pyref.BN254_G1
    n = 6000
    rng = np.random.default_rng(chunk * 1000 + digits)
    alphabet = cref.gen_scalars(curve.cid, 900 + digits, digits)
    sc = alphabet[rng.integers(0, digits, n)]
    pts = cref.gen_points(curve.cid, 901, 128)
    pts = np.tile(pts, (n // 128 + 1, 1))[:n]
    for k, v in (("window_bits", 13), ("chunk", chunk), ("merge_slice", slice_), ("merge_wave_th", wave_th)):
        ctx.set_option(k, v)
    try:
        out = ctx.msm(curve.cid, sc, pts)
        counts = ctx.last_merge_counts()
    finally:
        for k in ("window_bits", "chunk", "merge_slice", "merge_wave_th"):
            ctx.set_option(k, 0)
    assert canon(curve, out) == canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    pieces = n / digits / chunk                       # of a typical bucket
    if pieces > 40:
        assert counts[2] > 0, counts                  # long buckets were sliced
        if slice_ and pieces > 2 * slice_:
            assert counts[3] > 0, counts              # and the multi-slice final pass ran
    elif 10 < pieces < 30:
        assert counts[1] > 0, counts
    elif 3.5 < pieces < 7:
        assert counts[0] > 0, counts


def test_lhs_edge_record_merge_long_buckets(fctx):
    """negabase digits: B - 1 buckets per position, so every bucket is long; small slices force several per bucket"""
    ctx = fctx
    curve = pyref.GRUMPKIN
    n = 5000
    pts = cref.gen_points(curve.cid, 910, n)
    sc = cref.gen_scalars(curve.cid, 911, n, half=True)
    jac = cref.aff_to_jac(curve.cid, pts)
    ctx.set_option("merge_slice", 40); ctx.set_option("chunk", 4)
    try:
        carry, carries = ctx.lhs_msm(curve.cid, sc, jac, 5)
        counts = ctx.last_merge_counts()
    finally:
        ctx.set_option("merge_slice", 0); ctx.set_option("chunk", 0)
    ecarry, ecarries = cref.lhs_msm(curve.cid, sc, jac, 5)
    assert canon(curve, carry) == canon(curve, ecarry)
    for i in range(carries.shape[0]):
        assert canon(curve, carries[i]) == canon(curve, ecarries[i])
    assert counts[2] > 0 and counts[3] > 0, counts


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_batch_equals_separate_calls(ctx, curve):
    """lemsm_msm_batch_device: K MSMs over one set of points, pipelined over two lanes inside the library -- the results of
    K separate calls, in order; a non-canonical scalar in one call fails the batch with that call's status"""
    n = 7000
    pts = cref.gen_points(curve.cid, 950, n)
    dp = ctx.to_device(pts)
    scs = [cref.gen_scalars(curve.cid, 960 + k, n) for k in range(5)]
    dss = [ctx.to_device(s) for s in scs]
    got = ctx.msm_batch_device(curve.cid, [d.ptr for d in dss], dp.ptr, n)
    assert got.shape == (5, 12)
    for k in range(5):
        assert canon(curve, got[k]) == canon(curve, cref.best_multiexp(curve.cid, scs[k], pts, 8)), k
        assert canon(curve, got[k]) == canon(curve, ctx.msm_device(curve.cid, dss[k].ptr, dp.ptr, n)), k
    assert ctx.msm_batch_device(curve.cid, [], dp.ptr, n).shape[0] == 0
    one = ctx.msm_batch_device(curve.cid, [dss[2].ptr], dp.ptr, n)
    assert canon(curve, one[0]) == canon(curve, got[2])
    bad = scs[3].copy(); bad[4321] = 0xff
    dbad = ctx.to_device(bad)
    with pytest.raises(api.ScalarOutOfRange) as e:
        ctx.msm_batch_device(curve.cid, [dss[0].ptr, dss[1].ptr, dbad.ptr, dss[4].ptr], dp.ptr, n)
    assert e.value.index == 4321
    # the context (both lanes) is usable afterwards
    again = ctx.msm_batch_device(curve.cid, [d.ptr for d in dss[:3]], dp.ptr, n)
    for k in range(3):
        assert canon(curve, again[k]) == canon(curve, got[k])


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_batch_with_bases_host_scalars(ctx, curve):
    """lemsm_msm_batch_with_bases: K MSMs over resident bases with the scalar vectors in host memory (uploads pipelined with the
    compute): the results of K lemsm_msm_with_bases calls; shorter than the bases; a bad scalar fails the batch with its index"""
    n = 9000
    pts = cref.gen_points(curve.cid, 970, n)
    bases = ctx.bases_upload(curve.cid, pts)
    scs = [cref.gen_scalars(curve.cid, 980 + k, n) for k in range(5)]
    got = ctx.msm_batch_with_bases(bases, scs)
    assert got.shape == (5, 12)
    for k in range(5):
        assert canon(curve, got[k]) == canon(curve, cref.best_multiexp(curve.cid, scs[k], pts, 8)), k
        assert canon(curve, got[k]) == canon(curve, ctx.msm_with_bases(bases, scs[k])), k
    m = 6001
    short = ctx.msm_batch_with_bases(bases, [s[:m] for s in scs[:3]])
    for k in range(3):
        assert canon(curve, short[k]) == canon(curve, cref.best_multiexp(curve.cid, scs[k][:m], pts[:m], 8)), k
    assert ctx.msm_batch_with_bases(bases, []).shape[0] == 0
    one = ctx.msm_batch_with_bases(bases, [scs[2]])
    assert canon(curve, one[0]) == canon(curve, got[2])
    bad = scs[3].copy(); bad[1234] = 0xff
    with pytest.raises(api.ScalarOutOfRange) as e:
        ctx.msm_batch_with_bases(bases, [scs[0], bad, scs[4]])
    assert e.value.index == 1234
    again = ctx.msm_batch_with_bases(bases, scs[:2])
    for k in range(2):
        assert canon(curve, again[k]) == canon(curve, got[k])


def test_msm_all_zero_scalars_and_all_identity_points(fctx):
    ctx = fctx
    curve = pyref.GRUMPKIN
    n = 5000
    pts = cref.gen_points(curve.cid, 600, 16); pts = np.tile(pts, (n // 16 + 1, 1))[:n]
    zeros = np.zeros((n, 32), np.uint8)
    assert canon(curve, ctx.msm(curve.cid, zeros, pts)) == bytes(64)
    sc = cref.gen_scalars(curve.cid, 601, n)
    assert canon(curve, ctx.msm(curve.cid, sc, np.zeros((n, 8), np.uint64))) == bytes(64)
    # one real contribution among identities / zeros
    sc2 = zeros.copy(); sc2[n - 1] = sc[n - 1]
    exp = cref.scalar_mul(curve.cid, int.from_bytes(sc[n - 1].tobytes(), "little"), pts[n - 1])
    assert canon(curve, ctx.msm(curve.cid, sc2, pts)) == canon(curve, exp)


def test_msm_rejects_non_canonical_scalar(ctx):
    """best_multiexp reads to_repr() bytes, which are always < order; a non-canonical scalar is
    reported with its index instead of being bucketed"""
    curve = pyref.BN254_G1
    n = 300
    pts = cref.gen_points(curve.cid, 1, 8); pts = np.tile(pts, (n // 8 + 1, 1))[:n]
    sc = cref.gen_scalars(curve.cid, 2, n)
    for bad in (curve.order, curve.order + 5, (1 << 256) - 1):
        s2 = sc.copy(); s2[123] = np.frombuffer(int(bad).to_bytes(32, "little"), np.uint8)
        with pytest.raises(api.ScalarOutOfRange) as ei:
            ctx.msm(curve.cid, s2, pts)
        assert ei.value.index == 123
    s2 = sc.copy(); s2[7] = np.frombuffer(int(curve.order - 1).to_bytes(32, "little"), np.uint8)
    assert canon(curve, ctx.msm(curve.cid, s2, pts)) == canon(curve, cref.best_multiexp(curve.cid, s2, pts, 4))


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,slab_bits", [(20000, 12), (4096 * 3, 12), (4097, 12), (70001, 14)])
def test_msm_host_entry_slab_pipeline(fctx, curve, n, slab_bits):
    """the host-pointer entry uploads slab k+1 while slab k is accumulated; several slabs with a
    ragged last one give the oracle's result and the device-pointer entry's"""
    ctx = fctx
    ctx.set_option("host_slab_bits", slab_bits)
    try:
        pts = cref.gen_points(curve.cid, 7000 + n, n)
        sc = cref.gen_scalars(curve.cid, 8000 + n, n)
        out = ctx.msm(curve.cid, sc, pts)
        assert canon(curve, out) == canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
        ds, dp = ctx.to_device(sc), ctx.to_device(pts)
        assert canon(curve, ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)) == canon(curve, out)
    finally:
        ctx.set_option("host_slab_bits", 0)


@pytest.mark.parametrize("groups", [0, 3])
def test_msm_device_entry_multi_slab(fctx, groups):
    """n above the slab size (2^24 pairs by default; shrunk here): slabs run back to back on one
    queue reusing one workspace (or drained per slab with pipelined groups) and add up exactly"""
    ctx = fctx
    curve = pyref.BN254_G1
    n = 5 * 4096 + 123
    pts = cref.gen_points(curve.cid, 31, n); sc = cref.gen_scalars(curve.cid, 32, n)
    exp = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    ctx.set_option("slab_bits", 12); ctx.set_option("groups", groups)
    try:
        assert canon(curve, ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)) == exp
        s2 = sc.copy(); s2[4 * 4096 + 5] = np.frombuffer(int(curve.order).to_bytes(32, "little"), np.uint8)
        d2 = ctx.to_device(s2)
        with pytest.raises(api.ScalarOutOfRange) as ei:
            ctx.msm_device(curve.cid, d2.ptr, dp.ptr, n)
        assert ei.value.index == 4 * 4096 + 5
    finally:
        ctx.set_option("slab_bits", 0); ctx.set_option("groups", 0)


@pytest.mark.parametrize("n,slab_bits,abi_points", [(2 * 4096 + 1, 12, 0), (3 * 4096 + 4095, 12, 2), (8 * 4096, 12, 0), (9 * 4096 + 7, 12, 0)])
def test_msm_slabs_share_one_tail(fctx, n, slab_bits, abi_points):
    """several slabs per call: by default every slab accumulates into its own bucket area and the call runs ONE bucket
    reduction (k_sum_slabs + one pyramid, one block of records); option slab_tail = 2 runs a reduction per slab and adds
    the records on the host.  Both equal the oracle; sizes: a one-point ragged last slab (its accumulate form differs from
    the full slabs'), exactly 8 slabs, 10 slabs (above the limit: a tail per slab either way); also through the sharded
    entry, whose exchange carries one record block per rank.  (64 Ki-point slabs: test_msm_slabs_share_one_tail_64k.)"""
    ctx = fctx
    curve = pyref.BN254_G1                                          # (Grumpkin: test_msm_slabs_share_one_tail_64k)
    pts = cref.gen_points(curve.cid, 4100 + n % 97, n); sc = cref.gen_scalars(curve.cid, 4200 + n % 89, n)
    if n > 3 * 4096:
        sc[4096:4096 + 300] = sc[7]; pts[4096 + 5] = 0            # a long bucket and an identity inside the second slab
    exp = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    ctx.set_option("slab_bits", slab_bits); ctx.set_option("abi_points", abi_points)
    try:
        for mode in (0, 2):
            ctx.set_option("slab_tail", mode)
            assert canon(curve, ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)) == exp, mode
            assert canon(curve, ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, 3)) == exp, mode
    finally:
        ctx.set_option("slab_bits", 0); ctx.set_option("abi_points", 0); ctx.set_option("slab_tail", 0)
    ds.free(); dp.free()


def test_msm_slabs_share_one_tail_64k(ctx):
    """the same with 64 Ki-point slabs (three and a ragged bit), default arithmetic, one curve: the slab size at which the sort
    takes its full-size path (4096-entry ranges, k_binsort)"""
    curve = pyref.GRUMPKIN
    n = 2 * 65536 + 100
    pts = cref.gen_points(curve.cid, 4301, n); sc = cref.gen_scalars(curve.cid, 4302, n)
    exp = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    ctx.set_option("slab_bits", 16)
    try:
        for mode in (0, 2):
            ctx.set_option("slab_tail", mode)
            assert canon(curve, ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)) == exp, mode
    finally:
        ctx.set_option("slab_bits", 0); ctx.set_option("slab_tail", 0)
    ds.free(); dp.free()


@pytest.mark.parametrize("n,opts,host", [(130972, {"slab_bits": 16, "merge_slice": 33}, False), (262017, {"host_slab_bits": 16, "merge_wave_th": 1}, True),
                                         ((1 << 19) + 60000, {}, True)],
                         ids=["device-2^16-slabs-ragged-last", "host-2^16-slabs-ragged-last", "host-default-slabs-last-slab-below-2^16"])
def test_msm_ragged_last_slab_workspace(ctx, n, opts, host):
    """regression (found by tests/fuzz_gpu.py): a last slab of fewer than 2^16 pairs uses shorter pass-1 ranges,
    hence MORE of them than a full slab of 2^16..2^19 pairs; the workspace is sized over every slab's layout"""
    curve = pyref.BN254_G1
    pts = cref.gen_points(curve.cid, 11, 300)[np.random.default_rng(1).integers(0, 300, n)]
    sc = cref.gen_scalars(curve.cid, 12, n)
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        if host:
            got = ctx.msm(curve.cid, sc, pts)
        else:
            ds, dp = ctx.to_device(sc), ctx.to_device(pts)
            got = ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)
    finally:
        for k in opts:
            ctx.set_option(k, 0)
    assert canon(curve, got) == canon(curve, cref.best_multiexp(curve.cid, sc, pts, 16))


def test_msm_host_entry_bad_scalar_in_later_slab(ctx):
    curve = pyref.BN254_G1
    n = 3 * 4096 + 17
    ctx.set_option("host_slab_bits", 12)
    try:
        pts = cref.gen_points(curve.cid, 1, 8); pts = np.tile(pts, (n // 8 + 1, 1))[:n]
        sc = cref.gen_scalars(curve.cid, 2, n)
        for idx in (5, 4096, 2 * 4096 + 9, n - 1):
            s2 = sc.copy(); s2[idx] = np.frombuffer(int(curve.order).to_bytes(32, "little"), np.uint8)
            with pytest.raises(api.ScalarOutOfRange) as ei:
                ctx.msm(curve.cid, s2, pts)
            assert ei.value.index == idx
        s2 = sc.copy()
        for idx in (2 * 4096 + 9, 4100):     # two offenders: the first one is reported
            s2[idx] = np.frombuffer(int(curve.order + 1).to_bytes(32, "little"), np.uint8)
        with pytest.raises(api.ScalarOutOfRange) as ei:
            ctx.msm(curve.cid, s2, pts)
        assert ei.value.index == 4100
    finally:
        ctx.set_option("host_slab_bits", 0)


def test_msm_length_mismatch(ctx):
    pts = cref.gen_points(0, 1, 3); sc = cref.gen_scalars(0, 2, 2)
    with pytest.raises(api.LengthMismatch, match="incompatible amount of coefficients"):
        ctx.msm(0, sc, pts)


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_msm_point_partials_sum(ctx, parts):
    """the alternative multi-GPU partition (pairs split across ranks): the ranks'
    Jacobian partials, summed by lemsm_jacobian_sum, are the same group element as the whole MSM"""
    from halo2_liam_eagen_msm_amd import dist as ldist
    curve = pyref.BN254_G1
    n = 5003
    pts = cref.gen_points(curve.cid, 61, n); sc = cref.gen_scalars(curve.cid, 62, n)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    partials = []
    for r in range(parts):
        a, b = ldist.point_range(n, parts, r)
        partials.append(ctx.msm_device(curve.cid, ds.ptr + a * 32, dp.ptr + a * 64, b - a))
    got = api.jacobian_sum(curve.cid, np.stack(partials))
    assert canon(curve, got) == canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    assert canon(curve, got) == canon(curve, ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n))


# ------------------------------------------------------------------ window sharding
@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_msm_window_partials_combine(ctx, parts):
    curve = pyref.BN254_G1
    n = 2000
    pts = cref.gen_points(curve.cid, 41, n); sc = cref.gen_scalars(curve.cid, 42, n)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    W, rec = ctx.msm_plan(curve.cid, n)
    bounds = [W * i // parts for i in range(parts + 1)]
    chunks = [ctx.msm_partial_device(curve.cid, ds.ptr, dp.ptr, n, bounds[i], bounds[i + 1]) for i in range(parts)]
    allp = np.concatenate(chunks)
    assert allp.size == W * rec
    out = ctx.msm_combine(curve.cid, n, allp)
    assert canon(curve, out) == canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    assert canon(curve, ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)) == canon(curve, out)


# ------------------------------------------------------------------ negabase + lhs
@pytest.mark.parametrize("base", [3, 4, 5, 16, 17, 255])
def test_negbase_batch_matches_oracle(ctx, base):
    c = pyref.GRUMPKIN
    n = 3000
    sc = cref.gen_scalars(c.cid, 50 + base, n, half=True)
    bound = pyref.scalar_bound(c.order)
    edge = [0, 1, base - 1, base, base + 1, bound - 1, bound - 2, (1 << 126), (1 << 127) - 1 if (1 << 127) - 1 < bound else bound - 3]
    for i, v in enumerate(edge):
        sc[i] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
    d = api.num_digits(c.cid, base)
    assert d == pyref.num_digits(c.order, base)
    got = ctx.negbase_decompose_batch(sc, base, d)
    exp = cref.negbase_decompose_batch(sc, base, d)
    assert np.array_equal(got, exp)
    # single-scalar mirror of the reference function, incl. the reference's own recomposition test
    x = 0xDEADBEEF
    digs = api.negbase_decompose(x, 17, ctx)
    assert digs == pyref.negbase_decompose(x, 17)
    acc = 0
    for dg in reversed(digs):
        acc = acc * (-17) + dg
    assert acc == x
    assert api.negbase_decompose(0, 5, ctx) == []


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("base,n", [(5, 400), (16, 700), (3, 50), (255, 120), (17, 33)])
def test_lhs_matches_oracle(fctx, curve, base, n):
    ctx = fctx
    pts_aff = cref.gen_points(curve.cid, 60 + n, n)
    sc = cref.gen_scalars(curve.cid, 61 + n, n, half=True)
    pts = jacobian_with_random_z(curve, pts_aff, 62 + n)
    carry, carries = ctx.lhs_msm(curve.cid, sc, pts, base)
    ecarry, ecarries = cref.lhs_msm(curve.cid, sc, cref.aff_to_jac(curve.cid, pts_aff), base)
    assert canon(curve, carry) == canon(curve, ecarry)
    assert carries.shape == ecarries.shape
    for i in range(carries.shape[0]):
        assert canon(curve, carries[i]) == canon(curve, ecarries[i]), i
    # lhs_test: equals best_multiexp on the same inputs (src/argument_witness_calc.rs:144-147)
    assert canon(curve, carry) == canon(curve, ctx.msm(curve.cid, sc, pts_aff))


@pytest.mark.parametrize("base,n", [(16, 70000), (17, 66000), (3, 66000), (18, 66000)])
def test_lhs_few_bins_replicated_cursors_many_blocks(ctx, base, n):
    """r03: with at most 16 buckets per digit position (base <= 17) pass 1 counts and ranks in 16 copies of its LDS cursors
    (k_count1, k_scatter1); base 18 (17 bins) takes the single-cursor form.  Several pass-1 blocks per position, all carries."""
    curve = pyref.GRUMPKIN
    seed_pts = cref.gen_points(curve.cid, 90, 64)
    pts_aff = np.tile(seed_pts, ((n + 63) // 64, 1))[:n].copy()
    sc = cref.gen_scalars(curve.cid, 91 + n, n, half=True)
    pts = cref.aff_to_jac(curve.cid, pts_aff)
    ecarry, ecarries = cref.lhs_msm(curve.cid, sc, pts, base)
    carry, carries = ctx.lhs_msm(curve.cid, sc, pts, base)
    assert canon(curve, carry) == canon(curve, ecarry)
    assert carries.shape == ecarries.shape
    for i in range(carries.shape[0]):
        assert canon(curve, carries[i]) == canon(curve, ecarries[i]), i


def test_lhs_reference_test_shape(fctx):
    ctx = fctx
    """lhs_test itself: Grumpkin, base 5, one (scalar, point) pair replicated; here n = 2000"""
    c = pyref.GRUMPKIN
    n = 2000
    pt = cref.gen_points(c.cid, 7, 1); s = cref.gen_scalars(c.cid, 8, 1, half=True)
    pts_aff = np.repeat(pt, n, axis=0); sc = np.repeat(s, n, axis=0)
    pts = jacobian_with_random_z(c, pts_aff[:1], 9).repeat(n, axis=0)
    a = api.best_multiexp(sc, pts_aff, "grumpkin", ctx)
    b, _ = api.compute_lhs_witness(sc, pts, 5, "grumpkin", ctx)
    assert canon(c, a) == canon(c, b)
    k = int_of(np.frombuffer(s.tobytes(), np.uint64)) * n % c.order
    assert canon(c, a) == c.canonical(c.mul(k, c.raw_to_affine(pt[0].tobytes())))


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("base,n", [(5, 60), (16, 45), (3, 30)])
def test_lhs_witness_point_lists_match_reference_loop(ctx, curve, base, n):
    """api.compute_lhs_witness_inputs rebuilds the `tmp` lists of src/argument_witness_calc.rs:108-127
    from GPU outputs by indexing only; compared point by point with a big-int restatement of that loop,
    and each list sums to the identity (what makes it a principal divisor)."""
    rng = pyref.SplitMix64(4100 + base + n + curve.cid)
    pts = pyref.gen_points(curve, rng, n)
    sc = pyref.gen_scalars_half(rng, n, curve.order)
    sc[0] = 0; sc[1] = 1                                       # empty and one-digit decompositions
    d = pyref.num_digits(curve.order, base)
    # the reference's loop, restated on integers
    digs = [pyref.negbase_digits_padded(x, base, d)[::-1] for x in sc]
    carry = None; expect = []
    for i in range(d):
        tmp = []
        if carry is not None:
            tmp += [curve.neg(carry)] * base
        carry = curve.mul(base, curve.neg(carry)) if carry is not None else None
        for j in range(n):
            k = digs[j][i]
            if k:
                m = curve.mul(k, pts[j]); tmp.append(m); carry = curve.add(carry, m)
        tmp.append(curve.neg(carry) if carry is not None else None)
        expect.append(tmp)
    sc_b = np.frombuffer(pyref.scalars_to_bytes(sc), np.uint8).reshape(-1, 32)
    pj = np.array([np.frombuffer(curve.affine_to_jacobian_raw(p_, 1 + rng.next256() % (curve.fp - 1)), np.uint64) for p_ in pts])
    got_carry, lists = api.compute_lhs_witness_inputs(sc_b, pj, base, curve.cid, ctx)
    assert canon(curve, got_carry) == curve.canonical(carry)
    assert len(lists) == d
    for i in range(d):
        assert lists[i].shape == (len(expect[i]), 8), i
        for k, e in enumerate(expect[i]):
            assert lists[i][k].tobytes() == curve.affine_to_raw(e), (i, k)
        acc = None
        for row in lists[i]:
            acc = curve.add(acc, curve.raw_to_affine(row.tobytes()))
        assert acc is None, i


def test_lhs_errors(ctx):
    c = pyref.GRUMPKIN
    pts = cref.aff_to_jac(c.cid, cref.gen_points(c.cid, 5, 5))
    sc = cref.gen_scalars(c.cid, 6, 5, half=True)
    bad = sc.copy(); bad[3] = np.frombuffer(int(pyref.scalar_bound(c.order)).to_bytes(32, "little"), np.uint8)
    with pytest.raises(api.ScalarOutOfRange) as ei:
        ctx.lhs_msm(c.cid, bad, pts, 5)
    assert ei.value.index == 3
    with pytest.raises(api.LengthMismatch):
        ctx.lhs_msm(c.cid, sc[:4], pts, 5)
    with pytest.raises(api.BadBase):
        ctx.lhs_msm(c.cid, sc, pts, 2)
    # empty input: identity
    carry, carries = ctx.lhs_msm(c.cid, sc[:0], pts[:0], 5)
    assert canon(c, carry) == bytes(64)


@pytest.mark.parametrize("parts", [1, 2, 4, 8])
def test_lhs_position_partials_combine(ctx, parts):
    c = pyref.BN254_G1
    n, base = 500, 16
    pts = cref.gen_points(c.cid, 71, n); sc = cref.gen_scalars(c.cid, 72, n, half=True)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    d, rec = ctx.lhs_plan(c.cid, base)
    bounds = [d * i // parts for i in range(parts + 1)]
    allp = np.concatenate([ctx.lhs_partial_device(c.cid, ds.ptr, dp.ptr, n, base, bounds[i], bounds[i + 1]) for i in range(parts)])
    carry, carries = ctx.lhs_combine(c.cid, base, allp)
    ecarry, ecarries = cref.lhs_msm(c.cid, sc, cref.aff_to_jac(c.cid, pts), base)
    assert canon(c, carry) == canon(c, ecarry)
    assert all(canon(c, carries[i]) == canon(c, ecarries[i]) for i in range(d))


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_precompute_multiplicities(ctx, curve):
    pts_aff = cref.gen_points(curve.cid, 81, 5)
    pts = jacobian_with_random_z(curve, pts_aff, 82)
    pts[4] = 0   # identity
    for base in (2, 5, 16):
        got = ctx.precompute_multiplicities(curve.cid, pts, base)
        for j in range(5):
            exp = cref.precompute_multiplicities(curve.cid, pts[j], base)
            for k in range(base - 1):
                assert canon(curve, got[j, k]) == canon(curve, exp[k]), (base, j, k)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_precompute_multiplicities_affine_table(ctx, curve):
    """the fixed-table form of src/config.rs:542-560: affine (x,y) of k*P_j"""
    n = 37
    pts_aff = cref.gen_points(curve.cid, 91, n)
    pts = jacobian_with_random_z(curve, pts_aff, 92)
    pts[5] = 0   # identity input -> identity multiples
    for base in (2, 5, 16, 255):
        got = ctx.precompute_multiplicities_affine(curve.cid, pts, base)
        assert got.shape == (n, base - 1, 8)
        for j in (0, 5, 17, n - 1):
            pa = None if j == 5 else curve.raw_to_affine(pts_aff[j].tobytes())
            for k in (1, 2, base - 1) if base > 2 else (1,):
                exp = curve.mul(k, pa)
                assert curve.raw_to_affine(got[j, k - 1].tobytes()) == exp, (base, j, k)


@pytest.mark.parametrize("n", [4095, 4096, 4097, 8191, 8193, 12289, 65535, 65537])
def test_msm_ragged_sizes_around_block_and_tile_boundaries(ctx, n):
    """sizes straddling the pass-1 range (4096), the pass-2 tile (8192) and 2^16; c = 16 forced so that
    the big-bucket-count path (LB = 7, 256 bins per window) runs at small n"""
    curve = pyref.BN254_G1
    q = cref.gen_points(curve.cid, 300 + n % 7, 1)[0]
    dp = ctx.gen_walk(curve.cid, q, n)
    sc = cref.gen_scalars(curve.cid, 301 + n, n)
    ds = ctx.to_device(sc)
    exp = canon(curve, cref.scalar_mul(curve.cid, cref.walk_dot(curve.cid, sc), q))
    assert canon(curve, ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)) == exp
    for cb in (16, 17):
        ctx.set_option("window_bits", cb)
        try:
            out1 = ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)
            out2 = ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)
        finally:
            ctx.set_option("window_bits", 0)
        assert canon(curve, out1) == exp, cb
        assert canon(curve, out2) == exp, cb  # repeatable (atomics reorder the summation, never the group element)


def test_lhs_large_walk_relation(ctx):
    """2^16-point negabase path (base 16, the bench configuration at reduced n) through the device entry"""
    curve = pyref.GRUMPKIN
    n, base = 1 << 16, 16
    q = cref.gen_points(curve.cid, 411, 1)[0]
    dp = ctx.gen_walk(curve.cid, q, n)
    sc = cref.gen_scalars(curve.cid, 412, n, half=True)
    ds = ctx.to_device(sc)
    carry, carries = ctx.lhs_msm_device(curve.cid, ds.ptr, dp.ptr, n, base)
    exp = canon(curve, cref.scalar_mul(curve.cid, cref.walk_dot(curve.cid, sc), q))
    assert canon(curve, carry) == exp
    assert canon(curve, carries[-1]) == exp
    # carry_i recursion: carry_i = -B*carry_{i-1} + S_i  =>  all carries are multiples of Q; check the first non-trivial one
    d = api.num_digits(curve.cid, base)
    digs = ctx.negbase_decompose_batch(sc, base, d)
    top = int(sum(int(digs[j, d - 1]) * (j + 1) for j in range(n)) % curve.order)
    assert canon(curve, carries[0]) == canon(curve, cref.scalar_mul(curve.cid, top, q))


# ------------------------------------------------------------------ larger sizes: properties
@pytest.mark.parametrize("curve,logn", [(pyref.BN254_G1, 16), (pyref.GRUMPKIN, 16), (pyref.BN254_G1, 20), (pyref.GRUMPKIN, 22), (pyref.BN254_G1, 24)],
                         ids=["bn254-2^16", "grumpkin-2^16", "bn254-2^20", "grumpkin-2^22", "bn254-2^24-bench-size-17-bit-windows"])
def test_msm_walk_relation_large(ctx, curve, logn):
    """P_i = (i+1) Q  =>  sum s_i P_i == (sum s_i (i+1)) Q, checked with one scalar multiplication;
    the device-generated points are spot-checked against the oracle."""
    n = 1 << logn
    q = cref.gen_points(curve.cid, 123, 1)[0]
    dp = ctx.gen_walk(curve.cid, q, n)
    pts_head = dp.download(np.uint64, 64 * 1024).reshape(-1, 8)
    exp_head = cref.gen_walk(curve.cid, q, 1024)
    assert np.array_equal(pts_head, exp_head)
    sc = cref.gen_scalars(curve.cid, 124 + logn, n)
    ds = ctx.to_device(sc)
    out = ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)
    dot = cref.walk_dot(curve.cid, sc)
    assert canon(curve, out) == canon(curve, cref.scalar_mul(curve.cid, dot, q))
    if logn <= 16:
        pts = dp.download(np.uint64).reshape(-1, 8)
        assert canon(curve, out) == canon(curve, cref.best_multiexp(curve.cid, sc, pts, 16))
        # linearity: MSM(s) + MSM(t) == MSM(s+t mod order)
        sc2 = cref.gen_scalars(curve.cid, 999, n)
        ssum = np.zeros_like(sc)
        a = [int.from_bytes(sc[i].tobytes(), "little") for i in range(0, n, 1)]
        b = [int.from_bytes(sc2[i].tobytes(), "little") for i in range(0, n, 1)]
        ssum = np.frombuffer(b"".join(((x + y) % curve.order).to_bytes(32, "little") for x, y in zip(a, b)), np.uint8).reshape(-1, 32)
        o2 = ctx.msm(curve.cid, sc2, pts); o3 = ctx.msm(curve.cid, ssum, pts)
        assert canon(curve, cref.jac_add(curve.cid, out, o2)) == canon(curve, o3)


# ------------------------------------------------------------------ BASELINE.json configs at full size
def _walk_carries_expected(curve, digs, d, base, q):
    """scalars c_i of the reference's carry recursion (src/argument_witness_calc.rs:118-125) when P_j = (j+1) Q:
    c_i = -B c_{i-1} + sum_j digit_{j,i} (j+1)  (digits MSB first), so that carry_i == c_i Q"""
    n = digs.shape[0]
    idx = np.arange(1, n + 1, dtype=np.int64)
    c = 0
    out = []
    for i in range(d):
        t = int((digs[:, d - 1 - i].astype(np.int64) * idx).sum())      # digit < 256, j+1 < 2^26: fits int64
        c = (-base * c + t) % curve.order
        out.append(c)
    return out


def test_lhs_bn254_2p20_base16_full_size(ctx):
    """BASELINE.json configs[1] at full size: 2^20-point BN254 G1 compute_lhs_witness MSM core, negabase B = 16
    (w = 4), half-width scalars.  Every one of the d = 33 per-digit carries (src/argument_witness_calc.rs:127)
    is checked through the walk identity carry_i == c_i Q."""
    curve = pyref.BN254_G1
    n, base = 1 << 20, 16
    q = cref.gen_points(curve.cid, 511, 1)[0]
    dp = ctx.gen_walk(curve.cid, q, n)
    sc = cref.gen_scalars(curve.cid, 512, n, half=True)
    ds = ctx.to_device(sc)
    carry, carries = ctx.lhs_msm_device(curve.cid, ds.ptr, dp.ptr, n, base)
    assert ctx.last_truncated_count() == 0
    exp = canon(curve, cref.scalar_mul(curve.cid, cref.walk_dot(curve.cid, sc), q))
    assert canon(curve, carry) == exp and canon(curve, carries[-1]) == exp
    d = api.num_digits(curve.cid, base)
    assert d == 33 and carries.shape[0] == d
    digs = ctx.negbase_decompose_batch(sc, base, d)
    for i, c in enumerate(_walk_carries_expected(curve, digs, d, base, q)):
        assert canon(curve, carries[i]) == canon(curve, cref.scalar_mul(curve.cid, c, q)), i


def test_msm_bn254_2p26_window_sharded_x8(ctx):
    """BASELINE.json configs[3]'s workload on one GPU: 2^26-point BN254 G1 MSM (4 slabs of 2^24), 16-bit windows,
    the eight window ranges an 8-GPU run hands to its ranks computed one after the other with
    lemsm_msm_partial_device, then lemsm_msm_combine; checked by the walk identity."""
    from halo2_liam_eagen_msm_amd import dist as ldist
    curve = pyref.BN254_G1
    n = 1 << 26
    q = cref.gen_points(curve.cid, 611, 1)[0]
    dp = ctx.gen_walk(curve.cid, q, n)
    sc = cref.gen_scalars(curve.cid, 612, n)
    ds = ctx.to_device(sc)
    exp = canon(curve, cref.scalar_mul(curve.cid, cref.walk_dot(curve.cid, sc), q))
    del sc
    ctx.set_option("window_bits", 16)
    try:
        W, rec = ctx.msm_plan(curve.cid, n)
        assert W == 16
        parts = [ctx.msm_partial_device(curve.cid, ds.ptr, dp.ptr, n, *ldist.window_range(W, 8, r)) for r in range(8)]
        allp = np.concatenate(parts)
        assert allp.size == W * rec
        out = ctx.msm_combine(curve.cid, n, allp)
    finally:
        ctx.set_option("window_bits", 0)
    assert canon(curve, out) == exp
    # the same through the C ABI's sharded entry (raw device records, the exchange simulated on one GPU)
    assert canon(curve, ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, 8)) == exp
    ds.free(); dp.free()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_lhs_multi_slab_digit_stride(fctx, curve):
    """regression: with several slabs per call the position-major digit matrix keeps the row stride
    of the WHOLE call; slab_bits = 12 and n = 3*4096+5 make four slabs"""
    ctx = fctx
    n, base = 3 * 4096 + 5, 5
    pts = cref.gen_points(curve.cid, 711, n); sc = cref.gen_scalars(curve.cid, 712, n, half=True)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    ecarry, ecarries = cref.lhs_msm(curve.cid, sc, cref.aff_to_jac(curve.cid, pts), base)
    ctx.set_option("slab_bits", 12)
    try:
        carry, carries = ctx.lhs_msm_device(curve.cid, ds.ptr, dp.ptr, n, base)
    finally:
        ctx.set_option("slab_bits", 0)
    assert canon(curve, carry) == canon(curve, ecarry)
    for i in range(carries.shape[0]):
        assert canon(curve, carries[i]) == canon(curve, ecarries[i]), i


def test_lhs_above_one_slab(ctx):
    """n > 2^24 on the negabase path (two slabs at the default slab size): final carry and first carry by the walk identity"""
    curve = pyref.GRUMPKIN
    n, base = (1 << 24) + 4097, 16
    q = cref.gen_points(curve.cid, 811, 1)[0]
    dp = ctx.gen_walk(curve.cid, q, n)
    sc = cref.gen_scalars(curve.cid, 812, n, half=True)
    ds = ctx.to_device(sc)
    carry, carries = ctx.lhs_msm_device(curve.cid, ds.ptr, dp.ptr, n, base)
    assert canon(curve, carry) == canon(curve, cref.scalar_mul(curve.cid, cref.walk_dot(curve.cid, sc), q))
    d = api.num_digits(curve.cid, base)
    digs = ctx.negbase_decompose_batch(sc, base, d)
    exp = _walk_carries_expected(curve, digs, d, base, q)
    for i in (0, 1, d - 1):
        assert canon(curve, carries[i]) == canon(curve, cref.scalar_mul(curve.cid, exp[i], q)), i
    ds.free(); dp.free()


def test_partial_entries_accept_empty_ranges(ctx):
    """a rank beyond the last window / digit position (world > W) owns an empty range: no output, no error"""
    curve = pyref.BN254_G1
    n = 500
    pts = cref.gen_points(curve.cid, 911, n); sc = cref.gen_scalars(curve.cid, 912, n, half=True)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    W, _ = ctx.msm_plan(curve.cid, n)
    assert ctx.msm_partial_device(curve.cid, ds.ptr, dp.ptr, n, W, W).size == 0
    assert ctx.msm_partial_device(curve.cid, ds.ptr, dp.ptr, n, 3, 3).size == 0
    d, _ = ctx.lhs_plan(curve.cid, 16)
    assert ctx.lhs_partial_device(curve.cid, ds.ptr, dp.ptr, n, 16, d, d).size == 0


def test_negbase_truncation_is_counted(ctx):
    """digits beyond d are dropped like `.take(d)` (src/argument_witness_calc.rs:99); the count of scalars it happened to is exposed"""
    sc = np.zeros((4, 32), np.uint8)
    sc[0, 0] = 1; sc[1, :2] = 255; sc[2, :8] = 255; sc[3, :15] = 255
    ctx.negbase_decompose_batch(sc, 16, 40)
    assert ctx.last_truncated_count() == 0
    digs = ctx.negbase_decompose_batch(sc, 16, 5)
    assert ctx.last_truncated_count() == 2
    assert digs.shape == (4, 5)


# ------------------------------------------------------------------ multi-GPU entries of the C ABI, rehearsed on one GPU
@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_msm_sharded_sim_matches_oracle(fctx, curve, world):
    """lemsm_debug_msm_sharded_sim = the code of lemsm_msm_sharded_device with the ranks' pipelines run one after the
    other and their raw record areas placed where the all-gather puts them (uneven window ranges at world 3)"""
    ctx = fctx
    n = 3000
    pts = cref.gen_points(curve.cid, 1001, n); sc = cref.gen_scalars(curve.cid, 1002 + world, n)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    exp = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    assert canon(curve, ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, world)) == exp


def test_msm_sharded_sim_multi_slab_and_more_ranks_than_windows(ctx):
    """several slabs per rank (their records are summed after the exchange) and world > W (ranks with no window)"""
    curve = pyref.BN254_G1
    n = 3 * 4096 + 77
    pts = cref.gen_points(curve.cid, 1011, n); sc = cref.gen_scalars(curve.cid, 1012, n)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    exp = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    ctx.set_option("slab_bits", 12)
    try:
        assert canon(curve, ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, 4)) == exp
        W, _ = ctx.msm_plan(curve.cid, 4096)
        assert canon(curve, ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, W + 3)) == exp
    finally:
        ctx.set_option("slab_bits", 0)
    assert canon(curve, ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, 0, 4)) == bytes(64)


@pytest.mark.parametrize("world,base", [(1, 5), (2, 16), (8, 16), (40, 16), (3, 255)])
def test_lhs_sharded_sim_matches_oracle(fctx, world, base):
    """digit-position sharding of the compute_lhs_witness core; world 40 > d = 33 leaves ranks without a position"""
    ctx = fctx
    curve = pyref.GRUMPKIN
    n = 1500
    pts = cref.gen_points(curve.cid, 1021, n); sc = cref.gen_scalars(curve.cid, 1022 + world, n, half=True)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    ecarry, ecarries = cref.lhs_msm(curve.cid, sc, cref.aff_to_jac(curve.cid, pts), base)
    carry, carries = ctx.debug_lhs_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, base, world)
    assert canon(curve, carry) == canon(curve, ecarry)
    for i in range(carries.shape[0]):
        assert canon(curve, carries[i]) == canon(curve, ecarries[i]), i


def test_lhs_sharded_sim_reports_out_of_range_scalar(ctx):
    curve = pyref.GRUMPKIN
    n = 300
    pts = cref.gen_points(curve.cid, 1031, n); sc = cref.gen_scalars(curve.cid, 1032, n, half=True)
    sc[123] = np.frombuffer((math.isqrt(curve.order) + 2).to_bytes(32, "little"), np.uint8)
    ds, dp = ctx.to_device(sc), ctx.to_device(pts)
    with pytest.raises(api.ScalarOutOfRange) as ei:
        ctx.debug_lhs_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, 16, 4)
    assert ei.value.index == 123


def test_msm_sharded_sim_2p24_x8_walk(ctx):
    """the bench's N = 8 shape on one GPU (16-bit windows pinned by the sharded entry itself, 2 windows per rank)"""
    curve = pyref.BN254_G1
    n = 1 << 24
    q = cref.gen_points(curve.cid, 1041, 1)[0]
    dp = ctx.gen_walk(curve.cid, q, n)
    sc = cref.gen_scalars(curve.cid, 1042, n)
    ds = ctx.to_device(sc)
    exp = canon(curve, cref.scalar_mul(curve.cid, cref.walk_dot(curve.cid, sc), q))
    assert canon(curve, ctx.debug_msm_sharded_sim(curve.cid, ds.ptr, dp.ptr, n, 8)) == exp
    ds.free(); dp.free()


# ------------------------------------------------------------------ resident bases
@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_with_resident_bases(ctx, curve):
    """lemsm_bases_upload + lemsm_msm_with_bases: bases uploaded once, scalars staged slab by slab; a prefix of the bases
    may be used (n <= resident count), more scalars than bases is the reference's length assert (:88)"""
    n = 20000
    pts = cref.gen_points(curve.cid, 1051, n)
    bases = ctx.bases_upload(curve.cid, pts)
    ctx.set_option("host_slab_bits", 12)
    try:
        for m, seed in ((n, 1052), (4097, 1053), (1, 1054)):
            sc = cref.gen_scalars(curve.cid, seed, m)
            assert canon(curve, ctx.msm_with_bases(bases, sc)) == canon(curve, cref.best_multiexp(curve.cid, sc, pts[:m], 8)), m
    finally:
        ctx.set_option("host_slab_bits", 0)
    assert canon(curve, ctx.msm_with_bases(bases, np.zeros((0, 32), np.uint8))) == bytes(64)
    with pytest.raises(api.LengthMismatch):
        ctx.msm_with_bases(bases, cref.gen_scalars(curve.cid, 1055, n + 1))
    bases.free()


# ------------------------------------------------------------------ prepare_scalar_witness / table_entry_by_id
def _witness_cases(seed, count, bits):
    rng = pyref.SplitMix64(seed)
    vals = [0, 1, 2, 3, 4, 5, 15, 16, 17, 255, 256, (1 << 64) - 1, 1 << 64]
    vals += [rng.next256() >> (256 - bits) for _ in range(count)]
    vals += [-(rng.next256() >> (256 - bits)) for _ in range(count // 2)]
    return vals


def _witness_batch(ctx, vals, base, nd, lt):
    sc = np.frombuffer(b"".join(abs(v).to_bytes(32, "little") for v in vals), np.uint8).reshape(-1, 32)
    neg = np.array([1 if v < 0 else 0 for v in vals], np.uint8)
    return ctx.prepare_scalar_witness_batch(sc, neg, base, nd, lt)


@pytest.mark.parametrize("base,nd,lt,bits", [(5, 56, 7, 120), (16, 33, 5, 120), (16, 33, 6, 100), (3, 82, 9, 126), (17, 33, 1, 126), (255, 17, 4, 126)])
def test_prepare_scalar_witness_matches_reference_semantics(ctx, base, nd, lt, bits):
    """entry by entry against the verbatim restatement of src/negbase_utils.rs:79-124 (limb index i % logtable + 1 as the
    reference computes it), for scalars the reference does not panic on; signed inputs included"""
    vals = []
    for v in _witness_cases(1100 + base + lt, 150, bits):
        try:
            pyref.prepare_scalar_witness(v, base, nd, lt)
            vals.append(v)
        except pyref.RefPanic:
            pass
    assert len(vals) > 60
    arr = _witness_batch(ctx, vals, base, nd, lt)
    cols = (nd + lt - 1) // lt + 1
    assert arr.shape == (len(vals), base, cols)
    for j, v in enumerate(vals):
        exp = pyref.prepare_scalar_witness(v, base, nd, lt)
        for r in range(base):
            for c in range(cols):
                e = arr[j, r, c]
                val = (int(e["hi"]) << 64) | int(e["lo"])
                kind = api.ENTRY_KINDS[int(e["kind"])]
                want = exp[r][c]
                assert kind == want[0], (v, r, c)
                if kind == "Scalar":
                    assert val == (v if abs(v) < (1 << 127) else val)
                elif kind == "Bucket":
                    assert val == want[1], (v, r, c)
                else:
                    assert (val, int(e["mask"])) == (want[1], want[2]), (v, r, c)
    # the reference-named single-scalar entry returns the same nested list
    assert api.prepare_scalar_witness(vals[20], base, nd, lt, ctx=ctx) == pyref.prepare_scalar_witness(vals[20], base, nd, lt)


@pytest.mark.parametrize("base,nd,lt", [(5, 56, 15), (16, 33, 15), (16, 33, 5), (16, 10, 5), (255, 17, 17), (3, 82, 40)])
def test_prepare_scalar_witness_reports_the_reference_panics(ctx, base, nd, lt):
    """where the reference panics (assert :81, index out of bounds :98-101, i128 / u32 overflow) the status names the
    kind and the FIRST offending scalar, exactly as the restatement finds them"""
    vals = _witness_cases(1200 + base + lt, 120, 127)
    kinds = []
    for v in vals:
        try:
            pyref.prepare_scalar_witness(v, base, nd, lt); kinds.append(None)
        except pyref.RefPanic as e:
            kinds.append(e.kind)
    first = next((j for j, k in enumerate(kinds) if k), None)
    assert first is not None, "case list must contain a panicking scalar"
    exc = {"too_many_digits": api.TooManyDigits, "index": api.RefIndexOutOfBounds, "overflow": api.RefArithmeticOverflow}[kinds[first]]
    with pytest.raises(exc) as ei:
        _witness_batch(ctx, vals, base, nd, lt)
    assert ei.value.index == first
    # every scalar on its own: same verdict as the restatement
    for j, v in enumerate(vals[:60]):
        if kinds[j] is None:
            _witness_batch(ctx, [v], base, nd, lt)
        else:
            with pytest.raises({"too_many_digits": api.TooManyDigits, "index": api.RefIndexOutOfBounds, "overflow": api.RefArithmeticOverflow}[kinds[j]]):
                _witness_batch(ctx, [v], base, nd, lt)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("base", [3, 5, 16, 255])
def test_table_entries_match_reference_semantics(ctx, curve, base):
    """table_entry_by_id as the reference computes it (src/negbase_utils.rs:58-77: the extra factor -base included), in the
    base field of the curve, raw Montgomery limbs out"""
    p = curve.fp
    got = ctx.table_entries(curve.cid, base, 0, 1 << 12)
    rinv = pow(1 << 256, -1, p)
    for idx in list(range(70)) + [255, 256, 1023, 4095]:
        assert int_of(got[idx]) * rinv % p == pyref.table_entry_by_id(base, idx, p), idx
    hi = ctx.table_entries(curve.cid, base, (1 << 40) + 5, 3)
    for t in range(3):
        assert int_of(hi[t]) * rinv % p == pyref.table_entry_by_id(base, (1 << 40) + 5 + t, p)
    assert np.array_equal(api.table_entry_by_id(base, 37, curve.cid, ctx=ctx), got[37])


# ------------------------------------------------------------------ divisor witness (second half)
from oracle import divisor as dv   # noqa: E402


def _fr_fft():
    g = pyref.GRUMPKIN
    head = int.from_bytes(bytes.fromhex(load_json("fr_mont_chains.json")["omega_pow"]["head"]), "little")
    return dv.FrFft(g.fp, head * pow(1 << 256, -1, g.fp) % g.fp)       # omega_pow(0) of src/precomputed_fft_data.rs


def _from_mont(arr, p):
    rinv = pow(1 << 256, -1, p)
    a = np.ascontiguousarray(arr, np.uint64).reshape(-1, 4)
    return [int.from_bytes(a[i].tobytes(), "little") * rinv % p for i in range(a.shape[0])]


def _to_mont_rows(vals, p):
    return np.frombuffer(b"".join(((v << 256) % p).to_bytes(32, "little") for v in vals), np.uint64).reshape(-1, 4).copy()


def _aff_rows(curve, pts):
    return np.frombuffer(b"".join(curve.affine_to_raw(q) for q in pts), np.uint64).reshape(-1, 8).copy()


def _sqrt_mod(a, p):
    """Tonelli-Shanks (p = r has 2-adicity 28)"""
    if a % p == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2; s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p; i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c, t, r = i, b * b % p, t * b * b % p, r * b % p
    return r


@pytest.mark.parametrize("logn", [10, 11, 14, 18, 19, 21])
def test_gpu_ntt_tiled_equals_stagewise(ctx, logn):
    """the LDS-tiled passes (up to 10 stages per launch, strided tiles above 2^10) give bit for bit what one launch per stage
    gives, forward and inverse, at every pass structure (1, 2 and 3 passes, short last groups)"""
    p = pyref.GRUMPKIN.fp
    rng = np.random.Generator(np.random.PCG64(1350 + logn))
    nseq = 2 if logn <= 19 else 1
    data = rng.integers(0, 1 << 62, size=(nseq << logn, 4), dtype=np.uint64)     # < 2^254: valid (if unusual) Montgomery limbs
    data[:, 3] &= np.uint64((1 << 60) - 1)
    outs = {}
    for mode in (0, 2):
        ctx.set_option("ntt_tiled", mode)
        try:
            f = ctx.debug_ntt(data, logn, False)
            outs[mode] = (f, ctx.debug_ntt(f, logn, True))
        finally:
            ctx.set_option("ntt_tiled", 0)
    assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[2][1])
    assert np.array_equal(outs[0][1], data)


@pytest.mark.parametrize("logn", [1, 2, 5, 9, 12])
def test_gpu_ntt_is_the_reference_fft(ctx, logn):
    """the device transform over bn256::Fr equals best_fft with the reference's own twiddle omega_pow(S - logn)
    (src/regular_functions_utils.rs:111-124, constants of src/precomputed_fft_data.rs); inverse(forward(x)) == x"""
    g = pyref.GRUMPKIN; p = g.fp
    fft = _fr_fft()
    rng = pyref.SplitMix64(1300 + logn)
    nseq = 3
    vals = [rng.next256() % p for _ in range(nseq << logn)]
    got = ctx.debug_ntt(_to_mont_rows(vals, p), logn, False)
    gi = _from_mont(got, p)
    for s in range(nseq):
        seq = vals[s << logn: (s + 1) << logn]
        dv.best_fft(seq, fft.omega[fft.S - logn], logn, p)
        assert gi[s << logn: (s + 1) << logn] == seq, s
    back = ctx.debug_ntt(got, logn, True)
    assert _from_mont(back, p) == vals


def _check_witness(ctx, O, jac_pts, aff_rows):
    """GPU witness of the points vs the restatement: lengths exactly, coefficients after normalisation"""
    p = O.p
    a, b, outp = ctx.divisor_witness(api.GRUMPKIN, aff_rows, True, True)
    w = O.normalise(O.compute_divisor_witness(jac_pts))
    assert (a.shape[0], b.shape[0]) == (len(w[0]), len(w[1]))
    assert _from_mont(a, p) == w[0]
    assert _from_mont(b, p) == w[1]
    assert not outp.any()
    return w


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 9, 31, 64, 200, 1024])
def test_divisor_witness_matches_reference_restatement(ctx, n):
    """compute_divisor_witness (src/regular_functions_utils.rs:476-480) of n random points and minus their sum"""
    g = pyref.GRUMPKIN
    O = dv.DivisorOracle(g, _fr_fft())
    rng = pyref.SplitMix64(1400 + n)
    pts = pyref.gen_points(g, rng, n)
    s = None
    for q in pts:
        s = g.add(s, q)
    pts = pts + [g.neg(s)]
    w = _check_witness(ctx, O, [O.from_affine(q, 1 + rng.next256() % (g.fp - 1)) for q in pts], _aff_rows(g, pts))
    for q in pts[:20]:
        assert O.rf_ev(w, O.from_affine(q)) == 0                       # the reference's own assertion, randpoints_witness_test :661


@pytest.mark.parametrize("n", [3, 4, 15, 16, 17, 255, 256, 1000, 4095, 4096])
def test_divisor_witness_half_size_transforms_equal_full_size(ctx, n):
    """option dw_wrap: a level whose longest part has 2^k + 1 coefficients runs on 2^k-point transforms (top coefficient
    unfolded from the value at x = 0) -- the same witness, coefficient for coefficient, as with the next power of two"""
    g = pyref.GRUMPKIN
    q = cref.gen_points(g.cid, 1700 + n, 1)[0]
    rows = ctx.gen_walk(g.cid, q, n).download(np.uint64).reshape(-1, 8)
    res = []
    try:
        for mode in (0, 2):
            ctx.set_option("dw_wrap", mode)
            a, b, outp = ctx.divisor_witness(api.GRUMPKIN, rows, False, True)
            res.append((a.copy(), b.copy(), outp.copy()))
    finally:
        ctx.set_option("dw_wrap", 0)
    assert res[0][0].shape == res[1][0].shape and res[0][1].shape == res[1][1].shape
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all()


@pytest.mark.parametrize("n", [3, 8, 9, 31, 64, 100, 1000, 2049, 6001, 20000])
def test_divisor_witness_child_evaluation_reuse_equals_whole_transforms(ctx, n):
    """option dw_reuse: from the second merge level on a node transforms its children onto the ODD half of its domain only
    and reads the even half from the evaluations the level below left (a passed-through child -- the ragged right edge of
    a tree -- through a side transform): the same witness, coefficient for coefficient, as with whole transforms; lists with
    an identity and a repeated point; the counter says the reuse path really ran"""
    g = pyref.GRUMPKIN
    q = cref.gen_points(g.cid, 1750 + n, 1)[0]
    rows = ctx.gen_walk(g.cid, q, n).download(np.uint64).reshape(-1, 8).copy()
    if n >= 9:
        rows[2] = 0; rows[n - 3] = rows[1]
    res = []
    try:
        for mode in (0, 2):
            ctx.set_option("dw_reuse", mode)
            a, b, outp = ctx.divisor_witness(api.GRUMPKIN, rows, False, True)
            res.append((a.copy(), b.copy(), outp.copy(), ctx.divisor_last_reuse_levels()))
    finally:
        ctx.set_option("dw_reuse", 0)
    assert res[0][0].shape == res[1][0].shape and res[0][1].shape == res[1][1].shape
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all()
    assert res[1][3] == 0
    if n >= 64:
        assert res[0][3] >= 2, res[0][3]


@pytest.mark.parametrize("n", [2, 7, 64, 1000, 3000, 9001])
def test_divisor_witness_lazy_field_numerators_equal_strict_field(ctx, n):
    """option dw_pw_lazy: the pointwise numerators / denominators in the lazy 29-bit field give the witness of the
    strict-field formulas (the default), coefficient for coefficient; with an identity and a repeated point in the list (product mode, doubling)"""
    g = pyref.GRUMPKIN
    q = cref.gen_points(g.cid, 1850 + n, 1)[0]
    rows = ctx.gen_walk(g.cid, q, n).download(np.uint64).reshape(-1, 8).copy()
    if n >= 7:
        rows[3] = 0; rows[n - 2] = rows[0]
    res = []
    try:
        for mode in (0, 1):
            ctx.set_option("dw_pw_lazy", mode)
            a, b, outp = ctx.divisor_witness(api.GRUMPKIN, rows, False, True)
            res.append((a.copy(), b.copy(), outp.copy()))
    finally:
        ctx.set_option("dw_pw_lazy", 0)
    assert res[0][0].shape == res[1][0].shape and res[0][1].shape == res[1][1].shape
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all()


@pytest.mark.parametrize("knob", ["dw_ntt_lazy", "dw_halves"])
@pytest.mark.parametrize("n", [2, 7, 64, 1000, 3000, 9001, 40000])
def test_divisor_witness_ab_knobs_equal_default(ctx, knob, n):
    """options dw_ntt_lazy (transform butterflies in the lazy 29-bit field: sums reduced mid-pass, canonical on store) and
    dw_halves (a level's pointwise chain as two halves of its nodes on two queues) give the default path's witness,
    coefficient for coefficient; lists long enough for strided transform passes (2^11 and above) and, as a batch of two
    trees, for levels with several nodes at every size"""
    g = pyref.GRUMPKIN
    q = cref.gen_points(g.cid, 1950 + n, 1)[0]
    rows = ctx.gen_walk(g.cid, q, n).download(np.uint64).reshape(-1, 8).copy()
    if n >= 7:
        rows[3] = 0; rows[n - 2] = rows[0]
    lists = [rows, rows[: max(1, n // 3)]]
    res = []
    try:
        for mode in (0, 1):
            ctx.set_option(knob, mode)
            res.append(ctx.divisor_witness_batch(api.GRUMPKIN, lists, False, True))
    finally:
        ctx.set_option(knob, 0)
    for t in range(2):
        for part in range(3):
            assert np.array_equal(np.asarray(res[0][t][part]), np.asarray(res[1][t][part])), (t, part)


def test_divisor_witness_batch_reuse_with_ragged_trees(ctx):
    """a forest of trees of different shapes (every kind of ragged right edge at once) with and without reuse"""
    g = pyref.GRUMPKIN
    counts = [1, 2, 3, 5, 6, 7, 12, 13, 33, 64, 65, 127, 200, 513]
    q = cref.gen_points(g.cid, 1790, 1)[0]
    rows = ctx.gen_walk(g.cid, q, sum(counts)).download(np.uint64).reshape(-1, 8).copy()
    res = []
    try:
        for mode in (0, 2):
            ctx.set_option("dw_reuse", mode)
            lists = []; o = 0
            for c in counts:
                lists.append(rows[o:o + c]); o += c
            res.append(ctx.divisor_witness_batch(api.GRUMPKIN, lists, False, True))
    finally:
        ctx.set_option("dw_reuse", 0)
    assert len(res[0]) == len(res[1]) == len(counts)
    for t in range(len(counts)):
        for part in range(3):
            assert np.array_equal(np.asarray(res[0][t][part]), np.asarray(res[1][t][part])), (t, part)


@pytest.mark.parametrize("n", [2, 5, 16, 17, 300, 1024, 5000])
def test_divisor_witness_fused_load_store_equals_separate_kernels(ctx, n):
    """option dw_fuse: the first forward pass gathers from the coefficient arrays and the last inverse pass scatters into
    them -- the same witness, coefficient for coefficient, as with k_load / k_store; lists with identities and repeats too"""
    g = pyref.GRUMPKIN
    q = cref.gen_points(g.cid, 1800 + n, 1)[0]
    rows = ctx.gen_walk(g.cid, q, n).download(np.uint64).reshape(-1, 8).copy()
    if n >= 5:
        rows[1] = 0; rows[n - 2] = rows[0]                   # an identity and a repeated point: lengths off the regular pattern
    res = []
    try:
        for mode in (0, 2):
            ctx.set_option("dw_fuse", mode)
            a, b, outp = ctx.divisor_witness(api.GRUMPKIN, rows, False, True)
            res.append((a.copy(), b.copy(), outp.copy()))
    finally:
        ctx.set_option("dw_fuse", 0)
    assert res[0][0].shape == res[1][0].shape and res[0][1].shape == res[1][1].shape
    assert (res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all() and (res[0][2] == res[1][2]).all()


def test_divisor_witness_10000_points(ctx):
    """the size of randpoints_witness_test (:650-662): 10 000 points (a walk k Q, so that the oracle needs no 10 000 scalar
    multiplications) and minus their sum; full coefficient comparison and the vanishing assertion on a sample"""
    g = pyref.GRUMPKIN
    O = dv.DivisorOracle(g, _fr_fft())
    n = 10000
    q = cref.gen_points(g.cid, 1500, 1)[0]
    rows = ctx.gen_walk(g.cid, q, n).download(np.uint64).reshape(-1, 8)
    pts = [g.raw_to_affine(rows[i].tobytes()) for i in range(n)]
    s = None
    for t in pts:
        s = g.add(s, t)
    pts.append(g.neg(s))
    w = _check_witness(ctx, O, [O.from_affine(t) for t in pts], _aff_rows(g, pts))
    for t in pts[:10] + pts[-5:]:
        assert O.rf_ev(w, O.from_affine(t)) == 0


def test_divisor_witness_reference_test_shapes(ctx):
    """randpoints_witness_test's actual shape (repeat(..).take(n): ONE point n times, :654) and witness_with_zeros_test's
    list (:668: identities, a point, its negative, repeats)"""
    g = pyref.GRUMPKIN
    O = dv.DivisorOracle(g, _fr_fft())
    a = pyref.gen_points(g, pyref.SplitMix64(1600), 1)[0]
    n = 1000
    pts = [a] * n + [g.neg(g.mul(n, a))]
    _check_witness(ctx, O, [O.from_affine(t, 5) for t in pts], _aff_rows(g, pts))
    na = g.neg(a)
    lst = [None, None, None, a, a, na, None, na, a, na]
    w = _check_witness(ctx, O, [O.from_affine(t) for t in lst], _aff_rows(g, lst))
    assert [len(w[0]), len(w[1])] == [7, 5]
    # lone points, pairs with an identity on either side, P and -P adjacent (output identity -> plain products)
    for lst in ([a, na], [None, a, na], [a, None, None, na], [a, a, g.neg(g.mul(2, a))], [None], [], [a, na, a, na, a, na, None]):
        _check_witness(ctx, O, [O.from_affine(t) for t in lst], _aff_rows(g, lst) if lst else np.zeros((0, 8), np.uint64))


def test_divisor_witness_panics_of_the_reference_are_statuses(ctx):
    g = pyref.GRUMPKIN
    O = dv.DivisorOracle(g, _fr_fft())
    a, b = pyref.gen_points(g, pyref.SplitMix64(1700), 2)
    with pytest.raises(api.SumNotIdentity):                              # :478
        ctx.divisor_witness(api.GRUMPKIN, _aff_rows(g, [a, b]), True, True)
    (wa, wb), outp = api.compute_divisor_witness_partial(_aff_rows(g, [a, b]), "grumpkin", ctx)
    assert g.raw_to_affine(outp.tobytes()) == g.neg(g.add(a, b))       # _partial returns the output point instead
    w = O.normalise(O.compute_divisor_witness_partial([O.from_affine(a), O.from_affine(b)])[0])
    assert (_from_mont(wa, g.fp), _from_mont(wb, g.fp)) == w
    lst = [None, None, None, None, a, g.neg(a)]                          # empty() x empty(): usize underflow at :55
    with pytest.raises(dv.RefPanic):
        O.compute_divisor_witness([O.from_affine(t) for t in lst])
    with pytest.raises(api.RefArithmeticOverflow):
        ctx.divisor_witness(api.GRUMPKIN, _aff_rows(g, lst), True, True)
    with pytest.raises(api.LemsmError):                                  # no FftPrecomp for BN254 G1's base field
        ctx.divisor_witness(api.BN254_G1, np.zeros((2, 8), np.uint64), True, True)


def test_divisor_witness_coset_retry_when_an_output_sits_on_the_domain(ctx):
    """the evaluation domain is the coset 7 omega^i; a node output whose x equals a domain point makes a denominator zero:
    detected, the level is redone on the next coset, the result is unchanged.  (With g = 1 the Grumpkin generator, x = 1,
    would already do it.)"""
    g = pyref.GRUMPKIN; p = g.fp
    O = dv.DivisorOracle(g, _fr_fft())
    fft = _fr_fft()
    pt = None
    for j in range(4):                       # level 1 runs transforms of size 4 on {7 w4^j}
        x = 7 * pow(fft.omega[fft.S - 2], j, p) % p
        y = _sqrt_mod((x * x * x + g.b) % p, p)
        if y:
            pt = (x, y); break
    assert pt is not None and g.is_on_curve(pt)
    other = pyref.gen_points(g, pyref.SplitMix64(1800), 3)
    # leaves: (O, pt) -> output -pt with x on the domain; (other0, other1); then the rest so that everything sums to zero
    lst = [None, pt, other[0], other[1], other[2]]
    s = None
    for t in lst:
        s = g.add(s, t)
    lst.append(g.neg(s))
    _check_witness(ctx, O, [O.from_affine(t) for t in lst], _aff_rows(g, lst))
    _check_witness(ctx, O, [O.from_affine(t) for t in lst], _aff_rows(g, lst))   # and again on the moved coset
    gen = g.gen                               # the generator itself (x = 1)
    lst = [gen, other[0], g.neg(g.add(gen, other[0]))]
    _check_witness(ctx, O, [O.from_affine(t) for t in lst], _aff_rows(g, lst))


@pytest.mark.parametrize("base,n", [(5, 7), (16, 12), (3, 5), (5, 40)])
def test_compute_lhs_witness_full_return_value(ctx, base, n):
    """compute_lhs_witness (src/argument_witness_calc.rs:87-136) in full: the carry and all d RegularFunctions (reversed order,
    :132) against the restatement; arbitrary-Z Jacobian inputs as the reference's callers pass them"""
    g = pyref.GRUMPKIN; p = g.fp
    O = dv.DivisorOracle(g, _fr_fft())
    rng = pyref.SplitMix64(1900 + base + n)
    sc = pyref.gen_scalars_half(rng, n, g.order)
    sc[0] = 0
    pts = pyref.gen_points(g, rng, n)
    jac = [O.from_affine(q, 1 + rng.next256() % (p - 1)) for q in pts]
    ecarry, efns = dv.compute_lhs_witness(O, sc, jac, base)
    scb = np.frombuffer(pyref.scalars_to_bytes(sc), np.uint8).reshape(-1, 32)
    jrows = np.frombuffer(b"".join(g.affine_to_jacobian_raw(O.to_affine(j), j[2]) for j in jac), np.uint64).reshape(-1, 12)
    carry, fns = api.compute_lhs_witness(scb, jrows, base, "grumpkin", ctx)
    assert canon(g, carry) == g.canonical(O.to_affine(ecarry))
    assert len(fns) == len(efns) == api.num_digits(g.cid, base)
    for f, (got, exp) in enumerate(zip(fns, efns)):
        e = O.normalise(exp)
        assert (_from_mont(got[0], p), _from_mont(got[1], p)) == e, f


@pytest.mark.parametrize("base,n", [(5, 37), (16, 300)])
def test_lhs_witness_device_entry_equals_host_entry(ctx, base, n):
    """lemsm_lhs_witness_device (scalars and affine points in HBM, coefficients left in HBM) returns what lemsm_lhs_witness
    does for the same points handed over as host Jacobian rows: carry, index and every coefficient"""
    g = pyref.GRUMPKIN
    rng = pyref.SplitMix64(2600 + base + n)
    sc = pyref.gen_scalars_half(rng, n, g.order)
    scb = np.frombuffer(pyref.scalars_to_bytes(sc), np.uint8).reshape(-1, 32)
    q = cref.gen_points(g.cid, 2601, 1)[0]
    dp = ctx.gen_walk(g.cid, q, n)
    aff = dp.download(np.uint64).reshape(-1, 8)
    aff[3] = 0                                              # an identity among the points
    dp.upload(aff)
    jac = cref.aff_to_jac(g.cid, aff)
    carry_h, fns = ctx.lhs_witness(g.cid, scb, jac, base, True)
    ds = ctx.to_device(scb)
    carry_d, index, out = ctx.lhs_witness_device(g.cid, ds.ptr, dp.ptr, n, base, True)
    assert canon(g, carry_d) == canon(g, carry_h)
    flat = out.download(np.uint64).reshape(-1, 4)
    assert index.shape[0] == len(fns)
    for f, (a, b) in enumerate(fns):
        oa, la, ob, lb = (int(v) for v in index[f])
        assert (la, lb) == (a.shape[0], b.shape[0]), f
        assert (flat[oa: oa + la] == a).all() and (flat[ob: ob + lb] == b).all(), f


def test_lhs_witness_function_ranges_are_the_full_call_in_pieces(ctx):
    """lemsm_lhs_witness_device_range: the d merge trees are independent -- three ranks' shares (dist.window_range), an empty
    share among them, put together are the full call: same carry from every share, the same coefficients function by function"""
    from halo2_liam_eagen_msm_amd import dist as ldist
    g = pyref.GRUMPKIN
    n, base = 200, 5
    rng = pyref.SplitMix64(2700)
    sc = pyref.gen_scalars_half(rng, n, g.order)
    scb = np.frombuffer(pyref.scalars_to_bytes(sc), np.uint8).reshape(-1, 32)
    q = cref.gen_points(g.cid, 2701, 1)[0]
    dp = ctx.gen_walk(g.cid, q, n)
    ds = ctx.to_device(scb)
    carry, index, out = ctx.lhs_witness_device(g.cid, ds.ptr, dp.ptr, n, base, True)
    full = out.download(np.uint64).reshape(-1, 4)
    d = index.shape[0]
    seen = 0
    for world, rank in ((3, 0), (3, 1), (3, 2), (d + 5, 1), (d + 5, d + 4)):
        f0, f1 = ldist.window_range(d, world, rank)
        c2, ix2, out2 = ctx.lhs_witness_device(g.cid, ds.ptr, dp.ptr, n, base, True, None, (f0, f1))
        assert canon(g, c2) == canon(g, carry)
        part = out2.download(np.uint64).reshape(-1, 4)
        for f in range(d):
            oa, la, ob, lb = (int(v) for v in index[f]); pa, qa, pb, qb = (int(v) for v in ix2[f])
            if f0 <= f < f1:
                assert (qa, qb) == (la, lb), f
                assert (part[pa: pa + qa] == full[oa: oa + la]).all() and (part[pb: pb + qb] == full[ob: ob + lb]).all(), f
                seen += world == 3
            else:
                assert (qa, qb) == (0, 0), f
    assert seen == d


def test_divisor_witness_2p20_vanishes_on_its_points(ctx):
    """full size (the point count of configs[1]'s per-digit lists): 2^20 walk points k Q and minus their sum
    (n (n + 1) / 2 Q); the size-independent property the reference asserts (randpoints_witness_test :661): the
    witness vanishes on every point -- checked on a sample by Horner evaluation in Python integers"""
    g = pyref.GRUMPKIN; p = g.fp
    O = dv.DivisorOracle(g, None)
    n = 1 << 20
    q = cref.gen_points(g.cid, 2000, 1)[0]
    dp = ctx.gen_walk(g.cid, q, n)
    rows = np.zeros((n + 1, 8), np.uint64)
    rows[:n] = dp.download(np.uint64).reshape(-1, 8)
    qa = g.raw_to_affine(q.tobytes())
    last = g.neg(g.mul(n * (n + 1) // 2 % g.order, qa))
    rows[n] = np.frombuffer(g.affine_to_raw(last), np.uint64)
    a, b, outp = ctx.divisor_witness(api.GRUMPKIN, rows, True, True)
    assert not outp.any() and a.shape[0] + b.shape[0] >= n
    w = (_from_mont(a, p), _from_mont(b, p))
    for idx in (0, 1, 12345, n - 1, n):
        assert O.rf_ev(w, O.from_affine(g.raw_to_affine(rows[idx].tobytes()))) == 0, idx
    dp.free()


def test_validate_points_option(ctx):
    """option validate_points: an input that is not on the curve (here y + 1, and the (x != 0, y == 0) shape the hot kernel
    would silently take for the identity) is refused with its index; off (the default) the caller is trusted like the
    reference's from_raw_bytes_unchecked"""
    curve = pyref.BN254_G1
    n = 5000
    pts = cref.gen_points(curve.cid, 2101, n); sc = cref.gen_scalars(curve.cid, 2102, n)
    good = canon(curve, cref.best_multiexp(curve.cid, sc, pts, 8))
    bad = pts.copy(); bad[3210, 4] += np.uint64(1)
    bad2 = pts.copy(); bad2[77, 4:] = 0
    ctx.set_option("validate_points", 1)
    try:
        assert canon(curve, ctx.msm(curve.cid, sc, pts)) == good
        for arr, idx in ((bad, 3210), (bad2, 77)):
            with pytest.raises(api.LemsmError) as ei:
                ctx.msm(curve.cid, sc, arr)
            assert ei.value.status == 6 and ctx.lib.lemsm_last_bad_index(ctx.h) == idx
            dp = ctx.to_device(arr); ds = ctx.to_device(sc)
            with pytest.raises(api.LemsmError):
                ctx.msm_device(curve.cid, ds.ptr, dp.ptr, n)
            with pytest.raises(api.LemsmError):
                ctx.lhs_msm_device(curve.cid, ds.ptr, dp.ptr, n, 16)
    finally:
        ctx.set_option("validate_points", 0)
    ctx.msm(curve.cid, sc, bad2)      # unchecked: no error (the point is skipped)


def test_divisor_witness_batch_is_a_forest(ctx):
    """lemsm_divisor_witness_batch: lists of very different lengths (0, 1, 2, 5, 64, 333 points; one of them not summing
    to zero, asked for as _partial) advance together level by level and each equals its own single-list witness"""
    g = pyref.GRUMPKIN; p = g.fp
    O = dv.DivisorOracle(g, _fr_fft())
    rng = pyref.SplitMix64(2200)
    lists = []
    for n in (0, 1, 2, 5, 64, 333, 7):
        pts = pyref.gen_points(g, rng, n)
        lists.append(pts)
    res = ctx.divisor_witness_batch(api.GRUMPKIN, [_aff_rows(g, l) if l else np.zeros((0, 8), np.uint64) for l in lists], False, True)
    assert len(res) == len(lists)
    for l, (a, b, outp) in zip(lists, res):
        w, out = O.compute_divisor_witness_partial([O.from_affine(q) for q in l])
        w = O.normalise(w)
        assert (_from_mont(a, p), _from_mont(b, p)) == w, len(l)
        assert g.raw_to_affine(outp.tobytes()) == O.to_affine(out), len(l)
    with pytest.raises(api.SumNotIdentity):
        ctx.divisor_witness_batch(api.GRUMPKIN, [_aff_rows(g, lists[3])], True, True)
