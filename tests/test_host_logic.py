"""Host-side logic of the product library, runnable without a GPU: digit counts, plans, the
Horner tails (lemsm_msm_combine / lemsm_lhs_combine) fed with oracle-built window records, and
canonicalisation.  No device entry point is called."""
import ctypes
import os
import sys

import numpy as np
import pytest

import hostref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from helpers import CURVES, canon
from halo2_liam_eagen_msm_amd import _lib, api
from oracle import cref, pyref


def test_num_digits_matches_reference_formula():
    for c in CURVES:
        for base in (3, 4, 5, 7, 16, 17, 255):
            assert api.num_digits(c.cid, base) == pyref.num_digits(c.order, base)
    with pytest.raises(api.BadBase):
        api.num_digits(0, 1)


def test_scalar_helpers_mirror_reference():
    assert api.logb_ceil(0, 5) == 0 and api.logb_ceil(24, 5) == 2 and api.logb_ceil(25, 5) == 3
    assert api.id_by_digit(0) is None and api.id_by_digit(7) == 6 and api.digit_by_id(6) == 7
    assert api.order("bn254_g1") == pyref.R_BN254 and api.order("grumpkin") == pyref.P_BN254


def test_msm_plan_windows():
    for c in CURVES:
        for n in (1, 100, 1 << 12, 1 << 20, 1 << 24, 1 << 26):
            W, L, rec = hostref.msm_plan(c, n)
            cbits = L + 1
            assert rec == 128 and 3 <= cbits <= 17
            K = sum((1 << (cbits - 1)) << (cbits * w) for w in range(W - 1))
            assert ((c.order - 1 + K) >> (cbits * (W - 1))) <= (1 << (cbits - 1))   # top window fits the buckets
            assert (c.order - 1 + K) >> (cbits * W) == 0
        assert hostref.msm_plan(c, 1 << 23)[0] == 16    # 16-bit windows: 16 windows, 8/4/2 per rank on 2/4/8 GPUs
        assert hostref.msm_plan(c, 1 << 24)[0] == 15    # 17-bit windows from 2^24 pairs (single-GPU default)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 7, 40, 300])
def test_msm_combine_with_oracle_records(curve, n):
    rng = pyref.SplitMix64(900 + n)
    pts = pyref.gen_points(curve, rng, min(n, 12)) * (n // 12 + 1)
    pts = pts[:n]
    sc = pyref.gen_scalars_full(rng, n, curve.order)
    sc[0] = curve.order - 1
    W, rec, recs = hostref.msm_records(curve, sc, pts)
    out = hostref.msm_combine(curve, n, b"".join(recs))
    assert canon(curve, out) == curve.canonical(curve.msm_naive(sc, pts))
    assert api.jacobian_to_canonical(curve.cid, out) == canon(curve, out)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("base", [3, 5, 16, 255])
def test_lhs_combine_with_oracle_records(curve, base):
    n = 25
    rng = pyref.SplitMix64(950 + base)
    pts = pyref.gen_points(curve, rng, n)
    sc = pyref.gen_scalars_half(rng, n, curve.order)
    d, rec, recs = hostref.lhs_records(curve, sc, pts, base)
    carry, carries = hostref.lhs_combine(curve, base, b"".join(recs))
    ecarry, ecarries = pyref.lhs_msm(curve, sc, pts, base)
    assert canon(curve, carry) == curve.canonical(ecarry)
    for i in range(d):
        assert canon(curve, carries[i]) == curve.canonical(ecarries[i]), i


def test_jacobian_to_canonical_identity_and_scaling():
    c = pyref.GRUMPKIN
    assert api.jacobian_to_canonical(c.cid, np.zeros(12, np.uint64)) == bytes(64)
    pt = c.mul(12345, c.gen)
    for z in (1, 2, 0xABCDEF):
        j = np.frombuffer(c.affine_to_jacobian_raw(pt, z), np.uint64)
        assert api.jacobian_to_canonical(c.cid, j) == c.canonical(pt)


def test_status_strings_and_bad_args():
    lib = _lib.load()
    assert lib.lemsm_strerror(1) == b"incompatible amount of coefficients"
    assert lib.lemsm_strerror(2) == b"scalar out of range"
    d = ctypes.c_uint32()
    assert lib.lemsm_num_digits(7, 5, ctypes.byref(d)) == _lib.LEMSM_ERR_BAD_CURVE
    assert lib.lemsm_lhs_plan(0, 2, None, None) == _lib.LEMSM_ERR_BAD_BASE


def test_product_fails_loudly_without_gpu():
    """no CPU fallback: without a usable gfx950 device context creation raises"""
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.lemsm_create(0, ctypes.byref(h))
    if rc == 0:          # running on a GPU box
        lib.lemsm_destroy(h)
    else:
        assert rc == _lib.LEMSM_ERR_HIP
        with pytest.raises(api.LemsmError):
            api.Context(0)


def test_jacobian_sum_host():
    """lemsm_jacobian_sum: host-side fold of partial results (no GPU): identity handling,
    P + (-P), doubling (equal inputs), random sums vs the big-int oracle"""
    import numpy as np
    from halo2_liam_eagen_msm_amd import api
    from oracle import cref, pyref
    for curve in (pyref.BN254_G1, pyref.GRUMPKIN):
        rng = pyref.SplitMix64(991 + curve.cid)
        pts = pyref.gen_points(curve, rng, 6)
        def jac(pt):
            z = 1 + rng.next256() % (curve.fp - 1)
            return np.frombuffer(curve.affine_to_jacobian_raw(pt, z), np.uint64)
        ident = np.zeros(12, np.uint64)
        assert cref.jac_to_canonical(curve.cid, api.jacobian_sum(curve.cid, np.zeros((0, 12), np.uint64))) == bytes(64)
        assert cref.jac_to_canonical(curve.cid, api.jacobian_sum(curve.cid, np.stack([ident, ident]))) == bytes(64)
        neg = (pts[0][0], (-pts[0][1]) % curve.fp)
        assert cref.jac_to_canonical(curve.cid, api.jacobian_sum(curve.cid, np.stack([jac(pts[0]), jac(neg)]))) == bytes(64)
        got = api.jacobian_sum(curve.cid, np.stack([jac(pts[1]), ident, jac(pts[1])]))
        assert cref.jac_to_canonical(curve.cid, got) == curve.canonical(curve.add(pts[1], pts[1]))
        acc = None
        for p_ in pts:
            acc = curve.add(acc, p_)
        got = api.jacobian_sum(curve.cid, np.stack([jac(p_) for p_ in pts]))
        assert cref.jac_to_canonical(curve.cid, got) == curve.canonical(acc)


# ------------------------------------------------------------------ challenge post-processing helpers
def _mont(v, p):
    return np.frombuffer(((v << 256) % p).to_bytes(32, "little"), np.uint64).copy()


def _unmont(limbs, p):
    return int.from_bytes(np.ascontiguousarray(limbs, np.uint64).tobytes(), "little") * pow(1 << 256, -1, p) % p


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_to_curve_x_y_from_x_slope(curve):
    """src/config.rs:166-187 against big-integer arithmetic: to_curve_x returns its input when x^3 + b is a square and reports
    the reference's endless loop otherwise; y_from_x is sqrt_alt (a root of x^3 + b, or of ROOT_OF_UNITY (x^3 + b) for a
    non-residue); slope = 3 x^2 / (2 y) with the y = 0 panic as an error"""
    p = curve.fp
    # ROOT_OF_UNITY = g^t, t the odd part of p - 1, g the multiplicative generator halo2curves uses (3 for Fq, 7 for Fr)
    t = p - 1
    while t % 2 == 0:
        t //= 2
    root = pow(3 if curve.cid == 0 else 7, t, p)
    rng = pyref.SplitMix64(3100 + curve.cid)
    seen = {True: 0, False: 0}
    for _ in range(60):
        x = rng.next256() % p
        rhs = (x * x * x + curve.b) % p
        sq = rhs == 0 or pow(rhs, (p - 1) // 2, p) == 1
        seen[sq] += 1
        y, flag = api.y_from_x(_mont(x, p), curve.cid)
        yv = _unmont(y, p)
        assert flag == sq
        assert yv * yv % p == (rhs if sq else root * rhs % p)
        if sq:
            assert _unmont(api.to_curve_x(_mont(x, p), curve.cid), p) == x
            if yv:
                s = api.slope(np.concatenate([_mont(x, p), y]), curve.cid)
                assert _unmont(s, p) == 3 * x * x * pow(2 * yv, -1, p) % p
        else:
            with pytest.raises(api.WouldNotTerminate):
                api.to_curve_x(_mont(x, p), curve.cid)
    assert seen[True] > 10 and seen[False] > 10
    gx, gy = curve.gen
    y, flag = api.y_from_x(_mont(gx, p), curve.cid)
    assert flag and _unmont(y, p) in (gy, p - gy)
    with pytest.raises(ZeroDivisionError):
        api.slope(np.concatenate([_mont(gx, p), np.zeros(4, np.uint64)]), curve.cid)


# ------------------------------------------------------------------ sanitizer runs (CPU only)
def test_host_tail_under_asan_ubsan(tmp_path):
    """the product's host tail (csrc/hosttail.hpp, hostmath.hpp, hostpool.hpp: record conversion, window / final Horner,
    pyramid task tables, merge queue layout, worker pool) compiled alone with -fsanitize=address,undefined and driven on
    synthetic data, the task tables simulated on integers (tests/host_tail_check.cpp)"""
    import subprocess
    exe = tmp_path / "host_tail_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-pthread", "-o", str(exe), os.path.join(ROOT, "tests", "host_tail_check.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host tail ok" in out.stdout, out.stdout + out.stderr
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr


def test_oracle_c_under_asan_ubsan():
    """the C oracle (test infrastructure) rebuilt with -fsanitize=address,undefined (`make -C oracle asan`) and the golden-vector
    tests of tests/test_oracle_golden.py run against that build in a child interpreter with libasan preloaded"""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1",
               LEMSM_ORACLE_LIB=os.path.join(ROOT, "oracle", "_build", "liblemsm_oracle_asan.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider",
                          "-k", "not asan"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
