"""Pins the oracle (oracle/pyref.py and oracle/c) before anything trusts it.

* the reference's only byte-level golden data: the 192 raw-Montgomery bn256::Fr constants of
  /root/reference/src/precomputed_fft_data.rs, through tests/golden/fr_mont_chains.json (heads +
  SHA-256 of each table) and, when the reference is mounted, through the file itself;
* the public EIP-196 vector 2*(1,2) on BN254 G1;
* SURVEY.md Appendix A micro-KATs for negbase_decompose and the digit-count table;
* every property the reference's own tests assert (negbase_test, lhs_test);
* committed seeded vectors (tests/golden/msm_vectors.json).
The reference holds no MSM known-answer vectors: MSM byte parity is "parity unpinned" by the
reference and is defined as equality of group elements in canonical affine form.
"""
import json
import math
import os
import random

import numpy as np
import pytest

from helpers import CURVES, canon, golden_points_raw, golden_scalars, load_json, regenerate_chain_digests
from oracle import cref, pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REF_FILE = "/root/reference/src/precomputed_fft_data.rs"


def test_fr_montmul_chains_python():
    chains = load_json("fr_mont_chains.json")
    c = pyref.GRUMPKIN   # coordinate field of Grumpkin is Fr
    mm = lambda a, b: c.montmul(int.from_bytes(a, "little"), int.from_bytes(b, "little")).to_bytes(32, "little")
    got = regenerate_chain_digests(mm, chains)
    for name in ("omega_pow", "omega_pow_inv", "half_pow"):
        assert got[name] == chains[name]["sha256"], name


def test_fr_montmul_chains_c_oracle():
    chains = load_json("fr_mont_chains.json")
    def mm(a, b):
        return cref.montmul(1, np.frombuffer(a, np.uint64), np.frombuffer(b, np.uint64)).tobytes()
    got = regenerate_chain_digests(mm, chains)
    for name in ("omega_pow", "omega_pow_inv", "half_pow"):
        assert got[name] == chains[name]["sha256"], name


@pytest.mark.skipif(not os.path.exists(REF_FILE), reason="reference not mounted (GPU box)")
def test_fr_constants_against_reference_file():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import parse_reference_tables
    t = parse_reference_tables(REF_FILE)
    c = pyref.GRUMPKIN
    r = pyref.R_BN254
    w = [int.from_bytes(b, "little") for b in t["omega_pow"]]
    wi = [int.from_bytes(b, "little") for b in t["omega_pow_inv"]]
    h = [int.from_bytes(b, "little") for b in t["half_pow"]]
    one = (1 << 256) % r
    assert all(v < r for v in w + wi + h)
    for i in range(63):
        assert c.montmul(w[i], w[i]) == w[i + 1]
        assert c.montmul(wi[i], wi[i]) == wi[i + 1]
        assert cref.montmul(1, np.frombuffer(t["omega_pow"][i], np.uint64), np.frombuffer(t["omega_pow"][i], np.uint64)).tobytes() == t["omega_pow"][i + 1]
    for i in range(64):
        assert c.montmul(w[i], wi[i]) == one
    assert h[0] == one and (2 * c.from_mont(h[1])) % r == 1
    for i in range(1, 63):
        assert c.montmul(h[i], h[1]) == h[i + 1]
    assert w[28] == one and c.from_mont(w[27]) == r - 1
    assert c.from_mont(w[0]) == 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C


def test_field_constants():
    for cid, mod in ((0, pyref.P_BN254), (1, pyref.R_BN254)):
        m, inv, R, R2 = cref.field_consts(cid)
        assert int.from_bytes(m.tobytes(), "little") == mod
        assert int.from_bytes(R.tobytes(), "little") == (1 << 256) % mod
        assert int.from_bytes(R2.tobytes(), "little") == (1 << 512) % mod
        assert (inv * mod + 1) % (1 << 64) == 0


def test_eip196_doubling_and_generators():
    c = pyref.BN254_G1
    assert c.mul(2, (1, 2)) == (
        1368015179489954701390400359078579693043519447331113978918064868415326638035,
        9918110051302171585080402603319702774565515993150576347155970296011118125764)
    for c in CURVES:
        assert c.is_on_curve(c.gen)
        assert c.mul(c.order, c.gen) is None
        g = cref.generator(c.cid)
        assert c.raw_to_affine(g.tobytes()) == c.gen
        for k in (1, 2, 3, 1234567, c.order - 1, c.order):
            assert cref.jac_to_canonical(c.cid, cref.scalar_mul(c.cid, k, g)) == c.canonical(c.mul(k, c.gen))


NEGBASE_KATS = [
    (0, 5, []), (1, 5, [1]), (4, 5, [4]), (5, 5, [0, 4, 1]), (7, 5, [2, 4, 1]), (24, 5, [4, 1, 1]),
    (25, 5, [0, 0, 1]), (15, 16, [15]), (16, 16, [0, 15, 1]), (255, 16, [15, 1, 1]),
    (2**32 - 1, 17, [0, 8, 11, 4, 16, 2, 8, 7, 1]),
]


def test_negbase_kats():
    for x, b, exp in NEGBASE_KATS:
        assert pyref.negbase_decompose(x, b) == exp
        assert cref.negbase_decompose(x, b) == exp
    assert len(pyref.negbase_decompose(2**126, 16)) == 33


def test_negbase_recomposition_property():
    """the reference's negbase_test (src/negbase_utils.rs:126-134), over many inputs and bases"""
    rnd = random.Random(7)
    for base in (3, 4, 5, 16, 17, 255):
        for _ in range(300):
            x = rnd.getrandbits(rnd.choice([1, 8, 32, 64, 126, 127]))
            d = pyref.negbase_decompose(x, base)
            acc = 0
            for dig in reversed(d):
                acc = acc * (-base) + dig
            assert acc == x
            assert all(0 <= v < base for v in d)
            assert cref.negbase_decompose(x, base) == d


def test_digit_count_table():
    for c in CURVES:
        for base, d in ((3, 82), (4, 65), (5, 56), (7, 47), (16, 33), (17, 33), (255, 17)):
            assert pyref.num_digits(c.order, base) == d
            dd, bound = cref.num_digits(c.cid, base)
            assert dd == d and bound == math.isqrt(c.order) + 2
    # worst cases never exceed d for base >= 3 (for base 2 they can: the reference truncates)
    for base in (3, 5, 16, 255):
        d = pyref.num_digits(pyref.R_BN254, base)
        bound = pyref.scalar_bound(pyref.R_BN254)
        for x in (bound - 1, bound - 2, bound // 2, (bound - 1) // base):
            assert len(pyref.negbase_decompose(x, base)) <= d


def test_committed_msm_vectors_both_oracles():
    vecs = load_json("msm_vectors.json")
    for v in vecs["msm"]:
        c = pyref.CURVES[v["curve"]]
        pts = golden_points_raw(c, v["points"])
        sc = golden_scalars(v["scalars"])
        exp = bytes.fromhex(v["expected"])
        assert canon(c, cref.best_multiexp(c.cid, sc, pts, 1)) == exp
        assert canon(c, cref.best_multiexp(c.cid, sc, pts, 4)) == exp
        assert canon(c, cref.msm_naive(c.cid, sc, pts)) == exp
    for v in vecs["lhs"]:
        c = pyref.CURVES[v["curve"]]
        pts = golden_points_raw(c, v["points"])
        sc = golden_scalars(v["scalars"])
        carry, carries = cref.lhs_msm(c.cid, sc, cref.aff_to_jac(c.cid, pts), v["base"])
        assert canon(c, carry) == bytes.fromhex(v["expected_carry"])
        for i, h in enumerate(v["expected_carries_msb_first"]):
            assert canon(c, carries[i]) == bytes.fromhex(h)
        d = pyref.num_digits(c.order, v["base"])
        digs = cref.negbase_decompose_batch(sc, v["base"], d)
        assert digs.tolist() == v["digits_lsb_first"]


def test_lhs_test_shape_cpu():
    """the reference's lhs_test (src/argument_witness_calc.rs:138-148): one scalar and one point
    replicated (iter::repeat evaluates once), base 5, Grumpkin: lhs == best_multiexp."""
    c = pyref.GRUMPKIN
    n = 300
    pt = cref.gen_points(c.cid, 77, 1)
    s = cref.gen_scalars(c.cid, 78, 1, half=True)
    pts = np.repeat(pt, n, axis=0); sc = np.repeat(s, n, axis=0)
    a = cref.best_multiexp(c.cid, sc, pts, 4)
    b, _ = cref.lhs_msm(c.cid, sc, cref.aff_to_jac(c.cid, pts), 5)
    assert cref.jac_eq(c.cid, a, b)
    k = int.from_bytes(s[0].tobytes(), "little") * n % c.order
    assert canon(c, a) == c.canonical(c.mul(k, c.raw_to_affine(pt[0].tobytes())))


def test_lhs_rejects_out_of_range_and_mismatch():
    c = pyref.GRUMPKIN
    pts = cref.gen_points(c.cid, 5, 3)
    sc = cref.gen_scalars(c.cid, 6, 3, half=True)
    sc[1] = np.frombuffer(int(pyref.scalar_bound(c.order)).to_bytes(32, "little"), np.uint8)
    with pytest.raises(ValueError, match="scalar 1 out of range"):
        cref.lhs_msm(c.cid, sc, cref.aff_to_jac(c.cid, pts), 5)
    with pytest.raises(ValueError):
        cref.lhs_msm(c.cid, sc[:2], cref.aff_to_jac(c.cid, pts), 5)


def test_walk_relation():
    c = pyref.BN254_G1
    q = cref.gen_points(0, 5, 1)[0]
    w = cref.gen_walk(0, q, 300)
    qa = c.raw_to_affine(q.tobytes())
    assert c.raw_to_affine(w[299].tobytes()) == c.mul(300, qa)
    sc = cref.gen_scalars(0, 11, 300)
    dot = cref.walk_dot(0, sc)
    assert canon(c, cref.best_multiexp(0, sc, w, 2)) == c.canonical(c.mul(dot, qa))


# ------------------------------------------------------------------ compiled witness restatement vs the Python one
def _omega0():
    chains = json.load(open(os.path.join(ROOT, "tests", "golden", "fr_mont_chains.json")))
    head = bytes.fromhex(chains["omega_pow"]["head"])
    return np.frombuffer(head, np.uint64).copy(), int.from_bytes(head, "little")


def _jac_ints(pts_jac, p):
    ri = pow(1 << 256, -1, p)
    out = []
    for row in np.ascontiguousarray(pts_jac, np.uint64).reshape(-1, 12):
        b = row.tobytes()
        out.append(tuple(int.from_bytes(b[32 * i:32 * i + 32], "little") * ri % p for i in range(3)))
    return out


@pytest.mark.parametrize("n,threads", [(1, 1), (2, 1), (3, 2), (7, 1), (31, 3), (64, 2), (150, 4), (301, 3)])
def test_c_divisor_witness_equals_python_restatement(n, threads):
    """oracle/c/witness_oracle.inc (the timed CPU baseline of the witness path: schoolbook below 32 coefficients, radix-2 FFT
    with the pinned omega above, threads over the pairs of a level) against oracle/divisor.py, coefficient for coefficient
    after normalisation, lengths exactly -- n >= 64 takes the products through the FFT"""
    from oracle import divisor as dv
    omega_raw, head = _omega0()
    c = pyref.GRUMPKIN
    O = dv.DivisorOracle(c, dv.FrFft(c.fp, head * pow(1 << 256, -1, c.fp) % c.fp))
    pts = cref.aff_to_jac(1, cref.gen_points(1, 4000 + n, n))
    # close the list with minus the sum so that the witness exists (:478)
    acc = np.zeros(12, np.uint64)
    for row in pts:
        acc = cref.jac_add(1, acc, row)
    neg = acc.copy()
    yb = int.from_bytes(neg[4:8].tobytes(), "little")
    if int.from_bytes(neg[8:12].tobytes(), "little"):
        neg[4:8] = np.frombuffer(((c.fp - yb) % c.fp).to_bytes(32, "little"), np.uint64)
    full = np.vstack([pts, neg.reshape(1, 12)])
    st, a, b = cref.divisor_witness(full, omega_raw, threads)
    assert st == 0
    want = O.compute_divisor_witness(_jac_ints(full, c.fp))
    assert (len(a), len(b)) == (len(want[0]), len(want[1]))
    assert O.normalise((a, b)) == O.normalise(want)
    # without the closing point the reference panics (:478)
    st, _, _ = cref.divisor_witness(pts, omega_raw, threads)
    assert st == 1


@pytest.mark.parametrize("n,base,threads", [(5, 5, 1), (40, 16, 4), (130, 3, 3)])
def test_c_lhs_witness_equals_python_restatement(n, base, threads):
    from oracle import divisor as dv
    omega_raw, head = _omega0()
    c = pyref.GRUMPKIN
    O = dv.DivisorOracle(c, dv.FrFft(c.fp, head * pow(1 << 256, -1, c.fp) % c.fp))
    pts = cref.aff_to_jac(1, cref.gen_points(1, 4100 + n, n))
    sc = cref.gen_scalars(1, 4200 + n, n, half=True)
    st, carry, fs = cref.lhs_witness(sc, pts, base, omega_raw, threads)
    assert st == 0
    ecarry, efs = dv.compute_lhs_witness(O, [int.from_bytes(s.tobytes(), "little") for s in sc], _jac_ints(pts, c.fp), base)
    assert cref.jac_to_canonical(1, carry) == c.canonical(O.to_affine(ecarry))
    assert len(fs) == len(efs)
    for f, (got, want) in enumerate(zip(fs, efs)):
        assert (len(got[0]), len(got[1])) == (len(want[0]), len(want[1])), f
        assert O.normalise(got) == O.normalise(want), f
