"""Independent big-integer oracle (pure Python ints) for the MSM witness path.

TEST INFRASTRUCTURE ONLY.  Nothing under `halo2_liam_eagen_msm_amd/` may import
this file; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg use it, and only as the checker.

What it restates (paths relative to /root/reference):
  * negbase_decompose            src/negbase_utils.rs:20-36
  * id_by_digit / digit_by_id    src/negbase_utils.rs:46-56
  * logb_ceil                    src/argument_witness_calc.rs:32-40
  * precompute_multiplicities    src/argument_witness_calc.rs:43-51
  * order / digit count d        src/argument_witness_calc.rs:54-56, 89-91
  * compute_lhs_witness MSM core src/argument_witness_calc.rs:87-127,132-134
    (the Horner-in-(-B) carry recursion; the divisor-witness call at :129 is
    out of scope)
  * best_multiexp                third-party halo2_proofs (source absent from
    /root/reference, unpinned git dependency, Cargo.toml:10): its contract
    `sum_i coeffs[i] * bases[i]` is what is restated here (naive and windowed),
    since the reference compares group elements only
    (src/argument_witness_calc.rs:144-147).

Pinning status: the reference holds NO MSM golden vectors, so MSM
byte-level parity is "parity unpinned" by reference fixtures.  What IS pinned:
the r-modulus Montgomery multiplier against the 192 raw-Montgomery constants of
src/precomputed_fft_data.rs (tests/test_oracle_golden.py), the BN254 G1 group
law against the public EIP-196 doubling vector 2*(1,2), and every property the
reference's own tests assert (negbase_test recomposition, lhs_test equality).

Arithmetic here is affine short-Weierstrass with modular inverses via pow(),
deliberately unlike the Montgomery/Jacobian code in oracle/c and in the HIP
kernels, so that an error in one is not mirrored in the other.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

# BN254 base-field modulus p and scalar-field modulus r (SURVEY.md §8c constants).
P_BN254 = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R_BN254 = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

MONT_R = 1 << 256

Point = Optional[Tuple[int, int]]  # None = identity


@dataclass(frozen=True)
class Curve:
    name: str
    cid: int            # id shared with the C oracle and the C ABI
    fp: int             # coordinate field modulus
    order: int          # group order == scalar field modulus
    b: int              # y^2 = x^3 + b
    gen: Tuple[int, int]

    def is_on_curve(self, pt: Point) -> bool:
        if pt is None:
            return True
        x, y = pt
        return (y * y - (x * x * x + self.b)) % self.fp == 0

    # ---- affine group law --------------------------------------------------
    def neg(self, a: Point) -> Point:
        if a is None:
            return None
        return (a[0], (-a[1]) % self.fp)

    def add(self, a: Point, b: Point) -> Point:
        if a is None:
            return b
        if b is None:
            return a
        p = self.fp
        x1, y1 = a
        x2, y2 = b
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = (3 * x1 * x1) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        y3 = (lam * (x1 - x3) - y1) % p
        return (x3, y3)

    def mul(self, k: int, a: Point) -> Point:
        k %= self.order
        acc: Point = None
        addend = a
        while k:
            if k & 1:
                acc = self.add(acc, addend)
            addend = self.add(addend, addend)
            k >>= 1
        return acc

    def msm_naive(self, scalars: Sequence[int], pts: Sequence[Point]) -> Point:
        assert len(scalars) == len(pts)
        acc: Point = None
        for s, q in zip(scalars, pts):
            acc = self.add(acc, self.mul(s, q))
        return acc

    def msm_windowed(self, scalars: Sequence[int], pts: Sequence[Point], c: int = 8) -> Point:
        """Bucket MSM (the contract of halo2 `best_multiexp`)."""
        assert len(scalars) == len(pts)
        nbits = self.order.bit_length()
        segments = (nbits + c - 1) // c
        acc: Point = None
        for seg in reversed(range(segments)):
            for _ in range(c):
                acc = self.add(acc, acc)
            buckets: List[Point] = [None] * ((1 << c) - 1)
            for s, q in zip(scalars, pts):
                k = ((s % self.order) >> (seg * c)) & ((1 << c) - 1)
                if k:
                    buckets[k - 1] = self.add(buckets[k - 1], q)
            running: Point = None
            for bkt in reversed(buckets):
                running = self.add(running, bkt)
                acc = self.add(acc, running)
        return acc

    # ---- Montgomery raw-limb (de)serialisation ------------------------------
    def to_mont(self, x: int) -> int:
        return (x * MONT_R) % self.fp

    def from_mont(self, xm: int) -> int:
        return (xm * pow(MONT_R, -1, self.fp)) % self.fp

    def montmul(self, am: int, bm: int) -> int:
        return (am * bm * pow(MONT_R, -1, self.fp)) % self.fp

    def affine_to_raw(self, pt: Point) -> bytes:
        """n x (x[4], y[4]) raw Montgomery u64 limbs, identity = (0,0)."""
        if pt is None:
            return bytes(64)
        return self.to_mont(pt[0]).to_bytes(32, "little") + self.to_mont(pt[1]).to_bytes(32, "little")

    def raw_to_affine(self, raw: bytes) -> Point:
        xm = int.from_bytes(raw[:32], "little")
        ym = int.from_bytes(raw[32:64], "little")
        if xm == 0 and ym == 0:
            return None
        return (self.from_mont(xm), self.from_mont(ym))

    def jacobian_raw_to_affine(self, raw: bytes) -> Point:
        """(x,y,z) raw Montgomery, x_aff = X/Z^2, y_aff = Y/Z^3; z == 0 is identity."""
        X = self.from_mont(int.from_bytes(raw[0:32], "little"))
        Y = self.from_mont(int.from_bytes(raw[32:64], "little"))
        Z = self.from_mont(int.from_bytes(raw[64:96], "little"))
        if Z == 0:
            return None
        zi = pow(Z, -1, self.fp)
        return (X * zi * zi % self.fp, Y * zi * zi * zi % self.fp)

    def affine_to_jacobian_raw(self, pt: Point, z: int = 1) -> bytes:
        """Jacobian with an arbitrary non-zero Z (reference passes hash_to_curve
        output with arbitrary Z: src/regular_functions_utils.rs:447-451)."""
        if pt is None:
            return bytes(96)
        z %= self.fp
        assert z != 0
        X = pt[0] * z * z % self.fp
        Y = pt[1] * z * z * z % self.fp
        return b"".join(self.to_mont(v).to_bytes(32, "little") for v in (X, Y, z))

    def canonical(self, pt: Point) -> bytes:
        """Canonical comparison form: affine x||y, 32-byte LE canonical ints; identity = zeros."""
        if pt is None:
            return bytes(64)
        return pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")


BN254_G1 = Curve("bn254_g1", 0, P_BN254, R_BN254, 3, (1, 2))
GRUMPKIN = Curve(
    "grumpkin", 1, R_BN254, P_BN254, (-17) % R_BN254,
    (1, 0x2CF135E7506A45D632D270D45F1181294833FC48D823F272C),
)
CURVES = {c.name: c for c in (BN254_G1, GRUMPKIN)}


# ---------------------------------------------------------------------------
# negabase decomposition  (src/negbase_utils.rs:20-36)
# ---------------------------------------------------------------------------
def negbase_decompose(x: int, base: int) -> List[int]:
    """Digits d_i in [0, base), LSB first, x = sum d_i (-base)^i; [] for 0.

    Follows the reference loop: digit = x % base with Rust's truncated remainder
    fixed up by +base when negative (:24-28); x = -((x - digit)/base) (:32).
    Python's % is already the non-negative remainder, which equals the fixed-up
    value.
    """
    acc: List[int] = []
    while x != 0:
        digit = x % base
        acc.append(digit)
        x = -((x - digit) // base)
    return acc


def id_by_digit(digit: int) -> Optional[int]:       # src/negbase_utils.rs:46-51
    return None if digit == 0 else digit - 1


def digit_by_id(idx: int) -> int:                   # src/negbase_utils.rs:54-56
    return idx + 1


def logb_ceil(x: int, base: int) -> int:            # src/argument_witness_calc.rs:32-40
    i = 0
    while x > 0:
        x //= base
        i += 1
    return i


def scalar_bound(order: int) -> int:                # src/argument_witness_calc.rs:90
    return math.isqrt(order) + 2


def num_digits(order: int, base: int) -> int:       # src/argument_witness_calc.rs:91
    return logb_ceil(scalar_bound(order), base) + 1


def negbase_digits_padded(x: int, base: int, d: int) -> List[int]:
    """LSB-first digits padded with zeros / truncated to d (chain(repeat(0)).take(d), :99)."""
    ds = negbase_decompose(x, base)
    return (ds + [0] * d)[:d]


def precompute_multiplicities(curve: Curve, pt: Point, base: int) -> List[Point]:
    """[1*P .. (base-1)*P] by repeated addition (src/argument_witness_calc.rs:43-51)."""
    acc = pt
    ret = []
    for _ in range(1, base):
        ret.append(acc)
        acc = curve.add(acc, pt)
    return ret


def lhs_msm(curve: Curve, scalars: Sequence[int], pts: Sequence[Point], base: int):
    """MSM core of compute_lhs_witness (src/argument_witness_calc.rs:87-127,132-134).

    Returns (carry, per_digit_carries) where per_digit_carries[i] is `carry`
    after digit position i (MSB first), i.e. the value negated and pushed at :127.
    Raises ValueError where the reference asserts (:88, :97).
    """
    if len(scalars) != len(pts):
        raise ValueError("incompatible amount of coefficients")
    sq_p = scalar_bound(curve.order)
    d = logb_ceil(sq_p, base) + 1
    for s in scalars:
        if not (0 <= s < sq_p):
            raise ValueError("scalar out of range")
    digits = [list(reversed(negbase_digits_padded(s, base, d))) for s in scalars]   # MSB first (:101)
    pre = [precompute_multiplicities(curve, q, base) for q in pts]                  # :103
    carry: Point = None
    carries: List[Point] = []
    for i in range(d):
        carry = curve.mul(base, curve.neg(carry))                                   # :118
        for j in range(len(pts)):
            k = id_by_digit(digits[j][i])
            if k is not None:
                carry = curve.add(carry, pre[j][k])                                 # :123
        carries.append(carry)
    return carry, carries


# ---------------------------------------------------------------------------
# deterministic synthetic inputs (SplitMix64; mirrored in oracle/c and csrc)
# ---------------------------------------------------------------------------
MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def next256(self) -> int:
        return self.next() | (self.next() << 64) | (self.next() << 128) | (self.next() << 192)


def gen_scalars_full(rng: SplitMix64, n: int, order: int) -> List[int]:
    """256 random bits reduced mod the scalar-field order (SURVEY §8d full-width path)."""
    return [rng.next256() % order for _ in range(n)]


def gen_scalars_half(rng: SplitMix64, n: int, order: int) -> List[int]:
    """Uniform in [0, isqrt(order)) as gen_random_coeff (src/argument_witness_calc.rs:65-79)."""
    sq = math.isqrt(order)
    return [rng.next256() % sq for _ in range(n)]


def gen_points(curve: Curve, rng: SplitMix64, n: int) -> List[Point]:
    return [curve.mul(1 + rng.next256() % (curve.order - 1), curve.gen) for _ in range(n)]


def scalars_to_bytes(scalars: Iterable[int]) -> bytes:
    return b"".join(int(s).to_bytes(32, "little") for s in scalars)


# ---------------------------------------------------------------------------------------------
# prepare_scalar_witness / table_entry_by_id  (src/negbase_utils.rs:58-124), restated verbatim -- including the
# places where the reference panics in a debug build (cargo test): assert :81, slice index out of bounds :98-101,
# i128 / u32 overflow inside num_traits::pow and `+=`.  Parity means what the code does (limb index i % logtable + 1,
# the extra (-base) factor of table_entry_by_id), not upstream's presumable intent.
# ---------------------------------------------------------------------------------------------
class RefPanic(Exception):
    """the reference would panic here; kind in {"too_many_digits", "index", "overflow", "div0"}"""

    def __init__(self, kind):
        super().__init__(kind)
        self.kind = kind


_I128 = (-(1 << 127), (1 << 127) - 1)
_U32 = (0, (1 << 32) - 1)


def _chk(v, rng):
    if not (rng[0] <= v <= rng[1]):
        raise RefPanic("overflow")
    return v


def _num_traits_pow(base, exp, rng):
    """num_traits::pow(base, exp) (exponentiation by squaring) with Rust's debug overflow checks on every `*`"""
    if exp == 0:
        return 1
    while exp & 1 == 0:
        base = _chk(base * base, rng)
        exp >>= 1
    if exp == 1:
        return base
    acc = base
    while exp > 1:
        exp >>= 1
        base = _chk(base * base, rng)
        if exp & 1 == 1:
            acc = _chk(acc * base, rng)
    return acc


def negbase_decompose_signed(x: int, base: int):
    """src/negbase_utils.rs:20-36 for a signed BigInt: digit = x % base (truncated remainder, fixed up), x = -((x - digit) / base)"""
    acc = []
    while x != 0:
        digit = abs(x) % base
        if x < 0:
            digit = -digit          # num-bigint: remainder takes the sign of the dividend
        if digit < 0:
            digit += base           # :25-28
        acc.append(digit)
        q = (x - digit)
        assert q % base == 0
        x = -(q // base)            # exact division
    return acc


def prepare_scalar_witness(sc: int, base: int, num_digits: int, logtable: int):
    """src/negbase_utils.rs:79-124 -> list (base rows) of lists (num_limbs+1) of
    ("Scalar", sc) | ("Bucket", i128) | ("Limb", i128, u32)"""
    digits = negbase_decompose_signed(sc, base)
    if not len(digits) <= num_digits:
        raise RefPanic("too_many_digits")                                   # :81
    if logtable == 0:
        raise RefPanic("div0")                                              # :82
    num_limbs = (num_digits + logtable - 1) // logtable
    ret = [[[0, 0] for _ in range(num_limbs + 1)] for _ in range(base)]

    def cell(r, c):
        if c >= len(ret[r]):
            raise RefPanic("index")
        return ret[r][c]

    for i in range(len(digits)):
        if digits[i] == 0:
            continue
        idx = digits[i] - 1                                                  # id_by_digit :46-51
        k = i % logtable
        # (the right operand of a primitive `+=` is evaluated before the place expression)
        v = _num_traits_pow(-base, i, _I128); c = cell(idx + 1, 0); c[0] = _chk(c[0] + v, _I128)       # :97
        v = _num_traits_pow(-base, k, _I128); c = cell(idx + 1, k + 1); c[0] = _chk(c[0] + v, _I128)   # :98
        v = _num_traits_pow(2, k, _U32); c = cell(idx + 1, k + 1); c[1] = _chk(c[1] + v, _U32)         # :99
        v = _num_traits_pow(-base, k, _I128); c = cell(0, k + 1); c[0] = _chk(c[0] + v, _I128)         # :100
        v = _num_traits_pow(2, k, _U32); c = cell(0, k + 1); c[1] = _chk(c[1] + v, _U32)               # :101
    out = []
    for i in range(base):
        row = []
        for j in range(num_limbs + 1):
            if i == 0 and j == 0:
                row.append(("Scalar", sc))
            elif j == 0:
                row.append(("Bucket", ret[i][j][0]))
            else:
                row.append(("Limb", ret[i][j][0], ret[i][j][1]))
        out.append(row)
    return out


def table_entry_by_id(base: int, idx: int, p: int) -> int:
    """src/negbase_utils.rs:58-77 in the field of order p (canonical integer)"""
    if idx == 0:
        return 0
    b = (-base) % p
    acc = 0
    bits = []
    while idx > 0:
        bits.append(idx & 1)
        idx >>= 1
    l = len(bits)
    for i in range(l):
        if bits[l - i - 1] == 1:
            acc = (acc + 1) % p
        acc = acc * b % p
    return acc
