"""Big-integer restatement of the reference's divisor-witness algebra (TEST INFRASTRUCTURE ONLY: imported by
tests/ and nothing else; see oracle/pyref.py's header for the rules).

What it restates (paths relative to /root/reference, file src/regular_functions_utils.rs):
  Polynomial::mul_naive :54-62, mul_fft :102-129 (with best_fft's radix-2 transform and the reference's own
  FftPrecomp twiddles omega_pow / omega_pow_inv / half_pow -- src/precomputed_fft_data.rs, the 192 constants this
  repository pins), Mul dispatch :209-216, Add :178-195, kate_div :45-47 (halo2 kate_division: synthetic division
  by (x - b), remainder dropped, length - 1), RegularFunction Mul :266-273, linefunc :285-303, projective_coords
  :426-431, Propagation::{from_point, empty, from_pair, merge, group_merge} :319-405,
  compute_divisor_witness_partial :453-467, compute_divisor_witness :476-480,
and compute_lhs_witness' second return value, src/argument_witness_calc.rs:105-134.

Points are Jacobian triples (X, Y, Z) of Python ints, the identity has Z == 0, exactly as the reference's
`jacobian_coordinates()`.  A witness is defined up to a non-zero scalar of the field: linefunc works on PROJECTIVE
coordinates derived from whatever Jacobian representative a point happens to have (:287-288, :426-431), so two
correct implementations differ by a scalar factor; `normalise()` divides by the coefficient of the term of
highest pole order and is what tests compare.  Polynomial LENGTHS (trailing zero coefficients included) follow the
reference's rules exactly and are compared exactly.

Where the reference panics this raises RefPanic: sum of the points not the identity (:478), group_merge of an empty
list (:382), usize underflow of `a.len() + b.len() - 1` for two empty polynomials in mul_naive (:55, debug build).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from .pyref import RefPanic

Jac = Tuple[int, int, int]


class FrFft:
    """FftPrecomp for bn256::Fr (src/precomputed_fft_data.rs): built from the root of unity the reference's table starts
    with (omega_pow[0], passed in by the caller from the pinned golden chain head) -- S = 28."""
    S = 28

    def __init__(self, p: int, root_of_unity: int):
        self.p = p
        self.omega = [root_of_unity]
        for _ in range(63):
            self.omega.append(self.omega[-1] * self.omega[-1] % p)
        inv = pow(root_of_unity, -1, p)
        self.omega_inv = [inv]
        for _ in range(63):
            self.omega_inv.append(self.omega_inv[-1] * self.omega_inv[-1] % p)
        half = pow(2, -1, p)
        self.half = [pow(half, i, p) for i in range(64)]
        assert self.omega[28] == 1 and self.omega[27] == p - 1


def log2_floor(num: int) -> int:                                     # :197-207
    assert num > 0
    pw = 0
    while (1 << (pw + 1)) <= num:
        pw += 1
    return pw


def best_fft(a: List[int], omega: int, log_n: int, p: int) -> None:
    """halo2 arithmetic::best_fft restated as the radix-2 decimation-in-time transform it computes:
    a[k] <- sum_j a[j] * omega^(j k), in place (bit-reversal permutation, then log_n butterfly stages)."""
    n = 1 << log_n
    assert len(a) == n
    for k in range(n):
        rk = int(format(k, "0%db" % log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), p)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % p
                a[k + j + m] = (a[k + j] - t) % p
                a[k + j] = (a[k + j] + t) % p
                w = w * w_m % p
        m *= 2


class PolyRing:
    def __init__(self, p: int, fft: Optional[FrFft]):
        self.p = p
        self.fft = fft
        self._bits = (2 * p.bit_length() + 40 + 7) // 8 * 8      # Kronecker slot (whole bytes): room for a sum of 2^40 products

    # ---- Polynomial ----------------------------------------------------------------------------
    def add(self, a: List[int], b: List[int]) -> List[int]:          # :178-195
        n = max(len(a), len(b))
        return [((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % self.p for i in range(n)]

    def scale(self, a: List[int], s: int) -> List[int]:
        return [x * s % self.p for x in a]

    def mul_naive(self, a: List[int], b: List[int]) -> List[int]:    # :54-62 (result via Kronecker substitution: exact)
        ln = len(a) + len(b) - 1
        if ln < 0:
            raise RefPanic("overflow")                               # usize underflow, :55
        if not a or not b:
            return [0] * ln
        s = self._bits
        A = sum(x << (s * i) for i, x in enumerate(a)) if len(a) < 64 else int.from_bytes(b"".join(x.to_bytes(s // 8, "little") for x in a), "little")
        B = sum(x << (s * i) for i, x in enumerate(b)) if len(b) < 64 else int.from_bytes(b"".join(x.to_bytes(s // 8, "little") for x in b), "little")
        C = A * B
        mask = (1 << s) - 1
        return [((C >> (s * i)) & mask) % self.p for i in range(ln)]

    def mul_schoolbook(self, a: List[int], b: List[int]) -> List[int]:
        """:54-62 literally (cross-check of the Kronecker form on small inputs)"""
        ln = len(a) + len(b) - 1
        if ln < 0:
            raise RefPanic("overflow")
        ret = [0] * ln
        for i in range(len(a)):
            for j in range(len(b)):
                ret[i + j] = (ret[i + j] + a[i] * b[j]) % self.p
        return ret

    def mul_fft(self, a: List[int], b: List[int]) -> List[int]:      # :102-129
        f = self.fft
        length = len(a) + len(b) - 1
        loglength = log2_floor(length) + 1
        padded = 1 << loglength
        aa = (a + [0] * padded)[:padded]
        bb = (b + [0] * padded)[:padded]
        assert f.S >= loglength                                      # :110
        omega = f.omega[f.S - loglength]
        omega_inv = f.omega_inv[f.S - loglength]
        scaling = f.half[loglength]
        best_fft(aa, omega, loglength, self.p)
        best_fft(bb, omega, loglength, self.p)
        prod = [x * y % self.p * scaling % self.p for x, y in zip(aa, bb)]
        best_fft(prod, omega_inv, loglength, self.p)
        return prod[:length]

    def mul(self, a: List[int], b: List[int], use_fft: bool = False) -> List[int]:   # :209-216
        if len(a) < 32 or len(b) < 32:
            return self.mul_naive(a, b)
        # the product is the same polynomial either way (exact arithmetic); the literal FFT path is exercised by the
        # tests on moderate sizes (use_fft=True) against the Kronecker form used here for speed
        return self.mul_fft(a, b) if use_fft else self.mul_naive(a, b)

    def kate_div(self, a: List[int], b: int) -> List[int]:           # :45-47, halo2 kate_division
        """quotient of a(x) by (x - b), remainder dropped; len(a) - 1 coefficients"""
        if not a:
            raise RefPanic("overflow")                               # vec![0; a.len() - 1]
        q = [0] * (len(a) - 1)
        tmp = 0
        for i in range(len(a) - 1, 0, -1):
            lead = (a[i] + tmp) % self.p
            q[i - 1] = lead
            tmp = lead * b % self.p
        return q

    def ev(self, a: List[int], x: int) -> int:
        acc = 0
        for c in reversed(a):
            acc = (acc * x + c) % self.p
        return acc


class DivisorOracle:
    """RegularFunction / Propagation algebra over one curve (y^2 = x^3 + b, a = 0)."""

    def __init__(self, curve, fft: Optional[FrFft] = None, use_fft: bool = False):
        self.c = curve
        self.p = curve.fp
        self.R = PolyRing(curve.fp, fft)
        self.use_fft = use_fft

    # ---- points (Jacobian) ---------------------------------------------------------------------
    def to_affine(self, pt: Jac):
        X, Y, Z = pt
        if Z % self.p == 0:
            return None
        zi = pow(Z, -1, self.p)
        return (X * zi * zi % self.p, Y * zi * zi * zi % self.p)

    def from_affine(self, a, z: int = 1) -> Jac:
        if a is None:
            return (0, 1, 0)
        return (a[0] * z * z % self.p, a[1] * z * z * z % self.p, z % self.p)

    def is_identity(self, pt: Jac) -> bool:
        return pt[2] % self.p == 0

    def padd(self, a: Jac, b: Jac) -> Jac:
        return self.from_affine(self.c.add(self.to_affine(a), self.to_affine(b)))

    def pneg(self, a: Jac) -> Jac:
        return (a[0], (-a[1]) % self.p, a[2])

    def peq(self, a: Jac, b: Jac) -> bool:
        return self.to_affine(a) == self.to_affine(b)

    def projective_coords(self, pt: Jac):                            # :426-431
        x, y, z = pt
        zsq = z * z % self.p
        return (x * z % self.p, y % self.p, z * zsq % self.p)

    # ---- RegularFunction = (a, b): a(x) + y b(x) ------------------------------------------------
    def rf_mul(self, f, g):                                          # :266-273
        R = self.R
        subst = [self.c.b % self.p, 0, 0, 1]                         # x^3 + a x + b with a = 0
        m = lambda u, v: R.mul(u, v, self.use_fft)
        a = R.add(m(f[0], g[0]), m(m(f[1], g[1]), subst))
        b = R.add(m(f[0], g[1]), m(f[1], g[0]))
        return (a, b)

    def from_line(self, a: int, b: int, c: int):                     # :244-246: a x + b y + c
        return ([c % self.p, a % self.p], [b % self.p])

    def linefunc(self, a: Jac, b: Jac):                              # :285-303
        p = self.p
        ax, ay, az = self.projective_coords(a)
        bx, by, bz = self.projective_coords(b)
        lz = (ax * by - ay * bx) % p
        lx = (ay * bz - az * by) % p
        ly = (az * bx - ax * bz) % p
        if lx or ly or lz:
            return self.from_line(lx, ly, lz)
        c = self.pneg(self.padd(a, b))
        cx, cy, cz = self.projective_coords(c)
        return self.from_line(ay * cz - az * cy, az * cx - ax * cz, ax * cy - ay * cx)

    def rf_ev(self, f, pt: Jac) -> int:                              # :228-237
        x, y = self.to_affine(pt)
        return (self.R.ev(f[0], x) + self.R.ev(f[1], x) * y) % self.p

    # ---- Propagation = (inputs, output, wtns) ---------------------------------------------------
    def empty(self):                                                 # :324-326
        return ([], (0, 1, 0), ([1], []))

    def from_point(self, pt: Jac):                                   # :319-322
        if self.is_identity(pt):
            return self.empty()
        return ([pt], self.pneg(pt), self.linefunc(pt, self.pneg(pt)))

    def from_pair(self, p1: Jac, p2: Jac):                           # :328-331
        if self.is_identity(p1):
            return self.from_point(p2)
        return ([p1, p2], self.pneg(self.padd(p1, p2)), self.linefunc(p1, p2))

    def merge(self, A, B):                                           # :333-360
        inputs = A[0] + B[0]
        output = self.padd(A[1], B[1])
        if self.is_identity(A[1]) or self.is_identity(B[1]):
            return (inputs, output, self.rf_mul(A[2], B[2]))
        numerator = self.rf_mul(A[2], self.rf_mul(B[2], self.linefunc(self.pneg(A[1]), self.pneg(B[1]))))
        ax = self.to_affine(A[1])[0]
        bx = self.to_affine(B[1])[0]
        R = self.R
        wt = (R.kate_div(R.kate_div(numerator[0], ax), bx), R.kate_div(R.kate_div(numerator[1], ax), bx))
        return (inputs, output, wt)

    def group_merge(self, arr):                                      # :380-405
        if len(arr) == 0:
            raise RefPanic("empty")
        while len(arr) > 1:
            nxt = []
            for i in range(0, len(arr) - 1, 2):
                nxt.append(self.merge(arr[i], arr[i + 1]))
            if len(arr) & 1:
                nxt.append(arr[-1])
            arr = nxt
        return arr[0]

    def compute_divisor_witness_partial(self, pts: Sequence[Jac]):   # :453-467
        if len(pts) == 0:
            return (([1], []), (0, 1, 0))
        tmp = []
        i = 0
        while i < len(pts) - 1:
            tmp.append(self.from_pair(pts[i], pts[i + 1]))
            i += 2
        if i == len(pts) - 1:
            tmp.append(self.from_point(pts[i]))
        ret = self.group_merge(tmp)
        return (ret[2], ret[1])

    def compute_divisor_witness(self, pts: Sequence[Jac]):           # :476-480
        w, out = self.compute_divisor_witness_partial(pts)
        if not self.is_identity(out):
            raise RefPanic("sum_not_identity")
        return w

    # ---- comparison form ------------------------------------------------------------------------
    def normalise(self, f):
        """divide by the coefficient of the term of highest pole order at infinity (x^i: 2i, y x^i: 2i + 3);
        the zero function stays as it is.  Lengths are kept."""
        a, b = f
        best = None
        for i, v in enumerate(a):
            if v % self.p:
                best = max(best or (-1, 0), (2 * i, v))
        for i, v in enumerate(b):
            if v % self.p:
                best = max(best or (-1, 0), (2 * i + 3, v))
        if best is None:
            return ([0] * len(a), [0] * len(b))
        inv = pow(best[1], -1, self.p)
        return ([x * inv % self.p for x in a], [x * inv % self.p for x in b])


def compute_lhs_witness(oracle: DivisorOracle, scalars: Sequence[int], pts: Sequence[Jac], base: int):
    """src/argument_witness_calc.rs:87-136 in full: (carry, Vec<RegularFunction>) -- the MSM core as in
    oracle/pyref.py's lhs_msm plus the per-digit divisor witnesses of :129, reversed (:132)."""
    from . import pyref
    c = oracle.c
    if len(scalars) != len(pts):
        raise RefPanic("len")
    sq_p = pyref.scalar_bound(c.order)                               # :90
    d = pyref.num_digits(c.order, base)                              # :91
    for s in scalars:
        if not s < sq_p:
            raise RefPanic("range")                                  # :97
    digits = [pyref.negbase_digits_padded(s, base, d)[::-1] for s in scalars]   # :99-101
    pre = []
    for pt in pts:
        row = [pt]
        for _ in range(base - 2):
            row.append(oracle.padd(row[-1], pt))
        pre.append(row)                                              # precompute_multiplicities :43-51
    carry = (0, 1, 0)
    ret = []
    for i in range(d):
        tmp = []
        if not oracle.is_identity(carry):
            for _ in range(base):
                tmp.append(oracle.pneg(carry))
        carry = oracle.from_affine(c.mul(base, c.neg(oracle.to_affine(carry))))   # -carry * base  :118
        for j in range(len(pts)):
            dg = digits[j][i]
            if dg:
                tmp.append(pre[j][dg - 1])
                carry = oracle.padd(carry, pre[j][dg - 1])
        tmp.append(oracle.pneg(carry))
        ret.append(oracle.compute_divisor_witness(tmp))
    ret.reverse()
    return carry, ret
