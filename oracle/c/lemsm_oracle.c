/*
 * CPU restatement (plain C, 4x64-bit Montgomery limbs) of the reference's MSM
 * witness path.  TEST INFRASTRUCTURE ONLY: nothing in the product library
 * (halo2_liam_eagen_msm_amd/csrc) links, includes or calls this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load the
 * resulting liblemsm_oracle.so, and only as the checker / timed CPU baseline.
 *
 * Restated from (paths relative to /root/reference):
 *   negbase_decompose             src/negbase_utils.rs:20-36
 *   id_by_digit                   src/negbase_utils.rs:46-51
 *   logb_ceil, order, d           src/argument_witness_calc.rs:32-40,54-56,89-91
 *   precompute_multiplicities     src/argument_witness_calc.rs:43-51
 *   compute_lhs_witness MSM core  src/argument_witness_calc.rs:87-127,132-134
 *   best_multiexp/multiexp_serial third-party halo2_proofs::arithmetic (git
 *       https://github.com/levs57/halo2, NO pinned revision, Cargo.toml:10;
 *       source absent from /root/reference).  Its published algorithm is restated:
 *       split into T chunks of n/T, serial Pippenger per chunk with window
 *       c = 1 (len<4) | 3 (len<32) | ceil(ln len), 256/c+1 segments, 2^c-1
 *       buckets with running-sum reduction, partial sums folded serially.
 *   field/curve arithmetic        third-party halo2curves (unpinned, Cargo.toml:11):
 *       4x u64 little-endian limbs, Montgomery R = 2^256 (confirmed by the raw
 *       constants in src/precomputed_fft_data.rs), Jacobian coordinates.
 *
 * Pinning: r-modulus montmul is checked against all 192 constants of
 * src/precomputed_fft_data.rs (tests/test_oracle_golden.py); the group law is
 * checked against oracle/pyref.py (independent affine big-int arithmetic) and
 * the public EIP-196 vector 2*(1,2).  The reference holds no MSM golden
 * vectors: MSM byte-level parity is "parity unpinned" by reference fixtures
 *  and is defined as equality of group elements in canonical
 * affine form.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

typedef struct { u64 l[4]; } fe;

typedef struct {
    fe N;        /* modulus */
    u64 inv;     /* -N^-1 mod 2^64 */
    fe R;        /* 2^256 mod N  (Montgomery one) */
    fe R2;       /* 2^512 mod N */
    fe b;        /* curve constant b, Montgomery form */
    fe gx, gy;   /* generator, Montgomery form */
    fe order;    /* group order (scalar field modulus), plain integer */
} curve_t;

static const u64 P_LIMBS[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 R_LIMBS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

static curve_t CURVES[2];
static int curves_ready = 0;

/* ---------- 256-bit helpers ---------- */
static int u256_geq(const u64 *a, const u64 *b) {
    for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static u64 u256_add(u64 *r, const u64 *a, const u64 *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r[i] = (u64)c; c >>= 64; }
    return (u64)c;
}
static u64 u256_sub(u64 *r, const u64 *a, const u64 *b) {
    u64 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d; borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
static int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }

/* ---------- field ---------- */
static void f_add(const curve_t *c, fe *r, const fe *a, const fe *b) {
    u64 t[4]; u64 carry = u256_add(t, a->l, b->l);
    if (carry || u256_geq(t, c->N.l)) u256_sub(t, t, c->N.l);
    memcpy(r->l, t, 32);
}
static void f_sub(const curve_t *c, fe *r, const fe *a, const fe *b) {
    u64 t[4]; if (u256_sub(t, a->l, b->l)) u256_add(t, t, c->N.l);
    memcpy(r->l, t, 32);
}
static void f_neg(const curve_t *c, fe *r, const fe *a) {
    if (fe_is_zero(a)) { *r = *a; return; }
    u64 t[4]; u256_sub(t, c->N.l, a->l); memcpy(r->l, t, 32);
}
static void f_dbl(const curve_t *c, fe *r, const fe *a) { f_add(c, r, a, a); }

/* CIOS Montgomery product a*b*R^-1 mod N */
static void f_mul(const curve_t *c, fe *r, const fe *a, const fe *b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 cur = (u128)a->l[j] * b->l[i] + t[j] + carry;
            t[j] = (u64)cur; carry = cur >> 64;
        }
        u128 cur = (u128)t[4] + carry; t[4] = (u64)cur; t[5] = (u64)(cur >> 64);
        u64 m = t[0] * c->inv;
        cur = (u128)m * c->N.l[0] + t[0]; carry = cur >> 64;
        for (int j = 1; j < 4; j++) {
            cur = (u128)m * c->N.l[j] + t[j] + carry;
            t[j - 1] = (u64)cur; carry = cur >> 64;
        }
        cur = (u128)t[4] + carry; t[3] = (u64)cur; t[4] = t[5] + (u64)(cur >> 64);
    }
    if (t[4] || u256_geq(t, c->N.l)) u256_sub(t, t, c->N.l);
    memcpy(r->l, t, 32);
}
static void f_sqr(const curve_t *c, fe *r, const fe *a) { f_mul(c, r, a, a); }

static void f_pow(const curve_t *c, fe *r, const fe *a, const u64 *e) {
    fe acc = c->R, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) f_mul(c, &acc, &acc, &base);
        f_sqr(c, &base, &base);
    }
    *r = acc;
}
static void f_inv(const curve_t *c, fe *r, const fe *a) {
    u64 e[4]; u64 two[4] = {2, 0, 0, 0};
    u256_sub(e, c->N.l, two);
    f_pow(c, r, a, e);
}
static void f_to_mont(const curve_t *c, fe *r, const fe *a) { f_mul(c, r, a, &c->R2); }
static void f_from_mont(const curve_t *c, fe *r, const fe *a) {
    fe one = {{1, 0, 0, 0}}; f_mul(c, r, a, &one);
}

static void curve_setup(curve_t *c, const u64 *N, const u64 *order, int b_neg, u64 b_abs, const u64 *gy_plain) {
    memcpy(c->N.l, N, 32); memcpy(c->order.l, order, 32);
    u64 x = 1; for (int i = 0; i < 6; i++) x *= 2 - N[0] * x;   /* Newton: x = N^-1 mod 2^64 */
    c->inv = (u64)0 - x;
    u64 t[4] = {1, 0, 0, 0};
    for (int i = 0; i < 512; i++) {
        u64 carry = u256_add(t, t, t);
        if (carry || u256_geq(t, N)) u256_sub(t, t, N);
        if (i == 255) memcpy(c->R.l, t, 32);
    }
    memcpy(c->R2.l, t, 32);
    fe bb = {{b_abs, 0, 0, 0}}; f_to_mont(c, &c->b, &bb);
    if (b_neg) f_neg(c, &c->b, &c->b);
    c->gx = c->R;
    fe gy; memcpy(gy.l, gy_plain, 32); f_to_mont(c, &c->gy, &gy);
}
static void ensure_curves(void) {
    if (curves_ready) return;
    static const u64 gy_bn[4] = {2, 0, 0, 0};
    /* Grumpkin generator y: 0x2cf135e7506a45d632d270d45f1181294833fc48d823f272c */
    static const u64 gy_gr[4] = {0x833fc48d823f272cULL, 0x2d270d45f1181294ULL, 0xcf135e7506a45d63ULL, 0x2ULL};
    curve_setup(&CURVES[0], P_LIMBS, R_LIMBS, 0, 3, gy_bn);    /* BN254 G1: y^2 = x^3 + 3 over p, order r */
    curve_setup(&CURVES[1], R_LIMBS, P_LIMBS, 1, 17, gy_gr);   /* Grumpkin: y^2 = x^3 - 17 over r, order p */
    curves_ready = 1;
}
static const curve_t *get_curve(int cid) { ensure_curves(); return (cid == 0 || cid == 1) ? &CURVES[cid] : NULL; }

/* ---------- Jacobian points (X,Y,Z), identity Z = 0 ---------- */
typedef struct { fe x, y, z; } jac;
typedef struct { fe x, y; } aff;   /* identity = (0,0) */

static int aff_is_id(const aff *a) { return fe_is_zero(&a->x) && fe_is_zero(&a->y); }
static int jac_is_id(const jac *a) { return fe_is_zero(&a->z); }
static void jac_set_id(jac *a) { memset(a, 0, sizeof(*a)); }

static void jac_double(const curve_t *c, jac *r, const jac *p) {
    if (jac_is_id(p)) { jac_set_id(r); return; }
    /* a = 0: dbl-2009-l */
    fe A, B, C, D, E, F, t, x3, y3, z3;
    f_sqr(c, &A, &p->x); f_sqr(c, &B, &p->y); f_sqr(c, &C, &B);
    f_add(c, &t, &p->x, &B); f_sqr(c, &t, &t); f_sub(c, &t, &t, &A); f_sub(c, &t, &t, &C); f_dbl(c, &D, &t);
    f_dbl(c, &E, &A); f_add(c, &E, &E, &A);
    f_sqr(c, &F, &E);
    f_dbl(c, &t, &D); f_sub(c, &x3, &F, &t);
    f_mul(c, &z3, &p->y, &p->z); f_dbl(c, &z3, &z3);
    f_sub(c, &t, &D, &x3); f_mul(c, &y3, &E, &t);
    f_dbl(c, &t, &C); f_dbl(c, &t, &t); f_dbl(c, &t, &t); f_sub(c, &y3, &y3, &t);
    r->x = x3; r->y = y3; r->z = z3;
}
static void jac_add(const curve_t *c, jac *r, const jac *p, const jac *q) {
    if (jac_is_id(p)) { *r = *q; return; }
    if (jac_is_id(q)) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
    f_sqr(c, &z1z1, &p->z); f_sqr(c, &z2z2, &q->z);
    f_mul(c, &u1, &p->x, &z2z2); f_mul(c, &u2, &q->x, &z1z1);
    f_mul(c, &t, &q->z, &z2z2); f_mul(c, &s1, &p->y, &t);
    f_mul(c, &t, &p->z, &z1z1); f_mul(c, &s2, &q->y, &t);
    f_sub(c, &h, &u2, &u1); f_sub(c, &rr, &s2, &s1);
    if (fe_is_zero(&h)) {
        if (fe_is_zero(&rr)) { jac_double(c, r, p); return; }
        jac_set_id(r); return;
    }
    fe hh, hhh, v, x3, y3, z3;
    f_sqr(c, &hh, &h); f_mul(c, &hhh, &hh, &h); f_mul(c, &v, &u1, &hh);
    f_sqr(c, &x3, &rr); f_sub(c, &x3, &x3, &hhh); f_dbl(c, &t, &v); f_sub(c, &x3, &x3, &t);
    f_sub(c, &t, &v, &x3); f_mul(c, &y3, &rr, &t); f_mul(c, &t, &s1, &hhh); f_sub(c, &y3, &y3, &t);
    f_mul(c, &z3, &p->z, &q->z); f_mul(c, &z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}
static void jac_from_aff(const curve_t *c, jac *r, const aff *a) {
    if (aff_is_id(a)) { jac_set_id(r); return; }
    r->x = a->x; r->y = a->y; r->z = c->R;
}
static void jac_add_aff(const curve_t *c, jac *r, const jac *p, const aff *q) {
    if (aff_is_id(q)) { *r = *p; return; }
    if (jac_is_id(p)) { jac_from_aff(c, r, q); return; }
    fe z1z1, u2, s2, h, rr, t;
    f_sqr(c, &z1z1, &p->z);
    f_mul(c, &u2, &q->x, &z1z1);
    f_mul(c, &t, &p->z, &z1z1); f_mul(c, &s2, &q->y, &t);
    f_sub(c, &h, &u2, &p->x); f_sub(c, &rr, &s2, &p->y);
    if (fe_is_zero(&h)) {
        if (fe_is_zero(&rr)) { jac_double(c, r, p); return; }
        jac_set_id(r); return;
    }
    fe hh, hhh, v, x3, y3, z3;
    f_sqr(c, &hh, &h); f_mul(c, &hhh, &hh, &h); f_mul(c, &v, &p->x, &hh);
    f_sqr(c, &x3, &rr); f_sub(c, &x3, &x3, &hhh); f_dbl(c, &t, &v); f_sub(c, &x3, &x3, &t);
    f_sub(c, &t, &v, &x3); f_mul(c, &y3, &rr, &t); f_mul(c, &t, &p->y, &hhh); f_sub(c, &y3, &y3, &t);
    f_mul(c, &z3, &p->z, &h);
    r->x = x3; r->y = y3; r->z = z3;
}
static void jac_neg(const curve_t *c, jac *r, const jac *p) { r->x = p->x; r->z = p->z; f_neg(c, &r->y, &p->y); }
static void jac_to_aff(const curve_t *c, aff *r, const jac *p) {
    if (jac_is_id(p)) { memset(r, 0, sizeof(*r)); return; }
    fe zi, zi2, zi3;
    f_inv(c, &zi, &p->z); f_sqr(c, &zi2, &zi); f_mul(c, &zi3, &zi2, &zi);
    f_mul(c, &r->x, &p->x, &zi2); f_mul(c, &r->y, &p->y, &zi3);
}
/* k * P for a small/large plain-integer k given as 4 limbs (double-and-add, MSB first) */
static void jac_mul_limbs(const curve_t *c, jac *r, const jac *p, const u64 *k) {
    jac acc; jac_set_id(&acc);
    for (int i = 255; i >= 0; i--) {
        jac_double(c, &acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) jac_add(c, &acc, &acc, p);
    }
    *r = acc;
}
static int jac_eq(const curve_t *c, const jac *p, const jac *q) {
    if (jac_is_id(p) || jac_is_id(q)) return jac_is_id(p) && jac_is_id(q);
    fe z1z1, z2z2, a, b, t;
    f_sqr(c, &z1z1, &p->z); f_sqr(c, &z2z2, &q->z);
    f_mul(c, &a, &p->x, &z2z2); f_mul(c, &b, &q->x, &z1z1);
    if (!fe_eq(&a, &b)) return 0;
    f_mul(c, &t, &z2z2, &q->z); f_mul(c, &a, &p->y, &t);
    f_mul(c, &t, &z1z1, &p->z); f_mul(c, &b, &q->y, &t);
    return fe_eq(&a, &b);
}

/* ================= exported API (ctypes) ================= */
enum { ORC_OK = 0, ORC_LEN_MISMATCH = 1, ORC_SCALAR_OUT_OF_RANGE = 2, ORC_BAD_BASE = 3, ORC_BAD_CURVE = 4, ORC_TOO_MANY_DIGITS = 5 };

int orc_field_consts(int cid, u64 *modulus, u64 *inv, u64 *R, u64 *R2) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    memcpy(modulus, c->N.l, 32); *inv = c->inv; memcpy(R, c->R.l, 32); memcpy(R2, c->R2.l, 32);
    return ORC_OK;
}
int orc_montmul(int cid, const u64 *a, const u64 *b, u64 *out) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    fe x, y, r; memcpy(x.l, a, 32); memcpy(y.l, b, 32); f_mul(c, &r, &x, &y); memcpy(out, r.l, 32);
    return ORC_OK;
}
int orc_generator(int cid, u64 *out_aff) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    memcpy(out_aff, c->gx.l, 32); memcpy(out_aff + 4, c->gy.l, 32); return ORC_OK;
}
int orc_is_on_curve_aff(int cid, const u64 *pt) {
    const curve_t *c = get_curve(cid); if (!c) return 0;
    aff a; memcpy(&a, pt, 64); if (aff_is_id(&a)) return 1;
    fe l, r2; f_sqr(c, &l, &a.y); f_sqr(c, &r2, &a.x); f_mul(c, &r2, &r2, &a.x); f_add(c, &r2, &r2, &c->b);
    return fe_eq(&l, &r2);
}
/* Jacobian raw Montgomery -> canonical affine bytes (x||y, 32-byte LE canonical; identity zeros) */
int orc_jac_to_canonical(int cid, const u64 *jac_in, uint8_t *out64) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    jac p; memcpy(&p, jac_in, 96); aff a; jac_to_aff(c, &a, &p);
    fe x, y; f_from_mont(c, &x, &a.x); f_from_mont(c, &y, &a.y);
    if (jac_is_id(&p)) { memset(out64, 0, 64); return ORC_OK; }
    memcpy(out64, x.l, 32); memcpy(out64 + 32, y.l, 32); return ORC_OK;
}
int orc_jac_to_aff_raw(int cid, const u64 *jac_in, u64 *aff_out) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    jac p; memcpy(&p, jac_in, 96); aff a; jac_to_aff(c, &a, &p); memcpy(aff_out, &a, 64); return ORC_OK;
}
int orc_jac_add(int cid, const u64 *a, const u64 *b, u64 *out) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    jac p, q, r; memcpy(&p, a, 96); memcpy(&q, b, 96); jac_add(c, &r, &p, &q); memcpy(out, &r, 96); return ORC_OK;
}
int orc_jac_eq(int cid, const u64 *a, const u64 *b) {
    const curve_t *c = get_curve(cid); if (!c) return 0;
    jac p, q; memcpy(&p, a, 96); memcpy(&q, b, 96); return jac_eq(c, &p, &q);
}
/* k (32-byte LE plain integer) * affine raw point -> Jacobian raw */
int orc_scalar_mul(int cid, const uint8_t *k32, const u64 *pt_aff, u64 *out_jac) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    aff a; memcpy(&a, pt_aff, 64); jac p, r; jac_from_aff(c, &p, &a);
    u64 k[4]; memcpy(k, k32, 32); jac_mul_limbs(c, &r, &p, k); memcpy(out_jac, &r, 96); return ORC_OK;
}

/* ---- SplitMix64 synthetic inputs (same generator as oracle/pyref.py) ---- */
static u64 splitmix_next(u64 *s) {
    u64 z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
/* x mod m for 256-bit x (m > 2^253): conditional subtraction suffices after at most 5 steps */
static void u256_mod(u64 *x, const u64 *m) { while (u256_geq(x, m)) u256_sub(x, x, m); }

/* Batch affine normalisation (Montgomery's trick) of n Jacobian points, none of which is the identity
   unless flagged; identity maps to (0,0). */
static void batch_to_aff(const curve_t *c, aff *out, const jac *in, size_t n) {
    fe *pref = (fe *)malloc(sizeof(fe) * (n + 1));
    fe acc = c->R;
    for (size_t i = 0; i < n; i++) { pref[i] = acc; if (!jac_is_id(&in[i])) f_mul(c, &acc, &acc, &in[i].z); }
    fe inv; f_inv(c, &inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (jac_is_id(&in[i])) { memset(&out[i], 0, sizeof(aff)); continue; }
        fe zi, zi2, zi3; f_mul(c, &zi, &inv, &pref[i]); f_mul(c, &inv, &inv, &in[i].z);
        f_sqr(c, &zi2, &zi); f_mul(c, &zi3, &zi2, &zi);
        f_mul(c, &out[i].x, &in[i].x, &zi2); f_mul(c, &out[i].y, &in[i].y, &zi3);
    }
    free(pref);
}
/* P_i = k_i * G, k_i = 1 + (256 random bits mod (order-1)); affine raw Montgomery out */
int orc_gen_points(int cid, u64 seed, size_t n, u64 *out_aff) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    jac g; g.x = c->gx; g.y = c->gy; g.z = c->R;
    jac *tmp = (jac *)malloc(sizeof(jac) * (n ? n : 1));
    u64 om1[4]; u64 one[4] = {1, 0, 0, 0}; u256_sub(om1, c->order.l, one);
    for (size_t i = 0; i < n; i++) {
        u64 k[4]; for (int j = 0; j < 4; j++) k[j] = splitmix_next(&seed);
        u256_mod(k, om1); u256_add(k, k, one);
        jac_mul_limbs(c, &tmp[i], &g, k);
    }
    batch_to_aff(c, (aff *)out_aff, tmp, n);
    free(tmp); return ORC_OK;
}
/* P_i = (i+1) * Q for i in [0,n): a cheap walk giving valid curve points with a known discrete-log
   relation, so that sum s_i P_i == (sum s_i (i+1)) * Q can be checked with one scalar mul. */
int orc_gen_walk(int cid, const u64 *q_aff, size_t n, u64 *out_aff) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    aff q; memcpy(&q, q_aff, 64);
    const size_t CH = 4096;
    jac *tmp = (jac *)malloc(sizeof(jac) * CH);
    jac cur; jac_set_id(&cur);
    for (size_t base = 0; base < n; base += CH) {
        size_t m = n - base < CH ? n - base : CH;
        for (size_t i = 0; i < m; i++) { jac_add_aff(c, &cur, &cur, &q); tmp[i] = cur; }
        batch_to_aff(c, (aff *)out_aff + base, tmp, m);
    }
    free(tmp); return ORC_OK;
}
/* sum_i s_i * (i+1) mod order, scalars 32-byte LE canonical; out 32-byte LE */
int orc_walk_dot(int cid, const uint8_t *scalars, size_t n, uint8_t *out32) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    /* accumulate sum s_i*(i+1) in 384 bits, reduce at the end by binary long division */
    u64 acc[6] = {0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        u64 s[4]; memcpy(s, scalars + 32 * i, 32);
        u64 w = (u64)(i + 1); u128 carry = 0;
        for (int j = 0; j < 4; j++) { u128 cur = (u128)s[j] * w + acc[j] + carry; acc[j] = (u64)cur; carry = cur >> 64; }
        for (int j = 4; j < 6; j++) { u128 cur = (u128)acc[j] + carry; acc[j] = (u64)cur; carry = cur >> 64; }
    }
    /* reduce: r = acc mod order via shift-subtract over 384 bits */
    u64 r[4] = {0, 0, 0, 0};
    for (int bit = 383; bit >= 0; bit--) {
        u64 top = r[3] >> 63;
        r[3] = (r[3] << 1) | (r[2] >> 63); r[2] = (r[2] << 1) | (r[1] >> 63);
        r[1] = (r[1] << 1) | (r[0] >> 63); r[0] = (r[0] << 1) | ((acc[bit >> 6] >> (bit & 63)) & 1);
        if (top || u256_geq(r, c->order.l)) u256_sub(r, r, c->order.l);
    }
    memcpy(out32, r, 32); return ORC_OK;
}
/* scalars: full = 256 random bits mod order; half = mod isqrt(order) (gen_random_coeff,
   src/argument_witness_calc.rs:65-79).  isqrt is passed in by the caller (computed in Python). */
int orc_gen_scalars(int cid, u64 seed, size_t n, const u64 *modulus4, uint8_t *out) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    for (size_t i = 0; i < n; i++) {
        u64 k[4]; for (int j = 0; j < 4; j++) k[j] = splitmix_next(&seed);
        if (modulus4[3] >> 61) { u256_mod(k, modulus4); }
        else {  /* small modulus: shift-subtract long division */
            u64 r[4] = {0, 0, 0, 0};
            for (int bit = 255; bit >= 0; bit--) {
                r[3] = (r[3] << 1) | (r[2] >> 63); r[2] = (r[2] << 1) | (r[1] >> 63);
                r[1] = (r[1] << 1) | (r[0] >> 63); r[0] = (r[0] << 1) | ((k[bit >> 6] >> (bit & 63)) & 1);
                if (u256_geq(r, modulus4)) u256_sub(r, r, modulus4);
            }
            memcpy(k, r, 32);
        }
        memcpy(out + 32 * i, k, 32);
    }
    return ORC_OK;
}

/* ---- negabase decomposition (src/negbase_utils.rs:20-36) ----
 * x is a signed big integer held as sign + magnitude (4 limbs suffice: callers pass < 2^127,
 * and |x| shrinks every step).  digit = x mod B in [0,B) (Rust's truncated remainder plus the
 * +B fix-up of :24-28); x <- -((x - digit) / B) (:32).  Digits LSB first; empty for 0. */
static u64 mag_divmod_small(u64 *m, u64 b) {   /* m <- m / b, returns m mod b */
    u128 rem = 0;
    for (int i = 3; i >= 0; i--) { u128 cur = (rem << 64) | m[i]; m[i] = (u64)(cur / b); rem = cur % b; }
    return (u64)rem;
}
static int mag_is_zero(const u64 *m) { return (m[0] | m[1] | m[2] | m[3]) == 0; }
static size_t negbase_decompose_raw(const u64 *x4, uint8_t base, uint8_t *digits, size_t cap) {
    u64 m[4]; memcpy(m, x4, 32); int neg = 0; size_t len = 0;
    while (!mag_is_zero(m)) {
        u64 q[4]; memcpy(q, m, 32);
        u64 rem = mag_divmod_small(q, base);          /* |x| = q*B + rem */
        u64 digit;
        if (!neg) { digit = rem; /* (x - digit)/B = q ; x <- -q */ memcpy(m, q, 32); neg = 1; }
        else {
            /* x = -(q*B + rem); digit = (B - rem) mod B; (x - digit)/B = -(q + (rem?1:0)); x <- +(q + (rem?1:0)) */
            digit = rem ? base - rem : 0;
            memcpy(m, q, 32);
            if (rem) { u64 one[4] = {1, 0, 0, 0}; u256_add(m, m, one); }
            neg = 0;
        }
        if (mag_is_zero(m)) neg = 0;
        if (len < cap) digits[len] = (uint8_t)digit;
        len++;
    }
    return len;
}
int orc_negbase_decompose(const uint8_t *x32, uint8_t base, uint8_t *digits, size_t cap, size_t *out_len) {
    if (base < 2) return ORC_BAD_BASE;
    u64 x[4]; memcpy(x, x32, 32);
    *out_len = negbase_decompose_raw(x, base, digits, cap);
    return ORC_OK;
}
/* isqrt(order)+2 as 4 limbs (src/argument_witness_calc.rs:90): bitwise integer square root */
static void scalar_bound(const curve_t *c, u64 *out) {
    /* find largest s with s*s <= order, s < 2^128 */
    u64 s[2] = {0, 0};
    for (int bit = 127; bit >= 0; bit--) {
        u64 t[2] = {s[0], s[1]}; t[bit >> 6] |= 1ULL << (bit & 63);
        /* t*t as 256-bit */
        u128 p00 = (u128)t[0] * t[0], p01 = (u128)t[0] * t[1], p11 = (u128)t[1] * t[1];
        u64 sq[4]; u128 acc;
        sq[0] = (u64)p00; acc = (p00 >> 64) + (u64)p01 + (u64)p01;
        sq[1] = (u64)acc; acc = (acc >> 64) + (p01 >> 64) + (p01 >> 64) + (u64)p11;
        sq[2] = (u64)acc; acc = (acc >> 64) + (p11 >> 64);
        sq[3] = (u64)acc;
        if ((acc >> 64) == 0 && u256_geq(c->order.l, sq)) { s[0] = t[0]; s[1] = t[1]; }
    }
    u64 r[4] = {s[0], s[1], 0, 0}; u64 two[4] = {2, 0, 0, 0}; u256_add(r, r, two); memcpy(out, r, 32);
}
static unsigned logb_ceil4(const u64 *x4, uint8_t base) {   /* src/argument_witness_calc.rs:32-40 */
    u64 m[4]; memcpy(m, x4, 32); unsigned i = 0;
    while (!mag_is_zero(m)) { mag_divmod_small(m, base); i++; }
    return i;
}
int orc_num_digits(int cid, uint8_t base, uint32_t *d, u64 *bound4) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    if (base < 2) return ORC_BAD_BASE;
    u64 b[4]; scalar_bound(c, b); *d = logb_ceil4(b, base) + 1; if (bound4) memcpy(bound4, b, 32);
    return ORC_OK;
}
/* batch: digits[n][d], LSB first, zero padded / truncated to d exactly like chain(repeat(0)).take(d) (:99) */
int orc_negbase_decompose_batch(const uint8_t *scalars, size_t n, uint8_t base, uint32_t d, uint8_t *digits) {
    if (base < 2) return ORC_BAD_BASE;
    for (size_t i = 0; i < n; i++) {
        u64 x[4]; memcpy(x, scalars + 32 * i, 32);
        memset(digits + (size_t)d * i, 0, d);
        negbase_decompose_raw(x, base, digits + (size_t)d * i, d);
    }
    return ORC_OK;
}
/* precompute_multiplicities (src/argument_witness_calc.rs:43-51): [1P..(B-1)P], first add is P+P */
int orc_precompute_multiplicities(int cid, const u64 *pt_jac, uint8_t base, u64 *out_jac) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    jac p, acc; memcpy(&p, pt_jac, 96); acc = p;
    for (unsigned k = 1; k < base; k++) { memcpy(out_jac + 12 * (k - 1), &acc, 96); jac_add(c, &acc, &acc, &p); }
    return ORC_OK;
}
/* MSM core of compute_lhs_witness (src/argument_witness_calc.rs:87-127,132-134), serial like the
 * reference: per digit position MSB->LSB, carry <- (-carry)*B, then += digit*P_j via the precomputed
 * multiples.  out_carries (optional) receives carry after every digit position, d x 12 limbs.
 * bad_index (optional) receives the index of the first out-of-range scalar (:97). */
int orc_lhs_msm(int cid, const uint8_t *scalars, const u64 *pts_jac, size_t n, uint8_t base,
                u64 *out_carry, u64 *out_carries, size_t *bad_index) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    if (base < 2) return ORC_BAD_BASE;
    u64 bound[4]; scalar_bound(c, bound);
    unsigned d = logb_ceil4(bound, base) + 1;
    for (size_t i = 0; i < n; i++) {
        u64 x[4]; memcpy(x, scalars + 32 * i, 32);
        if (u256_geq(x, bound)) { if (bad_index) *bad_index = i; return ORC_SCALAR_OUT_OF_RANGE; }
    }
    uint8_t *digits = (uint8_t *)calloc((size_t)d * (n ? n : 1), 1);
    orc_negbase_decompose_batch(scalars, n, base, d, digits);
    jac *pre = (jac *)malloc(sizeof(jac) * (size_t)(base - 1) * (n ? n : 1));
    for (size_t j = 0; j < n; j++) orc_precompute_multiplicities(cid, pts_jac + 12 * j, base, (u64 *)(pre + (size_t)(base - 1) * j));
    jac carry; jac_set_id(&carry);
    u64 bl[4] = {base, 0, 0, 0};
    for (unsigned i = 0; i < d; i++) {
        jac nc; jac_neg(c, &nc, &carry); jac_mul_limbs(c, &carry, &nc, bl);      /* :118 */
        for (size_t j = 0; j < n; j++) {
            uint8_t dg = digits[(size_t)d * j + (d - 1 - i)];                      /* reversed: MSB first (:101) */
            if (dg) jac_add(c, &carry, &carry, &pre[(size_t)(base - 1) * j + (dg - 1)]);   /* :120-125 */
        }
        if (out_carries) memcpy(out_carries + 12 * (size_t)i, &carry, 96);
    }
    memcpy(out_carry, &carry, 96);
    free(pre); free(digits); return ORC_OK;
}

/* ---- best_multiexp restatement (halo2_proofs::arithmetic, see header) ---- */
static unsigned get_at(unsigned segment, unsigned cbits, const uint8_t *bytes32) {
    unsigned skip_bits = segment * cbits, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    u64 v = 0; for (unsigned i = 0; i < 8 && skip_bytes + i < 32; i++) v |= (u64)bytes32[skip_bytes + i] << (8 * i);
    v >>= skip_bits - skip_bytes * 8;
    return (unsigned)(v % (1ULL << cbits));
}
static void multiexp_serial(const curve_t *c, const uint8_t *scalars, const aff *bases, size_t n, jac *acc) {
    unsigned cb = n < 4 ? 1 : (n < 32 ? 3 : (unsigned)ceil(log((double)n)));
    unsigned segments = 256 / cb + 1;
    size_t nb = ((size_t)1 << cb) - 1;
    jac *buckets = (jac *)malloc(sizeof(jac) * nb);
    for (unsigned seg = segments; seg-- > 0;) {
        for (unsigned k = 0; k < cb; k++) jac_double(c, acc, acc);
        memset(buckets, 0, sizeof(jac) * nb);
        for (size_t i = 0; i < n; i++) {
            unsigned k = get_at(seg, cb, scalars + 32 * i);
            if (k) jac_add_aff(c, &buckets[k - 1], &buckets[k - 1], &bases[i]);
        }
        jac running; jac_set_id(&running);
        for (size_t k = nb; k-- > 0;) { jac_add(c, &running, &running, &buckets[k]); jac_add(c, acc, acc, &running); }
    }
    free(buckets);
}
typedef struct { const curve_t *c; const uint8_t *s; const aff *b; size_t n; jac acc; } me_job;
static void *me_thread(void *arg) { me_job *j = (me_job *)arg; jac_set_id(&j->acc); multiexp_serial(j->c, j->s, j->b, j->n, &j->acc); return NULL; }

int orc_best_multiexp(int cid, const uint8_t *scalars, const u64 *bases_aff, size_t n, int threads, u64 *out_jac) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    if (threads < 1) threads = 1;
    jac total; jac_set_id(&total);
    if (n > (size_t)threads) {
        size_t chunk = n / threads, nchunks = (n + chunk - 1) / chunk;
        me_job *jobs = (me_job *)malloc(sizeof(me_job) * nchunks);
        pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * nchunks);
        for (size_t k = 0; k < nchunks; k++) {
            size_t off = k * chunk, len = n - off < chunk ? n - off : chunk;
            jobs[k].c = c; jobs[k].s = scalars + 32 * off; jobs[k].b = (const aff *)bases_aff + off; jobs[k].n = len;
            pthread_create(&tids[k], NULL, me_thread, &jobs[k]);
        }
        for (size_t k = 0; k < nchunks; k++) { pthread_join(tids[k], NULL); jac_add(c, &total, &total, &jobs[k].acc); }
        free(jobs); free(tids);
    } else {
        multiexp_serial(c, scalars, (const aff *)bases_aff, n, &total);
    }
    memcpy(out_jac, &total, 96); return ORC_OK;
}
/* independent cross-check: sum of double-and-add products */
int orc_msm_naive(int cid, const uint8_t *scalars, const u64 *bases_aff, size_t n, u64 *out_jac) {
    const curve_t *c = get_curve(cid); if (!c) return ORC_BAD_CURVE;
    jac total; jac_set_id(&total);
    for (size_t i = 0; i < n; i++) {
        aff a; memcpy(&a, bases_aff + 8 * i, 64); jac p, r; jac_from_aff(c, &p, &a);
        u64 k[4]; memcpy(k, scalars + 32 * i, 32); jac_mul_limbs(c, &r, &p, k); jac_add(c, &total, &total, &r);
    }
    memcpy(out_jac, &total, 96); return ORC_OK;
}

/* compiled, threaded restatement of the divisor-witness path (compute_lhs_witness in full): see the file */
#include "witness_oracle.inc"
