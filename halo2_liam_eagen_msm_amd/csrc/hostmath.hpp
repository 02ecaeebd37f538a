// Host-side field / group arithmetic of the product library (4 x 64-bit Montgomery limbs).
// Used only for the O(windows * bits) tail of an MSM: turning the per-window
// (total, U_0..U_{L-1}) points the GPU produces into window sums and running the final
// Horner recursion -- a few hundred dependent group operations, which are latency-bound
// (~15 us each on one GPU wave, ~0.4 us here).  All O(n) work stays on the GPU.
//
// This is product code and deliberately separate from oracle/ (which it never includes).
#pragma once
#include <stdint.h>
#include <string.h>

namespace lemsm {
namespace host {

typedef uint64_t u64;
typedef unsigned __int128 u128;

struct FqParams64 {
  static constexpr u64 N[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr u64 R[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
  static constexpr u64 R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
  static constexpr u64 NINV = 0x87d20782e4866389ULL;
};
struct FrParams64 {
  static constexpr u64 N[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr u64 R[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
  static constexpr u64 R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
  static constexpr u64 NINV = 0xc2e1f593efffffffULL;
};

struct fe { u64 l[4]; };

template <class P>
struct HF {
  static bool is_zero(const fe& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
  static bool eq(const fe& a, const fe& b) { return memcmp(&a, &b, sizeof(fe)) == 0; }
  static fe zero() { fe r; memset(&r, 0, sizeof r); return r; }
  static fe one() { fe r; for (int i = 0; i < 4; i++) r.l[i] = P::R[i]; return r; }
  static bool geq_n(const u64* t) {
    for (int i = 3; i >= 0; i--) { if (t[i] > P::N[i]) return true; if (t[i] < P::N[i]) return false; }
    return true;
  }
  static void sub_n(u64* t) {
    u64 bw = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - P::N[i] - bw; t[i] = (u64)d; bw = (u64)(d >> 64) & 1; }
  }
  static fe add(const fe& a, const fe& b) {
    fe r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (u64)c; c >>= 64; }
    if (c || geq_n(r.l)) sub_n(r.l);
    return r;
  }
  static fe sub(const fe& a, const fe& b) {
    fe r; u64 bw = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - bw; r.l[i] = (u64)d; bw = (u64)(d >> 64) & 1; }
    if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + P::N[i]; r.l[i] = (u64)c; c >>= 64; } }
    return r;
  }
  static fe neg(const fe& a) { return is_zero(a) ? a : sub(zero(), a); }
  static fe dbl(const fe& a) { return add(a, a); }
  static fe mul(const fe& a, const fe& b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 carry = 0;
      for (int j = 0; j < 4; j++) { u128 cur = (u128)a.l[j] * b.l[i] + t[j] + carry; t[j] = (u64)cur; carry = cur >> 64; }
      u128 cur = (u128)t[4] + carry; t[4] = (u64)cur; t[5] = (u64)(cur >> 64);
      u64 m = t[0] * P::NINV;
      cur = (u128)m * P::N[0] + t[0]; carry = cur >> 64;
      for (int j = 1; j < 4; j++) { cur = (u128)m * P::N[j] + t[j] + carry; t[j - 1] = (u64)cur; carry = cur >> 64; }
      cur = (u128)t[4] + carry; t[3] = (u64)cur; t[4] = t[5] + (u64)(cur >> 64);
    }
    if (t[4] || geq_n(t)) sub_n(t);
    fe r; memcpy(r.l, t, 32); return r;
  }
  static fe sqr(const fe& a) { return mul(a, a); }
  static fe inv(const fe& a) {
    u64 e[4]; u64 bw = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)P::N[i] - (i == 0 ? 2 : 0) - bw; e[i] = (u64)d; bw = (u64)(d >> 64) & 1; }
    fe acc = one(), base = a;
    for (int i = 0; i < 256; i++) { if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base); base = sqr(base); }
    return acc;
  }
};

// XYZZ point on the host; identity: zz == 0.  Same 128-byte layout as the device records.
struct pt { fe x, y, zz, zzz; };

template <class P>
struct HG {
  typedef HF<P> F;
  static pt identity() { pt p; memset(&p, 0, sizeof p); return p; }
  static bool is_identity(const pt& p) { return F::is_zero(p.zz); }
  static pt neg(const pt& p) { pt r = p; r.y = F::neg(p.y); return r; }
  static pt dbl(const pt& p) {
    if (is_identity(p)) return p;
    fe U = F::dbl(p.y), V = F::sqr(U), W = F::mul(U, V), S = F::mul(p.x, V);
    fe t = F::sqr(p.x), M = F::add(F::dbl(t), t);
    pt r;
    r.x = F::sub(F::sub(F::sqr(M), S), S);
    r.y = F::sub(F::mul(M, F::sub(S, r.x)), F::mul(W, p.y));
    r.zz = F::mul(V, p.zz); r.zzz = F::mul(W, p.zzz);
    return r;
  }
  static pt add(const pt& a, const pt& b) {
    if (is_identity(a)) return b;
    if (is_identity(b)) return a;
    fe U1 = F::mul(a.x, b.zz), U2 = F::mul(b.x, a.zz), S1 = F::mul(a.y, b.zzz), S2 = F::mul(b.y, a.zzz);
    fe Pp = F::sub(U2, U1), R = F::sub(S2, S1);
    if (F::is_zero(Pp)) { if (F::is_zero(R)) return dbl(a); return identity(); }
    fe PP = F::sqr(Pp), PPP = F::mul(Pp, PP), Q = F::mul(U1, PP);
    pt r;
    r.x = F::sub(F::sub(F::sub(F::sqr(R), PPP), Q), Q);
    r.y = F::sub(F::mul(R, F::sub(Q, r.x)), F::mul(S1, PPP));
    r.zz = F::mul(F::mul(a.zz, b.zz), PP);
    r.zzz = F::mul(F::mul(a.zzz, b.zzz), PPP);
    return r;
  }
  static pt mul_small(const pt& p, uint32_t k) {
    if (k == 0 || is_identity(p)) return identity();
    int top = 31; while (!((k >> top) & 1)) top--;
    pt acc = p;
    for (int i = top - 1; i >= 0; i--) { acc = dbl(acc); if ((k >> i) & 1) acc = add(acc, p); }
    return acc;
  }
  // Jacobian (X', Y', Z') with x = X'/Z'^2, y = Y'/Z'^3:  Z' = ZZZ, X' = X*ZZ^2, Y' = Y*ZZZ^2
  // (ZZ^3 == ZZZ^2).  No inversion; identity -> (0,0,0).
  static void to_jacobian(const pt& p, u64 out[12]) {
    if (is_identity(p)) { memset(out, 0, 96); return; }
    fe X = F::mul(p.x, F::sqr(p.zz)), Y = F::mul(p.y, F::sqr(p.zzz));
    memcpy(out, X.l, 32); memcpy(out + 4, Y.l, 32); memcpy(out + 8, p.zzz.l, 32);
  }
  static pt from_jacobian(const u64 in[12]) {
    pt p; fe Z; memcpy(p.x.l, in, 32); memcpy(p.y.l, in + 4, 32); memcpy(Z.l, in + 8, 32);
    if (F::is_zero(Z)) return identity();
    p.zz = F::sqr(Z); p.zzz = F::mul(p.zz, Z); return p;
  }
  static void to_affine(const pt& p, u64 out[8]) {
    if (is_identity(p)) { memset(out, 0, 64); return; }
    fe zi3 = F::inv(p.zzz);                          // 1/ZZZ
    fe y = F::mul(p.y, zi3);
    fe zi2 = F::sqr(F::mul(zi3, p.zz));              // (ZZ/ZZZ)^2 = Z^-2 = 1/ZZ
    fe x = F::mul(p.x, zi2);
    memcpy(out, x.l, 32); memcpy(out + 4, y.l, 32);
  }
};

}  // namespace host
}  // namespace lemsm
