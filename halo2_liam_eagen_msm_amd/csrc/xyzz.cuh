// Short-Weierstrass (a = 0) group law in extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), complete for the cases the reference's own
// tests drive: equal inputs (P+P), opposite inputs (P + -P) and the identity
// (SURVEY.md §4 "critical quirk": lhs_test feeds 10 000 copies of one point, so every
// bucket accumulation starts with a doubling; /root/reference/src/argument_witness_calc.rs:141-142).
//
// Replaces the Jacobian `+`/`double` of halo2curves' G1 types that
// compute_lhs_witness / best_multiexp lean on
// (/root/reference/src/argument_witness_calc.rs:43-51,118-125).  Only the group
// element matters for parity, not the coordinate system.
//
// Identity is encoded ZZ == 0 (all limbs zero).  Affine inputs encode the
// identity as (0,0) like the C ABI (include/lemsm.h).
#pragma once
#include "field32.cuh"

namespace lemsm {

template <class F>
struct XYZZ {
  typedef F F_;
  typedef typename F::fe fe;
  static constexpr bool CONVERTED_DOMAIN = false;
  static __device__ __forceinline__ bool aff_is_identity(const fe& x, const fe& y) { return F::is_zero(x) && F::is_zero(y); }
  struct pt { fe x, y, zz, zzz; };
  struct aff { fe x, y; };

  static __device__ __forceinline__ void set_identity(pt& p) {
    F::set_zero(p.x); F::set_zero(p.y); F::set_zero(p.zz); F::set_zero(p.zzz);
  }
  static __device__ __forceinline__ bool is_identity(const pt& p) { return F::is_zero(p.zz); }
  static __device__ __forceinline__ bool aff_is_identity(const aff& a) {
    return F::is_zero(a.x) && F::is_zero(a.y);
  }
  static __device__ __forceinline__ void from_affine(pt& r, const aff& a) {
    if (aff_is_identity(a)) { set_identity(r); return; }
    r.x = a.x; r.y = a.y; F::set_one(r.zz); F::set_one(r.zzz);
  }

  // r = 2*(x,y) for an affine, non-identity point with y != 0 (mdbl-2008-s-1, a = 0).
  static __device__ __noinline__ void dbl_affine(pt& r, const fe& x, const fe& y) {
    fe U, V, W, S, M, t;
    F::dbl(U, y);                 // U = 2y
    F::sqr(V, U);                 // V = U^2
    F::mul(W, U, V);              // W = U*V
    F::mul(S, x, V);              // S = x*V
    F::sqr(t, x); F::dbl(M, t); F::add(M, M, t);   // M = 3x^2
    F::sqr(r.x, M); F::sub(r.x, r.x, S); F::sub(r.x, r.x, S);   // X3 = M^2 - 2S
    F::sub(t, S, r.x); F::mul(t, M, t); F::mul(U, W, y); F::sub(r.y, t, U);   // Y3 = M(S-X3) - W*y
    r.zz = V; r.zzz = W;
  }
  // r = 2*p for XYZZ p, non-identity (dbl-2008-s-1, a = 0).  y == 0 cannot occur on
  // prime-order curves; if it did the formula yields ZZ3 = 0 = identity, which is correct.
  static __device__ __noinline__ void dbl(pt& r, const pt& p) {
    fe U, V, W, S, M, t, x3;
    F::dbl(U, p.y);
    F::sqr(V, U);
    F::mul(W, U, V);
    F::mul(S, p.x, V);
    F::sqr(t, p.x); F::dbl(M, t); F::add(M, M, t);
    F::sqr(x3, M); F::sub(x3, x3, S); F::sub(x3, x3, S);
    F::sub(t, S, x3); F::mul(t, M, t); F::mul(U, W, p.y);
    F::sub(r.y, t, U);
    r.x = x3;
    F::mul(r.zz, V, p.zz);
    F::mul(r.zzz, W, p.zzz);
  }

  // interface shared with XYZZ29 (k_accum1): this field already works in the ABI's domain
  static __device__ __forceinline__ void madd_abi(pt& acc, const fe& x2, const fe& y2, bool& empty) { madd(acc, x2, y2, empty); }
  static __device__ __forceinline__ void madd(pt& acc, const fe& x2, const fe& y2, bool& empty) { madd(acc, x2, y2); empty = is_identity(acc); }
  static __device__ __forceinline__ void unscale(pt&) {}
  static __device__ __forceinline__ void scale(pt&) {}

  // acc += (x2,y2) affine, non-identity input; acc may be anything (madd-2008-s + special cases).
  static __device__ __forceinline__ void madd(pt& acc, const fe& x2, const fe& y2) {
    if (is_identity(acc)) {
      acc.x = x2; acc.y = y2; F::set_one(acc.zz); F::set_one(acc.zzz);
      return;
    }
    fe U2, S2, P, R;
    F::mul(U2, x2, acc.zz);
    F::mul(S2, y2, acc.zzz);
    F::sub(P, U2, acc.x);
    F::sub(R, S2, acc.y);
    if (F::is_zero(P)) {                 // same x: doubling or cancellation (rare; kept out of line)
      if (F::is_zero(R)) { pt res; fe xx = x2, yy = y2; dbl_affine(res, xx, yy); acc = res; }   // acc's address must not escape
      else set_identity(acc);
      return;
    }
    fe PP, PPP, Q, t;
    F::sqr(PP, P);
    F::mul(PPP, P, PP);
    F::mul(Q, acc.x, PP);
    F::sqr(t, R); F::sub(t, t, PPP); F::sub(t, t, Q); F::sub(t, t, Q);   // X3
    F::mul(U2, acc.y, PPP);              // Y1*PPP (reuse U2)
    acc.x = t;
    F::sub(t, Q, t); F::mul(t, R, t); F::sub(acc.y, t, U2);             // Y3
    F::mul(acc.zz, acc.zz, PP);
    F::mul(acc.zzz, acc.zzz, PPP);
  }

  // acc += q, both XYZZ, any inputs (add-2008-s + special cases).
  static __device__ __forceinline__ void add(pt& acc, const pt& q) {
    if (is_identity(q)) return;
    if (is_identity(acc)) { acc = q; return; }
    fe U1, U2, S1, S2, P, R;
    F::mul(U1, acc.x, q.zz);
    F::mul(U2, q.x, acc.zz);
    F::mul(S1, acc.y, q.zzz);
    F::mul(S2, q.y, acc.zzz);
    F::sub(P, U2, U1);
    F::sub(R, S2, S1);
    if (F::is_zero(P)) {
      if (F::is_zero(R)) { pt t = acc, res; dbl(res, t); acc = res; }
      else set_identity(acc);
      return;
    }
    fe PP, PPP, Q, t;
    F::sqr(PP, P);
    F::mul(PPP, P, PP);
    F::mul(Q, U1, PP);
    F::sqr(t, R); F::sub(t, t, PPP); F::sub(t, t, Q); F::sub(t, t, Q);
    acc.x = t;
    F::sub(t, Q, t); F::mul(t, R, t); F::mul(S1, S1, PPP); F::sub(acc.y, t, S1);
    F::mul(t, acc.zz, q.zz); F::mul(acc.zz, t, PP);
    F::mul(t, acc.zzz, q.zzz); F::mul(acc.zzz, t, PPP);
  }

  static __device__ __forceinline__ void neg(pt& p) { F::neg(p.y, p.y); }

  // One step of the wave-level inclusive scan over points (GFX9 DPP, the sequence the compiler uses for wave
  // reductions): steps 0..3 fetch the lane 1, 2, 4, 8 places below inside the 16-lane row, step 4 gives rows 1 and 3
  // the last lane of the row below (row_bcast:15), step 5 gives rows 2 and 3 lane 31 (row_bcast:31); a lane without a
  // source receives the identity (all-zero limbs).  DPP moves are plain VALU instructions: a 36-limb fetch costs a
  // few hundred cycles where 36 ds_bpermute cost ~10 000 (tools/ubench/add_latency.hip), and in a scan every lane keeps
  // adding real data, which matters: an addition only a few lanes take runs 1.7-3x slower than a full-wave one.
  template <int CTRL, int ROW_MASK>
  static __device__ __forceinline__ void dpp_move(pt& r, const pt& p) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      r.x.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)p.x.v[i], CTRL, ROW_MASK, 0xf, false);
      r.y.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)p.y.v[i], CTRL, ROW_MASK, 0xf, false);
      r.zz.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)p.zz.v[i], CTRL, ROW_MASK, 0xf, false);
      r.zzz.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)p.zzz.v[i], CTRL, ROW_MASK, 0xf, false);
    }
  }
  static __device__ __forceinline__ void scan_fetch(pt& r, const pt& p, int step) {
    switch (step) {
      case 0: dpp_move<0x111, 0xf>(r, p); break;   // row_shr:1
      case 1: dpp_move<0x112, 0xf>(r, p); break;   // row_shr:2
      case 2: dpp_move<0x114, 0xf>(r, p); break;   // row_shr:4
      case 3: dpp_move<0x118, 0xf>(r, p); break;   // row_shr:8
      case 4: dpp_move<0x142, 0xa>(r, p); break;   // row_bcast:15 into rows 1, 3
      default: dpp_move<0x143, 0xc>(r, p); break;  // row_bcast:31 into rows 2, 3
    }
  }

  // memory format of an XYZZ point: 4 x 32 bytes (x, y, zz, zzz), raw Montgomery limbs
  static constexpr u32 PT_BYTES = 128;
  static __device__ __forceinline__ void load(pt& p, const void* mem) {
    const char* m = reinterpret_cast<const char*>(mem);
    F::load(p.x, m); F::load(p.y, m + 32); F::load(p.zz, m + 64); F::load(p.zzz, m + 96);
  }
  static constexpr int RAW_WORDS = 8;
  static __device__ __forceinline__ void load_raw(uint4 (&r)[RAW_WORDS], const void* mem) {
    const uint4* q = reinterpret_cast<const uint4*>(mem);
#pragma unroll
    for (int i = 0; i < RAW_WORDS; i++) r[i] = q[i];
  }
  static __device__ __forceinline__ void from_raw(pt& p, const uint4 (&r)[RAW_WORDS]) {
    F::from_words(p.x, r[0], r[1]); F::from_words(p.y, r[2], r[3]); F::from_words(p.zz, r[4], r[5]); F::from_words(p.zzz, r[6], r[7]);
  }
  static __device__ __forceinline__ void store(void* mem, const pt& p) {
    char* m = reinterpret_cast<char*>(mem);
    F::store(m, p.x); F::store(m + 32, p.y); F::store(m + 64, p.zz); F::store(m + 96, p.zzz);
  }
};

}  // namespace lemsm
