// XYZZ group law over the lazy radix-2^29 field (field29.cuh): the arithmetic of the hot
// kernels (k_accum1, k_segreduce, k_pyramid).  Same formulas and the same complete case
// handling as xyzz.cuh; what differs is how the field operations are composed so that no
// operand of a product ever exceeds the limb bound of field29.cuh:
//   * differences of two normalised values (|limb| < 2^29) feed products directly,
//   * X3 = R^2 - PPP - 2Q is ONE Montgomery reduction: (2N - PPP - 2Q) rides in on the upper
//     columns of R*R, so X3 comes out normalised,
//   * Y3 = R*(Q - X3) - Y1*PPP is ONE reduction of a two-product column sum.
// A mixed addition is 7 multiplications + 2 squarings worth of products and 9 reductions
// (~2,000 VALU instructions).
//
// Points in memory: 160 bytes of raw limbs (see load/store below); identity = all zero.
#pragma once
#include "field29.cuh"

namespace lemsm {

template <class F>
struct XYZZ29 {
  typedef F F_;
  typedef typename F::fe fe;
  static constexpr bool CONVERTED_DOMAIN = true;   // coordinates are x*2^261, see field29.cuh
  // (0,0) is the identity.  Both curves have prime order, so no valid point has y == 0: testing y
  // alone is exact for valid input and halves the per-entry cost of the test in k_accum1.
  static __device__ __forceinline__ bool aff_is_identity(const fe&, const fe& y) { return F::limbs_zero(y); }
  struct pt { fe x, y, zz, zzz; };

  static __device__ __forceinline__ void set_identity(pt& p) {
    F::set_zero(p.x); F::set_zero(p.y); F::set_zero(p.zz); F::set_zero(p.zzz);
  }
  // zz of a valid non-identity point is never == 0 mod N, and the identity is always written
  // as literal zeros, so the limb pattern test is exact.
  static __device__ __forceinline__ bool is_identity(const pt& p) { return F::limbs_zero(p.zz); }

  // hi = 2N - ppp - 2q  (lazy limbs; only ever added into the upper columns of a product)
  static __device__ __forceinline__ void hi_term(fe& t, const fe& ppp, const fe& q) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      i32 n2 = (i32)((((u32)F::P_::N[i] << 1) & (u32)F::MASK) | (i ? ((u32)F::P_::N[i - 1] >> 28) : 0u));
      t.l[i] = n2 - ppp.l[i] - 2 * q.l[i];
    }
  }

  // P == 0 (mod N)  <=>  PP = P^2 * 2^-261 is 0 or N as an integer (PP in [0, 1.2N)).
  static __device__ __forceinline__ bool pp_is_zero(const fe& pp) {
    if (pp.l[0] != 0 && pp.l[0] != F::P_::N[0]) return false;
    i32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { z |= pp.l[i]; e |= pp.l[i] ^ F::P_::N[i]; }
    return z == 0 || e == 0;
  }

  // r = 2 * (X, Y, ZZ, ZZZ) (dbl-2008-s-1, a = 0); for an affine input pass zz = zzz = one and
  // AFFINE = true.  Rare path (equal points meet in a bucket): generous normalisation, out of line.
  template <bool AFFINE>
  static __device__ __noinline__ void dbl_impl(pt& r, const pt& p) {
    fe U, V, W, S, M, t, nW, x3;
    F::add(U, p.y, p.y); F::wnorm(U);                 // U = 2Y
    F::sqr(V, U);                                     // V = U^2
    F::mul(W, U, V);                                  // W = U*V
    F::mul(S, p.x, V);                                // S = X*V
    F::sqr(t, p.x); F::add(M, t, t); F::add(M, M, t); F::wnorm(M);   // M = 3X^2
    // X3 = M^2 - 2S :  hi = 2N - 0 - 2S
    fe zero; F::set_zero(zero);
    hi_term(t, zero, S);
    F::sqr_addhi(x3, M, t);
    // Y3 = M*(S - X3) - W*Y
    F::sub(t, S, x3); F::neg(nW, W);
    fe y = p.y; F::wnorm(y);
    F::mul2(r.y, M, t, nW, y);
    r.x = x3;
    if (AFFINE) { r.zz = V; r.zzz = W; }
    else { F::mul(r.zz, V, p.zz); F::mul(r.zzz, W, p.zzz); }
  }

  // acc += (x2, y2): affine, non-identity, 2^261-domain normalised coordinates; y2 may carry
  // a negation (|limb| < 2^29).
  // `empty` (k_accum1 only): the caller's record of acc == identity, so that the hot loop tests one
  // flag instead of nine limbs; madd keeps it current (set on cancellation, cleared otherwise).
  static __device__ __forceinline__ void madd(pt& acc, const fe& x2, const fe& y2, bool& empty) {
    if (empty) {
      acc.x = x2; acc.y = y2; F::wnorm(acc.y); F::set_one(acc.zz); F::set_one(acc.zzz);
      empty = false;
      return;
    }
    madd_nonempty(acc, x2, y2, empty);
  }
  static __device__ __forceinline__ void madd(pt& acc, const fe& x2, const fe& y2) {
    bool empty = is_identity(acc);
    madd(acc, x2, y2, empty);
  }
  static __device__ __forceinline__ void madd_nonempty(pt& acc, const fe& x2, const fe& y2, bool& empty) {
    fe U2, S2, P, R, PP;
    F::mul(U2, x2, acc.zz);
    F::mul(S2, y2, acc.zzz);
    F::sub(P, U2, acc.x);
    F::sub(R, S2, acc.y);
    F::sqr(PP, P);
    if (pp_is_zero(PP)) {                 // same x: doubling or cancellation
      if (F::is_zero_mod(R)) {
        // res/a are the only objects whose address escapes to the out-of-line call: acc itself must
        // stay in registers (an address-taken acc lives in scratch and is re-loaded every iteration:
        // measured 90 GB of scratch traffic per 2^24-point MSM)
        pt a, res; a.x = x2; a.y = y2; F::set_one(a.zz); F::set_one(a.zzz);
        dbl_impl<true>(res, a);
        acc = res;
      } else { set_identity(acc); empty = true; }
      return;
    }
    fe PPP, Q, t, nY;
    F::mul(PPP, P, PP);
    F::mul(Q, acc.x, PP);
    hi_term(t, PPP, Q);
    F::neg(nY, acc.y);
    F::sqr_addhi(acc.x, R, t);                     // X3 = R^2 - PPP - 2Q
    F::sub(t, Q, acc.x);
    F::mul2(acc.y, R, t, nY, PPP);                 // Y3 = R(Q - X3) - Y1*PPP
    F::mul(acc.zz, acc.zz, PP);
    F::mul(acc.zzz, acc.zzz, PPP);
  }

  // k_accum1's second form of madd: the incoming point is in the C ABI's domain (x2a = x*2^256,
  // canonical limbs; y2a possibly negated) and the accumulator is (X, Y, 32*ZZ, 32*ZZZ) in the 2^261
  // domain.  ZZ and ZZZ only ever multiply the incoming coordinates (U2 = x2*ZZ, S2 = y2*ZZZ) and
  // their own updates (ZZ*PP, ZZZ*PPP), so the factor 32 = 2^261/2^256 rides along for free and the
  // per-MSM conversion pass over all points (0.43 ms, 2 GB of HBM traffic at 2^24 -- replicated on
  // every rank of a window-sharded run) is not needed.  What it costs: opening a segment converts
  // its first point (2 x mul32, ~100 instructions) inside a divergent branch, which a wave pays
  // whenever ANY lane opens a segment -- so this form is chosen when segments are long or the call
  // covers few windows (run_group), and the plain form + conversion pass otherwise.
  static __device__ __forceinline__ void madd_abi(pt& acc, const fe& x2a, const fe& y2a, bool& empty) {
    if (empty) {
      F::mul32(acc.x, x2a); F::mul32(acc.y, y2a); F::set_c266(acc.zz); F::set_c266(acc.zzz);
      empty = false;
      return;
    }
    fe U2, S2, P, R, PP;
    F::mul(U2, x2a, acc.zz);
    F::mul(S2, y2a, acc.zzz);
    F::sub(P, U2, acc.x);
    F::sub(R, S2, acc.y);
    F::sqr(PP, P);
    if (pp_is_zero(PP)) {                 // same x: doubling or cancellation
      if (F::is_zero_mod(R)) {
        // (only a and res have their address taken, see madd)
        pt a, res; F::from_abi(a.x, x2a); F::from_abi(a.y, y2a); F::set_one(a.zz); F::set_one(a.zzz);
        dbl_impl<true>(res, a);
        F::from_abi(res.zz, res.zz); F::from_abi(res.zzz, res.zzz);   // times 32
        acc = res;
      } else { set_identity(acc); empty = true; }
      return;
    }
    fe PPP, Q, t, nY;
    F::mul(PPP, P, PP);
    F::mul(Q, acc.x, PP);
    hi_term(t, PPP, Q);
    F::neg(nY, acc.y);
    F::sqr_addhi(acc.x, R, t);                     // X3 = R^2 - PPP - 2Q
    F::sub(t, Q, acc.x);
    F::mul2(acc.y, R, t, nY, PPP);                 // Y3 = R(Q - X3) - Y1*PPP
    F::mul(acc.zz, acc.zz, PP);
    F::mul(acc.zzz, acc.zzz, PPP);
  }
  // (X, Y, 32 ZZ, 32 ZZZ) <-> plain XYZZ.  In the ABI form k_accum1 writes what it holds (its flush is
  // a divergent branch too) and the consumers convert on load: edge records in k_segreduce, bucket
  // sums in the first pyramid step; bucket_sum[] then holds the scaled form throughout.
  static __device__ __forceinline__ void unscale(pt& p) {
    if (is_identity(p)) return;
    F::div32(p.zz, p.zz); F::div32(p.zzz, p.zzz);
  }
  static __device__ __forceinline__ void scale(pt& p) {
    if (is_identity(p)) return;
    F::from_abi(p.zz, p.zz); F::from_abi(p.zzz, p.zzz);
  }

  // acc += q (both XYZZ, any inputs)
  static __device__ __forceinline__ void add(pt& acc, const pt& q) {
    if (is_identity(q)) return;
    if (is_identity(acc)) { acc = q; return; }
    fe U1, U2, S1, S2, P, R, PP;
    F::mul(U1, acc.x, q.zz);
    F::mul(U2, q.x, acc.zz);
    F::mul(S1, acc.y, q.zzz);
    F::mul(S2, q.y, acc.zzz);
    F::sub(P, U2, U1);
    F::sub(R, S2, S1);
    F::sqr(PP, P);
    if (pp_is_zero(PP)) {
      if (F::is_zero_mod(R)) { pt t = acc, res; dbl_impl<false>(res, t); acc = res; }
      else set_identity(acc);
      return;
    }
    fe PPP, Q, t, nS;
    F::mul(PPP, P, PP);
    F::mul(Q, U1, PP);
    hi_term(t, PPP, Q);
    F::neg(nS, S1);
    F::sqr_addhi(acc.x, R, t);
    F::sub(t, Q, acc.x);
    F::mul2(acc.y, R, t, nS, PPP);
    F::mul(t, acc.zz, q.zz); F::mul(acc.zz, t, PP);
    F::mul(t, acc.zzz, q.zzz); F::mul(acc.zzz, t, PPP);
  }

  // ---- four lanes per addition (narrow pyramid steps: one dependent addition deep, the chip nearly empty) -------------------
  // The 14 products of the XYZZ addition have a critical path of four: lane q of a quad computes one product per stage and
  // DPP quad permutes carry the few values another lane needs (9 moves each).  ~1150 instructions deep instead of ~3300:
  //   stage 1   q0: U1 = X1 ZZ2    q1: U2 = X2 ZZ1     q2: S1 = Y1 ZZZ2     q3: S2 = Y2 ZZZ1     -> P = U2 - U1 (q0, q1), R = S2 - S1 (q2, q3)
  //   stage 2   q0: PP = P^2       q1: ZZ1 ZZ2         q2: RR = R^2         q3: ZZZ1 ZZZ2
  //   stage 3   q0: PPP = P PP     q1: ZZ3 = (..) PP   q2: Q = U1 PP                             -> q2: X3 = RR - PPP - 2Q
  //   stage 4   q0: S1 PPP                             q2: R (Q - X3)       q3: ZZZ3 = (..) PPP  -> q2: Y3 = R (Q - X3) - S1 PPP
  // Operands come straight from memory with lane-dependent addresses (no selects for stages 1 and 2); results are stored by the
  // lane that holds them.  Returns false -- for the whole quad -- when the pair needs the complete addition (an identity operand,
  // equal or opposite points): the caller lets one lane run add().
  template <int CTRL>
  static __device__ __forceinline__ void quad_move(fe& r, const fe& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (i32)__builtin_amdgcn_update_dpp(0, (int)a.l[i], CTRL, 0xf, 0xf, false);
  }
  static __device__ __forceinline__ void ld_elem(fe& r, const char* p, u32 k) {
    const i32* e = reinterpret_cast<const i32*>(p + 36u * k);
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = e[i];
  }
  static __device__ __forceinline__ void st_elem(char* p, u32 k, const fe& a) {
    i32* e = reinterpret_cast<i32*>(p + 36u * k);
#pragma unroll
    for (int i = 0; i < 9; i++) e[i] = a.l[i];
  }
  static __device__ __forceinline__ void sel(fe& r, bool c, const fe& a, const fe& b) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
  }
  static __device__ __forceinline__ bool add4_mem(const char* pa, const char* pb, char* dst, u32 q) {
    const bool odd = (q & 1u) != 0, hi = q >= 2u;
    fe A, B, C, E, r1, x1, D;
    ld_elem(A, odd ? pb : pa, hi ? 1u : 0u);            // q0: X1, q1: X2, q2: Y1, q3: Y2
    ld_elem(B, odd ? pa : pb, hi ? 3u : 2u);            // q0: ZZ2, q1: ZZ1, q2: ZZZ2, q3: ZZZ1
    ld_elem(C, pa, hi ? 3u : 2u);                       // ZZ1 | ZZZ1
    ld_elem(E, pb, hi ? 3u : 2u);                       // ZZ2 | ZZZ2
    int sp = (F::limbs_zero(C) || F::limbs_zero(E)) ? 1 : 0;   // the identity is all zeros
    F::mul(r1, A, B);
    quad_move<0xB1>(x1, r1);                            // the neighbour's product
    F::sub(D, x1, r1); F::cneg(D, D, odd);              // q0, q1: P = U2 - U1;  q2, q3: R = S2 - S1
    fe A2, B2, r2;
    sel(A2, odd, C, D); sel(B2, odd, E, D);
    F::mul(r2, A2, B2);                                 // q0: PP, q1: ZZ1 ZZ2, q2: RR, q3: ZZZ1 ZZZ2
    const int pz = pp_is_zero(r2) ? 1 : 0;
    sp |= __builtin_amdgcn_update_dpp(0, pz, 0x00, 0xf, 0xf, false);      // q0's verdict on P
    sp |= __builtin_amdgcn_update_dpp(0, sp, 0xB1, 0xf, 0xf, false);
    sp |= __builtin_amdgcn_update_dpp(0, sp, 0x4E, 0xf, 0xf, false);      // the same answer in all four lanes
    if (sp) return false;
    fe bPP, bU1, A3, r3;
    quad_move<0x00>(bPP, r2); quad_move<0x00>(bU1, r1);
    sel(A3, q == 0u, D, r2); sel(A3, q == 2u, bU1, A3);
    F::mul(r3, A3, bPP);                                // q0: PPP, q1: ZZ3, q2: Q
    fe bPPP, T, X3, V, bS1, A4, B4, r4, bY, Y3;
    quad_move<0x00>(bPPP, r3);
#pragma unroll
    for (int i = 0; i < 9; i++) T.l[i] = r2.l[i] - bPPP.l[i] - 2 * r3.l[i];
    F::wnorm(T); F::reduce_small(X3, T);                // q2: X3 = RR - PPP - 2Q, |X3| < 2N
    F::sub(V, r3, X3);                                  // q2: Q - X3
    quad_move<0xAA>(bS1, r1);                           // S1
    sel(A4, q == 0u, bS1, r2); sel(A4, q == 2u, D, A4);
    sel(B4, q == 2u, V, bPPP);
    F::mul(r4, A4, B4);                                 // q0: S1 PPP, q2: R (Q - X3), q3: ZZZ3
    quad_move<0x00>(bY, r4);
    F::sub(Y3, r4, bY); F::wnorm(Y3);                   // q2: Y3
    if (q == 1u) st_elem(dst, 2u, r3);
    if (q == 3u) st_elem(dst, 3u, r4);
    if (q == 2u) { st_elem(dst, 0u, X3); st_elem(dst, 1u, Y3); }
    return true;
  }

  // One step of the wave-level inclusive scan over points (GFX9 DPP, the sequence the compiler uses for wave
  // reductions): steps 0..3 fetch the lane 1, 2, 4, 8 places below inside the 16-lane row, step 4 gives rows 1 and 3
  // the last lane of the row below (row_bcast:15), step 5 gives rows 2 and 3 lane 31 (row_bcast:31); a lane without a
  // source receives the identity (all-zero limbs).  DPP moves are plain VALU instructions: a 36-limb fetch costs a
  // few hundred cycles where 36 ds_bpermute cost ~10 000 (tools/ubench/add_latency.hip), and in a scan every lane keeps
  // adding real data, which matters: an addition only a few lanes take runs 1.7-3x slower than a full-wave one.
  template <int CTRL, int ROW_MASK>
  static __device__ __forceinline__ void dpp_move(pt& r, const pt& p) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      r.x.l[i] = (i32)__builtin_amdgcn_update_dpp(0, (int)p.x.l[i], CTRL, ROW_MASK, 0xf, false);
      r.y.l[i] = (i32)__builtin_amdgcn_update_dpp(0, (int)p.y.l[i], CTRL, ROW_MASK, 0xf, false);
      r.zz.l[i] = (i32)__builtin_amdgcn_update_dpp(0, (int)p.zz.l[i], CTRL, ROW_MASK, 0xf, false);
      r.zzz.l[i] = (i32)__builtin_amdgcn_update_dpp(0, (int)p.zzz.l[i], CTRL, ROW_MASK, 0xf, false);
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

  // Memory format of a point of this arithmetic: the 36 raw limbs (x, y, zz, zzz; 9 x i32 each)
  // + 4 pad words = 160 bytes, NOT canonicalised: storing is ten 16-byte stores.  (Canonical
  // packing costs ~900 instructions per point and the flush branch of k_accum1 is taken by some
  // lane in a third of all iterations.)  The host reduces the few hundred result points
  // (lemsm.hip: from_device_records).  Identity = all zero.
  static constexpr u32 PT_BYTES = 160;
  static __device__ __forceinline__ void load(pt& p, const void* mem) {
    const uint4* q = reinterpret_cast<const uint4*>(mem);
    u32 w[40];
#pragma unroll
    for (int i = 0; i < 10; i++) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#pragma unroll
    for (int i = 0; i < 9; i++) { p.x.l[i] = (i32)w[i]; p.y.l[i] = (i32)w[9 + i]; p.zz.l[i] = (i32)w[18 + i]; p.zzz.l[i] = (i32)w[27 + i]; }
  }
  // the same in two halves: request the 16-byte words now, turn them into limbs later (prefetch across an addition)
  static constexpr int RAW_WORDS = 9;
  static __device__ __forceinline__ void load_raw(uint4 (&r)[RAW_WORDS], const void* mem) {
    const uint4* q = reinterpret_cast<const uint4*>(mem);
#pragma unroll
    for (int i = 0; i < RAW_WORDS; i++) r[i] = q[i];
  }
  static __device__ __forceinline__ void from_raw(pt& p, const uint4 (&r)[RAW_WORDS]) {
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) { w[4 * i] = r[i].x; w[4 * i + 1] = r[i].y; w[4 * i + 2] = r[i].z; w[4 * i + 3] = r[i].w; }
#pragma unroll
    for (int i = 0; i < 9; i++) { p.x.l[i] = (i32)w[i]; p.y.l[i] = (i32)w[9 + i]; p.zz.l[i] = (i32)w[18 + i]; p.zzz.l[i] = (i32)w[27 + i]; }
  }
  static __device__ __forceinline__ void store(void* mem, const pt& p) {
    uint4* q = reinterpret_cast<uint4*>(mem);
    u32 w[40];
#pragma unroll
    for (int i = 0; i < 9; i++) { w[i] = (u32)p.x.l[i]; w[9 + i] = (u32)p.y.l[i]; w[18 + i] = (u32)p.zz.l[i]; w[27 + i] = (u32)p.zzz.l[i]; }
    w[36] = w[37] = w[38] = w[39] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
  }
};

}  // namespace lemsm
