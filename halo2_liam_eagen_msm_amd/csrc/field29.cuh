// Lazy 256-bit Montgomery arithmetic for the hot kernels: 9 signed limbs of 29 bits
// (radix 2^29, 261 bits), Montgomery radix R' = 2^261, NO carry handling inside products.
//
// Why (profiles/r01_valu_rates_microbench.txt): on gfx950 v_mad_u64_u32 / v_mad_i64_i32 issue
// at the same rate as a 32-bit add, so cost = VALU instruction count.  With 32-bit limbs every
// product needs mad + addc (the 64-bit column accumulator overflows); with 29-bit limbs a
// column of up to 27 products (< 2^58 each) fits a signed 64-bit accumulator, so a product is
// ONE instruction and there is no conditional subtraction anywhere: ~215 instructions per
// modular multiplication instead of ~370.  The products are generated (tools/gen_field29.py) as
// one inline-asm block per column: left to itself hipcc hoists the products of later columns into
// side accumulators (extra 64-bit adds, spills), and it pads every asm statement with an s_nop.
//
// Domain: kernels using this field work on x * 2^261 mod N.  The C ABI's raw-Montgomery
// format is x * 2^256 (8 x u32).  Either k_convert_points multiplies every point by 2^5 once per
// MSM, or (long buckets / few windows per call) the accumulate kernel consumes the ABI form directly
// by carrying ZZ and ZZZ times 2^5 (XYZZ29::madd_abi).  The host multiplies the few hundred result
// coordinates by 2^-5 (hostmath.hpp).
//
// Invariants ("N" = normalised): limbs 0..7 in [0, 2^29), limb 8 a small signed value; the
// integer value V (any representative of the residue) satisfies |V| < 8N.  Every mont*() output
// is N.  A difference of two N values has |limb| < 2^29 and may be fed to mont*() directly.
// Bound: a column holds <= 18 data products + 9 reduction products, each < 2^58 in magnitude:
// 27 * 2^58 < 2^63.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lemsm {

typedef int32_t i32;
typedef int64_t i64;
typedef uint32_t u32;
typedef uint64_t u64;

struct Fq29Params {   // BN254 base field p
  static constexpr i32 N[9] = {0x187cfd47, 0x010460b6, 0x1c72a34f, 0x02d522d0, 0x1585d978, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};
  static constexpr u32 NINV = 0x04866389u;   // -N^-1 mod 2^29
  static constexpr i32 ONE[9] = {0x157ccc21, 0x141c2758, 0x185230d3, 0x014c0419, 0x0aa36fb9, 0x1d4240ce, 0x11d54c07, 0x052ac7a8, 0x000dc836};    // 2^261 mod N
  static constexpr i32 C266[9] = {0x13349ca1, 0x1a5d84a8, 0x0a3e5cac, 0x100249e0, 0x12b951e8, 0x0e92d304, 0x14cb95b3, 0x041b9d3d, 0x00058003};   // 2^266 mod N
  static constexpr i32 C256[9] = {0x058f0d9d, 0x1aea1c6e, 0x11c2cf74, 0x11d651eb, 0x1462c0a7, 0x11b7bc3c, 0x1cbd99ba, 0x183340fb, 0x000e0a77};   // 2^256 mod N
};
struct Fr29Params {   // BN254 scalar field r
  static constexpr i32 N[9] = {0x10000001, 0x1f0fac9f, 0x0e5c2450, 0x07d090f3, 0x1585d283, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};
  static constexpr u32 NINV = 0x0fffffffu;
  static constexpr i32 ONE[9] = {0x0fffff57, 0x1ea70ab4, 0x052c068b, 0x17504f49, 0x0aa8075b, 0x1d4240ce, 0x11d54c07, 0x052ac7a8, 0x000dc836};
  static constexpr i32 C266[9] = {0x0fffead7, 0x1d5444f4, 0x04438aa5, 0x03b4d096, 0x134c84da, 0x0e92d304, 0x14cb95b3, 0x041b9d3d, 0x00058003};
  static constexpr i32 C256[9] = {0x0ffffffb, 0x04b1a0e2, 0x18334a6b, 0x18ed2b3e, 0x1462e36f, 0x11b7bc3c, 0x1cbd99ba, 0x183340fb, 0x000e0a77};
};

template <class P>
struct Field29 {
  typedef P P_;
  struct fe { i32 l[9]; };
  static constexpr i32 MASK = (1 << 29) - 1;

  static __device__ __forceinline__ void set_zero(fe& r) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
  }
  static __device__ __forceinline__ void set_one(fe& r) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = P::ONE[i];
  }
  // exact test for the all-zero limb pattern (identity marker / zero-initialised memory)
  static __device__ __forceinline__ bool limbs_zero(const fe& a) {
    i32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i];
    return o == 0;
  }
  static __device__ __forceinline__ void sub(fe& r, const fe& a, const fe& b) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] - b.l[i];
  }
  static __device__ __forceinline__ void add(fe& r, const fe& a, const fe& b) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
  }
  static __device__ __forceinline__ void neg(fe& r, const fe& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = -a.l[i];
  }
  static __device__ __forceinline__ void cneg(fe& r, const fe& a, bool flag) {
    i32 s = flag ? -1 : 0;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (a.l[i] ^ s) - s;
  }
  // exact sequential carry propagation: limbs 0..7 into [0, 2^29), limb 8 keeps the sign
  static __device__ __forceinline__ void wnorm(fe& a) {
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      i32 v = a.l[i] + c;
      a.l[i] = v & MASK;
      c = v >> 29;
    }
    a.l[8] += c;
  }

  // Montgomery products, fully unrolled with one asm block per column (tools/gen_field29.py):
  //   mul(r,a,b)            r = a*b / 2^261
  //   sqr(r,a)              r = a*a / 2^261            (36 doubled cross products + 9 squares)
  //   mul_addhi(r,a,b,hi)   r = a*b / 2^261 + hi       (hi: lazy limbs, added on the upper columns)
  //   sqr_addhi(r,a,hi)     r = a*a / 2^261 + hi
  //   mul2(r,a,b,c,d)       r = (a*b + c*d) / 2^261
  // all results exactly normalised.
#include "field29_gen.inc"

  // canonical representative in [0, N) with normalised limbs.  |V| < 8N on entry.
  static __device__ __forceinline__ void canon(fe& a) {
    // + 8N  -> (0, 16N)
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      // 8N limb i = (N << 3) split over limbs; computed on the fly from N's limbs
      i32 n8 = (i32)((((u32)P::N[i] << 3) & (u32)MASK) | (i ? ((u32)P::N[i - 1] >> 26) : 0u));
      i32 v = a.l[i] + n8 + c;
      if (i < 8) { a.l[i] = v & MASK; c = v >> 29; } else a.l[i] = v;
    }
    // conditional subtraction of 8N, 4N, 2N, N
#pragma unroll
    for (int s = 3; s >= 0; s--) {
      fe t; i32 bw = 0;
#pragma unroll
      for (int i = 0; i < 9; i++) {
        i32 ns = (i32)((((u32)P::N[i] << s) & (u32)MASK) | ((i && s) ? ((u32)P::N[i - 1] >> (29 - s)) : 0u));
        i32 v = a.l[i] - ns + bw;
        if (i < 8) { t.l[i] = v & MASK; bw = v >> 29; } else t.l[i] = v;
      }
      bool keep = t.l[8] < 0;
#pragma unroll
      for (int i = 0; i < 9; i++) a.l[i] = keep ? a.l[i] : t.l[i];
    }
  }
  static __device__ __forceinline__ bool is_zero_mod(const fe& a) {
    fe t = a; canon(t); return limbs_zero(t);
  }

  // packed memory format: 8 x u32 = the canonical integer (value < 2^256)
  static __device__ __forceinline__ void unpack(fe& r, const u32 (&w)[8]) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
      u32 lo = w[wi] >> sh;
      u32 hi = (sh + 29 > 32 && wi + 1 < 8) ? (w[wi + 1] << (32 - sh)) : 0u;
      r.l[i] = (i32)((lo | hi) & (u32)MASK);
    }
  }
  static __device__ __forceinline__ void pack(u32 (&w)[8], const fe& a) {   // a canonical
#pragma unroll
    for (int j = 0; j < 8; j++) {
      int bit = 32 * j, li = bit / 29, sh = bit - 29 * li;   // word j starts at limb li, bit sh
      u32 v = (u32)a.l[li] >> sh;
      int got = 29 - sh;
      if (li + 1 < 9) v |= (u32)a.l[li + 1] << got;
      if (got + 29 < 32 && li + 2 < 9) v |= (u32)a.l[li + 2] << (got + 29);
      w[j] = v;
    }
  }
  static __device__ __forceinline__ void load(fe& r, const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unpack(r, w);
  }
  static __device__ __forceinline__ void from_words(fe& r, const uint4& a, const uint4& b) {
    u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unpack(r, w);
  }
  // canonicalises a copy and stores 32 bytes
  static __device__ __forceinline__ void store(void* p, const fe& a) {
    fe t = a; canon(t);
    u32 w[8]; pack(w, t);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
  // x*2^256 (ABI raw Montgomery) -> x*2^261; equally: times 32 inside the 2^261 domain
  static __device__ __forceinline__ void from_abi(fe& r, const fe& a) {
    fe c;
#pragma unroll
    for (int i = 0; i < 9; i++) c.l[i] = P::C266[i];
    mul(r, a, c);
  }
  // divide by 32 inside the 2^261 domain (multiply by the domain image of 2^-5, which is 2^256)
  static __device__ __forceinline__ void div32(fe& r, const fe& a) {
    fe c;
#pragma unroll
    for (int i = 0; i < 9; i++) c.l[i] = P::C256[i];
    mul(r, a, c);
  }
  // r = 32 * a as a lazy value, WITHOUT a Montgomery multiplication: 32a - q N with q estimated
  // from the top limbs, so r is congruent to 32a with |r| < 4N, limbs normalised (~50 instructions
  // instead of ~225).  a: canonical limbs, possibly negated as a whole (|limb| < 2^29).
  // Since 32 = 2^261 / 2^256 this also takes x*2^256 (the C ABI's form) to x*2^261.
  static __device__ __forceinline__ void mul32(fe& r, const fe& a) {
    const float inv = 32.0f / (float)P::N[8];
    i32 q = (i32)((float)a.l[8] * inv);            // |32a/N - q| < 2
    i64 t = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      t += (i64)a.l[i] * 32 - (i64)q * P::N[i];
      if (i < 8) { r.l[i] = (i32)t & MASK; t >>= 29; }
      else r.l[i] = (i32)t;
    }
  }
  // a - q N with q = round-to-zero(a / N) estimated from the top limb: congruent to a, |r| < 2N, limbs normalised (~40
  // instructions).  a: normalised limbs, |a| < 2^261 (the transforms' sums grow by a factor of two per stage; a product
  // with a canonical factor tolerates |a| < 128N: |a w| / 2^261 + N < 2N).
  static __device__ __forceinline__ void reduce_small(fe& r, const fe& a) {
    const float inv = 1.0f / (float)P::N[8];
    i32 q = (i32)((float)a.l[8] * inv);            // |a/N - q| < 1 + 2^-13
    i64 t = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      t += (i64)a.l[i] - (i64)q * P::N[i];
      if (i < 8) { r.l[i] = (i32)t & MASK; t >>= 29; }
      else r.l[i] = (i32)t;
    }
  }
  static __device__ __forceinline__ void set_c266(fe& r) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = P::C266[i];
  }
};

}  // namespace lemsm
