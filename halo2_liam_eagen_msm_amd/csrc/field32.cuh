// 256-bit Montgomery field arithmetic for gfx950, 8 x 32-bit limbs held in VGPRs.
//
// Replaces the third-party halo2curves field types the reference builds on
// (bn256::Fq / bn256::Fr, imported at /root/reference/src/argument_witness_calc.rs:21-23):
// 4 x u64 little-endian limbs, Montgomery R = 2^256 -- the same bytes, read here as
// 8 x u32.  Values are kept fully reduced in [0, N).
//
// gfx950 facts this file is shaped by (profiles/r01_valu_rates_microbench.txt):
// v_mad_u64_u32 issues at the same rate as v_add_u32 (one wave64 instruction per
// ~4.4 cycles per SIMD with >= 2 waves), so the cost model is plain instruction
// count.  hipcc does not fuse "acc += a*b; carry" into mad + addc (it emits a
// 64-bit compare/select sequence, ~5 instructions per product), hence the one
// inline-asm pair below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lemsm {

typedef uint32_t u32;
typedef uint64_t u64;

// BN254 base field p: coordinates of BN254 G1.
struct FqParams {
  static constexpr u32 N[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr u32 R[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr u32 R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
  static constexpr u32 NINV = 0xe4866389u;   // -N^-1 mod 2^32
};
// BN254 scalar field r: coordinates of Grumpkin.
struct FrParams {
  static constexpr u32 N[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr u32 R[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr u32 R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
  static constexpr u32 NINV = 0xefffffffu;
};

// acc(96 bit: hi:lo) += a * b.   One mad (64-bit accumulate, carry to vcc) + one addc.
__device__ __forceinline__ void mac96(u64& lo, u32& hi, u32 a, u32 b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
}

// Same with b a wave-uniform constant (modulus limb) kept in an SGPR.
__device__ __forceinline__ void mac96c(u64& lo, u32& hi, u32 a, u32 b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(lo), "+v"(hi) : "v"(a), "s"(b) : "vcc");
}

template <class P>
struct Field32 {
  struct fe { u32 v[8]; };

  static __device__ __forceinline__ void set_zero(fe& r) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
  }
  static __device__ __forceinline__ void set_one(fe& r) {
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = P::R[i];
  }
  static __device__ __forceinline__ bool is_zero(const fe& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
  }
  static __device__ __forceinline__ bool eq(const fe& a, const fe& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
  }
  // 32-byte aligned raw Montgomery limbs in memory.
  static __device__ __forceinline__ void load(fe& r, const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  }
  // the same from two 16-byte words already in registers (k_accum1 keeps the raw words in flight across a whole
  // mixed addition and unpacks them only when it needs the point)
  static __device__ __forceinline__ void from_words(fe& r, const uint4& a, const uint4& b) {
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  }
  static __device__ __forceinline__ void store(void* p, const fe& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
  }

  static __device__ __forceinline__ void add(fe& r, const fe& a, const fe& b) {
    u32 t[8], s[8], c = 0, bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = __builtin_subc(t[i], P::N[i], bw, &bw);
    // N < 2^254 so a+b never carries out of 256 bits; keep s when t >= N (no borrow)
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = bw ? t[i] : s[i];
  }
  static __device__ __forceinline__ void sub(fe& r, const fe& a, const fe& b) {
    u32 t[8], bw = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = __builtin_subc(a.v[i], b.v[i], bw, &bw);
    u32 mask = 0u - bw;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_addc(t[i], P::N[i] & mask, c, &c);
  }
  static __device__ __forceinline__ void dbl(fe& r, const fe& a) { add(r, a, a); }
  static __device__ __forceinline__ void neg(fe& r, const fe& a) {
    u32 bw = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) nz |= a.v[i];
    u32 mask = nz ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = __builtin_subc(P::N[i], a.v[i], bw, &bw) & mask;
  }
  // r = flag ? -a : a
  static __device__ __forceinline__ void cneg(fe& r, const fe& a, bool flag) {
    fe n; neg(n, a);
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = flag ? n.v[i] : a.v[i];
  }

  // Montgomery product a*b*2^-256 mod N, product scanning with a 96-bit column accumulator.
  static __device__ __forceinline__ void mul(fe& r, const fe& a, const fe& b) {
    u32 m[8];
    u64 lo = 0; u32 hi = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) mac96(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
      for (int i = 0; i < k; i++) mac96c(lo, hi, m[i], P::N[k - i]);
      m[k] = (u32)lo * P::NINV;
      mac96c(lo, hi, m[k], P::N[0]);
      lo = (lo >> 32) | ((u64)hi << 32); hi = 0;
    }
    u32 t[8];
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96c(lo, hi, m[i], P::N[k - i]);
      t[k - 8] = (u32)lo;
      lo = (lo >> 32) | ((u64)hi << 32); hi = 0;
    }
    // a,b < N < 2^254  =>  result < 2N < 2^255: no 257th bit, one conditional subtract
    u32 s[8], bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = __builtin_subc(t[i], P::N[i], bw, &bw);
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = bw ? t[i] : s[i];
  }
  static __device__ __forceinline__ void sqr(fe& r, const fe& a) { mul(r, a, a); }

  // a^-1 via Fermat (used only in batch normalisation, one per thread)
  static __device__ void inv(fe& r, const fe& a) {
    // exponent N-2
    u32 e[8];
    u32 bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = __builtin_subc(P::N[i], i == 0 ? 2u : 0u, bw, &bw);
    fe acc; set_one(acc);
    fe base = a;
#pragma unroll
    for (int w = 0; w < 8; w++) {
      u32 bits = e[w];
      for (int j = 0; j < 32; j++) {
        if (bits & 1u) mul(acc, acc, base);
        sqr(base, base);
        bits >>= 1;
      }
    }
    r = acc;
  }
};

}  // namespace lemsm
