// a^-1 for the strict 8x32-bit field (Montgomery form in, Montgomery form out) by Fermat in the lazy 29-bit field:
// 254 squarings of ~190 instructions and ~130 products of ~225 (~77 000 instructions) instead of field32's own inv
// (381 products of ~370: ~141 000).  One inversion's latency is the floor of every batched-inversion kernel here.
#pragma once
#include "field32.cuh"
#include "field29.cuh"

namespace lemsm {

template <class P32, class P29>
__device__ __noinline__ void inv_lazy(typename Field32<P32>::fe& r, const typename Field32<P32>::fe& a) {
  typedef Field29<P29> F29;
  typename F29::fe x, acc;
  F29::unpack(x, a.v); F29::from_abi(x, x);            // a 2^256 -> a 2^261 (the lazy field's domain)
  F29::set_one(acc);
  u32 e[8]; u32 bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) e[i] = __builtin_subc(P32::N[i], i == 0 ? 2u : 0u, bw, &bw);   // N - 2
  for (int w = 0; w < 8; w++) {
    u32 bits = e[w];
    for (int j = 0; j < 32; j++) {
      if (bits & 1u) F29::mul(acc, acc, x);
      F29::sqr(x, x);
      bits >>= 1;
    }
  }
  F29::div32(acc, acc);                                  // back to the 2^256 domain
  F29::canon(acc);
  F29::pack(r.v, acc);
}
// the lazy parameter set of a strict field
template <class F> struct Lazy29Of;
template <> struct Lazy29Of<Field32<FqParams>> { typedef FqParams P32; typedef Fq29Params P29; };
template <> struct Lazy29Of<Field32<FrParams>> { typedef FrParams P32; typedef Fr29Params P29; };
template <class F>
__device__ __forceinline__ void inv_via_lazy(typename F::fe& r, const typename F::fe& a) {
  inv_lazy<typename Lazy29Of<F>::P32, typename Lazy29Of<F>::P29>(r, a);
}

}  // namespace lemsm
