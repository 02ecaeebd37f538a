// Measured answer to "would batched-affine bucket additions with the prefix products resident in LDS beat the XYZZ
// mixed addition?" (VERDICT r1 item 5).  Arithmetic and LDS only -- points come from an L2-resident pool, as the
// accumulate kernel's gathers mostly do -- in the lazy 29-bit field both ways:
//   k_xyzz      : the shipped mixed addition (XYZZ29::madd_nonempty), one accumulator chain per lane
//   k_batch<B>  : B independent affine accumulators per lane; a round adds one point to each of them with ONE inversion
//                 (Montgomery's trick: B prefix products parked in LDS, Fermat inversion in the lane, back-substitution,
//                 5 products + 1 squaring per addition).  B = 8 is what LDS allows at 2 waves per SIMD (72 KiB per 256
//                 lanes); INV = false skips the inversion: the floor a perfectly amortised inversion would leave.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o batched_affine batched_affine.hip ; run: ./batched_affine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../halo2_liam_eagen_msm_amd/csrc/xyzz29.cuh"
using namespace lemsm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef Field29<Fq29Params> F;
typedef XYZZ29<F> G;

__device__ __forceinline__ void ldfe(F::fe& r, const int* p) {
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = p[i];
}
struct Exp { unsigned e[8]; };   // p - 2

__device__ __noinline__ void fermat_inv(F::fe& r, const F::fe& a, const Exp& ex) {
  F::fe x = a, acc; F::set_one(acc);
  for (int w = 0; w < 8; w++) {
    unsigned bits = ex.e[w];
    for (int j = 0; j < 32; j++) {
      if (bits & 1u) F::mul(acc, acc, x);
      F::sqr(x, x);
      bits >>= 1;
    }
  }
  r = acc;
}

// baseline: rounds * B mixed additions into ONE XYZZ accumulator per lane
template <int B>
__global__ __launch_bounds__(256) void k_xyzz(const int* __restrict__ pool, int* __restrict__ out, int rounds) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int* my = pool + (size_t)t * (B + 1) * 18;
  G::pt acc; ldfe(acc.x, my); ldfe(acc.y, my + 9); F::set_one(acc.zz); F::set_one(acc.zzz);
  bool empty = false;
  for (int r = 0; r < rounds; r++)
    for (int k = 0; k < B; k++) {
      F::fe x, y; ldfe(x, my + (k + 1) * 18); ldfe(y, my + (k + 1) * 18 + 9);
      G::madd_nonempty(acc, x, y, empty);
    }
  for (int i = 0; i < 9; i++) { out[(size_t)t * 36 + i] = acc.x.l[i]; out[(size_t)t * 36 + 9 + i] = acc.y.l[i]; out[(size_t)t * 36 + 18 + i] = acc.zz.l[i]; out[(size_t)t * 36 + 27 + i] = acc.zzz.l[i]; }
}

// batched affine: accumulators k = 0..B-1 each receive point k of the pool every round
template <int B, bool INV>
__global__ __launch_bounds__(256) void k_batch(const int* __restrict__ pool, int* __restrict__ out, int rounds, Exp ex) {
  __shared__ int park[B][9][256];                       // limb-major: lanes on consecutive banks
  const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x;
  const int* my = pool + (size_t)t * (B + 1) * 18;
  F::fe ax[B], ay[B];
#pragma unroll
  for (int k = 0; k < B; k++) { ldfe(ax[k], my); ldfe(ay[k], my + 9); ax[k].l[0] ^= k; }   // B different starting points
  for (int r = 0; r < rounds; r++) {
    F::fe run; F::set_one(run);
#pragma unroll
    for (int k = 0; k < B; k++) {
      F::fe x, d; ldfe(x, my + (k + 1) * 18);
      F::sub(d, x, ax[k]);
#pragma unroll
      for (int i = 0; i < 9; i++) park[k][i][lane] = run.l[i];
      F::mul(run, run, d);
    }
    F::fe inv;
    if (INV) fermat_inv(inv, run, ex); else inv = run;
#pragma unroll
    for (int k = B - 1; k >= 0; k--) {
      F::fe x, y, d, pk, ik, lam, t1, x3, y3;
      ldfe(x, my + (k + 1) * 18); ldfe(y, my + (k + 1) * 18 + 9);
      F::sub(d, x, ax[k]);
#pragma unroll
      for (int i = 0; i < 9; i++) pk.l[i] = park[k][i][lane];
      F::mul(ik, inv, pk); F::mul(inv, inv, d);
      F::sub(t1, y, ay[k]); F::mul(lam, t1, ik);
      F::sqr(x3, lam); F::sub(x3, x3, ax[k]); F::sub(x3, x3, x); F::wnorm(x3);
      F::sub(t1, ax[k], x3); F::wnorm(t1);
      F::mul(y3, lam, t1); F::sub(y3, y3, ay[k]); F::wnorm(y3);
      ax[k] = x3; ay[k] = y3;
    }
  }
#pragma unroll
  for (int k = 0; k < B; k++)
    for (int i = 0; i < 9; i++) { out[((size_t)t * B + k) * 18 + i] = ax[k].l[i]; out[((size_t)t * B + k) * 18 + 9 + i] = ay[k].l[i]; }
}

// consistency of the two formula sets on the same data: one accumulator, `steps` additions of point 0 both ways
__global__ void k_check(const int* __restrict__ pool, int steps, Exp ex, int* __restrict__ bad) {
  const int t = threadIdx.x;
  const int* my = pool + (size_t)t * 9 * 18;
  G::pt acc; ldfe(acc.x, my); ldfe(acc.y, my + 9); F::set_one(acc.zz); F::set_one(acc.zzz);
  F::fe ax = acc.x, ay = acc.y; bool empty = false;
  for (int s = 0; s < steps; s++) {
    F::fe x, y; ldfe(x, my + 18 * (1 + s % 8)); ldfe(y, my + 18 * (1 + s % 8) + 9);
    G::madd_nonempty(acc, x, y, empty);
    F::fe d, inv, lam, t1, x3, y3;
    F::sub(d, x, ax); F::wnorm(d); fermat_inv(inv, d, ex);
    F::sub(t1, y, ay); F::mul(lam, t1, inv);
    F::sqr(x3, lam); F::sub(x3, x3, ax); F::sub(x3, x3, x); F::wnorm(x3);
    F::sub(t1, ax, x3); F::wnorm(t1); F::mul(y3, lam, t1); F::sub(y3, y3, ay); F::wnorm(y3);
    ax = x3; ay = y3;
  }
  // X == x ZZ and Y == y ZZZ ?
  F::fe u, v; F::mul(u, ax, acc.zz); F::sub(u, u, acc.x); F::mul(v, ay, acc.zzz); F::sub(v, v, acc.y);
  if (!F::is_zero_mod(u) || !F::is_zero_mod(v)) atomicAdd(bad, 1);
}

template <class Kern, class... Args>
static double time_ms(Kern k, dim3 grid, dim3 blk, int reps, Args... args) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, grid, blk, 0, 0, args...); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, grid, blk, 0, 0, args...);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main() {
  const int B = 8, blocks = 256 * 8, threads = blocks * 256, rounds = 16;
  std::vector<int> pool((size_t)threads * (B + 1) * 18);
  srand(7);
  for (size_t i = 0; i < pool.size(); i++) pool[i] = (i % 9 == 8) ? (rand() & 0x1fffff) : (rand() & 0x1fffffff);   // normalised limbs, value < p
  int *d_pool, *d_out, *d_bad;
  CK(hipMalloc(&d_pool, pool.size() * 4)); CK(hipMalloc(&d_out, (size_t)threads * B * 18 * 4)); CK(hipMalloc(&d_bad, 4));
  CK(hipMemcpy(d_pool, pool.data(), pool.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(d_bad, 0, 4));
  Exp ex; const unsigned long long N[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  for (int i = 0; i < 4; i++) { ex.e[2 * i] = (unsigned)N[i]; ex.e[2 * i + 1] = (unsigned)(N[i] >> 32); }
  ex.e[0] -= 2;
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d_pool, 12, ex, d_bad);
  int bad = -1; CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
  printf("formula check (12 additions both ways, 64 lanes): %s\n", bad == 0 ? "affine chord addition == XYZZ mixed addition" : "MISMATCH");
  const double adds = (double)threads * B * rounds;
  double ms_x = time_ms(k_xyzz<B>, dim3(blocks), dim3(256), 5, (const int*)d_pool, d_out, rounds);
  double ms_b = time_ms(k_batch<B, true>, dim3(blocks), dim3(256), 3, (const int*)d_pool, d_out, rounds, ex);
  double ms_f = time_ms(k_batch<B, false>, dim3(blocks), dim3(256), 5, (const int*)d_pool, d_out, rounds, ex);
  printf("%d lanes x %d additions each, whole chip:\n", threads, B * rounds);
  printf("  XYZZ mixed addition (shipped)                          %8.3f ms  %7.1f G additions/s   1.00x\n", ms_x, adds / ms_x / 1e6, 1.0);
  printf("  batched affine, B = %d per lane, Fermat inversion       %8.3f ms  %7.1f G additions/s   %.2fx the time\n", B, ms_b, adds / ms_b / 1e6, ms_b / ms_x);
  printf("  batched affine, B = %d per lane, inversion left out     %8.3f ms  %7.1f G additions/s   %.2fx the time (floor of a perfectly amortised inversion)\n", B, ms_f, adds / ms_f / 1e6, ms_f / ms_x);
  return bad == 0 ? 0 : 1;
}
