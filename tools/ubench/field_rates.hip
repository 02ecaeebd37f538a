// Arithmetic-only throughput of the field / group primitives (no memory traffic): how many
// cycles per SIMD one Montgomery multiplication and one mixed addition cost, for both field
// implementations, at 1..4 waves per SIMD.  Feeds the VALU roofline in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../halo2_liam_eagen_msm_amd/csrc/xyzz.cuh"
#include "../../halo2_liam_eagen_msm_amd/csrc/xyzz29.cuh"
using namespace lemsm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef Field32<FqParams> F32;
typedef Field29<Fq29Params> F29;
typedef XYZZ<F32> G32;
typedef XYZZ29<F29> G29;

template <int WV> __global__ __launch_bounds__(256 * WV) void k_mul32(uint4* io, int iters, unsigned long long* cyc) {
  __shared__ unsigned pad_[20800]; if (iters < 0) pad_[threadIdx.x] = iters;   // > 80 KB: one block per CU -> exactly WV waves per SIMD
  unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
  F32::fe a, b; F32::load(a, io + 2 * (blockIdx.x * blockDim.x + threadIdx.x)); b = a;
  for (int i = 0; i < iters; i++) { F32::fe r; F32::mul(r, a, b); b = a; a = r; }
  F32::store(io + 2 * (blockIdx.x * blockDim.x + threadIdx.x), a);
  if ((threadIdx.x & 63) == 0) { unsigned wv_ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * wv_] = __builtin_amdgcn_s_memtime() - t0_; cyc[2 * wv_ + 1] = __builtin_amdgcn_s_memrealtime() - r0_; }
}
template <int WV> __global__ __launch_bounds__(256 * WV) void k_mul29(uint4* io, int iters, unsigned long long* cyc) {
  __shared__ unsigned pad_[20800]; if (iters < 0) pad_[threadIdx.x] = iters;   // > 80 KB: one block per CU -> exactly WV waves per SIMD
  unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
  F29::fe a, b; F29::load(a, io + 2 * (blockIdx.x * blockDim.x + threadIdx.x)); b = a;
  for (int i = 0; i < iters; i++) { F29::fe r; F29::mul(r, a, b); b = a; a = r; }
  F29::store(io + 2 * (blockIdx.x * blockDim.x + threadIdx.x), a);
  if ((threadIdx.x & 63) == 0) { unsigned wv_ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * wv_] = __builtin_amdgcn_s_memtime() - t0_; cyc[2 * wv_ + 1] = __builtin_amdgcn_s_memrealtime() - r0_; }
}
template <int WV> __global__ __launch_bounds__(256 * WV) void k_sqr29(uint4* io, int iters, unsigned long long* cyc) {
  __shared__ unsigned pad_[20800]; if (iters < 0) pad_[threadIdx.x] = iters;   // > 80 KB: one block per CU -> exactly WV waves per SIMD
  unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
  F29::fe a; F29::load(a, io + 2 * (blockIdx.x * blockDim.x + threadIdx.x));
  for (int i = 0; i < iters; i++) { F29::fe r; F29::sqr(r, a); a = r; }
  F29::store(io + 2 * (blockIdx.x * blockDim.x + threadIdx.x), a);
  if ((threadIdx.x & 63) == 0) { unsigned wv_ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * wv_] = __builtin_amdgcn_s_memtime() - t0_; cyc[2 * wv_ + 1] = __builtin_amdgcn_s_memrealtime() - r0_; }
}
template <int WV> __global__ __launch_bounds__(256 * WV) void k_madd32(uint4* io, int iters, unsigned long long* cyc) {
  __shared__ unsigned pad_[20800]; if (iters < 0) pad_[threadIdx.x] = iters;   // > 80 KB: one block per CU -> exactly WV waves per SIMD
  unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
  uint4* p = io + 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  F32::fe x, y; F32::load(x, p); F32::mul(y, x, x);
  G32::pt acc; G32::set_identity(acc);
  for (int i = 0; i < iters; i++) { G32::madd(acc, x, y); F32::add(x, x, y); }
  F32::store(p, acc.x);
  if ((threadIdx.x & 63) == 0) { unsigned wv_ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * wv_] = __builtin_amdgcn_s_memtime() - t0_; cyc[2 * wv_ + 1] = __builtin_amdgcn_s_memrealtime() - r0_; }
}
template <int WV> __global__ __launch_bounds__(256 * WV) void k_madd29(uint4* io, int iters, unsigned long long* cyc) {
  __shared__ unsigned pad_[20800]; if (iters < 0) pad_[threadIdx.x] = iters;   // > 80 KB: one block per CU -> exactly WV waves per SIMD
  unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
  uint4* p = io + 2 * (blockIdx.x * blockDim.x + threadIdx.x);
  F29::fe x, y; F29::load(x, p); F29::mul(y, x, x);
  G29::pt acc; G29::set_identity(acc);
  for (int i = 0; i < iters; i++) { G29::madd(acc, x, y); F29::fe t; F29::mul(t, x, y); x = t; }
  F29::store(p, acc.x);
  if ((threadIdx.x & 63) == 0) { unsigned wv_ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * wv_] = __builtin_amdgcn_s_memtime() - t0_; cyc[2 * wv_ + 1] = __builtin_amdgcn_s_memrealtime() - r0_; }
}

typedef void (*kern_t)(uint4*, int, unsigned long long*);
struct E { const char* name; kern_t k[4]; int iters; double sub; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  size_t nthr = (size_t)cus * 8 * 256;
  unsigned long long* d_cyc; CK(hipMalloc(&d_cyc, nthr / 64 * 16)); std::vector<unsigned long long> hc(nthr / 64 * 2);
  uint4* d; CK(hipMalloc(&d, nthr * 32));
  std::vector<unsigned> h(nthr * 8);
  for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  E es[] = {
    {"mul32 (strict 8x32)", {k_mul32<1>, k_mul32<2>, k_mul32<3>, k_mul32<4>}, 2000, 0},
    {"mul29 (lazy 9x29)", {k_mul29<1>, k_mul29<2>, k_mul29<3>, k_mul29<4>}, 2000, 0},
    {"sqr29", {k_sqr29<1>, k_sqr29<2>, k_sqr29<3>, k_sqr29<4>}, 2000, 0},
    {"madd32 (+1 add)", {k_madd32<1>, k_madd32<2>, k_madd32<3>, k_madd32<4>}, 300, 0},
    {"madd29 (+1 mul)", {k_madd29<1>, k_madd29<2>, k_madd29<3>, k_madd29<4>}, 300, 0},
  };
  printf("%-22s %6s %10s %14s %9s %18s %18s\n", "op", "w/SIMD", "ms", "ns/op/SIMD", "clk_GHz", "cyc/op/wave", "cyc/op/SIMD");
  const int wvs[4] = {1, 2, 3, 4};
  for (auto& e : es) for (int wi = 0; wi < 4; wi++) {
    const int wv = wvs[wi];
    CK(hipMemcpy(d, h.data(), nthr * 32, hipMemcpyHostToDevice));
    int blocks = cus;   // one block per CU of 256 * wv threads
    hipLaunchKernelGGL(e.k[wi], dim3(blocks), dim3(256 * wv), 0, 0, d, e.iters, d_cyc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d, h.data(), nthr * 32, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(e.k[wi], dim3(blocks), dim3(256 * wv), 0, 0, d, e.iters, d_cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ns = ms * 1e6 / ((double)e.iters * wv);   // per wave-op on one SIMD (wall)   // per wave-op on one SIMD
    const int nwv = blocks * 4 * wv;
    CK(hipMemcpy(hc.data(), d_cyc, (size_t)nwv * 16, hipMemcpyDeviceToHost));
    double c = 0, r = 0; for (int i = 0; i < nwv; i++) { c += (double)hc[2 * i]; r += (double)hc[2 * i + 1]; }
    c /= nwv; r /= nwv;
    printf("%-22s %6d %10.3f %14.1f %9.3f %18.1f %18.1f\n", e.name, wv, ms, ns, r > 0 ? c / r * 0.1 : 0.0, c / e.iters, c / e.iters / wv);
  }
  return 0;
}
