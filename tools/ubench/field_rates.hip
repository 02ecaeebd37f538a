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

template <int WV> __global__ __launch_bounds__(256, WV) void k_mul32(uint4* io, int iters) {
  F32::fe a, b; F32::load(a, io + 2 * (blockIdx.x * 256 + threadIdx.x)); b = a;
  for (int i = 0; i < iters; i++) { F32::fe r; F32::mul(r, a, b); b = a; a = r; }
  F32::store(io + 2 * (blockIdx.x * 256 + threadIdx.x), a);
}
template <int WV> __global__ __launch_bounds__(256, WV) void k_mul29(uint4* io, int iters) {
  F29::fe a, b; F29::load(a, io + 2 * (blockIdx.x * 256 + threadIdx.x)); b = a;
  for (int i = 0; i < iters; i++) { F29::fe r; F29::mul(r, a, b); b = a; a = r; }
  F29::store(io + 2 * (blockIdx.x * 256 + threadIdx.x), a);
}
template <int WV> __global__ __launch_bounds__(256, WV) void k_sqr29(uint4* io, int iters) {
  F29::fe a; F29::load(a, io + 2 * (blockIdx.x * 256 + threadIdx.x));
  for (int i = 0; i < iters; i++) { F29::fe r; F29::sqr(r, a); a = r; }
  F29::store(io + 2 * (blockIdx.x * 256 + threadIdx.x), a);
}
template <int WV> __global__ __launch_bounds__(256, WV) void k_madd32(uint4* io, int iters) {
  uint4* p = io + 2 * (blockIdx.x * 256 + threadIdx.x);
  F32::fe x, y; F32::load(x, p); F32::mul(y, x, x);
  G32::pt acc; G32::set_identity(acc);
  for (int i = 0; i < iters; i++) { G32::madd(acc, x, y); F32::add(x, x, y); }
  F32::store(p, acc.x);
}
template <int WV> __global__ __launch_bounds__(256, WV) void k_madd29(uint4* io, int iters) {
  uint4* p = io + 2 * (blockIdx.x * 256 + threadIdx.x);
  F29::fe x, y; F29::load(x, p); F29::mul(y, x, x);
  G29::pt acc; G29::set_identity(acc);
  for (int i = 0; i < iters; i++) { G29::madd(acc, x, y); F29::fe t; F29::mul(t, x, y); x = t; }
  F29::store(p, acc.x);
}

typedef void (*kern_t)(uint4*, int);
struct E { const char* name; kern_t k[4]; int iters; double sub; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  size_t nthr = (size_t)cus * 4 * 256;
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
  printf("%-22s %6s %10s %16s %16s\n", "op", "w/SIMD", "ms", "ns/op/SIMD", "cyc/op/SIMD@2.4");
  for (auto& e : es) for (int wv = 1; wv <= 4; wv++) {
    CK(hipMemcpy(d, h.data(), nthr * 32, hipMemcpyHostToDevice));
    int blocks = cus * wv;
    hipLaunchKernelGGL(e.k[wv - 1], dim3(blocks), dim3(256), 0, 0, d, e.iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(d, h.data(), nthr * 32, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(e.k[wv - 1], dim3(blocks), dim3(256), 0, 0, d, e.iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ns = ms * 1e6 / ((double)e.iters * wv);   // per wave-op on one SIMD
    printf("%-22s %6d %10.3f %16.1f %16.0f\n", e.name, wv, ms, ns, ns * 2.4);
  }
  return 0;
}
