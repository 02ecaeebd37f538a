// Register-only rate of the transform's butterfly -- (u, v) -> (u + v, (u - v) w), one field multiplication, one addition, one
// subtraction -- in the strict 8x32-bit Montgomery field the transforms use (its results are canonical, 32-byte elements
// are the ABI's limbs) and in the lazy 9x29-bit field of the MSM kernels (no carries inside the product, but every sum has
// to be carry-normalised before it can feed the next product).  This is the measured VALU ceiling of k_ntt_tile
// (butterflies per second with nothing but the arithmetic) and the A/B DESIGN.md section 6 argued without numbers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../halo2_liam_eagen_msm_amd/csrc/field32.cuh"
#include "../../halo2_liam_eagen_msm_amd/csrc/field29.cuh"
using namespace lemsm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef Field32<FrParams> F32;
typedef Field29<Fr29Params> F29;

__global__ __launch_bounds__(256) void k_bfly32(uint4* io, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  F32::fe u, v, w; F32::load(u, io + 2 * t); F32::load(v, io + 2 * (t + 1)); F32::load(w, io + 2 * (t + 2));
  for (int i = 0; i < iters; i++) { F32::fe x, y; F32::add(x, u, v); F32::sub(y, u, v); F32::mul(y, y, w); u = x; v = y; }
  F32::add(u, u, v); F32::store(io + 2 * t, u);
}
__global__ __launch_bounds__(256) void k_bfly29(uint4* io, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  F29::fe u, v, w; F29::load(u, io + 2 * t); F29::load(v, io + 2 * (t + 1)); F29::load(w, io + 2 * (t + 2));
  for (int i = 0; i < iters; i++) {
    F29::fe x, y; F29::add(x, u, v); F29::wnorm(x);          // the sum must be normalised: it is an operand of the next stage's sum / product
    F29::sub(y, u, v); F29::mul(y, y, w);                    // a difference of normalised values may enter the product as it is
    u = x; v = y;
  }
  F29::add(u, u, v); F29::store(io + 2 * t, u);
}

int main() {
  const int blocks = 256 * 12, iters = 2000;     // 3 waves per SIMD
  const size_t n = (size_t)blocks * 256 + 4;
  std::vector<unsigned> h(n * 8); srand(3);
  for (auto& x : h) x = (unsigned)rand() * 2654435761u; for (size_t i = 0; i < n; i++) h[8 * i + 7] &= 0x0fffffffu;
  uint4* d; CK(hipMalloc(&d, n * 32)); CK(hipMemcpy(d, h.data(), n * 32, hipMemcpyHostToDevice));
  for (int which = 0; which < 2; which++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_bfly32, dim3(blocks), dim3(256), 0, 0, d, iters);
      else hipLaunchKernelGGL(k_bfly29, dim3(blocks), dim3(256), 0, 0, d, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("%s: %.1f G butterflies/s (%d blocks x 256 threads x %d butterflies in %.2f ms)\n", which == 0 ? "strict 8x32-bit field (k_ntt_tile's)" : "lazy 9x29-bit field            ", (double)blocks * 256 * iters / ms / 1e6, blocks, iters, ms);
    }
  }
  return 0;
}
