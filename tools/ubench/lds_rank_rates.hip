// What do the LDS operations of k_scatter1 cost on their own?  Pass 1 of the bucket sort ranks and places every entry with
// ONE returning LDS atomic on its bin's cursor (512 bins per window at 2^24 points) and one 4-byte LDS store at the rank.
// This microbenchmark runs exactly that -- 2^24 x 15 = 2.5e8 placements as 61 440 block-rounds of 4096 entries, 256-thread
// blocks holding 21 KB of LDS (seven per CU, as the kernel has them) -- with nothing else around it: no global loads, no
// scans, no stores.  Variants: the returning atomic alone; atomic + staged store; a non-returning atomic; 64 bins; one bin
// (every lane on the same address); the staged store + read-back without any atomic.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_rank_rates tools/ubench/lds_rank_rates.hip && /tmp/lds_rank_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned int u32;

template <int MODE>
__global__ __launch_bounds__(256) void k_rank(u32 rounds, u32 bin_mask, u32* __restrict__ sink) {
  __shared__ u32 lcur[512];
  __shared__ u32 pad[256];          // (the kernel's delta[] + wsum[]: same LDS footprint, same blocks per CU)
  __shared__ u32 stage[4096];
  const u32 tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
  u32 acc = 0;
  pad[tid] = 0;
  for (u32 r = 0; r < rounds; r++) {
    lcur[tid] = (tid * 8u) & 4095u; lcur[tid + 256] = ((tid + 256u) * 8u) & 4095u;     // a bin's cursor starts at its run
    __syncthreads();
    u32 h = (gid * 2654435761u) ^ (r * 0x9e3779b9u);
#pragma unroll
    for (int k = 0; k < 16; k++) {
      h = h * 1664525u + 1013904223u;
      const u32 b = (h >> 16) & bin_mask;
      if (MODE == 0) acc += atomicAdd(&lcur[b], 1u);                                    // returning atomic (ds_add_rtn_u32)
      if (MODE == 1) { u32 q = atomicAdd(&lcur[b], 1u); stage[q & 4095u] = h; }         // + the staged store at the rank
      if (MODE == 2) __hip_atomic_fetch_add(&lcur[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // result unused: ds_add_u32
      if (MODE == 3) stage[(h >> 7) & 4095u] = h;                                       // random 4-byte LDS stores only
    }
    __syncthreads();
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int k = 0; k < 16; k++) acc += stage[tid + 256 * k];                         // the store loop's read of the staging buffer
    }
    if (MODE == 2) acc += lcur[tid];
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[gid] = acc + pad[tid];
}

template <int MODE>
static void run(const char* what, u32 bin_mask, int cus, double mhz) {
  const u32 blocks = (u32)cus * 7u, total_rounds = 61440u;
  const u32 rounds = (total_rounds + blocks - 1) / blocks;
  u32* sink; CK(hipMalloc(&sink, (size_t)blocks * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 4; it++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_rank<MODE>), dim3(blocks), dim3(256), 0, 0, rounds, bin_mask, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
  }
  const double placements = (double)blocks * rounds * 4096.0;
  const double per_clk_cu = placements / (best * 1e-3) / (mhz * 1e6) / cus;
  printf("%-58s %8.1f us for %.3g placements = %5.2f lane-ops per clock and CU (at %.0f MHz)\n", what, best * 1e3, placements, per_clk_cu, mhz);
  CK(hipFree(sink));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount; const double mhz = p.clockRate / 1e3;
  printf("%s, %d CUs, %.0f MHz; 7 blocks of 256 threads per CU, 4096 placements per block-round, 61 440 block-rounds (= pass 1 of a 2^24-point MSM)\n", p.name, cus, mhz);
  run<0>("returning atomic, 512 bins", 511u, cus, mhz);
  run<1>("returning atomic + staged store + read-back, 512 bins", 511u, cus, mhz);
  run<2>("non-returning atomic, 512 bins", 511u, cus, mhz);
  run<3>("random staged stores + read-back, no atomic", 511u, cus, mhz);
  run<0>("returning atomic, 64 bins", 63u, cus, mhz);
  run<0>("returning atomic, 8 bins", 7u, cus, mhz);
  run<0>("returning atomic, one bin (every lane the same address)", 0u, cus, mhz);
  run<1>("returning atomic + staged store + read-back, 256 bins", 255u, cus, mhz);
  return 0;
}
