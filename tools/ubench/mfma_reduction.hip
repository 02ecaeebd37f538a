// VERDICT r02 item 9, "kill early": could the constant product m * N of the lazy field's Montgomery reduction (81 of the
// ~160 multiply-adds of one field multiplication) run on the matrix pipe beside the VALU stream?
//   variant 0   the 81 v_mad_u64_u32 it would replace: columns acc[i + j] += m[i] * N[j], 9 x 9 limbs of 29 bits
//   variant 1   ONLY the VALU work the i8-MFMA route adds around the matrix instructions: pack the 9 limbs into 33 bytes
//               (9 dwords), move the B operand and the 32 x 32 i32 result tiles between the lane halves (v_permlane32_swap:
//               a lane's column of the product lives half in lane l, half in lane l +- 32), and fold the 66 byte-position
//               sums back into 29-bit limb columns (one 64-bit multiply-add each: the shift rides on the multiplier)
//   variant 2   variant 1 plus the 12 v_mfma_i32_32x32x32_i8 (3 row tiles x 2 lane tiles x 2 k-steps; Toeplitz tile of N's
//               bytes as the A operand, resident in registers)
// Timing only: the data flow has the real shape and a dependency from every iteration into the next, the numbers are not a
// product anybody checks.  3 waves per SIMD like k_accum1.  Build: hipcc --offload-arch=gfx950 -O3 mfma_reduction.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned u32; typedef unsigned long long u64;
__constant__ u32 NL[9] = {0x187cfd47, 0x010460b6, 0x1c72a34f, 0x02d522d0, 0x1585d978, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};

__global__ __launch_bounds__(256) void k_mads(u32* io, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  u32 m[9];
  for (int i = 0; i < 9; i++) m[i] = io[t * 9 + i] & 0x1fffffffu;
  u32 nl[9];
  for (int i = 0; i < 9; i++) nl[i] = NL[i];
  for (int it = 0; it < iters; it++) {
    u64 acc[17];
#pragma unroll
    for (int k = 0; k < 17; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
      for (int j = 0; j < 9; j++) acc[i + j] += (u64)m[i] * nl[j];
    // carry chain over the columns (every bit of every column counts, as in the reduction), then the next m: the same in every variant
    u64 c = 0; u32 out[17];
#pragma unroll
    for (int k = 0; k < 17; k++) { const u64 v = acc[k] + c; out[k] = (u32)v & 0x1fffffffu; c = v >> 29; }
#pragma unroll
    for (int i = 0; i < 9; i++) m[i] = (out[8 + i] ^ out[i]) & 0x1fffffffu;
  }
  for (int i = 0; i < 9; i++) io[t * 9 + i] = m[i];
}

template <bool WITH_MFMA>
__global__ __launch_bounds__(256) void k_route(u32* io, const int* toeplitz, int iters) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  u32 m[9];
  for (int i = 0; i < 9; i++) m[i] = io[t * 9 + i] & 0x1fffffffu;
  v4i A[3][2];                                       // Toeplitz tiles of N's bytes: 3 row tiles x 2 k-steps, resident
  for (int r = 0; r < 3; r++) for (int k = 0; k < 2; k++) for (int q = 0; q < 4; q++) A[r][k][q] = toeplitz[((r * 2 + k) * 64 + (threadIdx.x & 63)) * 4 + q];
  for (int it = 0; it < iters; it++) {
    // (1) 9 limbs of 29 bits -> the 261-bit integer as 9 dwords = 33 bytes (+ 7 zero dwords up to 64 bytes)
    u32 w[16];
#pragma unroll
    for (int j = 0; j < 9; j++) {
      const int bit = 32 * j, li = bit / 29, sh = bit - 29 * li;
      u32 v = m[li] >> sh;
      if (li + 1 < 9) v |= m[li + 1] << (29 - sh);
      if (29 - sh + 29 < 32 && li + 2 < 9) v |= m[li + 2] << (58 - sh);
      w[j] = v;
    }
#pragma unroll
    for (int j = 9; j < 16; j++) w[j] = 0;
    // (2) B operands: lanes 0..31 carry k 0..15 of a column, lanes 32..63 k 16..31 -- one half-swap per dword pair
    v4i B0[2], B1[2];                                // per k-step: lane tile 0 (columns 0..31), lane tile 1 (columns 32..63)
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        auto sw = __builtin_amdgcn_permlane32_swap(w[8 * k + q], w[8 * k + 4 + q], false, false);
        B0[k][q] = (int)sw[0]; B1[k][q] = (int)sw[1];
      }
    // (3) the product on the matrix pipe: 3 row tiles x 2 lane tiles, 2 k-steps each
    v16i C[3][2];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      if (WITH_MFMA) {
        v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        C[r][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[r][0], B0[0], z, 0, 0, 0);
        C[r][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[r][1], B0[1], C[r][0], 0, 0, 0);
        C[r][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[r][0], B1[0], z, 0, 0, 0);
        C[r][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[r][1], B1[1], C[r][1], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 16; e++) {              // stand-in values the compiler cannot merge (one add each)
          int x = B0[e & 1][e & 3] + A[r][0][e & 3], y = B1[e & 1][e & 3] + A[r][1][e & 3];
          asm volatile("" : "+v"(x)); asm volatile("" : "+v"(y));
          C[r][0][e] = x; C[r][1][e] = y;
        }
      }
    }
    // (4) a lane's column of the result sits half in lane l, half in lane l +- 32: 48 half-swaps bring it together
    u32 S[96];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        auto sw = __builtin_amdgcn_permlane32_swap((u32)C[r][0][e], (u32)C[r][1][e], false, false);
        S[r * 32 + (e / 4) * 8 + (e % 4)] = sw[0];        // rows (e / 4) * 8 + (e % 4): the lower half's
        S[r * 32 + (e / 4) * 8 + 4 + (e % 4)] = sw[1];    // and + 4: the upper half's
      }
    // (5) 66 byte-position sums (< 2^21) back into 29-bit limb columns: one 64-bit multiply-add each (shift = multiplier)
    u64 acc[19];
#pragma unroll
    for (int k = 0; k < 19; k++) acc[k] = 0;
#pragma unroll
    for (int i = 0; i < 66; i++) {
      const int bit = 8 * i, li = bit / 29, sh = bit - 29 * li;
      acc[li] += (u64)S[i] * (u32)(1u << sh);
    }
    u64 c = 0; u32 out[19];
#pragma unroll
    for (int k = 0; k < 19; k++) { const u64 v = acc[k] + c; out[k] = (u32)v & 0x1fffffffu; c = v >> 29; }
#pragma unroll
    for (int i = 0; i < 9; i++) m[i] = (out[8 + i] ^ out[i] ^ (i == 0 ? out[17] ^ out[18] : 0u)) & 0x1fffffffu;
  }
  for (int i = 0; i < 9; i++) io[t * 9 + i] = m[i];
}

int main() {
  const int blocks = 256 * 12, iters = 4000;         // 3 waves per SIMD
  const size_t n = (size_t)blocks * 256;
  std::vector<u32> h(n * 9); srand(5);
  for (auto& x : h) x = (u32)rand() * 2654435761u;
  std::vector<int> tz(3 * 2 * 64 * 4);
  for (auto& x : tz) x = rand();
  u32* d; int* dt;
  CK(hipMalloc(&d, n * 36)); CK(hipMalloc(&dt, tz.size() * 4));
  CK(hipMemcpy(dt, tz.data(), tz.size() * 4, hipMemcpyHostToDevice));
  const char* name[3] = {"81 v_mad_u64_u32 (the constant product on the VALU)", "i8-MFMA route, VALU side only (pack, half-swaps, recombination)",
                         "i8-MFMA route in full (+ 12 v_mfma_i32_32x32x32_i8)"};
  for (int which = 0; which < 3; which++) {
    CK(hipMemcpy(d, h.data(), n * 36, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_mads, dim3(blocks), dim3(256), 0, 0, d, iters);
      else if (which == 1) hipLaunchKernelGGL((k_route<false>), dim3(blocks), dim3(256), 0, 0, d, (const int*)dt, iters);
      else hipLaunchKernelGGL((k_route<true>), dim3(blocks), dim3(256), 0, 0, d, (const int*)dt, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)n / 64, per = ms * 1e6 / iters / (waves / 1024);     // ns per iteration per SIMD-slot (1024 SIMDs)
    printf("%-70s %8.2f ms  %7.1f ns per product and SIMD (3 waves interleaved: %.0f cycles at 2.4 GHz per wave-product)\n", name[which], ms, per, per * 2.4);
  }
  return 0;
}
