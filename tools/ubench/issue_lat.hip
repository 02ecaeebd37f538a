// Issue interval and dependent-issue latency of the instructions the 29-bit Montgomery multiplier is made of,
// measured WITHOUT the s_nop hipcc puts after every asm statement: each kernel runs ONE asm block of 32
// instructions per loop trip, either 32 independent destinations (8 registers x 4) or one dependent chain, at
// exactly 1..4 waves per SIMD (ONE block per CU, enforced by an LDS allocation, of 256 x W threads), and reports shader cycles (s_memtime) per instruction per wave and per SIMD.
//   build: hipcc -O2 --offload-arch=gfx950 -o issue_lat issue_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
typedef unsigned long long u64; typedef unsigned int u32; typedef long long i64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2048;

#define R4(x) x x x x
#define R8(x) x x x x x x x x
#define R32(x) R8(x) R8(x) R8(x) R8(x)

#define KBEGIN(name)                                                                                        \
  __global__ __launch_bounds__(1024) void name(u32* out, u64* cyc, u32 seed) {                             \
    __shared__ u32 pad_[20800]; if (seed == 0x7fffffffu) pad_[threadIdx.x] = seed;   /* > 80 KB: one block per CU */ \
    u32 a = seed * (threadIdx.x + 1) | 1u, b = (seed ^ (threadIdx.x * 2654435761u)) | 3u; (void)a; (void)b;
#define KLOOP                                                                                               \
    u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                           \
    for (int i = 0; i < ITERS; i++) {
#define KEND(SINK)                                                                                          \
    }                                                                                                       \
    u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(SINK);                                               \
    if ((threadIdx.x & 63) == 0) { u32 wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * wv] = t1 - t0; cyc[2 * wv + 1] = r1 - r0; } }

// ---- 64-bit multiply-add: 8 independent accumulators, 4 rounds per block
KBEGIN(k_mad_indep) i64 c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3; KLOOP
  asm volatile(R4("v_mad_i64_i32 %0, vcc, %8, %9, %0\n\tv_mad_i64_i32 %1, vcc, %8, %9, %1\n\tv_mad_i64_i32 %2, vcc, %8, %9, %2\n\tv_mad_i64_i32 %3, vcc, %8, %9, %3\n\t"
                  "v_mad_i64_i32 %4, vcc, %8, %9, %4\n\tv_mad_i64_i32 %5, vcc, %8, %9, %5\n\tv_mad_i64_i32 %6, vcc, %8, %9, %6\n\tv_mad_i64_i32 %7, vcc, %8, %9, %7\n\t")
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");
KEND(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
// ---- one dependent chain (what a column of the multiplier is)
KBEGIN(k_mad_dep) i64 c0 = a; KLOOP
  asm volatile(R32("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t") : "+v"(c0) : "v"(a), "v"(b) : "vcc");
KEND(c0)
// ---- two interleaved chains
KBEGIN(k_mad_dep2) i64 c0 = a; i64 c1 = b; KLOOP
  asm volatile(R8("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %1, vcc, %2, %3, %1\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %1, vcc, %2, %3, %1\n\t")
               : "+v"(c0), "+v"(c1) : "v"(a), "v"(b) : "vcc");
KEND(c0 ^ c1)
// ---- mad with an SGPR multiplier (the reduction's m_i * N[j] products take N[j] from SGPRs)
KBEGIN(k_mad_dep_sgpr) i64 c0 = a; u32 sconst = seed | 5u; KLOOP
  asm volatile(R32("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t") : "+v"(c0) : "v"(a), "s"(sconst) : "vcc");
KEND(c0)
// ---- simple 32-bit VALU
KBEGIN(k_add_indep) u32 c0 = a, c1 = b, c2 = a + 1, c3 = b + 1, c4 = a + 2, c5 = b + 2, c6 = a + 3, c7 = b + 3; KLOOP
  asm volatile(R4("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                  "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t")
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a));
KEND(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
KBEGIN(k_add_dep) u32 c0 = a; KLOOP asm volatile(R32("v_add_u32 %0, %0, %1\n\t") : "+v"(c0) : "v"(b)); KEND(c0)
KBEGIN(k_mullo_dep) u32 c0 = a; KLOOP asm volatile(R32("v_mul_lo_u32 %0, %0, %1\n\t") : "+v"(c0) : "v"(b)); KEND(c0)
KBEGIN(k_mullo_indep) u32 c0 = a, c1 = b, c2 = a + 1, c3 = b + 1; KLOOP
  asm volatile(R8("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4\n\t")
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b));
KEND(c0 ^ c1 ^ c2 ^ c3)
KBEGIN(k_ashr64_dep) i64 c0 = ((i64)a << 31) | b; KLOOP asm volatile(R32("v_ashrrev_i64 %0, 1, %0\n\t") : "+v"(c0)); KEND(c0)
// ---- the reduction step of one column as the multiplier runs it: m = (lo(acc) * NINV) & MASK ; acc += m * N0 ; acc >>= 29
KBEGIN(k_redstep) i64 c0 = a; u32 m = 0; KLOOP
  asm volatile(R8("v_mul_lo_u32 %1, %1, %2\n\tv_and_b32 %1, 0x1fffffff, %1\n\tv_mad_i64_i32 %0, vcc, %1, %3, %0\n\tv_ashrrev_i64 %0, 29, %0\n\t")
               : "+v"(c0), "+v"(m) : "v"(a), "v"(b) : "vcc");
KEND(c0 ^ m)
// ---- a realistic column: 9 dependent mads then the reduction step (13 instructions, one chain)
KBEGIN(k_column) i64 c0 = a; u32 m = 0; KLOOP
  asm volatile(R4("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\t"
                  "v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mad_i64_i32 %0, vcc, %2, %3, %0\n\t"
                  "v_mad_i64_i32 %0, vcc, %2, %3, %0\n\tv_mul_lo_u32 %1, %1, %2\n\tv_and_b32 %1, 0x1fffffff, %1\n\tv_mad_i64_i32 %0, vcc, %1, %3, %0\n\tv_ashrrev_i64 %0, 29, %0\n\t")
               : "+v"(c0), "+v"(m) : "v"(a), "v"(b) : "vcc");
KEND(c0 ^ m)

typedef void (*kern_t)(u32*, u64*, u32);
struct Entry { const char* name; kern_t k; int per_trip; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  std::vector<Entry> es = {
    {"v_mad_i64_i32 x8 independent", k_mad_indep, 32}, {"v_mad_i64_i32 one chain", k_mad_dep, 32}, {"v_mad_i64_i32 two chains", k_mad_dep2, 32},
    {"v_mad_i64_i32 chain, SGPR operand", k_mad_dep_sgpr, 32},
    {"v_add_u32 x8 independent", k_add_indep, 32}, {"v_add_u32 one chain", k_add_dep, 32},
    {"v_mul_lo_u32 x4 independent", k_mullo_indep, 32}, {"v_mul_lo_u32 one chain", k_mullo_dep, 32}, {"v_ashrrev_i64 one chain", k_ashr64_dep, 32},
    {"reduction step (mul_lo,and,mad,ashr)", k_redstep, 32}, {"column: 9 mads + reduction step", k_column, 52},
  };
  u32* d_out; u64* d_cyc;
  size_t maxthreads = (size_t)cus * 8 * 256;
  CK(hipMalloc(&d_out, maxthreads * 4)); CK(hipMalloc(&d_cyc, maxthreads / 64 * 16));
  std::vector<u64> h(maxthreads / 64 * 2);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%-40s %6s %9s %8s %16s %16s\n", "sequence", "w/SIMD", "wall_ms", "clk_GHz", "cyc/instr/wave", "cyc/instr/SIMD");
  for (auto& e : es) for (int wps : {1, 2, 3, 4}) {
    int blocks = cus;                      // one block per CU, 256 * wps threads: exactly wps waves on every SIMD
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256 * wps), 0, 0, d_out, d_cyc, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256 * wps), 0, 0, d_out, d_cyc, 12345u);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int nw = blocks * 4 * wps;
    CK(hipMemcpy(h.data(), d_cyc, (size_t)nw * 16, hipMemcpyDeviceToHost));
    double c = 0, r = 0; for (int i = 0; i < nw; i++) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    c /= nw; r /= nw;
    double n_instr = (double)ITERS * e.per_trip;
    printf("%-40s %6d %9.4f %8.3f %16.2f %16.2f\n", e.name, wps, ms, r > 0 ? c / r * 0.1 : 0.0, c / n_instr, c / n_instr / wps);
  }
  return 0;
}
