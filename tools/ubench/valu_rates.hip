// VALU integer / fp64 issue-rate microbenchmark for gfx950 (MI355X).
// Measures cycles per wave-instruction on one SIMD for the instructions a
// 256-bit Montgomery multiplier can be built from, at 1/2/4 waves per SIMD.
// Output feeds DESIGN.md's VALU roofline (mads/s peak measured, not assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;   // independent chains per iteration

typedef unsigned long long u64;
typedef unsigned int u32;

#define KERNEL_BEGIN(name) \
__global__ __launch_bounds__(256) void name(u32* out, u64* cyc, u32 seed) { \
  u32 a = seed * (threadIdx.x + 1) | 1u, b = seed ^ (threadIdx.x * 2654435761u) | 3u; \
  u64 t0 = __builtin_amdgcn_s_memtime(); u64 r0 = __builtin_amdgcn_s_memrealtime();

#define KERNEL_END(sink) \
  u64 t1 = __builtin_amdgcn_s_memtime(); u64 r1 = __builtin_amdgcn_s_memrealtime(); \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(sink); \
  if ((threadIdx.x & 63) == 0) { u32 wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; cyc[2 * wv] = t1 - t0; cyc[2 * wv + 1] = r1 - r0; } }

// ---- v_mad_u64_u32 : independent accumulators
KERNEL_BEGIN(k_mad_u64_u32)
  u64 acc[UNROLL];
  #pragma unroll
  for (int j = 0; j < UNROLL; j++) acc[j] = a + j;
  for (int i = 0; i < ITERS; i++) {
    #pragma unroll
    for (int j = 0; j < UNROLL; j++)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc");
  }
  u64 s = 0;
  #pragma unroll
  for (int j = 0; j < UNROLL; j++) s ^= acc[j];
KERNEL_END(s ^ (s >> 32))

// ---- v_mad_u64_u32 : one dependent chain (latency)
KERNEL_BEGIN(k_mad_u64_u32_dep)
  u64 acc = a;
  for (int i = 0; i < ITERS; i++) {
    #pragma unroll
    for (int j = 0; j < UNROLL; j++)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
  }
KERNEL_END(acc ^ (acc >> 32))

#define SIMPLE32(name, ASM) \
KERNEL_BEGIN(name) \
  u32 acc[UNROLL]; \
  _Pragma("unroll") for (int j = 0; j < UNROLL; j++) acc[j] = a + j; \
  for (int i = 0; i < ITERS; i++) { \
    _Pragma("unroll") for (int j = 0; j < UNROLL; j++) \
      asm volatile(ASM : "+v"(acc[j]) : "v"(a), "v"(b) : "vcc"); \
  } \
  u32 s = 0; \
  _Pragma("unroll") for (int j = 0; j < UNROLL; j++) s ^= acc[j]; \
KERNEL_END(s)

SIMPLE32(k_mul_lo_u32,     "v_mul_lo_u32 %0, %0, %1")
SIMPLE32(k_mul_hi_u32,     "v_mul_hi_u32 %0, %0, %1")
SIMPLE32(k_add_u32,        "v_add_u32 %0, %0, %1")
SIMPLE32(k_add_co_u32,     "v_add_co_u32 %0, vcc, %0, %1")
SIMPLE32(k_addc_co_u32,    "v_addc_co_u32 %0, vcc, %0, %1, vcc")
SIMPLE32(k_add3_u32,       "v_add3_u32 %0, %0, %1, %2")
SIMPLE32(k_mad_u32_u24,    "v_mad_u32_u24 %0, %1, %2, %0")
SIMPLE32(k_mul_u32_u24,    "v_mul_u32_u24 %0, %0, %1")
SIMPLE32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
SIMPLE32(k_mad_u32_u16,    "v_mad_u32_u16 %0, %1, %2, %0")
SIMPLE32(k_dot2_u32_u16,   "v_dot2_u32_u16 %0, %1, %2, %0")
SIMPLE32(k_dot4_u32_u8,    "v_dot4_u32_u8 %0, %1, %2, %0")
SIMPLE32(k_alignbit_b32,   "v_alignbit_b32 %0, %0, %1, 29")
SIMPLE32(k_and_b32,        "v_and_b32 %0, %0, %1")
SIMPLE32(k_lshl_add_u32,   "v_lshl_add_u32 %0, %0, 3, %1")
// plain fp32 VALU (MI355X_MICROARCH.md lists v_fma_f32 at 2 cycles per wave64 with >= 2 waves: reconcile)
SIMPLE32(k_fma_f32,        "v_fma_f32 %0, %1, %2, %0")
SIMPLE32(k_add_f32,        "v_add_f32 %0, %0, %1")
SIMPLE32(k_mul_f32,        "v_mul_f32 %0, %0, %1")
SIMPLE32(k_mac_f32,        "v_fmac_f32 %0, %1, %2")
SIMPLE32(k_mov_b32,        "v_mov_b32 %0, %1")
SIMPLE32(k_xor_b32,        "v_xor_b32 %0, %0, %1")
KERNEL_BEGIN(k_mad_u64_u32_pair)
  u64 acc[UNROLL]; u32 top[UNROLL];
  #pragma unroll
  for (int j = 0; j < UNROLL; j++) { acc[j] = a + j; top[j] = j; }
  for (int i = 0; i < ITERS; i++) {
    #pragma unroll
    for (int j = 0; j < UNROLL; j++)
      asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[j]), "+v"(top[j]) : "v"(a), "v"(b) : "vcc");
  }
  u64 s = 0;
  #pragma unroll
  for (int j = 0; j < UNROLL; j++) s ^= acc[j] + top[j];
KERNEL_END(s ^ (s >> 32))

#define SIMPLE64(name, ASM) \
KERNEL_BEGIN(name) \
  u64 acc[UNROLL]; \
  _Pragma("unroll") for (int j = 0; j < UNROLL; j++) acc[j] = ((u64)a << 20) + j; \
  u64 bb = ((u64)b << 13) | a; \
  for (int i = 0; i < ITERS; i++) { \
    _Pragma("unroll") for (int j = 0; j < UNROLL; j++) \
      asm volatile(ASM : "+v"(acc[j]) : "v"(bb), "v"(a) : "vcc"); \
  } \
  u64 s = 0; \
  _Pragma("unroll") for (int j = 0; j < UNROLL; j++) s ^= acc[j]; \
KERNEL_END(s ^ (s >> 32))

SIMPLE64(k_lshrrev_b64,  "v_lshrrev_b64 %0, 29, %0")
SIMPLE64(k_ashrrev_i64,  "v_ashrrev_i64 %0, 29, %0")
SIMPLE64(k_mad_i64_i32,  "v_mad_i64_i32 %0, vcc, %2, %2, %0")
SIMPLE64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
SIMPLE64(k_fma_f64,      "v_fma_f64 %0, %1, %1, %0")
SIMPLE64(k_mul_f64,      "v_mul_f64 %0, %0, %1")
SIMPLE64(k_add_f64,      "v_add_f64 %0, %0, %1")
SIMPLE64(k_pk_fma_f32,   "v_pk_fma_f32 %0, %1, %1, %0")
SIMPLE32(k_pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %0")

typedef void (*kern_t)(u32*, u64*, u32);
struct Entry { const char* name; kern_t k; int ops_per_asm; };

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs=%d clockRate=%d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_add_co_u32", k_add_co_u32, 1}, {"v_addc_co_u32", k_addc_co_u32, 1},
    {"v_add3_u32", k_add3_u32, 1}, {"v_and_b32", k_and_b32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1},
    {"v_alignbit_b32", k_alignbit_b32, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_fma_f32", k_fma_f32, 1}, {"v_fmac_f32", k_mac_f32, 1}, {"v_add_f32", k_add_f32, 1}, {"v_mul_f32", k_mul_f32, 1},
    {"v_mov_b32", k_mov_b32, 1}, {"v_xor_b32", k_xor_b32, 1},
    {"v_mad_i64_i32", k_mad_i64_i32, 1}, {"v_ashrrev_i64", k_ashrrev_i64, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_u64_u32(dep chain)", k_mad_u64_u32_dep, 1},
    {"v_mad_u64_u32+v_addc", k_mad_u64_u32_pair, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
    {"v_mad_u32_u16", k_mad_u32_u16, 1}, {"v_dot2_u32_u16", k_dot2_u32_u16, 1}, {"v_dot4_u32_u8", k_dot4_u32_u8, 1},
    {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
    {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1},
  };
  u32* d_out; u64* d_cyc;
  size_t maxthreads = (size_t)cus * 8 * 256;
  CK(hipMalloc(&d_out, maxthreads * 4)); CK(hipMalloc(&d_cyc, maxthreads / 64 * 16));
  std::vector<u64> h_cyc(maxthreads / 64 * 2);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // cyc/instr/SIMD(wave) = shader cycles one wave spent per instruction (s_memtime) / waves sharing its SIMD: the issue cost
  // at the MEASURED clock (s_memtime / s_memrealtime x 100 MHz), independent of any assumed frequency
  printf("%-28s %5s %10s %9s %20s %22s %14s\n", "instr", "w/SIMD", "wall_ms", "clk_GHz", "cyc/instr/wave", "cyc/instr/SIMD(=wave/w)", "Glaneops/s");
  for (auto& e : es) {
    for (int wps : {1, 2, 4}) {
      int blocks = cus * wps;   // 256 threads = 4 waves = 1 wave per SIMD per block
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 12345u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h_cyc.data(), d_cyc, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost));
      double avg = 0, avr = 0; for (int i = 0; i < blocks * 4; i++) { avg += (double)h_cyc[2 * i]; avr += (double)h_cyc[2 * i + 1]; } avg /= blocks * 4; avr /= blocks * 4;
      double n_instr = (double)ITERS * UNROLL;            // per wave
      // s_memtime ticks at 100MHz-ish const clock on some parts; report wall-derived too
      double winstr = n_instr * blocks * 4;               // wave-instructions total
      double ginstr = winstr * 64 / (ms * 1e-3) / 1e9;    // lane-ops/s
      // per-SIMD cycles per wave-instr (all wps waves interleaved): wall * clk / (n_instr * wps)
      double clk = avr > 0 ? avg / avr * 0.1 : 0.0;   // GHz
      printf("%-28s %5d %10.4f %9.3f %20.2f %22.2f %14.1f\n", e.name, wps, ms, clk, avg / n_instr, avg / n_instr / wps, ginstr);
    }
  }
  return 0;
}
