// Latency of ONE XYZZ + XYZZ addition of the lazy field as the tail kernels run it: a wave doing a chain of
// dependent additions, (a) every lane active, (b) only lane 0 taking the addition's body (the rest meet the
// early identity return, as in the last steps of a shuffle tree), (c) with the 36-limb shuffle of a tree step in
// front of every addition, (d) with a 160-byte point load in front of every addition.  Reports cycles
// (s_memtime) and wall time (s_memrealtime, 100 MHz) per addition for 1 wave per SIMD on `blocks` CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../halo2_liam_eagen_msm_amd/csrc/xyzz29.cuh"
using namespace lemsm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef XYZZ29<Field29<Fq29Params>> G;

template <int MODE>
__global__ __launch_bounds__(256) void k_chain(const char* __restrict__ pts, int iters, char* __restrict__ out, unsigned long long* cyc, unsigned flagmask, unsigned long long* ts, const char* __restrict__ big, unsigned big_points, char* __restrict__ bigout) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u;
  G::pt acc, q0; G::load(acc, pts + (size_t)gid * G::PT_BYTES); G::load(q0, pts + (size_t)(gid + 1) * G::PT_BYTES);
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int i = 0; i < iters; i++) {
    G::pt q;
    if (MODE == 0) q = q0;
    if (MODE == 1) { if (lane == 0) q = q0; else G::set_identity(q); }
    if (MODE == 2) { G::scan_fetch(q, acc, 0); if (lane != 0) G::set_identity(q); }
    if (MODE == 3) G::load(q, pts + (size_t)((gid * 7u + i * 64u) & 0xffffu) * G::PT_BYTES);
    if (MODE == 4) { G::load(q, pts + (size_t)((gid * 7u + i * 64u) & 0xffffu) * G::PT_BYTES); if ((lane & flagmask) != 0) G::set_identity(q); }   // runtime mask: lane 0 only when flagmask = 63
    if (MODE == 5) { q = q0; if ((lane & flagmask) != 0) G::set_identity(q); }
    if (MODE == 6) { G::scan_fetch(q, acc, 0); }
    if (MODE == 7 || MODE == 8) {
      // one pyramid item per step: two points in, one out, on cold data (big, each step a fresh region); 7 = AoS (a lane's
      // point is 160 contiguous bytes), 8 = the same points interleaved by 64 (word k of the wave's 64 points contiguous)
      const size_t region = ((size_t)i * 262144u + (size_t)(gid >> 6) * 128u) % ((size_t)big_points - 256u);   // 128 points per wave-step
      G::pt a2;
      if (MODE == 7) {
        G::load(acc, big + (region + 2u * lane) * G::PT_BYTES); G::load(a2, big + (region + 2u * lane + 1u) * G::PT_BYTES);
      } else {
        const uint4* b0 = reinterpret_cast<const uint4*>(big + region * G::PT_BYTES);
        uint4 r0[9], r1[9];
#pragma unroll
        for (int k = 0; k < 9; k++) { r0[k] = b0[(size_t)k * 64 + lane]; r1[k] = b0[(size_t)(10 + k) * 64 + lane]; }
        G::from_raw(acc, r0); G::from_raw(a2, r1);
      }
      q = a2;
    }                                            // butterfly step: every lane adds its partner's sum
    G::add(acc, q);
    if (MODE == 7) G::store(bigout + ((size_t)i * 65536u + gid) * G::PT_BYTES, acc);
    if (MODE == 8) {
      uint4* o = reinterpret_cast<uint4*>(bigout + ((size_t)i * 65536u + (gid & ~63u)) * G::PT_BYTES);
      u32 wv[36];
#pragma unroll
      for (int k = 0; k < 9; k++) { wv[k] = (u32)acc.x.l[k]; wv[9 + k] = (u32)acc.y.l[k]; wv[18 + k] = (u32)acc.zz.l[k]; wv[27 + k] = (u32)acc.zzz.l[k]; }
#pragma unroll
      for (int k = 0; k < 9; k++) o[(size_t)k * 64 + lane] = make_uint4(wv[4 * k], wv[4 * k + 1], wv[4 * k + 2], wv[4 * k + 3]);
    }
    if (ts && gid == 0) ts[i] = __builtin_amdgcn_s_memtime();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  G::store(out + (size_t)gid * G::PT_BYTES, acc);
  if (lane == 0) { cyc[2 * (gid >> 6)] = t1 - t0; cyc[2 * (gid >> 6) + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 16;
  const size_t NP = 1 << 18;   // >= 768 blocks x 256 threads + 1
  std::vector<int> h(NP * 40);
  srand(7);
  for (size_t i = 0; i < h.size(); i++) h[i] = (i % 40 >= 36) ? 0 : (int)(((unsigned)rand() * 2654435761u) & ((1u << 29) - 1));
  for (size_t p = 0; p < NP; p++) for (int f = 0; f < 4; f++) h[p * 40 + f * 9 + 8] &= 0x1fffff;   // top limb small: |V| < 8N
  char *d_p, *d_o; unsigned long long* d_c;
  CK(hipMalloc(&d_p, NP * 160)); CK(hipMalloc(&d_o, NP * 160)); CK(hipMalloc(&d_c, 1 << 16));
  CK(hipMemcpy(d_p, h.data(), NP * 160, hipMemcpyHostToDevice));
  const char* names[9] = {"all lanes add", "lane 0 adds, others identity", "shuffle + lane-0 add (tree step)", "load + add (serial merge step)", "load + add, lane 0 only (runtime mask)", "register q, lanes & mask == 0 add", "butterfly: shuffle(lane ^ 1) + all lanes add", "cold AoS: 2 loads + add + store", "cold interleaved-by-64: 2 loads + add + store"};
  unsigned long long* d_ts; CK(hipMalloc(&d_ts, 8 * 64));
  const unsigned BIGP = 1u << 22;   // 4M points = 640 MB: cold for every step
  char *d_big, *d_bigout; CK(hipMalloc(&d_big, (size_t)BIGP * 160)); CK(hipMalloc(&d_bigout, (size_t)65536 * 160 * 64));
  for (size_t off = 0; off < (size_t)BIGP * 160; off += NP * 160) CK(hipMemcpy(d_big + off, d_p, NP * 160, hipMemcpyDeviceToDevice));
  for (int blocks : {256, 768}) {
    for (int mode : {0, 3, 7, 8}) for (unsigned fm : {63u}) {
      std::vector<unsigned long long> c(2 * blocks * 4);
      for (int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        unsigned long long* tsp = (iters <= 64) ? d_ts : nullptr;
        if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, 63u, tsp, d_big, BIGP, d_bigout);
        if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, 63u, tsp, d_big, BIGP, d_bigout);
        if (mode == 2) hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, 63u, tsp, d_big, BIGP, d_bigout);
        if (mode == 3) hipLaunchKernelGGL(k_chain<3>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, 63u, tsp, d_big, BIGP, d_bigout);
        if (mode == 4) hipLaunchKernelGGL(k_chain<4>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, fm, tsp, d_big, BIGP, d_bigout);
        if (mode == 5) hipLaunchKernelGGL(k_chain<5>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, fm, tsp, d_big, BIGP, d_bigout);
        if (mode == 7) hipLaunchKernelGGL(k_chain<7>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, fm, tsp, d_big, BIGP, d_bigout);
        if (mode == 8) hipLaunchKernelGGL(k_chain<8>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, fm, tsp, d_big, BIGP, d_bigout);
        if (mode == 6) hipLaunchKernelGGL(k_chain<6>, dim3(blocks), dim3(256), 0, 0, d_p, iters, d_o, d_c, fm, tsp, d_big, BIGP, d_bigout);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost));
        if (rep == 2) {
          double cy = 0, tk = 0; for (int w = 0; w < blocks * 4; w++) { cy += c[2 * w]; tk += c[2 * w + 1]; }
          cy /= blocks * 4; tk /= blocks * 4;
          printf("blocks %3d  %-34s  %8.0f cycles/add  %6.2f us/add (in-kernel)  clock %.2f GHz  kernel %.1f us by events\n", blocks, names[mode], cy / iters,
                 tk / 100.0 / iters, cy / (tk * 10.0), ms * 1e3);
          if (tsp && blocks == 1) {
            std::vector<unsigned long long> t(iters); CK(hipMemcpy(t.data(), d_ts, iters * 8, hipMemcpyDeviceToHost));
            printf("      per-iteration cycles (wave 0):");
            for (int i = 1; i < iters; i++) printf(" %llu", t[i] - t[i - 1]);
            printf("\n");
          }
        }
      }
    }
  }
  return 0;
}
