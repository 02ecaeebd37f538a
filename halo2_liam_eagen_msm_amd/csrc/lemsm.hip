// C ABI implementation (include/lemsm.h): contexts, workspace, launch plans, the kernel
// pipeline and the host-side tail.  gfx950 only; there is no CPU path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "../../include/lemsm.h"
#include "hostmath.hpp"
#include "hostpool.hpp"
#include "hosttail.hpp"
#include <memory>
#include "kernels.cuh"
#include "rccl_dyn.hpp"
#include "roctx_dyn.hpp"
#include "divisor.cuh"
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <functional>
#include <thread>

using namespace lemsm;

namespace {

typedef Field32<FqParams> FqDev;   // BN254 G1 coordinates (strict 32-bit limbs: ABI-facing kernels)
typedef Field32<FrParams> FrDev;   // Grumpkin coordinates
typedef XYZZ<FqDev> GqStrict;      // point arithmetic of the pipeline, strict field (A/B + reference)
typedef XYZZ<FrDev> GrStrict;
typedef XYZZ29<Field29<Fq29Params>> GqLazy;   // lazy radix-2^29 field: the default hot path
typedef XYZZ29<Field29<Fr29Params>> GrLazy;

// scalar-field orders as 8 x u32 (BN254 G1: r, Grumpkin: p) and isqrt(order)+2
const u32 ORDER_R[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
const u32 ORDER_P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
// isqrt(r)+2 = 0x6f4d8248eeb859fcc6fb4e9fc7b81b1b ; isqrt(p)+2 = 0x6f4d8248eeb859fcc6fb4e9fc7b81b1c
const u32 BOUND_R[8] = {0xc7b81b1bu, 0xc6fb4e9fu, 0xeeb859fcu, 0x6f4d8248u, 0, 0, 0, 0};
const u32 BOUND_P[8] = {0xc7b81b1cu, 0xc6fb4e9fu, 0xeeb859fcu, 0x6f4d8248u, 0, 0, 0, 0};

const u32* order_of(int curve) { return curve == LEMSM_BN254_G1 ? ORDER_R : ORDER_P; }
const u32* bound_of(int curve) { return curve == LEMSM_BN254_G1 ? BOUND_R : BOUND_P; }

}  // namespace
// heavy kernels are instantiated in inst_*.hip
#define LEMSM_EXTERN_G(G)                                                                                   \
  extern template __global__ void lemsm::k_merge_pairs<G>(GroupPlan, u32, MqLayout, const u32*, const u32*, const u32*, const char*, char*, u32*, uint4*); \
  extern template __global__ void lemsm::k_merge_serial<G>(u32, MqLayout, const u32*, const uint4*, const char*, char*); \
  extern template __global__ void lemsm::k_merge_waves<G>(u32, MqLayout, const u32*, const uint4*, const char*, char*, char*, u32*); \
  extern template __global__ void lemsm::k_merge_final<G>(u32, MqLayout, const u32*, const uint4*, const char*, char*, u32*); \
  extern template __global__ void lemsm::k_pyramid<G>(const PyrTask*, u32, u32, u32, char*, u32); \
  extern template __global__ void lemsm::k_pyramid_first2<G>(PyrFirst2Args, const u32*, char*, u32*); \
  extern template __global__ void lemsm::k_pyramid_tail<G>(const PyrTask*, PyrTailArgs, const CopyTaskPod*, u32, char*, u32);
#define LEMSM_EXTERN_ACC(G, W) \
  extern template __global__ void lemsm::k_accum1<G, W>(GroupPlan, const u32*, const u32*, u32*, const uint4*, char*, u32*, char*);
LEMSM_EXTERN_G(GqStrict) LEMSM_EXTERN_G(GrStrict) LEMSM_EXTERN_G(GqLazy) LEMSM_EXTERN_G(GrLazy)
LEMSM_EXTERN_ACC(GqStrict, 4) LEMSM_EXTERN_ACC(GrStrict, 4)
LEMSM_EXTERN_ACC(GqLazy, 2) LEMSM_EXTERN_ACC(GqLazy, 3) LEMSM_EXTERN_ACC(GqLazy, 4)
LEMSM_EXTERN_ACC(GrLazy, 2) LEMSM_EXTERN_ACC(GrLazy, 3) LEMSM_EXTERN_ACC(GrLazy, 4)
extern template __global__ void lemsm::k_accum1<GqLazy, 3, true>(GroupPlan, const u32*, const u32*, u32*, const uint4*, char*, u32*, char*);
extern template __global__ void lemsm::k_accum1<GrLazy, 3, true>(GroupPlan, const u32*, const u32*, u32*, const uint4*, char*, u32*, char*);
extern template __global__ void lemsm::k_accum1<GqLazy, 3, true, true>(GroupPlan, const u32*, const u32*, u32*, const uint4*, char*, u32*, char*);
extern template __global__ void lemsm::k_accum1<GrLazy, 3, true, true>(GroupPlan, const u32*, const u32*, u32*, const uint4*, char*, u32*, char*);
extern template __global__ void lemsm::k_convert_points<Field29<Fq29Params>>(const uint4*, uint4*, u32);
extern template __global__ void lemsm::k_convert_points<Field29<Fr29Params>>(const uint4*, uint4*, u32);
namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};


struct PyrCacheEntry { DevBuf buf; PyrPlan pp; };   // task tables resident on the device

}  // namespace

// host-pointer entries: where one call's inputs live on the host and where their slabs are staged in HBM
struct HostStage { const uint8_t* h_scalars; const void* h_points; char* d_scalars; char* d_points; };

struct lemsm_ctx {
  int device = 0;
  u32 num_cus = 256;                   // hipDeviceProp_t::multiProcessorCount (grids of the one-block-per-CU kernels)
  hipStream_t stream = nullptr;        // accumulate stream (and everything single-stream)
  hipStream_t stream_sort = nullptr;   // digit + sort passes of the next window group (high priority)
  hipStream_t stream_tail = nullptr;   // edge-record levels + pyramid of the previous group
  std::vector<hipEvent_t> evpool;
  hipEvent_t wait_accum = nullptr;               // batch entries: the other lane's accumulation must have ended before this call's starts (run_group)
  hipEvent_t ev_call_done = nullptr;             // batch entries: everything this lane enqueued for its call has run
  hipEvent_t ev_batch_up = nullptr;              // lemsm_msm_batch_with_bases: this lane's scalars have arrived
  hipEvent_t dw_ev[2] = {nullptr, nullptr};      // divisor witness: the two half-level pointwise chains (divisor_abi.inc)
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // [4]: the digit pass's error words have reached the host
  DevBuf ws;        // workspace arena
  DevBuf in_s;      // staged scalars (host-pointer entries)
  DevBuf in_p;      // staged points
  DevBuf in_aux;    // staged Jacobian points / misc
  DevBuf gather;    // multi-GPU: all ranks' raw per-window records after the all-gather
  DevBuf fail_buf;  // multi-GPU: the zeroed stand-in a rank sends when its own pipeline failed before the exchange
  DevBuf dw_tab, dw_arena, dw_tmp;   // divisor witness: twiddle / coset tables, level workspace, tmp point list
  u32 dw_logn = 0, dw_gexp = 0;      // tables hold transforms up to 2^dw_logn with coset generator 7^dw_gexp
  double dw_ntt_ms = 0; u64 dw_ntt_bytes = 0, dw_ntt_bflies = 0; u32 dw_reuse_levels = 0;
  double dw_phase_ms[4] = {0, 0, 0, 0};   // lhs witness: MSM core, point lists, merge forest, coefficient download
  ncclComm_t comm = nullptr; int comm_size = 1, comm_rank = 0;   // lemsm_comm_init
  int plan_world = 1;   // ranks sharing the current call's windows: > 1 pins 16-bit windows (16 split evenly over 2/4/8 ranks, 15 do not)
  u32* h_small = nullptr;                        // 256 pinned bytes: error words of the negabase digit pass
  lemsm_ctx* peer = nullptr;                      // second lane of lemsm_msm_batch_device: own queues, workspace and pinned buffer on the same device
  void* h_pin = nullptr; size_t h_pin_cap = 0;   // pinned staging for the read-back of one call's records (one async copy, no pageable bounce)
  std::unique_ptr<lemsm::host::Pool> pool;        // host tail: per-window work of one call (hostpool.hpp); created on first use
  std::string last_error;
  long opt_host_threads = 0;
  long opt_window_bits = 0, opt_chunk = 0, opt_tile = 0, opt_field = 0, opt_accum_waves = 0, opt_groups = 0, opt_host_slab_bits = 0, opt_slab_bits = 0, opt_abi_points = 0, opt_stage2x = 0, opt_xcd_windows = 0, opt_entry_ring = 0, opt_validate_points = 0, opt_pyr_fuse = 0, opt_ntt_tiled = 0, opt_ws_canary = 0, opt_binsort = 0, opt_dw_wrap = 0, opt_dw_fuse = 0, opt_dw_kb = 0, opt_merge_slice = 0, opt_merge_wave_th = 0, opt_dbg_repeat = 0, opt_pyr_first2 = 0, opt_dw_reuse = 0, opt_dw_pw_lazy = 0, opt_dw_halves = 0, opt_dw_ntt_lazy = 0, opt_slab_tail = 0, opt_pyr_quad = 0, opt_scatter_lean = 0;
  u32 plan_slab_n = 0;                            // choose_lb: points of a FULL slab of the running call (every slab of a call, the ragged last one too, uses the same bin geometry)
  bool plan_ring = false;                         // make_group_plan: round the accumulate chunk to the entry ring's 16-entry blocks
  const struct HostStage* host_stage = nullptr;   // set by the host-pointer entries for the duration of one call
  double t_total_ms = 0, t_accum_ms = 0; int n_accum = 0;
  double host_us[4] = {0, 0, 0, 0};                // last call, host tail: wait for the device, record conversion, window sums, final Horner
  uint32_t dbg_stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // last group's in-kernel cycle stamps of the merge wave kernels (debug)
  uint64_t dbg_merge[4] = {0, 0, 0, 0};            // last call: buckets the edge-record merge queued as short / medium / long slices / multi-slice
  double accum_clock_mhz = 0;                     // shader clock the accumulate kernel of the last call sustained (in-kernel stamps)
  size_t bad_index = 0;
  size_t truncated = 0;                          // scalars of the last negabase pass whose expansion needed more than d digits
  std::map<std::vector<u32>, PyrCacheEntry> pyr_cache;   // keyed by (NBpad, nb, nbw, nbp, L, gw)
};

namespace {

#define HIPCHK(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                \
      return LEMSM_ERR_HIP;                                                                 \
    }                                                                                       \
  } while (0)

int fail(lemsm_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->last_error = msg;
  return code;
}

int reserve(lemsm_ctx* ctx, DevBuf& b, size_t bytes) {
  if (b.cap >= bytes) return LEMSM_OK;
  if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
  size_t want = bytes + bytes / 8 + 4096;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { b.p = nullptr; return fail(ctx, LEMSM_ERR_NOMEM, "hipMalloc failed: " + std::string(hipGetErrorString(e))); }
  b.cap = want;
  return LEMSM_OK;
}

// pinned host staging buffer of the context
int reserve_pinned(lemsm_ctx* ctx, size_t bytes) {
  if (ctx->h_pin_cap >= bytes) return LEMSM_OK;
  if (ctx->h_pin) { HIPCHK(ctx, hipHostFree(ctx->h_pin)); ctx->h_pin = nullptr; ctx->h_pin_cap = 0; }
  size_t want = bytes + bytes / 4 + 4096;
  hipError_t e = hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault);
  if (e != hipSuccess) { ctx->h_pin = nullptr; return fail(ctx, LEMSM_ERR_NOMEM, "hipHostMalloc failed: " + std::string(hipGetErrorString(e))); }
  ctx->h_pin_cap = want;
  return LEMSM_OK;
}

// host tail in parallel: jobs are independent pieces of ~10 us (one window's records); option "host_threads":
// 0 = auto (up to 8 threads including the caller, never more than the CPUs this process may use), 1 = serial
void host_parallel(lemsm_ctx* ctx, int njobs, const std::function<void(int)>& fn) {
  if (!ctx || njobs <= 1 || ctx->opt_host_threads == 1) { for (int i = 0; i < njobs; i++) fn(i); return; }
  int want = ctx->opt_host_threads > 1 ? (int)ctx->opt_host_threads : std::min(8, lemsm::host::Pool::usable_cpus());
  if (!ctx->pool || ctx->pool->workers() != want - 1) ctx->pool.reset(new lemsm::host::Pool(std::max(0, want - 1)));
  ctx->pool->run(njobs, fn);
}

// ---- big-integer helpers on 8 x u32 (host) ----
bool geq8(const u32* a, const u32* b) {
  for (int i = 7; i >= 0; i--) { if (a[i] > b[i]) return true; if (a[i] < b[i]) return false; }
  return true;
}
void add8(u32* r, const u32* a, const u32* b) {
  u64 c = 0;
  for (int i = 0; i < 8; i++) { c += (u64)a[i] + b[i]; r[i] = (u32)c; c >>= 32; }
}
u32 logb_ceil8(const u32* x, u32 base) {   // src/argument_witness_calc.rs:32-40
  u32 m[8]; memcpy(m, x, 32); u32 i = 0;
  for (;;) {
    u32 nz = 0; for (int k = 0; k < 8; k++) nz |= m[k];
    if (!nz) break;
    u64 rem = 0;
    for (int k = 7; k >= 0; k--) { u64 cur = (rem << 32) | m[k]; m[k] = (u32)(cur / base); rem = cur % base; }
    i++;
  }
  return i;
}
u32 next_pow2(u32 x) { u32 p = 1; while (p < x) p <<= 1; return p; }
u32 ilog2(u32 x) { u32 l = 0; while ((1u << l) < x) l++; return l; }

// ---- MSM plan (Pippenger, signed windows) ----
struct MsmPlan {
  u32 c, W, nb, nbp, L;
  u32 kadd[8];
};

// window bits for an n-point MSM: 16 for large n (2-byte windows, 16 windows of 2^15 buckets
// for 254-bit scalars: divisible by 1/2/4/8 GPUs); fewer for small n so buckets are not mostly empty.
u32 choose_c(const lemsm_ctx* ctx, size_t n) {
  if (ctx && ctx->opt_window_bits >= 2 && ctx->opt_window_bits <= 17) return (u32)ctx->opt_window_bits;
  u32 lg = 0; while (((size_t)1 << (lg + 1)) <= n) lg++;
  // 17-bit windows (15 windows of 2^16 buckets, 512 coarse bins each, one window group) from 2^24
  // points per slab: 6 % less accumulation, 0.5 ms more sort + tail -> 3 % faster end to end at
  // 2^24, a wash below (profiles/r01/p_c16_vs_c17_one_group.txt).  Window-sharded multi-GPU runs
  // pin 16 (bench.py sets window_bits = 16 for them): 16 windows split evenly over 2/4/8 ranks, 15 do not.
  if (lg >= 24) return (ctx && ctx->plan_world > 1) ? 16u : 17u;
  int c = (int)lg - 3;
  if (c < 3) c = 3;
  if (c > 16) c = 16;
  return (u32)c;
}

// host-pointer entries: log2 of the slab of pairs uploaded while the previous slab is accumulated.
// auto: a quarter of the input, between 2^19 (below that the fixed ~0.5 ms tail of a slab costs more
// than the overlap hides) and 2^21 (measured best at 2^24: profiles/r01/n_host_pointer_slab_pipeline.txt)
u32 host_slab_log(const lemsm_ctx* ctx, size_t n) {
  u32 lg = 0; while (((size_t)2 << lg) <= n) lg++;   // floor(log2 n)
  u32 auto_log = std::min(21u, std::max(19u, lg >= 2 ? lg - 2 : 0u));
  // resident bases (lemsm_msm_with_bases): only 32 B per pair cross PCIe, the upload of a slab takes half the time its
  // kernels do, so fewer, larger slabs win (each slab pays its own sort + tail): 2^22 measured best at 2^24
  // (profiles/r02/m_host_path_2p24.txt)
  if (ctx->host_stage && !ctx->host_stage->h_points) auto_log = std::min(22u, std::max(19u, lg >= 2 ? lg - 2 : 0u));
  return std::min<u32>(MAX_SLAB_LOG, ctx->opt_host_slab_bits ? (u32)ctx->opt_host_slab_bits : auto_log);
}

MsmPlan make_msm_plan(const lemsm_ctx* ctx, int curve, size_t n) {
  MsmPlan p; memset(&p, 0, sizeof p);
  // the window width follows the number of pairs one pass of the pipeline sees (a slab), not the call's total
  size_t n_pass = n;
  if (ctx && ctx->host_stage) n_pass = std::min(n, (size_t)1 << host_slab_log(ctx, n));
  else if (ctx && ctx->opt_slab_bits) n_pass = std::min(n, (size_t)1 << ctx->opt_slab_bits);
  p.c = choose_c(ctx, n_pass);
  p.nb = 1u << (p.c - 1);
  p.nbp = p.nb; p.L = ilog2(p.nbp);
  const u32* order = order_of(curve);
  // smallest W such that (order-1 + K_W) >> (c(W-1)) <= nb, K_W = sum_{w<W-1} 2^(c-1) 2^(cw)
  for (u32 W = (254 + p.c - 1) / p.c;; W++) {
    u32 K[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 w = 0; w + 1 < W; w++) {
      u32 bit = w * p.c + p.c - 1;
      if (bit < 256) K[bit >> 5] |= 1u << (bit & 31);
    }
    u32 s[9]; u64 cy = 0;
    for (int i = 0; i < 8; i++) { cy += (u64)order[i] + K[i]; s[i] = (u32)cy; cy >>= 32; }
    s[8] = (u32)cy;
    // top = s >> (c (W-1))
    u32 sh = p.c * (W - 1);
    bool fits = true;
    if (s[8]) fits = false;   // overflowed 256 bits (cannot happen for 254-bit orders)
    // compute top window value (may be up to 64 bits wide if W too small)
    u64 top = 0; bool big = false;
    for (int bit = 255; bit >= (int)sh; bit--) {
      u32 b = (s[bit >> 5] >> (bit & 31)) & 1u;
      if (top >> 62) big = true;
      top = (top << 1) | b;
    }
    if (big || top > p.nb) fits = false;
    if (sh >= 256) { fits = true; }
    if (fits) { p.W = W; memcpy(p.kadd, K, 32); break; }
  }
  if (p.c == 17 && p.W != 15) {   // the 17-bit digit kernel is written for 15 windows (254-bit orders)
    lemsm_ctx tmp_opts; tmp_opts.opt_window_bits = 16;
    return make_msm_plan(&tmp_opts, curve, n);
  }
  return p;
}

template <class F> struct FieldTag {};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Workspace carve for one window group.
struct GroupWs {
  u32* err;             // [0] = count of non-canonical scalars, [1] = ~(smallest offending index); first word of the group workspace
  char* arena;          // points arena (bucket sums, pyramid, out)
  u32* bin_total; u32* bin_cursor; u32* bin_start; u32* tile_prefix; u32* meta;
  u32* bucket_count; u32* bucket_cursor; u32* bucket_start;
  u32* block_counts;
  uint16_t* dig16;
  unsigned long long* signbm;
  uint4* tile_info;
  u32* entries; u32* sorted;
  u32* rec_key; char* rec_pt;      // k_accum1's pieces: the bucket a chunk owns, and its first / last partial sums (2 slots per chunk)
  u32* mq_cnt; uint4* mq_items; char* mq_partial; MqLayout mq;   // merge queues (kernels_ec.cuh)
  size_t zero_begin, zero_bytes;   // contiguous region to memset(0) per group (offset from base)
  size_t zero_bytes_counters;      // ... its leading part: everything but the bucket sums
  size_t total;
  std::vector<size_t> guards;      // option ws_canary: offsets of the 256-byte guard behind every sub-buffer
};

MqLayout make_mq_layout(const lemsm_ctx* ctx, u32 nthr1) {
  return make_mq_layout_t(nthr1, ctx && ctx->opt_merge_slice ? (u32)ctx->opt_merge_slice : 512u,
                          ctx && ctx->opt_merge_wave_th ? (u32)ctx->opt_merge_wave_th - 1u : 2048u);
}

const size_t WS_GUARD = 256;
GroupWs carve(char* base, const GroupPlan& pl, const ArenaLayout& ar, const MqLayout& mq, size_t ptb, bool guard = false) {
  GroupWs w; size_t off = 0;
  // option ws_canary (debug / fuzz): every sub-buffer is followed by a guard that run_group fills with a pattern and
  // checks when the group is done -- an overrun of ANY sub-buffer shows up, not only one past the group's end
  auto take = [&](size_t bytes) {
    size_t o = off;
    if (guard) { size_t g = align_up(off + bytes, 16); w.guards.push_back(g); off = align_up(g + WS_GUARD, 256); }
    else off = align_up(off + bytes, 256);
    return o;
  };
  u32 NBpad = pl.nbins << pl.LB;
  // zeroed region first: bin_total, bin_cursor, bucket_count, bucket_cursor, bucket sums
  size_t z0 = off;
  size_t o_err = take(64);
  size_t o_mqcnt = take(MQ_WORDS * 4);
  size_t o_bin_total = take(MAX_BINS * 4), o_bin_cursor = take(MAX_BINS * 4);
  size_t o_bcount = take((size_t)NBpad * 4), o_bcursor = take((size_t)NBpad * 4);
  size_t o_arena = take((size_t)ar.total_points * ptb);
  size_t zend = o_arena + (size_t)(ar.apyr_off - ar.bucket_off) * ptb;   // only the bucket sums need zeroing (every slab's area when the slabs share one tail)
  w.zero_bytes_counters = o_arena - z0;
  size_t o_bin_start = take((MAX_BINS + 1) * 4), o_tile_prefix = take((MAX_BINS + 1) * 4), o_meta = take(64);
  size_t o_bstart = take(((size_t)NBpad + 1) * 4);
  size_t o_blockc = take((size_t)pl.nblk1 * pl.nbins * 8);   // per (window, range, bin): counts, then the offsets claimed inside the bin
  size_t o_dig = take(pl.c ? (size_t)pl.dstride * (pl.w1 - pl.w0) * 2 + 64 : 16);
  size_t o_tinfo = take((size_t)pl.max_tiles * 16 + 16);
  size_t o_sbm = take(pl.c == 17 ? (size_t)(pl.w1 - pl.w0) * ((pl.n + 63) / 64) * 8 + 16 : 16);
  size_t Mmax = (size_t)pl.n * (pl.w1 - pl.w0);
  size_t o_entries = take(Mmax * 4 + 4 * (size_t)pl.L1 + 512), o_sorted = take(Mmax * 4 + 4 * (size_t)pl.L1 + 512);   // (+ slack: the entry ring stages whole 64-byte blocks)
  w.mq = mq;
  size_t o_rk = take((size_t)pl.nthr1 * 4 + 16), o_rp = take(2 * (size_t)pl.nthr1 * ptb + 256);
  size_t o_mqi = take(((size_t)mq.offF + mq.capF) * 16), o_mqp = take((size_t)mq.capP * ptb + 256);
  w.total = off;
  if (base) {
    w.err = (u32*)(base + o_err);
    w.bin_total = (u32*)(base + o_bin_total); w.bin_cursor = (u32*)(base + o_bin_cursor);
    w.bucket_count = (u32*)(base + o_bcount); w.bucket_cursor = (u32*)(base + o_bcursor);
    w.arena = base + o_arena;
    w.bin_start = (u32*)(base + o_bin_start); w.tile_prefix = (u32*)(base + o_tile_prefix); w.meta = (u32*)(base + o_meta);
    w.bucket_start = (u32*)(base + o_bstart); w.block_counts = (u32*)(base + o_blockc); w.dig16 = (uint16_t*)(base + o_dig); w.tile_info = (uint4*)(base + o_tinfo); w.signbm = (unsigned long long*)(base + o_sbm);
    w.entries = (u32*)(base + o_entries); w.sorted = (u32*)(base + o_sorted);
    w.rec_key = (u32*)(base + o_rk); w.rec_pt = base + o_rp;
    w.mq_cnt = (u32*)(base + o_mqcnt); w.mq_items = (uint4*)(base + o_mqi); w.mq_partial = base + o_mqp;
  }
  w.zero_begin = z0; w.zero_bytes = zend - z0;
  return w;
}

// local bucket bits of a pass-1 entry: <= 256 coarse bins per window (512 at nb = 2^16) -- and 512 as well when 256 bins
// would hold more entries each (n / bins for uniform digits) than one k_binsort block can take: at 2^24 points and
// 16-bit windows (the window-sharded multi-GPU plan) that halves the bins to 2^15 entries and keeps pass 2 in one block
u32 choose_lb(const lemsm_ctx* ctx, u32 nb, u32 n, u32 d) {
  // one geometry per call: a ragged last slab must not pick MORE bins per window than the full slabs the window groups
  // were sized for (found by the fuzz soak with a 3-entry test capacity: 20 windows x 512 bins > MAX_BINS)
  if (ctx && ctx->plan_slab_n) n = ctx->plan_slab_n;
  u32 LB = 0;
  while (((nb + (1u << LB) - 1) >> LB) > 256 && LB < MAX_LB) LB++;
  if (d == 0 && ctx && ctx->opt_binsort != 2 && LB > 0) {
    const u32 cap = ctx->opt_binsort > 2 ? std::min((u32)ctx->opt_binsort, (u32)BIN_CAP) : (u32)BIN_CAP;
    const u32 BW = (nb + (1u << LB) - 1) >> LB;
    if ((u64)n / BW > (u64)cap * 9 / 10 && 2 * BW <= BW_MAX && (u64)n / (2 * BW) <= (u64)cap * 9 / 10) LB--;
  }
  return LB;
}

GroupPlan make_group_plan(const lemsm_ctx* ctx, u32 n, u32 c, u32 nb, u32 W, u32 w0, u32 w1, u32 d) {
  GroupPlan g; memset(&g, 0, sizeof g);
  g.n = n; g.c = c; g.nb = nb; g.W = W; g.w0 = w0; g.w1 = w1; g.d = d; g.nstride = n;   // run_windows sets nstride to the call's n
  u32 LB = choose_lb(ctx, nb, n, d);
  g.LB = LB;
  g.BW = (nb + (1u << LB) - 1) >> LB;
  g.nbw = g.BW << LB;
  g.NB = (w1 - w0) * g.nbw;
  g.nbins = (w1 - w0) * g.BW;
  u32 spb = (n + 1023) / 1024;
  spb = std::max(256u, std::min((u32)STAGE, spb));
  spb = (spb + 255) / 256 * 256;
  if (n >= (1u << 16)) spb = STAGE;     // long (block, bin) runs -> full-line writes
  if (n >= (1u << 16) && d == 0 && ctx->opt_stage2x == 4) spb = 4 * STAGE;   // 1024-thread pass-1 blocks staging 16384 entries: 128-byte runs at 512 bins per window
  if (n >= (1u << 16) && d == 0 && ctx->opt_stage2x == 2) spb = 2 * STAGE;   // A/B knob: 64-byte runs at 512 bins per window; measured slower (fewer resident blocks), profiles/r01/y_scatter1_staging_ab.txt
  g.spb = spb;
  g.dstride = (n + 63u) / 64u * 64u;
  g.nblk1 = (n + spb - 1) / spb;
  if (g.nblk1 == 0) g.nblk1 = 1;
  size_t Mmax = (size_t)n * (w1 - w0);
  u32 T2 = ctx->opt_tile > 0 ? std::min((u32)ctx->opt_tile, (u32)STAGE2) : (u32)STAGE2;
  g.T2 = T2;
  g.max_tiles = (u32)(Mmax / T2) + g.nbins + 1;
  g.bin_cap = ctx->opt_binsort == 2 ? 0u : (ctx->opt_binsort > 2 ? std::min((u32)ctx->opt_binsort, (u32)BIN_CAP) : (u32)BIN_CAP);
  // short bins (uniform digits put n / BW entries in each): the 32-KiB variant of k_binsort, with room for 1.5x the mean
  if (g.bin_cap == BIN_CAP && g.BW && (u64)n * 3 / 2 / g.BW <= BIN_CAP_SMALL) g.bin_cap = BIN_CAP_SMALL;
  u32 L1;
  if (ctx->opt_chunk > 0) L1 = (u32)ctx->opt_chunk;
  else {
    // two full rounds of the accumulate kernel's 3 waves per SIMD (256 CUs x 4 SIMDs x 3 x 64 lanes x 2):
    // measured best at 2^20..2^22 (profiles/r01/q_chunk_sweep.txt); from 2^23 the cap of 256 applies
    size_t target = (size_t)256 * 4 * 3 * 64 * 2;
    size_t l = (Mmax + target - 1) / target;
    L1 = (u32)std::max((size_t)8, std::min((size_t)256, l));
  }
  // the entry ring of k_accum1 stages 16-entry blocks: when that form will run (run_windows sets the flag once it
  // has chosen the ABI form), chunks are a multiple of 16 entries
  if (ctx->plan_ring && ctx->opt_chunk == 0 && L1 >= 128) L1 = (L1 + 15u) & ~15u;   // (short chunks: the ring's per-chunk prologue costs more than it saves: profiles/r01/zz_entry_ring_small_sizes.txt)
  g.L1 = L1;
  g.nthr1 = (u32)((Mmax + L1 - 1) / L1);
  if (g.nthr1 == 0) g.nthr1 = 1;
  return g;
}

// max windows per group: (w1-w0) * BW <= MAX_BINS
u32 max_group_windows(const lemsm_ctx* ctx, u32 nb, u32 n, u32 d) {
  u32 LB = choose_lb(ctx, nb, n, d);
  u32 BW = (nb + (1u << LB) - 1) >> LB;
  u32 cap = MAX_BINS / BW;
  return cap == 0 ? 1 : cap;
}

// digit providers: produce the decoder the sort passes read (bucket, sign) through
struct PipProvider {
  const uint4* scalars; KAdd kadd;
  typedef PipDec Dec;
  // digit columns + pass-1 counts in one kernel
  int prepare(lemsm_ctx*, hipStream_t st, const GroupPlan& pl, uint16_t* dig16, unsigned long long* signbm, u32* block_counts,
              u32* bin_total, u32* err, Dec& dec) const {
    dec.dig16 = dig16; dec.signbm = nullptr;
    if (pl.c == 17) {
      dec.signbm = signbm;
      if (pl.nblk1 < 2048)
        hipLaunchKernelGGL((k_pip_digits<2, 1024>), dim3(pl.nblk1), dim3(1024), 0, st, scalars, kadd, pl, dig16, signbm, block_counts, bin_total, err);
      else
        hipLaunchKernelGGL((k_pip_digits<2>), dim3(pl.nblk1), dim3(256), 0, st, scalars, kadd, pl, dig16, signbm, block_counts, bin_total, err);
    } else if (pl.c == 16 && pl.W == 16) {
      if (pl.nblk1 < 1024)   // too few ranges to fill 256 CUs with 256-thread blocks
        hipLaunchKernelGGL((k_pip_digits<1, 1024>), dim3(pl.nblk1), dim3(1024), 0, st, scalars, kadd, pl, dig16, signbm, block_counts, bin_total, err);
      else
        hipLaunchKernelGGL((k_pip_digits<1>), dim3(pl.nblk1), dim3(256), 0, st, scalars, kadd, pl, dig16, signbm, block_counts, bin_total, err);
    } else {
      if (pl.nblk1 < 1024)
        hipLaunchKernelGGL((k_pip_digits<0, 1024>), dim3(pl.nblk1), dim3(1024), 0, st, scalars, kadd, pl, dig16, signbm, block_counts, bin_total, err);
      else
        hipLaunchKernelGGL((k_pip_digits<0>), dim3(pl.nblk1), dim3(256), 0, st, scalars, kadd, pl, dig16, signbm, block_counts, bin_total, err);
    }
    return LEMSM_OK;
  }
};
struct NegProvider {
  const uint8_t* digitsT;
  typedef NegDec Dec;
  int prepare(lemsm_ctx*, hipStream_t st, const GroupPlan& pl, uint16_t*, unsigned long long*, u32* block_counts, u32* bin_total, u32*, Dec& dec) const {
    dec.digitsT = digitsT;
    hipLaunchKernelGGL((k_count1<NegDec>), dim3(pl.w1 - pl.w0, pl.nblk1), dim3(256), 0, st, dec, pl, block_counts, bin_total);
    return LEMSM_OK;
  }
};

// The group's results leave its (reused) workspace in ONE small launch: the per-window records -> the call's record area,
// the digit pass's error words, k_accum1's clock stamps and the merge queue lengths -> the group's 64-byte status slot.
__global__ __launch_bounds__(256) void k_group_finish(const uint4* __restrict__ out_area, uint4* __restrict__ d_out, u32 n16,
                                                      const u32* __restrict__ err, const u32* __restrict__ clock, const u32* __restrict__ mq_cnt,
                                                      u32* __restrict__ slot) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) d_out[i] = out_area[i];
  if (blockIdx.x == 0 && threadIdx.x < 24) {
    const u32 k = threadIdx.x;
    slot[k] = k < 2 ? err[k] : (k < 4 ? 0u : (k < 12 ? clock[k - 4] : (k < 16 ? mq_cnt[k - 12] : mq_cnt[k - 8])));   // [16..23] = debug stamps mq_cnt[8..15]
  }
}

// Runs one window group [w0,w1): sort + accumulate + reduce; results (gw x (L+1) XYZZ points)
// are left in the arena's out area and copied to d_out (device) + gslot.
template <class G, class Prov>
int run_group(lemsm_ctx* ctx, const Prov& prov, const GroupPlan& pl, u32 nbp, u32 L, const void* d_points,
              bool abi /* d_points are in the C ABI's domain: k_accum1<.., true>, scaled outputs */,
              char* ws_base, char* d_out /* device, gw*(L+1)*PT_BYTES */, u32* d_slot /* device, this (slab, group)'s 64-byte status slot */, hipStream_t s_sort, hipStream_t s_acc,
              hipStream_t s_tail, hipEvent_t ev_sorted, hipEvent_t ev_acc0, hipEvent_t ev_acc1,
              hipEvent_t ev_points /* null, or: the converted points become ready on another queue */ = nullptr,
              u32 slab_k = 0, u32 nba = 1 /* > 1: the slabs of the call share one tail -- slab_k accumulates into its own bucket area, the
              last one (slab_k == nba - 1) adds the areas up and runs the pyramid */, u32 scaled_mask = 0 /* bit k: slab k's sums are in the ABI form */) {
  // Three queues: the sort passes of this group may run while the previous group accumulates
  // (s_sort), the accumulate kernels of all groups run back to back (s_acc), and this group's
  // edge-record levels + pyramid overlap the next group's accumulation (s_tail).
  hipStream_t st = s_sort;
  u32 gw = pl.w1 - pl.w0;
  u32 NBpad = pl.nbins << pl.LB;
  ArenaLayout ar = make_arena(NBpad, nbp, gw, L, nba);
  const bool abi_pyr = nba > 1 ? false : abi;      // k_sum_slabs leaves plain sums
  const bool last_slab = slab_k + 1 == nba;
  // pyramid task tables: built and uploaded once per plan shape, then reused
  std::vector<u32> pkey = {NBpad, pl.nb, pl.nbw, nbp, L, gw, abi_pyr ? 1u : 0u, nba};
  auto pit = ctx->pyr_cache.find(pkey);
  if (pit == ctx->pyr_cache.end()) {
    PyrCacheEntry ent;
    ent.pp = make_pyr_plan(ar, pl.nb, pl.nbw, nbp, L, abi_pyr);
    std::vector<PyrTask> flat;
    for (auto& s : ent.pp.steps) flat.insert(flat.end(), s.begin(), s.end());
    size_t tb = align_up(flat.size() * sizeof(PyrTask), 256);
    int rcr = reserve(ctx, ent.buf, tb + sizeof(CopyTask) + 64); if (rcr) return rcr;
    HIPCHK(ctx, hipMemcpy(ent.buf.p, flat.data(), flat.size() * sizeof(PyrTask), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy((char*)ent.buf.p + tb, &ent.pp.copy, sizeof(CopyTask), hipMemcpyHostToDevice));
    pit = ctx->pyr_cache.emplace(pkey, std::move(ent)).first;
  }
  const PyrPlan& pp = pit->second.pp;
  size_t ntasks_total = 0;
  for (auto& s : pp.steps) ntasks_total += s.size();
  PyrTask* d_tasks = (PyrTask*)pit->second.buf.p;
  CopyTask* d_copy = (CopyTask*)((char*)pit->second.buf.p + align_up(ntasks_total * sizeof(PyrTask), 256));
  const size_t ptb = G::PT_BYTES;
  const MqLayout mq = make_mq_layout(ctx, pl.nthr1);
  GroupWs w = carve(ws_base, pl, ar, mq, ptb, ctx->opt_ws_canary != 0);
  // the kernels below size LDS arrays and workspace slots by these limits: never launch a plan that exceeds them
  if (pl.nbins > MAX_BINS || pl.BW > BW_MAX || pl.LB > MAX_LB || pl.spb > 4 * STAGE || (pl.c && pl.dstride < pl.n) || pl.T2 > STAGE2 || pl.bin_cap > BIN_CAP)
    return fail(ctx, LEMSM_ERR_HIP, "internal: window-group plan exceeds a kernel limit (bins " + std::to_string(pl.nbins) + ", bins per window " + std::to_string(pl.BW) + ")");

  // k_pyramid_first2 recognises empty buckets from bucket_start[]: with it only the counters are zeroed, not the bucket sums
  const bool first2 = L >= 5 && ctx->opt_pyr_first2 == 1 && nba == 1;
  RoctxRange rg_group("lemsm: window group (digits + sort + accumulate + tail enqueued)");
  // (shared tail: the first slab clears every slab's bucket area, the later ones only the counters)
  // (r03: clearing the bucket sums on the second queue beside the digit and sort passes was measured and dropped -- the
  // digit kernel slows down by what the memset takes, 29.6 -> 47.8 us at 2^20: profiles/r03/s_scatter_lean_and_clear_beside_ab.txt)
  HIPCHK(ctx, hipMemsetAsync(ws_base + w.zero_begin, 0, (first2 || slab_k > 0) ? w.zero_bytes_counters : w.zero_bytes, st));
  for (size_t g : w.guards) HIPCHK(ctx, hipMemsetAsync(ws_base + g, 0xA5, WS_GUARD, st));

  typename Prov::Dec dec;
  { int rcp = prov.prepare(ctx, st, pl, w.dig16, w.signbm, w.block_counts, w.bin_total, w.err, dec); if (rcp) return rcp; }
  hipLaunchKernelGGL(k_binscan, dim3(1), dim3(1024), 0, st, pl, w.bin_total, w.bin_start, w.tile_prefix, w.meta);
  // >= 8 windows: 1-D grid, one XCD per window (see the kernel); else (windows, ranges)
  const u32 xw = (gw >= 8 && ctx->opt_xcd_windows != 1) ? 1u : 0u;
  dim3 g1 = xw ? dim3(8u * ((gw + 7) / 8) * pl.nblk1) : dim3(gw, pl.nblk1);
  if (pl.spb > 2 * STAGE)
    hipLaunchKernelGGL((k_scatter1<typename Prov::Dec, 4 * STAGE, 1024>), g1, dim3(1024), 0, st, dec, pl, w.block_counts, w.bin_start, w.entries, xw);
  else if (pl.spb > STAGE)
    hipLaunchKernelGGL((k_scatter1<typename Prov::Dec, 2 * STAGE>), g1, dim3(256), 0, st, dec, pl, w.block_counts, w.bin_start, w.entries, xw);
  else if (Prov::Dec::VEC && (pl.BW & 1u) == 0 && ctx->opt_scatter_lean == 1)   // A/B knob (needs an even number of bins per window): 899 us against 880 at 2^24, profiles/r03/s_scatter_lean_and_clear_beside_ab.txt
    hipLaunchKernelGGL((k_scatter1<typename Prov::Dec, STAGE, 256, true>), g1, dim3(256), 0, st, dec, pl, w.block_counts, w.bin_start, w.entries, xw);
  else
    hipLaunchKernelGGL((k_scatter1<typename Prov::Dec, STAGE>), g1, dim3(256), 0, st, dec, pl, w.block_counts, w.bin_start, w.entries, xw);
  // pass 2 only where a bin holds more than one bucket (LB > 0); with <= 256 buckets per window
  // (negabase digits) pass 1 already sorts exactly: bins are buckets
  const u32* d_sorted = w.entries;
  const u32* d_bstart = w.bin_start;
  if (pl.LB > 0) {
    if (pl.bin_cap && pl.bin_cap <= BIN_CAP_SMALL) hipLaunchKernelGGL((k_binsort<BIN_CAP_SMALL>), dim3(pl.nbins), dim3(1024), 0, st, pl, w.entries, w.bin_start, w.sorted, w.bucket_start);
    else if (pl.bin_cap) hipLaunchKernelGGL((k_binsort<BIN_CAP>), dim3(pl.nbins), dim3(1024), 0, st, pl, w.entries, w.bin_start, w.sorted, w.bucket_start);
    hipLaunchKernelGGL(k_tilemap, dim3((pl.max_tiles + 255) / 256), dim3(256), 0, st, pl, w.bin_start, w.tile_prefix, w.meta, w.tile_info);
    const u32 g2 = pl.bin_cap ? std::min(pl.max_tiles, 768u) : pl.max_tiles;   // with k_binsort the tiled kernels see the oversize bins only: a small grid walks the (usually empty) tile table
    hipLaunchKernelGGL(k_count2, dim3(g2), dim3(256), 0, st, pl, w.entries, w.tile_info, w.bucket_count);
    hipLaunchKernelGGL(k_bucketscan, dim3(((pl.nbins << pl.LB) + 255) / 256), dim3(256), 0, st, pl, w.bin_start, w.bucket_count, w.bucket_start);
    hipLaunchKernelGGL(k_scatter2, dim3(g2), dim3(256), 0, st, pl, w.entries, w.tile_info,
                       w.bucket_start, w.bucket_cursor, w.sorted);
    d_sorted = w.sorted; d_bstart = w.bucket_start;
  }

  HIPCHK(ctx, hipEventRecord(ev_sorted, s_sort));
  st = s_acc;
  HIPCHK(ctx, hipStreamWaitEvent(s_acc, ev_sorted, 0));
  if (ev_points) HIPCHK(ctx, hipStreamWaitEvent(s_acc, ev_points, 0));
  // batch entries: two accumulations side by side run ~25 % slower than one after the other (measured: 2 x 2^24 in 49 ms
  // against 2 x 19.3), so this call's waits for the other lane's; its digit and sort passes above did not
  if (ctx->wait_accum) HIPCHK(ctx, hipStreamWaitEvent(s_acc, ctx->wait_accum, 0));
  HIPCHK(ctx, hipEventRecord(ev_acc0, s_acc));
  {
    dim3 grid((pl.nthr1 + 255) / 256), blk(256);
    char* bsum = w.arena + ((size_t)ar.bucket_off + (size_t)slab_k * NBpad) * ptb;
    // register-budget variant of the accumulate kernel (lazy field: 2, 3 or 4 waves per SIMD)
    int wps = G::CONVERTED_DOMAIN ? (ctx->opt_accum_waves ? (int)ctx->opt_accum_waves : 3) : 4;
    if constexpr (G::CONVERTED_DOMAIN) {
      if (wps == 2) hipLaunchKernelGGL((k_accum1<G, 2>), grid, blk, 0, st, pl, d_sorted, d_bstart, w.meta, (const uint4*)d_points, bsum, w.rec_key, w.rec_pt);
      else if (wps == 4) hipLaunchKernelGGL((k_accum1<G, 4>), grid, blk, 0, st, pl, d_sorted, d_bstart, w.meta, (const uint4*)d_points, bsum, w.rec_key, w.rec_pt);
      else if (abi && (pl.L1 & 15u) == 0 && pl.L1 >= 32 && ctx->opt_entry_ring != 1) hipLaunchKernelGGL((k_accum1<G, 3, true, true>), grid, blk, 0, st, pl, d_sorted, d_bstart, w.meta, (const uint4*)d_points, bsum, w.rec_key, w.rec_pt);
      else if (abi) hipLaunchKernelGGL((k_accum1<G, 3, true>), grid, blk, 0, st, pl, d_sorted, d_bstart, w.meta, (const uint4*)d_points, bsum, w.rec_key, w.rec_pt);
      else hipLaunchKernelGGL((k_accum1<G, 3>), grid, blk, 0, st, pl, d_sorted, d_bstart, w.meta, (const uint4*)d_points, bsum, w.rec_key, w.rec_pt);
    } else {
      hipLaunchKernelGGL((k_accum1<G, 4>), grid, blk, 0, st, pl, d_sorted, d_bstart, w.meta, (const uint4*)d_points, bsum, w.rec_key, w.rec_pt);
    }
  }
  HIPCHK(ctx, hipEventRecord(ev_acc1, s_acc));
  st = s_tail;
  HIPCHK(ctx, hipStreamWaitEvent(s_tail, ev_acc1, 0));

  // edge-record merge (kernels_ec.cuh): pairs at once, longer buckets through the size-class queues
  {
    RoctxRange rg_tail("lemsm: edge-record merge");
    const u32 sc = abi ? 1u : 0u;
    char* bsum = w.arena + ((size_t)ar.bucket_off + (size_t)slab_k * NBpad) * ptb;
    hipLaunchKernelGGL((k_merge_pairs<G>), dim3((pl.nthr1 + 255) / 256), dim3(256), 0, st, pl, sc, mq, d_bstart, w.meta, w.rec_key, w.rec_pt, bsum, w.mq_cnt, w.mq_items);
    // the host does not know the queue lengths: these kernels run one block per CU whose waves take the items in turns
    hipLaunchKernelGGL((k_merge_serial<G>), dim3(std::min(ctx->num_cus, std::max(1u, (mq.capS + mq.capM + 63) / 64))), dim3(256), 0, st, sc, mq, w.mq_cnt, w.mq_items, w.rec_pt, bsum);
    hipLaunchKernelGGL((k_merge_waves<G>), dim3(std::min(ctx->num_cus, std::max(1u, (mq.capM + mq.capL + 3) / 4))), dim3(256), 0, st, sc, mq, w.mq_cnt, w.mq_items, w.rec_pt, w.mq_partial, bsum, w.mq_cnt);
    hipLaunchKernelGGL((k_merge_final<G>), dim3(std::min(ctx->num_cus, std::max(1u, (mq.capF + 3) / 4))), dim3(256), 0, st, sc, mq, w.mq_cnt, w.mq_items, w.mq_partial, bsum, w.mq_cnt);
  }
  RoctxRange rg_pyr("lemsm: bucket-reduction pyramid");
  if (nba > 1 && last_slab)
    hipLaunchKernelGGL((k_sum_slabs<G>), dim3((NBpad + 255) / 256), dim3(256), 0, st, w.arena + (size_t)ar.bucket_off * ptb, NBpad, nba, scaled_mask);
  // bucket reduction pyramid: one launch per step while a step is wide, then all remaining steps (and the copy of
  // U_{L-1}) in one launch of one block per window (k_pyramid_tail)
  if (last_slab) {
    size_t toff = 0;
    u32 s_begin = 1;
    if (first2) {   // steps 1 + 2 in one pass over the bucket sums (k_pyramid_first2); the task tables take over at step 3
      PyrFirst2Args fa; memset(&fa, 0, sizeof fa);
      fa.nwin = gw; fa.nb = pl.nb; fa.nbw = pl.nbw; fa.nbp = nbp; fa.bucket_off = ar.bucket_off; fa.wstride = nbp;
      fa.a2_off = ar.apyr_off + nbp / 2;          // offA(2)
      fa.r02_off = ar.rbuf_off + nbp / 4;         // offR(0, 2)
      fa.r11_off = ar.rbuf_off + nbp / 2;         // offR(1, 1)
      fa.scaled = abi_pyr ? 1u : 0u;
      const u32 threads = gw * (nbp / 8);
      hipLaunchKernelGGL((k_pyramid_first2<G>), dim3((threads + 255) / 256), dim3(256), 0, st, fa, d_bstart, w.arena, w.mq_cnt);
      toff = pp.steps[0].size() + pp.steps[1].size();
      s_begin = 3;
    }
    u32 first_fused = L + 1;
    // the last steps in ONE launch of one block per window (k_pyramid_tail) from where a window's whole step is at most
    // `lim` items: 256 by default -- one wave per SIMD of the block's CU, so a fused step costs one addition (~6.5 us)
    // where a launch of its own costs ~10; with more items per step the block's waves share SIMDs and a fused step
    // gets slower than a launch spread over the chip (option pyr_fuse = 2: 2048, measured slower; 1: never fuse)
    if (ctx->opt_pyr_fuse != 1) {
      const size_t lim = ctx->opt_pyr_fuse == 2 ? 2048 : 256;
      for (u32 s = s_begin; s <= L; s++) {
        bool fits = true;
        for (u32 q = s; q <= L; q++) if ((size_t)pp.steps[q - 1].size() * pp.step_max_count[q - 1] > lim) fits = false;
        if (fits && L - s + 1 <= 20) { first_fused = s; break; }
      }
      if (first_fused == L) first_fused = L + 1;   // a single step gains nothing
    }
    for (u32 s = s_begin; s <= L && s < first_fused; s++) {
      auto& tasks = pp.steps[s - 1];
      u32 maxc = pp.step_max_count[s - 1];
      size_t threads = (size_t)tasks.size() * maxc * gw;
      // a step that leaves most SIMDs without a wave is one addition deep: four lanes per addition (XYZZ29::add4_mem) make it ~2.6x shallower
      const u32 quad = (G::CONVERTED_DOMAIN && ctx->opt_pyr_quad != 2 && threads <= 32768 && !(s == 1 && abi_pyr)) ? 1u : 0u;
      if (quad) threads *= 4;
      hipLaunchKernelGGL((k_pyramid<G>), dim3((u32)((threads + 255) / 256)), dim3(256), 0, st, d_tasks + toff, (u32)tasks.size(), gw, maxc, w.arena, quad);
      toff += tasks.size();
    }
    const bool need_copy = !(L == 1 && abi_pyr);
    if (first_fused <= L) {
      PyrTailArgs ta; memset(&ta, 0, sizeof ta);
      ta.first = first_fused; ta.last = L;
      u32 o = (u32)toff;
      for (u32 s = first_fused; s <= L; s++) { ta.step_off[s - first_fused] = o; ta.max_count[s - first_fused] = pp.step_max_count[s - 1]; o += (u32)pp.steps[s - 1].size(); }
      ta.step_off[L - first_fused + 1] = o;
      hipLaunchKernelGGL((k_pyramid_tail<G>), dim3(gw), dim3(1024), 0, st, (const PyrTask*)d_tasks, ta, (const CopyTaskPod*)d_copy, need_copy ? 1u : 0u, w.arena,
                         (G::CONVERTED_DOMAIN && ctx->opt_pyr_quad != 2) ? 1u : 0u);
    } else if (need_copy) {
      u32 cthreads = gw * (u32)(ptb / 16);
      hipLaunchKernelGGL(k_copy_points, dim3((cthreads + 63) / 64), dim3(64), 0, st, d_copy, 1u, gw, (u32)ptb, w.arena);
    }
  }
  {
    const u32 n16 = last_slab ? (u32)((size_t)gw * (L + 1) * ptb / 16) : 0u;   // (a slab that is not the call's last leaves its status slot only)
    hipLaunchKernelGGL(k_group_finish, dim3(std::max(1u, (n16 + 255) / 256)), dim3(256), 0, st, (const uint4*)(w.arena + (size_t)ar.out_off * ptb), (uint4*)d_out, n16,
                       w.err, w.meta + META_CLOCK, w.mq_cnt, d_slot);
  }
  HIPCHK(ctx, hipGetLastError());
  if (!w.guards.empty()) {   // debug: drain and verify every guard before the workspace is reused
    std::vector<unsigned char> host(w.guards.size() * WS_GUARD);
    for (size_t i = 0; i < w.guards.size(); i++) HIPCHK(ctx, hipMemcpyAsync(host.data() + i * WS_GUARD, ws_base + w.guards[i], WS_GUARD, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(s_sort)); HIPCHK(ctx, hipStreamSynchronize(s_acc)); HIPCHK(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < host.size(); i++)
      if (host[i] != 0xA5) return fail(ctx, LEMSM_ERR_HIP, "workspace guard " + std::to_string(i / WS_GUARD) + " overwritten at byte " + std::to_string(i % WS_GUARD) + " (option ws_canary)");
  }
  return LEMSM_OK;
}

size_t group_ws_bytes(const lemsm_ctx* ctx, const GroupPlan& pl, u32 nbp, u32 L, size_t ptb, bool guard, u32 nba = 1) {
  u32 gw = pl.w1 - pl.w0;
  ArenaLayout ar = make_arena(pl.nbins << pl.LB, nbp, gw, L, nba);
  GroupWs w = carve(nullptr, pl, ar, make_mq_layout(ctx, pl.nthr1), ptb, guard);
  return w.total + 4096;
}

// Generic windowed bucket pipeline over window range [wb,we): fills host_out with
// (we-wb) x (L+1) XYZZ points, summed over slabs.

// One call's raw per-window records on the device: nslabs blocks of out_slab bytes, block k holding the
// records [total, U_0..U_{L-1}] of windows wb.. of slab k (nw_pad windows' worth of room, unused tail zeroed).
struct WinRun {
  size_t nslabs = 0, ng = 0, out_slab = 0, SLAB = 0, err_slot = 0;
  size_t nrec = 0;         // record blocks: nslabs, or 1 when the slabs share one tail (shared_tail())
  size_t err_stride = 1;   // status slots per slab: nw_pad, the same on every rank whatever its own number of window groups
  size_t err_cap = 0;      // bytes of the status-slot area (nslabs x err_stride slots, rounded up)
  char* d_out = nullptr; char* d_err = nullptr;
  u32 nw = 0, nw_pad = 0, L = 0; size_t ptb = 0;
  size_t send_bytes() const { return out_slab * nrec; }
  // multi-GPU: what one rank contributes to the all-gather -- its records, its status slots (so that every rank sees every
  // rank's non-canonical-scalar flags and all return the same status) and one 256-byte rank status (STATUS_BYTES)
  static constexpr size_t STATUS_BYTES = 256;
  size_t send_total() const { return out_slab * nrec + err_cap + STATUS_BYTES; }
};

// Several slabs, one tail: each slab accumulates into a bucket area of its own and the call runs ONE pyramid (k_sum_slabs)
// and leaves ONE block of records, where the slabs would otherwise each pay the ~0.3 ms latency chain of the tail and the
// host add their records (option slab_tail = 2: a tail per slab, as before).  Up to 8 slabs (8 x 75 MB of bucket areas
// per window group at c = 16); a function of the call's arguments and options alone, so every rank decides alike.
bool shared_tail(const lemsm_ctx* ctx, size_t nslabs) { return nslabs > 1 && nslabs <= 8 && ctx->opt_slab_tail != 2 && ctx->opt_groups <= 1; }

// The part of run_windows_enqueue's bookkeeping that fixes what a rank sends in the exchange (device-pointer entries):
// a function of the call's arguments and options alone, the same on every rank.
template <class G>
void win_sizes(const lemsm_ctx* ctx, size_t n, u32 nw_pad, u32 L, WinRun& wr) {
  const u32 slab_log = ctx->opt_slab_bits ? (u32)ctx->opt_slab_bits : MAX_SLAB_LOG;
  wr.SLAB = (size_t)1 << slab_log;
  wr.nslabs = n ? (n + wr.SLAB - 1) / wr.SLAB : 1;
  wr.nrec = shared_tail(ctx, wr.nslabs) ? 1 : wr.nslabs;
  wr.ptb = G::PT_BYTES; wr.L = L; wr.nw_pad = nw_pad; wr.nw = 0; wr.ng = 0;
  wr.out_slab = align_up((size_t)std::max(nw_pad, 1u) * (L + 1) * wr.ptb, 256);
  wr.err_slot = 96; wr.err_stride = std::max<size_t>(nw_pad, 1);
  wr.err_cap = align_up(wr.nslabs * wr.err_stride * wr.err_slot, 256);
  wr.d_out = nullptr; wr.d_err = nullptr;
}

// A rank that cannot take part in a collective its peers may already be waiting in tears the communicator down, so that
// they return LEMSM_ERR_RCCL instead of blocking for ever (ncclCommAbort; the communicator is unusable afterwards).
void comm_abort(lemsm_ctx* ctx) {
  if (!ctx->comm) return;
  Rccl& rc = Rccl::get();
  if (rc.CommAbort) (void)rc.CommAbort(ctx->comm); else (void)rc.CommDestroy(ctx->comm);
  ctx->comm = nullptr; ctx->comm_size = 1; ctx->comm_rank = 0;
  ctx->last_error += " [communicator aborted: this rank could not enter the exchange]";
}

// Enqueues the whole pipeline of one call (every slab, every window group) on the context's queues and leaves the
// raw records in the workspace (wr.d_out); nothing is read back.  nw_pad >= we - wb sizes the record area per slab
// (multi-GPU: the same for every rank, so that the all-gather sends equal counts).
template <class P64, class G, class MakeSrc>
int run_windows_enqueue(lemsm_ctx* ctx, MakeSrc make_src, size_t n, u32 c, u32 nb, u32 nbp, u32 L, u32 W, u32 wb, u32 we, u32 d,
                        const void* d_points, u32 nw_pad, WinRun& wr) {
  u32 nw = we - wb;
  if (nw_pad < nw) nw_pad = nw;
  // Slabs of points: every slab runs the whole pipeline and the per-window records of the slabs
  // are added on the host.  Device-resident inputs use the largest slab the 32-bit entry format
  // allows; host-pointer entries (ctx->host_stage) use small slabs so that the PCIe upload of slab
  // k+1 (upload queue, host blocked in the copy) overlaps the kernels of slab k (main queue).
  ctx->plan_ring = false;
  const HostStage* hs = ctx->host_stage;
  u32 slab_log = ctx->opt_slab_bits ? (u32)ctx->opt_slab_bits : MAX_SLAB_LOG;
  if (hs) slab_log = host_slab_log(ctx, n);
  const size_t SLAB = (size_t)1 << slab_log;
  const size_t nslabs = n ? (n + SLAB - 1) / SLAB : 1;
  const bool shared = shared_tail(ctx, nslabs);
  const u32 nba = shared ? (u32)nslabs : 1u;
  const size_t nrec = shared ? 1 : nslabs;
  ctx->plan_slab_n = (u32)std::min(SLAB, n);
  u32 gmax = max_group_windows(ctx, nb, ctx->plan_slab_n, d);
  const size_t ptb = G::PT_BYTES;
  // Window groups of this call.  Default: as few as the bin limit allows (one at c = 16).  With
  // option "groups" > 1 the sort / accumulate / tail of neighbouring groups run on three queues;
  // measured (profiles/r01/pipelined_groups_trace.txt) this does NOT pay on MI355X: k_accum1 is
  // power-bound, so sort kernels running beside it slow it down by as much as they hide.
  u32 ngroups = 1;
  if (ctx->opt_groups > 0) ngroups = std::min((u32)ctx->opt_groups, std::max(nw, 1u));
  u32 gsz = std::max(1u, std::min(gmax, (nw + ngroups - 1) / ngroups));
  struct Grp { u32 g0, g1; size_t off; };
  std::vector<Grp> groups;
  size_t ws_total = 0;
  if (n) {
    // a group's workspace must hold the layout of EVERY slab: the ragged last slab can need more of some
    // buffers than a full one (below 2^16 points the pass-1 ranges shrink, so there are more of them)
    u32 sn0 = (u32)std::min(SLAB, n), snl = (u32)(n - (nslabs - 1) * SLAB);
    for (u32 g0 = wb; g0 < we; g0 += gsz) {
      u32 g1 = std::min(we, g0 + gsz);
      size_t bytes = group_ws_bytes(ctx, make_group_plan(ctx, sn0, c, nb, W, g0, g1, d), nbp, L, ptb, ctx->opt_ws_canary != 0, nba);
      if (snl != sn0) bytes = std::max(bytes, group_ws_bytes(ctx, make_group_plan(ctx, snl, c, nb, W, g0, g1, d), nbp, L, ptb, ctx->opt_ws_canary != 0, nba));
      groups.push_back({g0, g1, ws_total});
      ws_total += align_up(bytes + 8192, 256);     // (+ slack for the 16-entry rounding of the accumulate chunk, see make_group_plan)
    }
  }
  const size_t ng = groups.size();
  const size_t out_slab = align_up((size_t)std::max(nw_pad, 1u) * (L + 1) * ptb, 256);   // one slab's records
  const size_t ERR_SLOT = 96;   // (win_sizes: the same) per (slab, group): err[2] of the digit pass, then k_accum1's clock stamps (4 x u64) at byte 16
  const size_t err_stride = std::max<size_t>(nw_pad, 1);      // slot (slab k, group gi) = k * err_stride + gi, gi < ng <= nw <= nw_pad
  const size_t err_bytes = align_up(nslabs * err_stride * ERR_SLOT, 256);
  size_t conv_bytes = G::CONVERTED_DOMAIN ? align_up(std::min(SLAB, n) * 64, 256) : 0;
  // sizes first (a rank that fails below still has to contribute a buffer of the agreed size to the collective)
  wr.nslabs = nslabs; wr.nrec = nrec; wr.ng = ng; wr.out_slab = out_slab; wr.SLAB = SLAB; wr.err_slot = ERR_SLOT; wr.err_stride = err_stride; wr.err_cap = err_bytes;
  wr.nw = nw; wr.nw_pad = nw_pad; wr.L = L; wr.ptb = ptb;
  int rc = reserve(ctx, ctx->ws, ws_total + conv_bytes + out_slab * nrec + err_bytes + WinRun::STATUS_BYTES + 4096);
  if (rc) return rc;
  char* ws_base = (char*)ctx->ws.p;
  char* d_conv = ws_base + ws_total;
  char* d_out = d_conv + conv_bytes;
  char* d_err = d_out + out_slab * nrec;
  wr.d_out = d_out; wr.d_err = d_err;
  while (ctx->evpool.size() < 3 * ng * nslabs + (hs ? nslabs : 0)) {
    hipEvent_t e; HIPCHK(ctx, hipEventCreate(&e)); ctx->evpool.push_back(e);
  }
  hipEvent_t* ev_up = ctx->evpool.data() + 3 * ng * nslabs;
  hipStream_t s_sort = ctx->stream_sort, s_acc = ctx->stream, s_tail = ctx->stream_tail;
  const bool one_queue = ctx->opt_groups <= 1;
  if (one_queue) { s_sort = s_acc; s_tail = s_acc; }   // one queue unless pipelining was asked for: exact event timing
  hipStream_t s_up = ctx->stream_sort;                 // upload queue of the host-pointer path (one_queue is forced there)
  if (hs && !one_queue) return fail(ctx, LEMSM_ERR_BAD_ARG, "option groups > 1 is only available on the device-pointer entries");
  if (!one_queue) HIPCHK(ctx, hipStreamSynchronize(s_acc));   // several queues: inputs staged / digits produced on the main stream are complete (one queue: stream order)
  HIPCHK(ctx, hipEventRecord(ctx->ev[0], s_acc));
  ctx->t_accum_ms = 0; ctx->n_accum = 0;
  if (nw_pad != nw || n == 0 || nw == 0)      // record slots no kernel writes (a rank with fewer windows than the widest one) read as identities
    HIPCHK(ctx, hipMemsetAsync(d_out, 0, out_slab * nrec, s_acc));
  HIPCHK(ctx, hipMemsetAsync(d_err, 0, err_bytes + WinRun::STATUS_BYTES, s_acc));   // status slots no group writes, and the rank status word (0 = ok)
  u32 scaled_mask = 0;       // shared tail: which slabs accumulated in the ABI form (the ragged last slab may choose differently)
  for (size_t k = 0; k < nslabs && n && nw; k++) {
    const size_t s0 = k * SLAB;
    u32 sn = (u32)std::min(SLAB, n - s0);
    if (hs) {
      HIPCHK(ctx, hipMemcpyAsync(hs->d_scalars + s0 * 32, hs->h_scalars + s0 * 32, (size_t)sn * 32, hipMemcpyHostToDevice, s_up));
      if (hs->h_points)   // (null: the bases are resident already, lemsm_msm_with_bases)
        HIPCHK(ctx, hipMemcpyAsync(hs->d_points + s0 * 64, (const char*)hs->h_points + s0 * 64, (size_t)sn * 64, hipMemcpyHostToDevice, s_up));
      HIPCHK(ctx, hipEventRecord(ev_up[k], s_up));
      HIPCHK(ctx, hipStreamWaitEvent(s_acc, ev_up[k], 0));
    }
    const char* pts = (const char*)d_points + s0 * 64;
    // Lazy field: either convert the slab's points to the 2^261 domain (one pass, 25 ps per point)
    // or let k_accum1 consume them as they are (madd_abi: ~100 instructions whenever ANY lane of a
    // wave opens a segment).  With S = average segment length the second costs P = 1-(1-1/S)^64 of
    // 110/2058 of the accumulation (68 ps per point and window): cheaper when P * windows < 7.
    bool abi = false;
    hipEvent_t ev_points = nullptr;
    ctx->plan_ring = false;
    if constexpr (G::CONVERTED_DOMAIN) {
      GroupPlan pl0 = make_group_plan(ctx, sn, c, nb, W, groups[0].g0, groups[0].g1, d);
      double S = std::max(1.0, std::min((double)pl0.L1, (double)sn / (double)nb));
      double P = 1.0 - std::pow(1.0 - 1.0 / S, 64.0);
      abi = P * (double)(groups[0].g1 - groups[0].g0) < 7.0 && (ctx->opt_accum_waves == 0 || ctx->opt_accum_waves == 3);
      if (ctx->opt_abi_points == 1) abi = false;
      if (ctx->opt_abi_points == 2) abi = (ctx->opt_accum_waves == 0 || ctx->opt_accum_waves == 3);
      ctx->plan_ring = abi && ctx->opt_entry_ring != 1;
      if (!abi) {
        if (one_queue && !hs) {
          // the conversion (HBM-bound, 2 x 64 B per point) runs on the second queue beside this slab's digit and sort passes
          // (latency- and atomic-bound); it may start once everything enqueued so far -- the previous slab's accumulation,
          // which still reads d_conv -- is done, and the accumulate kernel waits for it
          HIPCHK(ctx, hipEventRecord(ctx->ev[2], s_acc));
          HIPCHK(ctx, hipStreamWaitEvent(ctx->stream_sort, ctx->ev[2], 0));
          hipLaunchKernelGGL((k_convert_points<typename G::F_>), dim3((2 * sn + 255) / 256), dim3(256), 0, ctx->stream_sort, (const uint4*)pts, (uint4*)d_conv, sn);
          HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream_sort));
          ev_points = ctx->ev[3];
        } else {
          hipLaunchKernelGGL((k_convert_points<typename G::F_>), dim3((2 * sn + 255) / 256), dim3(256), 0, s_acc, (const uint4*)pts, (uint4*)d_conv, sn);
        }
        pts = d_conv;
      }
    }
    for (size_t gi = 0; gi < ng; gi++) {
      const Grp& gr = groups[gi];
      GroupPlan pl = make_group_plan(ctx, sn, c, nb, W, gr.g0, gr.g1, d);
      pl.nstride = (u32)n;   // negabase digit matrix: d rows of n columns, whatever the slab (lhs_partial_t bounds n < 2^32)
      auto src = make_src(s0, sn);
      hipEvent_t* ev = ctx->evpool.data() + 3 * (k * ng + gi);
      if (gi == 0 && abi) scaled_mask |= 1u << k;
      rc = run_group<G>(ctx, src, pl, nbp, L, pts, abi, ws_base + gr.off, d_out + (shared ? 0 : k) * out_slab + (size_t)(gr.g0 - wb) * (L + 1) * ptb,
                        (u32*)(d_err + (k * err_stride + gi) * ERR_SLOT), s_sort, s_acc, s_tail, ev[0], ev[1], ev[2], gi == 0 ? ev_points : nullptr,
                        shared ? (u32)k : 0u, nba, scaled_mask);
      if (rc) return rc;
    }
    if (!one_queue) {   // three queues share one workspace: drain before the next slab reuses it
      HIPCHK(ctx, hipStreamSynchronize(s_sort));
      HIPCHK(ctx, hipStreamSynchronize(s_acc));
      HIPCHK(ctx, hipStreamSynchronize(s_tail));
    }
    // one queue: the slabs' kernels are stream-ordered, so the workspace is reused without a drain
  }
  if (n == 0 || nw == 0) { wr.ng = 0; }
  return LEMSM_OK;
}

// The queue the records of run_windows_enqueue become final on (where a collective or the read-back is enqueued).
hipStream_t records_stream(const lemsm_ctx* ctx) { return ctx->opt_groups <= 1 ? ctx->stream : ctx->stream_tail; }

// This rank's own bookkeeping out of its status slots (host copy `slots`, wr.err_cap bytes): merge-queue counters, clock
// stamps and accumulate timings of the last call.
int collect_local_stats(lemsm_ctx* ctx, const WinRun& wr, const u32* slots) {
  u64 clk_cyc = 0, clk_ticks = 0;
  const size_t sw = wr.err_slot / 4;
  for (int q = 0; q < 4; q++) ctx->dbg_merge[q] = 0;
  for (size_t k = 0; k < wr.nslabs; k++)
    for (size_t gi = 0; gi < wr.ng; gi++) {
      const u32* ew = slots + (k * wr.err_stride + gi) * sw;
      const size_t i = k * wr.ng + gi;             // the events of (slab, group) are numbered densely
      for (int q = 0; q < 4; q++) ctx->dbg_merge[q] += ew[12 + q];
      for (int q = 0; q < 8; q++) ctx->dbg_stamps[q] = ew[16 + q];
      float a = 0; HIPCHK(ctx, hipEventElapsedTime(&a, ctx->evpool[3 * i + 1], ctx->evpool[3 * i + 2]));
      ctx->t_accum_ms += a; ctx->n_accum++;
      u64 ck[4]; memcpy(ck, ew + 4, 32);
      if (ck[2] > ck[0] && ck[3] > ck[1] && ck[2] - ck[0] < ((u64)1 << 40)) { clk_cyc += ck[2] - ck[0]; clk_ticks += ck[3] - ck[1]; }
    }
  ctx->accum_clock_mhz = clk_ticks ? (double)clk_cyc / (double)clk_ticks * 100.0 : 0.0;
  return LEMSM_OK;
}

// Non-canonical scalars (>= field order) are rejected, never bucketed: the smallest offending index over the status slots
// of one rank's area (all nslabs x err_stride slots: unused ones are zero), or SIZE_MAX.
size_t first_bad_scalar(const WinRun& wr, const u32* slots) {
  size_t best = SIZE_MAX;
  const size_t sw = wr.err_slot / 4;
  for (size_t k = 0; k < wr.nslabs; k++)
    for (size_t gi = 0; gi < wr.err_stride; gi++) {
      const u32* ew = slots + (k * wr.err_stride + gi) * sw;
      if (ew[0]) best = std::min(best, k * wr.SLAB + (size_t)(~ew[1]));
    }
  return best;
}

// Reads back this call's records (wr.d_out) together with its status slots, waits, and turns the digit pass's flags into
// the reference's panic sites.  raw_bytes = 0: the status slots only (one-GPU rehearsal of a rank).
int run_windows_finish(lemsm_ctx* ctx, const WinRun& wr, const char* d_raw, size_t raw_bytes, std::vector<char>& raw) {
  RoctxRange rg_fin("lemsm: read-back and wait");
  hipStream_t s_tail = records_stream(ctx);
  raw.resize(raw_bytes);
  const size_t err_bytes = wr.ng ? wr.err_cap : 0;
  // read-back through the context's pinned buffer: records and status slots sit next to each other in the workspace,
  // so a call is ONE asynchronous copy (pageable destinations made two staged copies with ~25 us gaps)
  { int rcp = reserve_pinned(ctx, raw_bytes + err_bytes + 64); if (rcp) return rcp; }
  char* hp = (char*)ctx->h_pin;
  if (d_raw == wr.d_out && raw_bytes == wr.send_bytes() && raw_bytes + err_bytes) {
    HIPCHK(ctx, hipMemcpyAsync(hp, d_raw, raw_bytes + err_bytes, hipMemcpyDeviceToHost, s_tail));   // d_err follows d_out (run_windows_enqueue)
  } else {
    if (raw_bytes) HIPCHK(ctx, hipMemcpyAsync(hp, d_raw, raw_bytes, hipMemcpyDeviceToHost, s_tail));
    if (err_bytes) HIPCHK(ctx, hipMemcpyAsync(hp + raw_bytes, wr.d_err, err_bytes, hipMemcpyDeviceToHost, s_tail));
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev[1], s_tail));
  const auto th0 = std::chrono::steady_clock::now();
  HIPCHK(ctx, hipEventSynchronize(ctx->ev[1]));
  ctx->host_us[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th0).count();
  if (raw_bytes) memcpy(raw.data(), hp, raw_bytes);
  float ms = 0; HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
  ctx->t_total_ms = ms;
  if (!err_bytes) return LEMSM_OK;
  std::vector<u32> slots(err_bytes / 4);
  memcpy(slots.data(), hp + raw_bytes, err_bytes);
  const size_t bad = first_bad_scalar(wr, slots.data());
  if (bad != SIZE_MAX) {
    ctx->bad_index = bad;
    return fail(ctx, LEMSM_ERR_SCALAR_OUT_OF_RANGE, "scalar is not a canonical field element (>= order)");
  }
  return collect_local_stats(ctx, wr, slots.data());
}


template <class P64, class G>
void from_device_records(const char* raw, size_t n, host::pt* out) { from_device_records_t<P64, G::CONVERTED_DOMAIN, G::PT_BYTES>(raw, n, out); }

// host: records of nw windows = sum over the slabs of one rank's record area (raw: nslabs blocks of out_slab bytes)
template <class P64, class G>
void sum_slab_records(lemsm_ctx* ctx, const char* raw, size_t out_slab, size_t nslabs, u32 nw, u32 L, std::vector<host::pt>& host_out,
                      bool fold = false /* leave the nw window sums S_w = total + sum_l 2^l U_l instead of the records */) {
  typedef host::HG<P64> HGp;
  const size_t ptb = G::PT_BYTES;
  RoctxRange rg_host("lemsm: host fold of the window records");
  host_out.assign(fold ? (size_t)nw : (size_t)nw * (L + 1), HGp::identity());
  host_parallel(ctx, (int)nw, [&](int w) {        // one job per window: its L + 1 records of every slab (and their Horner fold)
    std::vector<host::pt> tmp(L + 1), acc(L + 1);
    host::pt* dst = fold ? acc.data() : host_out.data() + (size_t)w * (L + 1);
    for (size_t k = 0; k < nslabs; k++) {
      from_device_records<P64, G>(raw + k * out_slab + (size_t)w * (L + 1) * ptb, L + 1, tmp.data());
      if (k == 0) memcpy(dst, tmp.data(), (L + 1) * sizeof(host::pt));
      else for (u32 i = 0; i <= L; i++) dst[i] = HGp::add(dst[i], tmp[i]);
    }
    if (fold) host_out[w] = window_sum<P64>(dst, L);
  });
}

// Generic windowed bucket pipeline over window range [wb,we): fills host_out with
// (we-wb) x (L+1) XYZZ points, summed over slabs.
template <class P64, class G, class MakeSrc>
int run_windows(lemsm_ctx* ctx, MakeSrc make_src, size_t n, u32 c, u32 nb, u32 nbp, u32 L, u32 W, u32 wb, u32 we, u32 d,
                const void* d_points, std::vector<host::pt>& host_out, bool fold = false /* window sums instead of records */) {
  typedef host::HG<P64> HGp;
  u32 nw = we - wb;
  host_out.assign(fold ? (size_t)nw : (size_t)nw * (L + 1), HGp::identity());
  if (n == 0 || nw == 0) return LEMSM_OK;
  WinRun wr;
  int rc = run_windows_enqueue<P64, G>(ctx, make_src, n, c, nb, nbp, L, W, wb, we, d, d_points, nw, wr);
  if (rc) return rc;
  std::vector<char> raw;
  rc = run_windows_finish(ctx, wr, wr.d_out, wr.send_bytes(), raw);
  if (rc) return rc;
  const auto th0 = std::chrono::steady_clock::now();
  sum_slab_records<P64, G>(ctx, raw.data(), wr.out_slab, wr.nrec, nw, L, host_out, fold);
  ctx->host_us[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th0).count();
  return LEMSM_OK;
}

// the window sums of one call, one pool job per window (L doublings + L + 1 additions each)
template <class P64>
void window_sums_par(lemsm_ctx* ctx, const host::pt* recs /* nw x (L+1) */, u32 nw, u32 L, std::vector<host::pt>& out) {
  out.resize(nw);
  host_parallel(ctx, (int)nw, [&](int w) { out[w] = window_sum<P64>(recs + (size_t)w * (L + 1), L); });
}

template <class P64>
void msm_combine_t(const MsmPlan& mp, const host::pt* sums /* W window sums */, u64 out[12]) { msm_combine_windows<P64>(mp.c, mp.W, sums, out); }

template <class P64, class G>
int msm_partial_t(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, u32 wb, u32 we,
                  std::vector<host::pt>& out) {
  MsmPlan mp = make_msm_plan(ctx, curve, n);
  if (wb > we || we > mp.W) return fail(ctx, LEMSM_ERR_BAD_ARG, "window range out of bounds");
  auto make_src = [&](size_t s0, u32) {
    PipProvider s; s.scalars = (const uint4*)((const char*)d_scalars + s0 * 32);
    memcpy(s.kadd.k, mp.kadd, 32); memcpy(s.kadd.order, order_of(curve), 32); return s;
  };
  // conversion of the raw records and this rank's share of the host tail (S_w = total + sum_l 2^l U_l for its own
  // windows) in one pool job per window
  int rc = run_windows<P64, G>(ctx, make_src, n, mp.c, mp.nb, mp.nbp, mp.L, mp.W, wb, we, 0, d_points, out, true);
  return rc;
}

// K MSMs over the same points, pipelined over two lanes (the context and its peer: own queues, workspace, pinned buffer):
// while the GPU works on call k, the host folds the records of call k - 1, and call k - 1's latency-bound tail (merge +
// pyramid: a few dozen dependent launches that occupy a fraction of the chip) runs beside call k's digit / sort passes
// and fills the wave slots its accumulation leaves.  Matches how a prover uses best_multiexp: many calls, one SRS.
template <class P64, class G>
int msm_batch_t(lemsm_ctx* ctx, int curve, const void* const* d_scalars, const void* d_points, size_t n, size_t K, u64* outs,
                const uint8_t* const* h_scalars = nullptr /* scalars in host memory instead (lemsm_msm_batch_with_bases): call k's are uploaded
                into its lane's staging buffer on the lane's second queue while the GPU still works on call k - 1 */) {
  lemsm_ctx* lane[2] = {ctx, ctx->peer};
  MsmPlan mp = make_msm_plan(ctx, curve, n);
  WinRun wr[2];
  auto enqueue = [&](size_t k) -> int {
    lemsm_ctx* c = lane[k & 1];
    const void* sc = h_scalars ? nullptr : d_scalars[k];
    if (h_scalars) {
      int rcr = reserve(c, c->in_s, n * 32); if (rcr) return rcr;      // (the lane's call k - 2 has been finished: the buffer is free)
      if (!c->ev_batch_up) HIPCHK(c, hipEventCreateWithFlags(&c->ev_batch_up, hipEventDisableTiming));
      HIPCHK(c, hipMemcpyAsync(c->in_s.p, h_scalars[k], n * 32, hipMemcpyHostToDevice, c->stream_sort));
      HIPCHK(c, hipEventRecord(c->ev_batch_up, c->stream_sort));
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_batch_up, 0));
      sc = c->in_s.p;
    }
    auto make_src = [&](size_t s0, u32) {
      PipProvider s; s.scalars = (const uint4*)((const char*)sc + s0 * 32);
      memcpy(s.kadd.k, mp.kadd, 32); memcpy(s.kadd.order, order_of(curve), 32); return s;
    };
    c->wait_accum = nullptr;
    if (k > 0) {
      lemsm_ctx* o = lane[(k - 1) & 1]; const WinRun& w = wr[(k - 1) & 1];
      // resident scalars: wait for the END OF THE ACCUMULATION of the call before (other lane), so that its tail runs beside this
      // call's accumulation.  Host scalars: wait for the whole call -- under a full-chip accumulation the tail's few dozen
      // dependent launches starve (measured: call k - 1 completed together with call k, and the host, waiting for it, could not
      // start the next upload: 24.5 ms per 2^24 MSM instead of the 19 the GPU needs)
      if (h_scalars) c->wait_accum = o->ev_call_done;
      else if (w.ng && w.nslabs && o->evpool.size() >= 3 * w.ng * w.nslabs) c->wait_accum = o->evpool[3 * (w.ng * w.nslabs - 1) + 2];
    }
    int rce = run_windows_enqueue<P64, G>(c, make_src, n, mp.c, mp.nb, mp.nbp, mp.L, mp.W, 0, mp.W, 0, d_points, mp.W, wr[k & 1]);
    c->wait_accum = nullptr;
    if (!rce && h_scalars) {
      if (!c->ev_call_done) HIPCHK(c, hipEventCreateWithFlags(&c->ev_call_done, hipEventDisableTiming));
      HIPCHK(c, hipEventRecord(c->ev_call_done, records_stream(c)));
    }
    return rce;
  };
  auto finish = [&](size_t k) -> int {
    lemsm_ctx* c = lane[k & 1];
    std::vector<char> raw;
    int rc = run_windows_finish(c, wr[k & 1], wr[k & 1].d_out, wr[k & 1].send_bytes(), raw);
    if (rc) { ctx->last_error = c->last_error; ctx->bad_index = c->bad_index; return rc; }
    std::vector<host::pt> sums;
    sum_slab_records<P64, G>(ctx, raw.data(), wr[k & 1].out_slab, wr[k & 1].nrec, mp.W, mp.L, sums, true);
    msm_combine_t<P64>(mp, sums.data(), outs + 12 * k);
    return LEMSM_OK;
  };
  int rc = LEMSM_OK; size_t enq = 0;
  const bool stamps = getenv("LEMSM_DEBUG_STAMPS") != nullptr;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tb = now_us();
  for (size_t k = 0; k < K && !rc; k++) {
    const double t0 = now_us();
    rc = enqueue(k);
    const double t1 = now_us();
    if (rc) { if (lane[k & 1] != ctx) ctx->last_error = lane[k & 1]->last_error; break; }
    enq = k + 1;
    if (k > 0) rc = finish(k - 1);
    if (stamps) fprintf(stderr, "[lemsm batch] call %zu: enqueue (upload + launches) %.0f us at %.0f, finish of the call before %.0f us\n", k, t1 - t0, t0 - tb, now_us() - t1);
  }
  if (!rc && enq == K && K) rc = finish(K - 1);
  if (rc) for (lemsm_ctx* c : lane) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }   // nothing of this call may still run when it returns
  return rc;
}

int check_curve(lemsm_ctx* ctx, int curve) {
  if (curve != LEMSM_BN254_G1 && curve != LEMSM_GRUMPKIN) return fail(ctx, LEMSM_ERR_BAD_CURVE, "unknown curve id");
  return LEMSM_OK;
}

int stage(lemsm_ctx* ctx, DevBuf& b, const void* host, size_t bytes) {
  int rc = reserve(ctx, b, bytes ? bytes : 16);
  if (rc) return rc;
  if (bytes) HIPCHK(ctx, hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return LEMSM_OK;
}

// ---- lhs (negabase) plan ----
struct LhsPlan { u32 base, d, nb, nbp, L; };
int make_lhs_plan(int curve, u32 base, LhsPlan& lp) {
  if (base < 3) return LEMSM_ERR_BAD_BASE;
  lp.base = base; lp.d = logb_ceil8(bound_of(curve), base) + 1;
  lp.nb = base - 1; lp.nbp = next_pow2(lp.nb); if (lp.nbp < 2) lp.nbp = 2;
  lp.L = ilog2(lp.nbp);
  return LEMSM_OK;
}

// digits (position-major, d x n) for the lhs path; leaves err words in ctx workspace tail
struct LhsDigits { uint8_t* digitsT; u32* err; };
struct Words12 { u32 w[12]; };
__global__ void k_set_words12(u32* __restrict__ dst, Words12 v) { if (threadIdx.x < 12) dst[threadIdx.x] = v.w[threadIdx.x]; }

// the range-check words of the digit pass (first offending index, truncation count) travel to the context's small
// pinned buffer right behind the digit kernel; lhs_err_words waits for that copy (long done when the MSM core returns)
int lhs_err_words(lemsm_ctx* ctx, size_t n, u32 err[2]) {
  err[0] = 0xffffffffu; err[1] = 0;
  if (!n) return LEMSM_OK;
  HIPCHK(ctx, hipEventSynchronize(ctx->ev[4]));
  err[0] = ctx->h_small[0]; err[1] = ctx->h_small[1];
  return LEMSM_OK;
}

int lhs_digits(lemsm_ctx* ctx, int curve, const void* d_scalars, size_t n, const LhsPlan& lp, DevBuf& buf, LhsDigits& out, u32 row_begin, u32 row_end) {
  size_t bytes = align_up((size_t)lp.d * n, 256) + 256 + 64;
  int rc = reserve(ctx, buf, bytes);
  if (rc) return rc;
  char* base = (char*)buf.p;
  out.digitsT = (uint8_t*)base;
  u32* bound_d = (u32*)(base + align_up((size_t)lp.d * n, 256));
  out.err = bound_d + 8;
  Words12 init;   // the bound and the initial error words ride in as kernel arguments: no pageable upload, no host wait
  memcpy(init.w, bound_of(curve), 32); init.w[8] = 0xffffffffu; init.w[9] = 0; init.w[10] = 0xffffffffu; init.w[11] = 0;
  hipLaunchKernelGGL(k_set_words12, dim3(1), dim3(64), 0, ctx->stream, bound_d, init);
  if (n) {
    hipLaunchKernelGGL(k_negbase_digits, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4*)d_scalars, (u32)n,
                       lp.base, lp.d, bound_d, 1, (uint8_t*)nullptr, out.digitsT, out.err, row_begin, row_end);
    if (!ctx->h_small) HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_small, 256, hipHostMallocDefault));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_small, out.err, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  }
  HIPCHK(ctx, hipGetLastError());
  return LEMSM_OK;
}

template <class P64, class G>
int lhs_partial_t(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, const LhsPlan& lp,
                  u32 pb, u32 pe, std::vector<host::pt>& out, size_t* bad_index) {
  if (pb > pe || pe > lp.d) return fail(ctx, LEMSM_ERR_BAD_ARG, "position range out of bounds");
  if (n >= ((size_t)1 << 32)) return fail(ctx, LEMSM_ERR_BAD_ARG, "n too large");
  LhsDigits dg;
  int rc = lhs_digits(ctx, curve, d_scalars, n, lp, ctx->in_aux, dg, pb, pe);
  if (rc) return rc;
  // the digit matrix is position-major over the whole n; slabs index it with an offset
  auto make_src = [&](size_t s0, u32) { NegProvider s; s.digitsT = dg.digitsT + s0; return s; };
  rc = run_windows<P64, G>(ctx, make_src, n, 0, lp.nb, lp.nbp, lp.L, lp.d, pb, pe, lp.d, d_points, out, true);
  if (rc) return rc;
  u32 err[2];
  { int rce = lhs_err_words(ctx, n, err); if (rce) return rce; }
  ctx->truncated = err[1];
  if (err[0] != 0xffffffffu) {
    if (bad_index) *bad_index = err[0];
    return fail(ctx, LEMSM_ERR_SCALAR_OUT_OF_RANGE, "scalar out of range (>= isqrt(order)+2)");
  }
  return LEMSM_OK;
}

template <class P64>
void lhs_combine_t(const LhsPlan& lp, const host::pt* sums, u64 out_carry[12], u64* out_carries) { lhs_combine_positions<P64>(lp.base, lp.d, sums, out_carry, out_carries); }

// field selection: option "field" 0 = lazy radix-2^29 (default), 1 = strict 32-bit limbs
int msm_partial_dispatch(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, u32 wb, u32 we,
                         std::vector<host::pt>& out) {
  if (ctx->opt_field == 1) {
    if (curve == LEMSM_BN254_G1) return msm_partial_t<host::FqParams64, GqStrict>(ctx, curve, d_scalars, d_points, n, wb, we, out);
    return msm_partial_t<host::FrParams64, GrStrict>(ctx, curve, d_scalars, d_points, n, wb, we, out);
  }
  if (curve == LEMSM_BN254_G1) return msm_partial_t<host::FqParams64, GqLazy>(ctx, curve, d_scalars, d_points, n, wb, we, out);
  return msm_partial_t<host::FrParams64, GrLazy>(ctx, curve, d_scalars, d_points, n, wb, we, out);
}
int lhs_partial_dispatch(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, const LhsPlan& lp,
                         u32 pb, u32 pe, std::vector<host::pt>& out, size_t* bad_index) {
  if (ctx->opt_field == 1) {
    if (curve == LEMSM_BN254_G1) return lhs_partial_t<host::FqParams64, GqStrict>(ctx, curve, d_scalars, d_points, n, lp, pb, pe, out, bad_index);
    return lhs_partial_t<host::FrParams64, GrStrict>(ctx, curve, d_scalars, d_points, n, lp, pb, pe, out, bad_index);
  }
  if (curve == LEMSM_BN254_G1) return lhs_partial_t<host::FqParams64, GqLazy>(ctx, curve, d_scalars, d_points, n, lp, pb, pe, out, bad_index);
  return lhs_partial_t<host::FrParams64, GrLazy>(ctx, curve, d_scalars, d_points, n, lp, pb, pe, out, bad_index);
}

// ---- multi-GPU: window / digit-position sharding with ONE all-gather of raw device records -------------------
// Rank r of `world` owns windows [W r / world, W (r+1) / world).  Every rank runs the pipeline for its windows over
// ALL points (replicated in its HBM), leaves the raw records [total, U_0..U_{L-1}] of each window (per slab) in its
// workspace, and ncclAllGather moves them device-to-device over xGMI -- no host round trip before the exchange.
// Every rank then reads the gathered buffer back once and runs the same fused host Horner.  EC addition is not an RCCL
// reduction operator, hence all-gather + local combine.  `sim` (tests / one-GPU rehearsal): the ranks' pipelines
// run one after the other on this GPU and their record areas are copied into the gathered buffer where the
// collective would have put them, so everything but the ncclAllGather call itself is exercised.
struct Exchange { int world, rank; bool sim; };
int validate_points(lemsm_ctx* ctx, int curve, const void* d_points, size_t n);   // (defined below)

#define RCCLCHK(ctx, expr)                                                                     \
  do {                                                                                         \
    ncclResult_t r_ = (expr);                                                                  \
    if (r_ != ncclSuccess) {                                                                   \
      (ctx)->last_error = std::string(#expr) + ": " + Rccl::get().GetErrorString(r_);          \
      return LEMSM_ERR_RCCL;                                                                   \
    }                                                                                          \
  } while (0)


// shared by the MSM and the lhs path: runs the rank's (or, simulated, every rank's) windows and returns the records of
// ALL W windows (W x (L+1) host points, slabs summed)
template <class P64, class G, class MakeSrc>
int sharded_records(lemsm_ctx* ctx, MakeSrc make_src, size_t n, u32 c, u32 nb, u32 nbp, u32 L, u32 W, u32 d, const void* d_points,
                    const Exchange& ex, std::vector<host::pt>& all, int rc_pre = LEMSM_OK /* a failure of this rank before the pipeline */) {
  u32 max_nw = 0;
  for (int r = 0; r < ex.world; r++) { u32 a, b; shard_range(W, ex.world, r, a, b); max_nw = std::max(max_nw, b - a); }
  WinRun wr; std::vector<char> raw;
  size_t sb = 0;
  if (!ex.sim) {
    // Collective-safe: whatever happens on this rank before the exchange -- a failed allocation, a plan over a kernel
    // limit, rejected input points (rc_pre) -- it still contributes a buffer of the agreed size, zeroed, with its status
    // in the rank status word, and every rank learns of the failure from the gathered buffer; no rank is left waiting in
    // ncclAllGather.  Only when not even that stand-in can be had is the communicator aborted (peers then return
    // LEMSM_ERR_RCCL).  The status slots travel too: every rank sees every rank's non-canonical-scalar flags, so all ranks
    // return the same status -- also a rank without windows of its own (world > W), which runs no digit pass.
    u32 a, b; shard_range(W, ex.world, ex.rank, a, b);
    int rc_local = rc_pre;
    if (!rc_local) rc_local = run_windows_enqueue<P64, G>(ctx, make_src, n, c, nb, nbp, L, W, a, b, d, d_points, max_nw, wr);
    else win_sizes<G>(ctx, n, max_nw, L, wr);      // the sizes every rank agrees on, without touching the device
    sb = wr.send_total();
    hipStream_t st = records_stream(ctx);
    const char* send = wr.d_out;
    if (rc_local) {
      if (reserve(ctx, ctx->fail_buf, sb) != LEMSM_OK) { comm_abort(ctx); return rc_local; }
      HIPCHK(ctx, hipMemsetAsync(ctx->fail_buf.p, 0, sb, st));
      HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)((char*)ctx->fail_buf.p + sb - WinRun::STATUS_BYTES), rc_local, 1, st));
      send = (const char*)ctx->fail_buf.p;
    }
    if (reserve(ctx, ctx->gather, sb * ex.world) != LEMSM_OK) { comm_abort(ctx); return rc_local ? rc_local : LEMSM_ERR_NOMEM; }
    {
      ncclResult_t r_ = Rccl::get().AllGather(send, ctx->gather.p, sb, ncclUint8, ctx->comm, st);
      if (r_ != ncclSuccess) { ctx->last_error = std::string("ncclAllGather: ") + Rccl::get().GetErrorString(r_); comm_abort(ctx); return LEMSM_ERR_RCCL; }
    }
    // one read-back of the whole gathered buffer
    { int rcp = reserve_pinned(ctx, sb * ex.world + 64); if (rcp) return rcp; }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pin, ctx->gather.p, sb * ex.world, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], st));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev[1]));
    const char* hp = (const char*)ctx->h_pin;
    std::vector<u32> status(ex.world);
    for (int r = 0; r < ex.world; r++) memcpy(&status[r], hp + (size_t)r * sb + sb - WinRun::STATUS_BYTES, 4);
    int failed_rank = -1;
    const int agreed = merge_rank_status(status.data(), ex.world, ex.rank, rc_local, &failed_rank);
    if (agreed != LEMSM_OK) {
      if (agreed != rc_local || !rc_local) ctx->last_error = "rank " + std::to_string(failed_rank) + " failed before the exchange with status " + std::to_string(status[failed_rank]) + " (" + lemsm_strerror((int)status[failed_rank]) + ")";
      return agreed;
    }
    float ms = 0; HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->t_total_ms = ms;
    size_t bad = SIZE_MAX;
    for (int r = 0; r < ex.world; r++) bad = std::min(bad, first_bad_scalar(wr, (const u32*)(hp + (size_t)r * sb + wr.send_bytes())));
    if (bad != SIZE_MAX) {
      ctx->bad_index = bad;
      return fail(ctx, LEMSM_ERR_SCALAR_OUT_OF_RANGE, "scalar is not a canonical field element (>= order)");
    }
    { int rcs = collect_local_stats(ctx, wr, (const u32*)(hp + (size_t)ex.rank * sb + wr.send_bytes())); if (rcs) return rcs; }
    raw.assign(hp, hp + sb * ex.world);
  } else {
    double t_total = 0, t_acc = 0; int n_acc = 0;
    for (int r = 0; r < ex.world; r++) {
      u32 a, b; shard_range(W, ex.world, r, a, b);
      int rc = run_windows_enqueue<P64, G>(ctx, make_src, n, c, nb, nbp, L, W, a, b, d, d_points, max_nw, wr);
      if (rc) return rc;
      sb = wr.send_total();      // records + status slots + rank status, as the collective moves them
      if (r == 0) { rc = reserve(ctx, ctx->gather, sb * ex.world); if (rc) return rc; }
      HIPCHK(ctx, hipMemcpyAsync((char*)ctx->gather.p + (size_t)r * sb, wr.d_out, sb, hipMemcpyDeviceToDevice, records_stream(ctx)));
      std::vector<char> none;
      rc = run_windows_finish(ctx, wr, nullptr, 0, none);
      if (rc) return rc;
      t_total += ctx->t_total_ms; t_acc += ctx->t_accum_ms; n_acc += ctx->n_accum;
    }
    ctx->t_total_ms = t_total; ctx->t_accum_ms = t_acc; ctx->n_accum = n_acc;
    raw.resize(sb * ex.world);
    HIPCHK(ctx, hipMemcpy(raw.data(), ctx->gather.p, raw.size(), hipMemcpyDeviceToHost));
  }
  all.assign((size_t)W * (L + 1), host::HG<P64>::identity());
  std::vector<host::pt> part;
  for (int r = 0; r < ex.world; r++) {
    u32 a, b; shard_range(W, ex.world, r, a, b);
    if (a == b) continue;
    sum_slab_records<P64, G>(ctx, raw.data() + (size_t)r * sb, wr.out_slab, wr.nrec, b - a, L, part);
    std::copy(part.begin(), part.end(), all.begin() + (size_t)a * (L + 1));
  }
  return LEMSM_OK;
}

template <class P64, class G>
int msm_sharded_t(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, const Exchange& ex, u64 out[12]) {
  MsmPlan mp = make_msm_plan(ctx, curve, n);
  auto make_src = [&](size_t s0, u32) {
    PipProvider s; s.scalars = (const uint4*)((const char*)d_scalars + s0 * 32);
    memcpy(s.kadd.k, mp.kadd, 32); memcpy(s.kadd.order, order_of(curve), 32); return s;
  };
  std::vector<host::pt> all;
  // (a rank whose input check fails still enters the exchange: sharded_records carries rc_pre to every rank)
  const int rc_pre = validate_points(ctx, curve, d_points, n);
  int rc = sharded_records<P64, G>(ctx, make_src, n, mp.c, mp.nb, mp.nbp, mp.L, mp.W, 0, d_points, ex, all, rc_pre);
  if (rc) return rc;
  std::vector<host::pt> sums;
  window_sums_par<P64>(ctx, all.data(), mp.W, mp.L, sums);
  msm_combine_t<P64>(mp, sums.data(), out);
  return LEMSM_OK;
}

template <class P64, class G>
int lhs_sharded_t(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, const LhsPlan& lp,
                  const Exchange& ex, u64 out_carry[12], u64* out_carries, size_t* bad_index) {
  if (n >= ((size_t)1 << 32)) return fail(ctx, LEMSM_ERR_BAD_ARG, "n too large");
  u32 pb = 0, pe = lp.d;
  if (!ex.sim) shard_range(lp.d, ex.world, ex.rank, pb, pe);
  LhsDigits dg; dg.digitsT = nullptr; dg.err = nullptr;
  // a failure before the exchange (input check, digit buffer) is carried into it, never returned ahead of the peers
  int rc_pre = validate_points(ctx, curve, d_points, n);
  if (!rc_pre) rc_pre = lhs_digits(ctx, curve, d_scalars, n, lp, ctx->in_aux, dg, pb, pe);   // every rank range-checks every scalar; it stores its own rows only
  auto make_src = [&](size_t s0, u32) { NegProvider s; s.digitsT = dg.digitsT + s0; return s; };
  std::vector<host::pt> all;
  int rc = sharded_records<P64, G>(ctx, make_src, n, 0, lp.nb, lp.nbp, lp.L, lp.d, lp.d, d_points, ex, all, rc_pre);
  if (rc) return rc;
  u32 err[2];
  { int rce = lhs_err_words(ctx, n, err); if (rce) return rce; }
  ctx->truncated = err[1];
  if (err[0] != 0xffffffffu) {
    if (bad_index) *bad_index = err[0];
    return fail(ctx, LEMSM_ERR_SCALAR_OUT_OF_RANGE, "scalar out of range (>= isqrt(order)+2)");
  }
  std::vector<host::pt> sums;
  window_sums_par<P64>(ctx, all.data(), lp.d, lp.L, sums);
  lhs_combine_t<P64>(lp, sums.data(), out_carry, out_carries);
  return LEMSM_OK;
}

int msm_sharded_dispatch(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, const Exchange& ex, u64 out[12]) {
  if (n == 0) { memset(out, 0, 96); return LEMSM_OK; }
  int saved = ctx->plan_world; ctx->plan_world = ex.world;
  int rc;
  if (ctx->opt_field == 1) {
    rc = curve == LEMSM_BN254_G1 ? msm_sharded_t<host::FqParams64, GqStrict>(ctx, curve, d_scalars, d_points, n, ex, out)
                                 : msm_sharded_t<host::FrParams64, GrStrict>(ctx, curve, d_scalars, d_points, n, ex, out);
  } else {
    rc = curve == LEMSM_BN254_G1 ? msm_sharded_t<host::FqParams64, GqLazy>(ctx, curve, d_scalars, d_points, n, ex, out)
                                 : msm_sharded_t<host::FrParams64, GrLazy>(ctx, curve, d_scalars, d_points, n, ex, out);
  }
  ctx->plan_world = saved;
  return rc;
}
int lhs_sharded_dispatch(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, const LhsPlan& lp,
                         const Exchange& ex, u64 out_carry[12], u64* out_carries, size_t* bad_index) {
  if (ctx->opt_field == 1) {
    if (curve == LEMSM_BN254_G1) return lhs_sharded_t<host::FqParams64, GqStrict>(ctx, curve, d_scalars, d_points, n, lp, ex, out_carry, out_carries, bad_index);
    return lhs_sharded_t<host::FrParams64, GrStrict>(ctx, curve, d_scalars, d_points, n, lp, ex, out_carry, out_carries, bad_index);
  }
  if (curve == LEMSM_BN254_G1) return lhs_sharded_t<host::FqParams64, GqLazy>(ctx, curve, d_scalars, d_points, n, lp, ex, out_carry, out_carries, bad_index);
  return lhs_sharded_t<host::FrParams64, GrLazy>(ctx, curve, d_scalars, d_points, n, lp, ex, out_carry, out_carries, bad_index);
}

template <class F>
__global__ __launch_bounds__(256) void k_precompute_mult(const uint4* __restrict__ jac, u32 n, u32 base, uint4* __restrict__ out) {
  typedef XYZZ<F> G; typedef typename F::fe fe;
  u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  fe X, Y, Z; F::load(X, jac + (size_t)j * 6); F::load(Y, jac + (size_t)j * 6 + 2); F::load(Z, jac + (size_t)j * 6 + 4);
  typename G::pt p;
  if (F::is_zero(Z)) G::set_identity(p);
  else { p.x = X; p.y = Y; F::sqr(p.zz, Z); F::mul(p.zzz, p.zz, Z); }
  typename G::pt acc = p;
  for (u32 k = 1; k < base; k++) {
    uint4* o = out + ((size_t)j * (base - 1) + (k - 1)) * 6;
    if (G::is_identity(acc)) { uint4 z = make_uint4(0, 0, 0, 0); for (int q = 0; q < 6; q++) o[q] = z; }
    else {
      fe t, xo, yo;
      F::sqr(t, acc.zz); F::mul(xo, acc.x, t);
      F::sqr(t, acc.zzz); F::mul(yo, acc.y, t);
      F::store(o, xo); F::store(o + 2, yo); F::store(o + 4, acc.zzz);
    }
    G::add(acc, p);
  }
}

// [1P .. (base-1)P] per input point as AFFINE points (the form src/config.rs:542-560 writes into its
// fixed column: the reference normalises every multiple with its own inversion, :550-554): one
// thread per point, Montgomery's trick over its base-1 multiples (prefix products in scratch [k][n]).
template <class F>
__global__ __launch_bounds__(256) void k_precompute_mult_affine(const uint4* __restrict__ jac, u32 n, u32 base,
                                                                uint4* __restrict__ out, char* __restrict__ scratch) {
  typedef XYZZ<F> G; typedef typename F::fe fe;
  u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  fe X, Y, Z; F::load(X, jac + (size_t)j * 6); F::load(Y, jac + (size_t)j * 6 + 2); F::load(Z, jac + (size_t)j * 6 + 4);
  typename G::pt p;
  if (F::is_zero(Z)) G::set_identity(p);
  else { p.x = X; p.y = Y; F::sqr(p.zz, Z); F::mul(p.zzz, p.zz, Z); }
  typename G::pt acc = p;
  fe pref; F::set_one(pref);
  for (u32 k = 1; k < base; k++) {
    char* s = scratch + ((size_t)(k - 1) * n + j) * 160;
    G::store(s, acc); F::store(s + 128, pref);
    if (!G::is_identity(acc)) F::mul(pref, pref, acc.zzz);
    G::add(acc, p);
  }
  fe inv; inv_via_lazy<F>(inv, pref);
  for (u32 k = base - 1; k >= 1; k--) {
    const char* s = scratch + ((size_t)(k - 1) * n + j) * 160;
    typename G::pt q; G::load(q, s); fe pr; F::load(pr, s + 128);
    uint4* o = out + ((size_t)j * (base - 1) + (k - 1)) * 4;
    if (G::is_identity(q)) { uint4 z = make_uint4(0, 0, 0, 0); o[0] = z; o[1] = z; o[2] = z; o[3] = z; continue; }
    fe izzz, izz, x, y;
    F::mul(izzz, inv, pr);
    F::mul(inv, inv, q.zzz);
    F::mul(izz, izzz, q.zz); F::sqr(izz, izz);
    F::mul(x, q.x, izz); F::mul(y, q.y, izzz);
    F::store(o, x); F::store(o + 2, y);
  }
}

// P_i = (i+1) Q, thread t owns [t*KB, (t+1)*KB); affine output through a per-thread batch inversion.
template <class F, int KB>
__global__ __launch_bounds__(256) void k_gen_walk(const uint4* __restrict__ q_aff, u32 n, uint4* __restrict__ out, char* __restrict__ scratch) {
  typedef XYZZ<F> G; typedef typename F::fe fe;
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  const u32 nthreads = gridDim.x * 256;
  const u64 j0 = (u64)t * KB;
  if (j0 >= n) return;
  const u32 cnt = (u32)min((u64)KB, (u64)n - j0);
  fe qx, qy; F::load(qx, q_aff); F::load(qy, q_aff + 2);
  // acc = (j0 + 1) * Q
  typename G::pt acc; G::set_identity(acc);
  u32 k = (u32)j0 + 1;
  for (int b = 31; b >= 0; b--) {
    if (!G::is_identity(acc)) { typename G::pt tmp = acc; G::dbl(acc, tmp); }
    if ((k >> b) & 1u) G::madd(acc, qx, qy);
  }
  fe pref; F::set_one(pref);
  for (u32 i = 0; i < cnt; i++) {
    if (i) G::madd(acc, qx, qy);
    char* s = scratch + ((size_t)i * nthreads + t) * 160;
    G::store(s, acc); F::store(s + 128, pref);
    F::mul(pref, pref, acc.zzz);     // walk points are never the identity for n < group order
  }
  fe inv; inv_via_lazy<F>(inv, pref);
  for (u32 i = cnt; i-- > 0;) {
    const char* s = scratch + ((size_t)i * nthreads + t) * 160;
    typename G::pt p; G::load(p, s); fe pr; F::load(pr, s + 128);
    fe izzz, izz, x, y;
    F::mul(izzz, inv, pr);
    F::mul(inv, inv, p.zzz);
    F::mul(izz, izzz, p.zz); F::sqr(izz, izz);
    F::mul(x, p.x, izz); F::mul(y, p.y, izzz);
    F::store(out + (j0 + i) * 4, x); F::store(out + (j0 + i) * 4 + 2, y);
  }
}

// ---- known-answer entries for the DEFAULT arithmetic (option field = 0): the lazy radix-2^29 field and its XYZZ law,
// reached directly instead of only through whole-MSM results.  Inputs and outputs keep the C ABI's form (x * 2^256,
// canonical): a value enters the lazy field through from_abi (x * 2^261, normalised) and leaves through div32 + canon.
template <class F29>
__device__ __forceinline__ void dbg_in29(typename F29::fe& r, const void* p) { typename F29::fe a; F29::load(a, p); F29::from_abi(r, a); }
template <class F29>
__device__ __forceinline__ void dbg_out29(void* p, const typename F29::fe& a) { typename F29::fe t; F29::div32(t, a); F29::store(p, t); }

// op 0: add, 1: sub, 2: neg(a), 4: sqr(a), 5: mul2 = a*b + b*a, 6: sqr_addhi = a^2 + b, 7: mul_addhi = a*b + b, 8: mul (as lemsm_debug_montmul),
// 9: mul32 / from_abi round trip = a (x -> 32x lazily, then back)
template <class F29>
__global__ void k_dbg_fieldop29(int op, const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename F29::fe x, y, r;
  dbg_in29<F29>(x, a + 2 * (size_t)i); dbg_in29<F29>(y, b + 2 * (size_t)i);
  if (op == 0) { F29::add(r, x, y); F29::wnorm(r); }
  else if (op == 1) F29::sub(r, x, y);
  else if (op == 2) F29::neg(r, x);
  else if (op == 4) F29::sqr(r, x);
  else if (op == 5) F29::mul2(r, x, y, y, x);
  else if (op == 6) F29::sqr_addhi(r, x, y);
  else if (op == 7) F29::mul_addhi(r, x, y, y);
  else if (op == 9) { typename F29::fe c; F29::load(c, a + 2 * (size_t)i); F29::mul32(r, c); }   // canonical x*2^256 -> lazy x*2^261
  else F29::mul(r, x, y);
  dbg_out29<F29>(out + 2 * (size_t)i, r);
}
// op 3: the inversion every batched-inversion kernel uses (inv_lazy: Fermat in the lazy field on a strict element)
template <class F>
__global__ void k_dbg_inv_lazy(const uint4* __restrict__ a, uint4* __restrict__ out, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename F::fe x, r; F::load(x, a + 2 * (size_t)i);
  inv_via_lazy<F>(r, x);
  F::store(out + 2 * (size_t)i, r);
}
// op 0: XYZZ29::madd, 1: add, 2: madd_abi (accumulator in the scaled form, incoming point as the ABI passes it)
template <class G29>
__global__ void k_dbg_pointop29(int op, const char* __restrict__ acc_in, const char* __restrict__ q_in, char* __restrict__ out, u32 n) {
  typedef typename G29::F_ F29;
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename G29::pt acc;
  const char* ap = acc_in + (size_t)i * 128;
  dbg_in29<F29>(acc.x, ap); dbg_in29<F29>(acc.y, ap + 32); dbg_in29<F29>(acc.zz, ap + 64); dbg_in29<F29>(acc.zzz, ap + 96);
  if (G29::is_identity(acc)) G29::set_identity(acc);
  if (op == 0 || op == 2) {
    typename F29::fe xa, ya; F29::load(xa, q_in + (size_t)i * 64); F29::load(ya, q_in + (size_t)i * 64 + 32);
    if (!(F29::limbs_zero(xa) && F29::limbs_zero(ya))) {
      if (op == 0) { typename F29::fe x, y; F29::from_abi(x, xa); F29::from_abi(y, ya); G29::madd(acc, x, y); }
      else { G29::scale(acc); bool empty = G29::is_identity(acc); G29::madd_abi(acc, xa, ya, empty); G29::unscale(acc); }
    }
  } else {
    typename G29::pt q; const char* qp = q_in + (size_t)i * 128;
    dbg_in29<F29>(q.x, qp); dbg_in29<F29>(q.y, qp + 32); dbg_in29<F29>(q.zz, qp + 64); dbg_in29<F29>(q.zzz, qp + 96);
    if (G29::is_identity(q)) G29::set_identity(q);
    G29::add(acc, q);
  }
  char* op_ = out + (size_t)i * 128;
  if (G29::is_identity(acc)) { uint4 z = make_uint4(0, 0, 0, 0); for (int k = 0; k < 8; k++) reinterpret_cast<uint4*>(op_)[k] = z; return; }
  dbg_out29<F29>(op_, acc.x); dbg_out29<F29>(op_ + 32, acc.y); dbg_out29<F29>(op_ + 64, acc.zz); dbg_out29<F29>(op_ + 96, acc.zzz);
}

// optional input validation (option "validate_points"): every affine point is (0,0) or satisfies y^2 = x^3 + b
template <class F>
__global__ __launch_bounds__(256) void k_validate_points(const uint4* __restrict__ pts, u32 n, int bcoef /* +3 or -17 */, u32* __restrict__ err) {
  u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  typedef typename F::fe fe;
  fe x, y; F::load(x, pts + (size_t)j * 4); F::load(y, pts + (size_t)j * 4 + 2);
  if (F::is_zero(x) && F::is_zero(y)) return;
  fe one, b, t, l, r; F::set_one(one); F::set_zero(b);
  int m = bcoef < 0 ? -bcoef : bcoef;
  for (int bit = 5; bit >= 0; bit--) { F::add(b, b, b); if ((m >> bit) & 1) F::add(b, b, one); }
  if (bcoef < 0) F::neg(b, b);
  F::sqr(l, y); F::sqr(t, x); F::mul(r, t, x); F::add(r, r, b);
  if (!F::eq(l, r)) atomicMin(err, j);
}

int validate_points(lemsm_ctx* ctx, int curve, const void* d_points, size_t n) {
  if (!ctx->opt_validate_points || n == 0) return LEMSM_OK;
  int rc = reserve(ctx, ctx->in_aux, 256); if (rc) return rc;
  u32 init = 0xffffffffu;
  HIPCHK(ctx, hipMemcpyAsync(ctx->in_aux.p, &init, 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  for (size_t s0 = 0; s0 < n; s0 += (size_t)1 << 30) {
    u32 cnt = (u32)std::min((size_t)1 << 30, n - s0);
    const uint4* p = (const uint4*)((const char*)d_points + s0 * 64);
    if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_validate_points<FqDev>), dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, p, cnt, 3, (u32*)ctx->in_aux.p);
    else hipLaunchKernelGGL((k_validate_points<FrDev>), dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, p, cnt, -17, (u32*)ctx->in_aux.p);
    u32 bad = 0xffffffffu;
    HIPCHK(ctx, hipMemcpyAsync(&bad, ctx->in_aux.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (bad != 0xffffffffu) { ctx->bad_index = s0 + bad; return fail(ctx, LEMSM_ERR_BAD_ARG, "point is not on the curve (option validate_points)"); }
  }
  return LEMSM_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

const char* lemsm_strerror(int s) {
  switch (s) {
    case LEMSM_OK: return "ok";
    case LEMSM_ERR_LEN_MISMATCH: return "incompatible amount of coefficients";
    case LEMSM_ERR_SCALAR_OUT_OF_RANGE: return "scalar out of range";
    case LEMSM_ERR_BAD_BASE: return "base must be in 3..=255";
    case LEMSM_ERR_BAD_CURVE: return "unknown curve";
    case LEMSM_ERR_HIP: return "HIP runtime error";
    case LEMSM_ERR_BAD_ARG: return "bad argument";
    case LEMSM_ERR_NOMEM: return "out of device memory";
    case LEMSM_ERR_TOO_MANY_DIGITS: return "too many digits";
    case LEMSM_ERR_RCCL: return "RCCL error";
    case LEMSM_ERR_INDEX_OUT_OF_BOUNDS: return "index out of bounds";
    case LEMSM_ERR_ARITH_OVERFLOW: return "arithmetic overflow";
    case LEMSM_ERR_SUM_NOT_IDENTITY: return "points do not sum to the identity";
    case LEMSM_ERR_WOULD_NOT_TERMINATE: return "the reference would loop forever";
    case LEMSM_ERR_DIVISION_BY_ZERO: return "division by zero";
    default: return "unknown status";
  }
}

int lemsm_create(int device, lemsm_ctx** out) {
  if (!out) return LEMSM_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0 || device < 0 || device >= count) return LEMSM_ERR_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return LEMSM_ERR_HIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return LEMSM_ERR_HIP;   // kernels are built for gfx950 only
  if (hipSetDevice(device) != hipSuccess) return LEMSM_ERR_HIP;
  lemsm_ctx* c = new lemsm_ctx();
  c->device = device;
  c->num_cus = prop.multiProcessorCount > 0 ? (u32)prop.multiProcessorCount : 256u;
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically lowest = highest priority
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, lo) != hipSuccess ||
        hipStreamCreateWithPriority(&c->stream_sort, hipStreamNonBlocking, hi) != hipSuccess ||
        hipStreamCreateWithPriority(&c->stream_tail, hipStreamNonBlocking, hi) != hipSuccess) { delete c; return LEMSM_ERR_HIP; }
  }
  for (int i = 0; i < 5; i++) if (hipEventCreate(&c->ev[i]) != hipSuccess) { delete c; return LEMSM_ERR_HIP; }
  *out = c;
  return LEMSM_OK;
}

void lemsm_destroy(lemsm_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream_sort) { (void)hipStreamSynchronize(ctx->stream_sort); (void)hipStreamDestroy(ctx->stream_sort); }
  if (ctx->stream_tail) { (void)hipStreamSynchronize(ctx->stream_tail); (void)hipStreamDestroy(ctx->stream_tail); }
  for (hipEvent_t e : ctx->evpool) (void)hipEventDestroy(e);
  if (ctx->comm) { (void)Rccl::get().CommDestroy(ctx->comm); ctx->comm = nullptr; }
  if (ctx->peer) { lemsm_destroy(ctx->peer); ctx->peer = nullptr; }
  ctx->pool.reset();
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  if (ctx->h_small) (void)hipHostFree(ctx->h_small);
  for (DevBuf* b : {&ctx->ws, &ctx->in_s, &ctx->in_p, &ctx->in_aux, &ctx->gather, &ctx->fail_buf, &ctx->dw_tab, &ctx->dw_arena, &ctx->dw_tmp}) if (b->p) (void)hipFree(b->p);
  for (auto& kv : ctx->pyr_cache) if (kv.second.buf.p) (void)hipFree(kv.second.buf.p);
  for (int i = 0; i < 5; i++) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
  for (int i = 0; i < 2; i++) if (ctx->dw_ev[i]) (void)hipEventDestroy(ctx->dw_ev[i]);
  if (ctx->ev_batch_up) (void)hipEventDestroy(ctx->ev_batch_up);
  if (ctx->ev_call_done) (void)hipEventDestroy(ctx->ev_call_done);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* lemsm_last_error(const lemsm_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int lemsm_set_option(lemsm_ctx* ctx, const char* name, long value) {
  if (!ctx || !name) return LEMSM_ERR_BAD_ARG;
  if (!strcmp(name, "window_bits")) { if (value != 0 && (value < 2 || value > 17)) return LEMSM_ERR_BAD_ARG; ctx->opt_window_bits = value; }
  else if (!strcmp(name, "chunk")) { if (value < 0 || value > 65536) return LEMSM_ERR_BAD_ARG; ctx->opt_chunk = value; }
  else if (!strcmp(name, "binsort")) { if (value < 0) return LEMSM_ERR_BAD_ARG; ctx->opt_binsort = value; }
  else if (!strcmp(name, "tile")) { if (value < 0 || (value && value < 256)) return LEMSM_ERR_BAD_ARG; ctx->opt_tile = value; }
  else if (!strcmp(name, "groups")) { if (value < 0 || value > 64) return LEMSM_ERR_BAD_ARG; ctx->opt_groups = value; }
  else if (!strcmp(name, "merge_slice")) { if (value != 0 && (value < 33 || value > 65536)) return LEMSM_ERR_BAD_ARG; ctx->opt_merge_slice = value; }
  else if (!strcmp(name, "dbg_repeat")) { ctx->opt_dbg_repeat = value; }
  else if (!strcmp(name, "pyr_first2")) { if (value < 0 || value > 1) return LEMSM_ERR_BAD_ARG; ctx->opt_pyr_first2 = value; }
  else if (!strcmp(name, "merge_wave_th")) { if (value < 0 || value > (1 << 24)) return LEMSM_ERR_BAD_ARG; ctx->opt_merge_wave_th = value; }
  else if (!strcmp(name, "abi_points")) { if (value < 0 || value > 2) return LEMSM_ERR_BAD_ARG; ctx->opt_abi_points = value; }
  else if (!strcmp(name, "stage2x")) { if (value < 0 || value > 4 || value == 3) return LEMSM_ERR_BAD_ARG; ctx->opt_stage2x = value; }
  else if (!strcmp(name, "xcd_windows")) { if (value < 0 || value > 1) return LEMSM_ERR_BAD_ARG; ctx->opt_xcd_windows = value; }
  else if (!strcmp(name, "ws_canary")) { if (value < 0 || value > 1) return LEMSM_ERR_BAD_ARG; ctx->opt_ws_canary = value; }
  else if (!strcmp(name, "dw_kb")) { if (value < 0 || value > 64) return LEMSM_ERR_BAD_ARG; ctx->opt_dw_kb = value; }
  else if (!strcmp(name, "dw_fuse")) { if (value != 0 && value != 2) return LEMSM_ERR_BAD_ARG; ctx->opt_dw_fuse = value; }
  else if (!strcmp(name, "dw_pw_lazy")) { if (value != 0 && value != 1) return LEMSM_ERR_BAD_ARG; ctx->opt_dw_pw_lazy = value; }
  else if (!strcmp(name, "scatter_lean")) { if (value < 0 || value > 1) return LEMSM_ERR_BAD_ARG; ctx->opt_scatter_lean = value; }   // 1: k_scatter1 with two adjacent bins per thread (A/B knob: measured no faster)
  else if (!strcmp(name, "pyr_quad")) { if (value != 0 && value != 2) return LEMSM_ERR_BAD_ARG; ctx->opt_pyr_quad = value; }
  else if (!strcmp(name, "slab_tail")) { if (value != 0 && value != 2) return LEMSM_ERR_BAD_ARG; ctx->opt_slab_tail = value; }
  else if (!strcmp(name, "dw_halves")) { if (value != 0 && value != 1) return LEMSM_ERR_BAD_ARG; ctx->opt_dw_halves = value; }
  else if (!strcmp(name, "dw_ntt_lazy")) { if (value != 0 && value != 1) return LEMSM_ERR_BAD_ARG; ctx->opt_dw_ntt_lazy = value; }
  else if (!strcmp(name, "dw_reuse")) { if (value != 0 && value != 2) return LEMSM_ERR_BAD_ARG; ctx->opt_dw_reuse = value; }
  else if (!strcmp(name, "dw_wrap")) { if (value != 0 && value != 2) return LEMSM_ERR_BAD_ARG; ctx->opt_dw_wrap = value; }
  else if (!strcmp(name, "ntt_tiled")) { if (value != 0 && value != 2) return LEMSM_ERR_BAD_ARG; ctx->opt_ntt_tiled = value; }
  else if (!strcmp(name, "pyr_fuse")) { if (value < 0 || value > 2) return LEMSM_ERR_BAD_ARG; ctx->opt_pyr_fuse = value; }
  else if (!strcmp(name, "validate_points")) { if (value < 0 || value > 1) return LEMSM_ERR_BAD_ARG; ctx->opt_validate_points = value; }
  else if (!strcmp(name, "entry_ring")) { if (value < 0 || value > 1) return LEMSM_ERR_BAD_ARG; ctx->opt_entry_ring = value; }
  else if (!strcmp(name, "slab_bits")) { if (value != 0 && (value < 12 || value > 24)) return LEMSM_ERR_BAD_ARG; ctx->opt_slab_bits = value; }
  else if (!strcmp(name, "host_slab_bits")) { if (value != 0 && (value < 12 || value > 24)) return LEMSM_ERR_BAD_ARG; ctx->opt_host_slab_bits = value; }
  else if (!strcmp(name, "accum_waves")) { if (value != 0 && (value < 2 || value > 4)) return LEMSM_ERR_BAD_ARG; ctx->opt_accum_waves = value; }
  else if (!strcmp(name, "host_threads")) { if (value < 0 || value > 64) return LEMSM_ERR_BAD_ARG; ctx->opt_host_threads = value; ctx->pool.reset(); }
  else if (!strcmp(name, "field")) { if (value != 0 && value != 1) return LEMSM_ERR_BAD_ARG; ctx->opt_field = value; }
  else return LEMSM_ERR_BAD_ARG;
  return LEMSM_OK;
}

size_t lemsm_last_bad_index(const lemsm_ctx* ctx) { return ctx ? ctx->bad_index : 0; }
size_t lemsm_last_truncated_count(const lemsm_ctx* ctx) { return ctx ? ctx->truncated : 0; }

int lemsm_last_timing(const lemsm_ctx* ctx, double out[3]) {
  if (!ctx || !out) return LEMSM_ERR_BAD_ARG;
  out[0] = ctx->t_total_ms; out[1] = ctx->t_accum_ms; out[2] = ctx->n_accum;
  return LEMSM_OK;
}

int lemsm_debug_last_merge_counts(const lemsm_ctx* ctx, uint64_t out[4]) {
  if (!ctx || !out) return LEMSM_ERR_BAD_ARG;
  for (int q = 0; q < 4; q++) out[q] = ctx->dbg_merge[q];
  if (getenv("LEMSM_DEBUG_STAMPS")) fprintf(stderr, "[lemsm] merge stamps (cycles): queues wave0: setup %u sum %u store %u n %u | final wave0: setup %u sum %u store %u n %u\n",
                                            ctx->dbg_stamps[0], ctx->dbg_stamps[1], ctx->dbg_stamps[2], ctx->dbg_stamps[3] & 0x7fffffffu, ctx->dbg_stamps[4], ctx->dbg_stamps[5], ctx->dbg_stamps[6], ctx->dbg_stamps[7]);
  return LEMSM_OK;
}

int lemsm_last_accum_clock_mhz(const lemsm_ctx* ctx, double* out) {
  if (!ctx || !out) return LEMSM_ERR_BAD_ARG;
  *out = ctx->accum_clock_mhz;
  return LEMSM_OK;
}

int lemsm_msm_plan(const lemsm_ctx* ctx, int curve, size_t n, uint32_t* num_windows, size_t* partial_bytes_per_window) {
  if (curve != LEMSM_BN254_G1 && curve != LEMSM_GRUMPKIN) return LEMSM_ERR_BAD_CURVE;
  MsmPlan mp = make_msm_plan(ctx, curve, n);
  if (num_windows) *num_windows = mp.W;
  if (partial_bytes_per_window) *partial_bytes_per_window = 128;   // one XYZZ window sum
  return LEMSM_OK;
}

int lemsm_msm_partial_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n,
                             uint32_t win_begin, uint32_t win_end, uint8_t* out_partials) {
  if (!ctx || (!out_partials && win_begin != win_end)) return LEMSM_ERR_BAD_ARG;   // an empty range (a rank beyond the last window) writes nothing
  int rc = check_curve(ctx, curve); if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<host::pt> out;
  rc = msm_partial_dispatch(ctx, curve, d_scalars, d_points, n, win_begin, win_end, out);
  if (rc) return rc;
  if (!out.empty()) memcpy(out_partials, out.data(), out.size() * sizeof(host::pt));
  return LEMSM_OK;
}

int lemsm_msm_combine(const lemsm_ctx* ctx, int curve, size_t n, const uint8_t* partials, uint64_t out[12]) {
  if (!partials || !out) return LEMSM_ERR_BAD_ARG;
  if (curve != LEMSM_BN254_G1 && curve != LEMSM_GRUMPKIN) return LEMSM_ERR_BAD_CURVE;
  MsmPlan mp = make_msm_plan(ctx, curve, n);
  std::vector<host::pt> recs((size_t)mp.W);
  memcpy(recs.data(), partials, recs.size() * sizeof(host::pt));
  if (curve == LEMSM_BN254_G1) msm_combine_t<host::FqParams64>(mp, recs.data(), out);
  else msm_combine_t<host::FrParams64>(mp, recs.data(), out);
  return LEMSM_OK;
}

int lemsm_msm_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, uint64_t out[12]) {
  if (!ctx || !out) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (n == 0) { memset(out, 0, 96); return LEMSM_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!(ctx->host_stage && ctx->host_stage->h_points)) { rc = validate_points(ctx, curve, d_points, n); if (rc) return rc; }
  MsmPlan mp = make_msm_plan(ctx, curve, n);
  // host tail: the W window sums in parallel (hostpool.hpp, inside msm_partial_t), then the serial Horner over the
  // windows (c (W - 1) doublings)
  std::vector<host::pt> sums;
  rc = msm_partial_dispatch(ctx, curve, d_scalars, d_points, n, 0, mp.W, sums);
  if (rc) return rc;
  const auto th0 = std::chrono::steady_clock::now();
  if (curve == LEMSM_BN254_G1) msm_combine_t<host::FqParams64>(mp, sums.data(), out);
  else msm_combine_t<host::FrParams64>(mp, sums.data(), out);
  ctx->host_us[3] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th0).count();
  if (getenv("LEMSM_DEBUG_STAMPS")) fprintf(stderr, "[lemsm] host tail (us): device wait %.1f, records + window sums %.1f, (unused) %.1f, Horner %.1f\n", ctx->host_us[0], ctx->host_us[1], ctx->host_us[2], ctx->host_us[3]);
  return LEMSM_OK;
}

struct lemsm_bases { lemsm_ctx* ctx; int device; int curve; size_t n; void* d_points; bool validated; };   // `device` kept here: the context may be gone when the bases are freed
// common part of lemsm_msm_batch_device (scalars resident: d_scalars) and lemsm_msm_batch_with_bases (scalars in host memory: h_scalars)
static int msm_batch_entry(lemsm_ctx* ctx, int curve, const void* const* d_scalars, const uint8_t* const* h_scalars, const lemsm_bases* bases,
                           size_t batch, const void* d_points, size_t n, uint64_t* outs) {
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (batch == 0) return LEMSM_OK;
  if (n == 0) { memset(outs, 0, batch * 96); return LEMSM_OK; }
  for (size_t k = 0; k < batch; k++) if (h_scalars ? !h_scalars[k] : !d_scalars[k]) return LEMSM_ERR_BAD_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (!(bases && bases->validated)) { rc = validate_points(ctx, curve, d_points, n); if (rc) return rc; }
  if (ctx->opt_field == 1 || ctx->opt_groups > 1 || batch == 1) {       // A/B arithmetic, pipelined window groups, or nothing to overlap: one call after the other
    for (size_t k = 0; k < batch; k++) {
      rc = h_scalars ? lemsm_msm_with_bases(ctx, bases, h_scalars[k], n, outs + 12 * k) : lemsm_msm_device(ctx, curve, d_scalars[k], d_points, n, outs + 12 * k);
      if (rc) return rc;
    }
    return LEMSM_OK;
  }
  if (!ctx->peer) { rc = lemsm_create(ctx->device, &ctx->peer); if (rc) return fail(ctx, rc, "lemsm_msm_batch_device: second lane could not be created"); }
  {   // the peer plans exactly like the context
    lemsm_ctx* p = ctx->peer;
    p->opt_window_bits = ctx->opt_window_bits; p->opt_chunk = ctx->opt_chunk; p->opt_tile = ctx->opt_tile; p->opt_field = ctx->opt_field;
    p->opt_accum_waves = ctx->opt_accum_waves; p->opt_groups = ctx->opt_groups; p->opt_slab_bits = ctx->opt_slab_bits; p->opt_abi_points = ctx->opt_abi_points;
    p->opt_stage2x = ctx->opt_stage2x; p->opt_xcd_windows = ctx->opt_xcd_windows; p->opt_entry_ring = ctx->opt_entry_ring; p->opt_pyr_fuse = ctx->opt_pyr_fuse;
    p->opt_ws_canary = ctx->opt_ws_canary; p->opt_binsort = ctx->opt_binsort; p->opt_merge_slice = ctx->opt_merge_slice; p->opt_merge_wave_th = ctx->opt_merge_wave_th;
    p->opt_pyr_first2 = ctx->opt_pyr_first2; p->opt_host_threads = ctx->opt_host_threads; p->plan_world = ctx->plan_world; p->opt_slab_tail = ctx->opt_slab_tail; p->opt_pyr_quad = ctx->opt_pyr_quad; p->opt_scatter_lean = ctx->opt_scatter_lean;
  }
  if (curve == LEMSM_BN254_G1) return msm_batch_t<host::FqParams64, GqLazy>(ctx, curve, d_scalars, d_points, n, batch, outs, h_scalars);
  return msm_batch_t<host::FrParams64, GrLazy>(ctx, curve, d_scalars, d_points, n, batch, outs, h_scalars);
}

int lemsm_msm_batch_device(lemsm_ctx* ctx, int curve, const void* const* d_scalars, size_t batch, const void* d_points, size_t n, uint64_t* outs) {
  if (!ctx || (batch && (!d_scalars || !outs))) return LEMSM_ERR_BAD_ARG;
  return msm_batch_entry(ctx, curve, d_scalars, nullptr, nullptr, batch, d_points, n, outs);
}

int lemsm_msm(lemsm_ctx* ctx, int curve, const uint8_t* scalars, const uint64_t* points, size_t n, uint64_t out[12]) {
  if (!ctx || !out || (n && (!scalars || !points))) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (n == 0) { memset(out, 0, 96); return LEMSM_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // staged slab by slab inside run_windows, the upload of one slab overlapping the kernels of the previous one
  rc = reserve(ctx, ctx->in_s, n * 32); if (rc) return rc;
  rc = reserve(ctx, ctx->in_p, n * 64); if (rc) return rc;
  HostStage hs{scalars, points, (char*)ctx->in_s.p, (char*)ctx->in_p.p};
  if (ctx->opt_validate_points) {   // validated inputs: the points are uploaded whole and checked before any slab runs
    HIPCHK(ctx, hipMemcpy(ctx->in_p.p, points, n * 64, hipMemcpyHostToDevice));
    hs.h_points = nullptr;
  }
  ctx->host_stage = &hs;
  rc = lemsm_msm_device(ctx, curve, ctx->in_s.p, ctx->in_p.p, n, out);
  ctx->host_stage = nullptr;
  return rc;
}
int lemsm_msm_bn254_g1(lemsm_ctx* ctx, const uint8_t* s, const uint64_t* p, size_t n, uint64_t out[12]) { return lemsm_msm(ctx, LEMSM_BN254_G1, s, p, n, out); }
int lemsm_msm_grumpkin(lemsm_ctx* ctx, const uint8_t* s, const uint64_t* p, size_t n, uint64_t out[12]) { return lemsm_msm(ctx, LEMSM_GRUMPKIN, s, p, n, out); }

int lemsm_num_digits(int curve, uint8_t base, uint32_t* d) {
  if (curve != LEMSM_BN254_G1 && curve != LEMSM_GRUMPKIN) return LEMSM_ERR_BAD_CURVE;
  if (base < 2 || !d) return LEMSM_ERR_BAD_BASE;
  *d = logb_ceil8(bound_of(curve), base) + 1;
  return LEMSM_OK;
}

int lemsm_negbase_decompose_batch(lemsm_ctx* ctx, const uint8_t* scalars, size_t n, uint8_t base, uint32_t d, uint8_t* digits) {
  if (!ctx || (n && (!scalars || !digits))) return LEMSM_ERR_BAD_ARG;
  if (base < 2) return fail(ctx, LEMSM_ERR_BAD_BASE, "base must be >= 2");
  if (n == 0 || d == 0) return LEMSM_OK;
  if (n >= ((size_t)1 << 32)) return fail(ctx, LEMSM_ERR_BAD_ARG, "n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = stage(ctx, ctx->in_s, scalars, n * 32); if (rc) return rc;
  size_t dig_bytes = align_up((size_t)n * d, 256);
  rc = reserve(ctx, ctx->in_aux, dig_bytes + 256); if (rc) return rc;
  char* b = (char*)ctx->in_aux.p;
  u32 init[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0xffffffffu, 0, 0xffffffffu, 0};
  HIPCHK(ctx, hipMemcpyAsync(b + dig_bytes, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  hipLaunchKernelGGL(k_negbase_digits, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4*)ctx->in_s.p, (u32)n,
                     (u32)base, d, (const u32*)(b + dig_bytes), 0, (uint8_t*)b, (uint8_t*)nullptr, (u32*)(b + dig_bytes) + 8, 0u, d);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(digits, b, (size_t)n * d, hipMemcpyDeviceToHost, ctx->stream));
  u32 errw[2] = {0, 0};
  HIPCHK(ctx, hipMemcpyAsync(errw, (u32*)(b + dig_bytes) + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->truncated = errw[1];
  return LEMSM_OK;
}

// prepare_scalar_witness over a slice (src/negbase_utils.rs:79-124); see k_scalar_witness for the entry format
int lemsm_prepare_scalar_witness_batch(lemsm_ctx* ctx, const uint8_t* scalars, const uint8_t* negative, size_t n, uint8_t base,
                                       uint32_t num_digits, uint32_t logtable, uint8_t* out_entries, size_t* bad_index) {
  if (!ctx || (n && (!scalars || !out_entries))) return LEMSM_ERR_BAD_ARG;
  if (base < 2) return fail(ctx, LEMSM_ERR_BAD_BASE, "base must be >= 2");
  if (logtable == 0) return fail(ctx, LEMSM_ERR_BAD_ARG, "logtable == 0: the reference divides by it (:82)");
  if (num_digits > 4096) return fail(ctx, LEMSM_ERR_BAD_ARG, "num_digits too large");
  if (n >= ((size_t)1 << 32)) return fail(ctx, LEMSM_ERR_BAD_ARG, "n too large");
  if (n == 0) return LEMSM_OK;
  const u32 d = num_digits, num_limbs = (num_digits + logtable - 1) / logtable, cols = num_limbs + 1;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // (-base)^e for e < d as i128, with the overflow the reference's pow(-(base as i128), e) would hit
  std::vector<WitnessPow> pw(std::max<u32>(d, 1));
  {
    unsigned __int128 mag = 1; bool ovf = false;
    for (u32 e = 0; e < d; e++) {
      pw[e].ovf = ovf ? 1u : 0u; pw[e].pad = 0;
      __int128 v = (e & 1u) ? -(__int128)mag : (__int128)mag;
      pw[e].lo = (unsigned long long)(unsigned __int128)v; pw[e].hi = (long long)(v >> 64);
      if (!ovf) { unsigned __int128 nx = mag * base; if (nx / base != mag || nx >= ((unsigned __int128)1 << 127)) ovf = true; else mag = nx; }
    }
  }
  const size_t dig_bytes = align_up((size_t)n * std::max<u32>(d, 1), 256), flag_bytes = align_up(n, 256), pow_bytes = align_up(pw.size() * sizeof(WitnessPow), 256);
  const size_t out_bytes = (size_t)n * base * cols * 24;
  int rc = stage(ctx, ctx->in_s, scalars, n * 32); if (rc) return rc;
  rc = reserve(ctx, ctx->in_aux, dig_bytes + 2 * flag_bytes + pow_bytes + 512); if (rc) return rc;
  rc = reserve(ctx, ctx->ws, out_bytes + 256); if (rc) return rc;
  char* b = (char*)ctx->in_aux.p;
  uint8_t* d_digits = (uint8_t*)b; uint8_t* d_neg = (uint8_t*)(b + dig_bytes); uint8_t* d_trunc = d_neg + flag_bytes;
  WitnessPow* d_pow = (WitnessPow*)(b + dig_bytes + 2 * flag_bytes);
  u32* d_words = (u32*)(b + dig_bytes + 2 * flag_bytes + pow_bytes);      // [0..7] bound (unused), [8..11] err, [12..13] fail (u64)
  u32 init[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0xffffffffu, 0, 0xffffffffu, 0, 0xffffffffu, 0xffffffffu, 0, 0};
  HIPCHK(ctx, hipMemcpyAsync(d_words, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d_pow, pw.data(), pw.size() * sizeof(WitnessPow), hipMemcpyHostToDevice, ctx->stream));
  if (negative) HIPCHK(ctx, hipMemcpyAsync(d_neg, negative, n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // (init / pw are host stack and heap objects)
  dim3 grid((u32)((n + 255) / 256)), blk(256);
  hipLaunchKernelGGL(k_negbase_digits, grid, blk, 0, ctx->stream, (const uint4*)ctx->in_s.p, (u32)n, (u32)base, d, (const u32*)d_words, 0,
                     d_digits, (uint8_t*)nullptr, d_words + 8, 0u, 0u, negative ? (const uint8_t*)d_neg : (const uint8_t*)nullptr, d_trunc);
  hipLaunchKernelGGL(k_scalar_witness, grid, blk, 0, ctx->stream, (const uint4*)ctx->in_s.p, negative ? (const uint8_t*)d_neg : (const uint8_t*)nullptr,
                     (const uint8_t*)d_digits, (const uint8_t*)d_trunc, (u32)n, (u32)base, d, logtable, num_limbs, (const WitnessPow*)d_pow,
                     (char*)ctx->ws.p, (unsigned long long*)(d_words + 12));
  HIPCHK(ctx, hipGetLastError());
  unsigned long long failw = 0;
  HIPCHK(ctx, hipMemcpyAsync(&failw, d_words + 12, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(out_entries, ctx->ws.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (failw != ~0ull) {
    size_t j = (size_t)(failw >> 4); u32 code = (u32)(failw & 15u);
    if (bad_index) *bad_index = j;
    ctx->bad_index = j;
    if (code == 1) return fail(ctx, LEMSM_ERR_TOO_MANY_DIGITS, "negabase expansion longer than num_digits (assert at src/negbase_utils.rs:81)");
    if (code == 2) return fail(ctx, LEMSM_ERR_INDEX_OUT_OF_BOUNDS, "limb index i % logtable + 1 > num_limbs (index out of bounds at src/negbase_utils.rs:98-101)");
    return fail(ctx, LEMSM_ERR_ARITH_OVERFLOW, "i128 / u32 overflow in pow or += (src/negbase_utils.rs:97-101, a panic in the reference's debug build)");
  }
  return LEMSM_OK;
}

// table_entry_by_id for ids [id_begin, id_begin + count) in the BASE field of `curve` (src/negbase_utils.rs:58-77; the
// circuit's native field, C::Base at src/config.rs:486); out: count x 4 limbs, raw Montgomery
int lemsm_table_entries(lemsm_ctx* ctx, int curve, uint8_t base, uint64_t id_begin, size_t count, uint64_t* out) {
  if (!ctx || (count && !out)) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (count == 0) return LEMSM_OK;
  if (count >= ((size_t)1 << 31)) return fail(ctx, LEMSM_ERR_BAD_ARG, "count too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  rc = reserve(ctx, ctx->ws, count * 32 + 256); if (rc) return rc;
  dim3 grid((u32)((count + 255) / 256)), blk(256);
  if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_table_entries<FqDev>), grid, blk, 0, ctx->stream, (u32)base, (unsigned long long)id_begin, (u32)count, (uint4*)ctx->ws.p);
  else hipLaunchKernelGGL((k_table_entries<FrDev>), grid, blk, 0, ctx->stream, (u32)base, (unsigned long long)id_begin, (u32)count, (uint4*)ctx->ws.p);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->ws.p, count * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LEMSM_OK;
}

int lemsm_lhs_plan(int curve, uint8_t base, uint32_t* num_positions, size_t* partial_bytes) {
  if (curve != LEMSM_BN254_G1 && curve != LEMSM_GRUMPKIN) return LEMSM_ERR_BAD_CURVE;
  LhsPlan lp; int rc = make_lhs_plan(curve, base, lp); if (rc) return rc;
  if (num_positions) *num_positions = lp.d;
  if (partial_bytes) *partial_bytes = 128;
  return LEMSM_OK;
}

int lemsm_lhs_partial_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, uint8_t base,
                             uint32_t pos_begin, uint32_t pos_end, uint8_t* out_partials, size_t* bad_index) {
  if (!ctx || (!out_partials && pos_begin != pos_end)) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  LhsPlan lp; rc = make_lhs_plan(curve, base, lp); if (rc) return fail(ctx, rc, "base must be in 3..=255");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<host::pt> out;
  rc = lhs_partial_dispatch(ctx, curve, d_scalars, d_points, n, lp, pos_begin, pos_end, out, bad_index);
  if (rc) return rc;
  if (!out.empty()) memcpy(out_partials, out.data(), out.size() * sizeof(host::pt));
  return LEMSM_OK;
}

int lemsm_lhs_combine(int curve, uint8_t base, const uint8_t* partials, uint64_t out_carry[12], uint64_t* out_carries) {
  if (!partials || !out_carry) return LEMSM_ERR_BAD_ARG;
  if (curve != LEMSM_BN254_G1 && curve != LEMSM_GRUMPKIN) return LEMSM_ERR_BAD_CURVE;
  LhsPlan lp; int rc = make_lhs_plan(curve, base, lp); if (rc) return rc;
  std::vector<host::pt> recs((size_t)lp.d);
  memcpy(recs.data(), partials, recs.size() * sizeof(host::pt));
  if (curve == LEMSM_BN254_G1) lhs_combine_t<host::FqParams64>(lp, recs.data(), out_carry, out_carries);
  else lhs_combine_t<host::FrParams64>(lp, recs.data(), out_carry, out_carries);
  return LEMSM_OK;
}

int lemsm_lhs_msm_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, uint8_t base,
                         uint64_t out_carry[12], uint64_t* out_carries, size_t* bad_index) {
  if (!ctx || !out_carry) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  LhsPlan lp; rc = make_lhs_plan(curve, base, lp); if (rc) return fail(ctx, rc, "base must be in 3..=255");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  rc = validate_points(ctx, curve, d_points, n); if (rc) return rc;
  std::vector<host::pt> recs;
  rc = lhs_partial_dispatch(ctx, curve, d_scalars, d_points, n, lp, 0, lp.d, recs, bad_index);
  if (rc) return rc;
  if (curve == LEMSM_BN254_G1) lhs_combine_t<host::FqParams64>(lp, recs.data(), out_carry, out_carries);
  else lhs_combine_t<host::FrParams64>(lp, recs.data(), out_carry, out_carries);
  return LEMSM_OK;
}

int lemsm_lhs_msm(lemsm_ctx* ctx, int curve, const uint8_t* scalars, const uint64_t* pts_jac, size_t n, uint8_t base,
                  uint64_t out_carry[12], uint64_t* out_carries, size_t* bad_index) {
  if (!ctx || !out_carry || (n && (!scalars || !pts_jac))) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (base < 3) return fail(ctx, LEMSM_ERR_BAD_BASE, "base must be in 3..=255");
  if (n >= ((size_t)1 << 32)) return fail(ctx, LEMSM_ERR_BAD_ARG, "n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  rc = stage(ctx, ctx->in_s, scalars, n * 32); if (rc) return rc;
  // Jacobian -> affine on the device (workspace holds the staged Jacobian points + scratch)
  const int KB = 32;
  size_t nthreads = ((n + KB - 1) / KB + 255) / 256 * 256;
  size_t jac_bytes = align_up(n * 96, 256), scr_bytes = (size_t)KB * nthreads * 32;
  rc = reserve(ctx, ctx->ws, jac_bytes + scr_bytes + 256); if (rc) return rc;
  rc = reserve(ctx, ctx->in_p, n ? n * 64 : 16); if (rc) return rc;
  if (n) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws.p, pts_jac, n * 96, hipMemcpyHostToDevice, ctx->stream));
    if (curve == LEMSM_BN254_G1)
      hipLaunchKernelGGL((k_jac_to_affine<FqDev, KB>), dim3((u32)(nthreads / 256)), dim3(256), 0, ctx->stream, (const uint4*)ctx->ws.p, (u32)n,
                         (uint4*)ctx->in_p.p, (char*)ctx->ws.p + jac_bytes);
    else
      hipLaunchKernelGGL((k_jac_to_affine<FrDev, KB>), dim3((u32)(nthreads / 256)), dim3(256), 0, ctx->stream, (const uint4*)ctx->ws.p, (u32)n,
                         (uint4*)ctx->in_p.p, (char*)ctx->ws.p + jac_bytes);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // the workspace is re-carved by the MSM below
  }
  return lemsm_lhs_msm_device(ctx, curve, ctx->in_s.p, ctx->in_p.p, n, base, out_carry, out_carries, bad_index);
}
int lemsm_lhs_msm_grumpkin(lemsm_ctx* ctx, const uint8_t* s, const uint64_t* p, size_t n, uint8_t base, uint64_t oc[12], uint64_t* ocs, size_t* bad) {
  return lemsm_lhs_msm(ctx, LEMSM_GRUMPKIN, s, p, n, base, oc, ocs, bad);
}
int lemsm_lhs_msm_bn254_g1(lemsm_ctx* ctx, const uint8_t* s, const uint64_t* p, size_t n, uint8_t base, uint64_t oc[12], uint64_t* ocs, size_t* bad) {
  return lemsm_lhs_msm(ctx, LEMSM_BN254_G1, s, p, n, base, oc, ocs, bad);
}

// ---- multi-GPU entries ---------------------------------------------------------------------------------------
int lemsm_comm_unique_id(uint8_t id[LEMSM_COMM_ID_BYTES]) {
  if (!id) return LEMSM_ERR_BAD_ARG;
  Rccl& R = Rccl::get();
  if (!R.ok()) return LEMSM_ERR_RCCL;
  static_assert(LEMSM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  if (R.GetUniqueId(&u) != ncclSuccess) return LEMSM_ERR_RCCL;
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return LEMSM_OK;
}

int lemsm_comm_init(lemsm_ctx* ctx, const uint8_t id[LEMSM_COMM_ID_BYTES], int nranks, int rank) {
  if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return LEMSM_ERR_BAD_ARG;
  Rccl& R = Rccl::get();
  if (!R.ok()) return fail(ctx, LEMSM_ERR_RCCL, "RCCL not loadable: " + R.error);
  if (ctx->comm) return fail(ctx, LEMSM_ERR_BAD_ARG, "context already has a communicator (lemsm_comm_destroy first)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ncclUniqueId u; memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  RCCLCHK(ctx, R.CommInitRank(&ctx->comm, nranks, u, rank));
  ctx->comm_size = nranks; ctx->comm_rank = rank;
  return LEMSM_OK;
}

int lemsm_comm_destroy(lemsm_ctx* ctx) {
  if (!ctx) return LEMSM_ERR_BAD_ARG;
  if (ctx->comm) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ncclResult_t r = Rccl::get().CommDestroy(ctx->comm);
    ctx->comm = nullptr; ctx->comm_size = 1; ctx->comm_rank = 0;
    if (r != ncclSuccess) return fail(ctx, LEMSM_ERR_RCCL, std::string("ncclCommDestroy: ") + Rccl::get().GetErrorString(r));
  }
  return LEMSM_OK;
}

int lemsm_comm_info(const lemsm_ctx* ctx, int* nranks, int* rank) {
  if (!ctx) return LEMSM_ERR_BAD_ARG;
  if (nranks) *nranks = ctx->comm ? ctx->comm_size : 0;
  if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
  return LEMSM_OK;
}

int lemsm_msm_sharded_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, uint64_t out[12]) {
  if (!ctx || !out) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (!ctx->comm) return fail(ctx, LEMSM_ERR_BAD_ARG, "no communicator: call lemsm_comm_init on every rank first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Exchange ex{ctx->comm_size, ctx->comm_rank, false};
  return msm_sharded_dispatch(ctx, curve, d_scalars, d_points, n, ex, out);
}

int lemsm_lhs_msm_sharded_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, uint8_t base,
                                 uint64_t out_carry[12], uint64_t* out_carries, size_t* bad_index) {
  if (!ctx || !out_carry) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  LhsPlan lp; rc = make_lhs_plan(curve, base, lp); if (rc) return fail(ctx, rc, "base must be in 3..=255");
  if (!ctx->comm) return fail(ctx, LEMSM_ERR_BAD_ARG, "no communicator: call lemsm_comm_init on every rank first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Exchange ex{ctx->comm_size, ctx->comm_rank, false};
  return lhs_sharded_dispatch(ctx, curve, d_scalars, d_points, n, lp, ex, out_carry, out_carries, bad_index);
}

// one-GPU rehearsal of the sharded entries (tests): all `world` ranks' pipelines on this context, one after the other
int lemsm_debug_msm_sharded_sim(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, int world, uint64_t out[12]) {
  if (!ctx || !out || world < 1 || world > 64) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Exchange ex{world, 0, true};
  return msm_sharded_dispatch(ctx, curve, d_scalars, d_points, n, ex, out);
}
int lemsm_debug_lhs_sharded_sim(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points, size_t n, uint8_t base, int world,
                                uint64_t out_carry[12], uint64_t* out_carries, size_t* bad_index) {
  if (!ctx || !out_carry || world < 1 || world > 256) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  LhsPlan lp; rc = make_lhs_plan(curve, base, lp); if (rc) return fail(ctx, rc, "base must be in 3..=255");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Exchange ex{world, 0, true};
  return lhs_sharded_dispatch(ctx, curve, d_scalars, d_points, n, lp, ex, out_carry, out_carries, bad_index);
}

// ---- resident bases (halo2's bases are a fixed SRS: upload once, then only scalars cross PCIe) ------------------

int lemsm_bases_upload(lemsm_ctx* ctx, int curve, const uint64_t* points_affine, size_t n, lemsm_bases** out) {
  if (!ctx || !out || (n && !points_affine)) return LEMSM_ERR_BAD_ARG;
  *out = nullptr;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, n ? n * 64 : 64);
  if (e != hipSuccess) return fail(ctx, LEMSM_ERR_NOMEM, hipGetErrorString(e));
  if (n) { e = hipMemcpy(d, points_affine, n * 64, hipMemcpyHostToDevice); if (e != hipSuccess) { (void)hipFree(d); return fail(ctx, LEMSM_ERR_HIP, hipGetErrorString(e)); } }
  // option validate_points: resident bases are checked ONCE, here, not on every call that uses them
  if (ctx->opt_validate_points) { rc = validate_points(ctx, curve, d, n); if (rc) { (void)hipFree(d); return rc; } }
  *out = new lemsm_bases{ctx, ctx->device, curve, n, d, ctx->opt_validate_points != 0};
  return LEMSM_OK;
}
void lemsm_bases_free(lemsm_bases* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);              // (never through b->ctx: a host may destroy the context first)
  (void)hipFree(b->d_points);
  delete b;
}
const void* lemsm_bases_device_ptr(const lemsm_bases* b) { return b ? b->d_points : nullptr; }

// sum_i scalars[i] * bases[i] for the first n bases; scalars are host memory, uploaded slab by slab under the kernels
int lemsm_msm_with_bases(lemsm_ctx* ctx, const lemsm_bases* bases, const uint8_t* scalars, size_t n, uint64_t out[12]) {
  if (!ctx || !bases || !out || (n && !scalars)) return LEMSM_ERR_BAD_ARG;
  if (bases->ctx != ctx) return fail(ctx, LEMSM_ERR_BAD_ARG, "bases belong to another context");
  if (n > bases->n) return fail(ctx, LEMSM_ERR_LEN_MISMATCH, "more scalars than resident bases");
  if (n == 0) { memset(out, 0, 96); return LEMSM_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = reserve(ctx, ctx->in_s, n * 32); if (rc) return rc;
  HostStage hs{scalars, nullptr, (char*)ctx->in_s.p, (char*)bases->d_points};
  ctx->host_stage = &hs;
  const long saved_validate = ctx->opt_validate_points;
  if (bases->validated) ctx->opt_validate_points = 0;   // checked at upload
  rc = lemsm_msm_device(ctx, bases->curve, ctx->in_s.p, bases->d_points, n, out);
  ctx->opt_validate_points = saved_validate;
  ctx->host_stage = nullptr;
  return rc;
}

// `batch` MSMs over the same resident bases, scalars in host memory: the upload of call k's scalars (its lane's second queue)
// and the host fold of call k - 1 run while the GPU accumulates call k - 1 / k; per call the longer of the two, not their sum.
int lemsm_msm_batch_with_bases(lemsm_ctx* ctx, const lemsm_bases* bases, const uint8_t* const* scalars, size_t batch, size_t n, uint64_t* outs) {
  if (!ctx || !bases || (batch && (!scalars || !outs))) return LEMSM_ERR_BAD_ARG;
  if (bases->ctx != ctx) return fail(ctx, LEMSM_ERR_BAD_ARG, "bases belong to another context");
  if (n > bases->n) return fail(ctx, LEMSM_ERR_LEN_MISMATCH, "more scalars than resident bases");
  return msm_batch_entry(ctx, bases->curve, nullptr, scalars, bases, batch, bases->d_points, n, outs);
}

// ---- one process, all GPUs of the node: what a Rust host binds -----------------------------
// One context + one communicator per device, one host thread per device for the duration of a call.
struct lemsm_node {
  std::vector<lemsm_ctx*> ctx;
  std::vector<void*> d_scalars, d_points;   // per device: replicated inputs (bases resident across calls)
  std::vector<size_t> cap_s;
  int curve = -1; size_t n_bases = 0;
  std::string last_error;
};

int lemsm_node_create(const int* devices, int ndev, lemsm_node** out) {
  if (!out || ndev < 1 || ndev > 64) return LEMSM_ERR_BAD_ARG;
  *out = nullptr;
  lemsm_node* nd = new lemsm_node();
  int rc = LEMSM_OK;
  for (int i = 0; i < ndev && !rc; i++) {
    lemsm_ctx* c = nullptr;
    rc = lemsm_create(devices ? devices[i] : i, &c);
    if (!rc) nd->ctx.push_back(c);
  }
  uint8_t id[LEMSM_COMM_ID_BYTES];
  if (!rc) rc = lemsm_comm_unique_id(id);
  if (!rc) {   // ncclCommInitRank blocks until every rank has joined: one thread per rank
    std::vector<int> rcs(ndev, 0);
    std::vector<std::thread> th;
    for (int i = 0; i < ndev; i++) th.emplace_back([&, i] { rcs[i] = lemsm_comm_init(nd->ctx[i], id, ndev, i); });
    for (auto& t : th) t.join();
    for (int i = 0; i < ndev; i++) if (rcs[i] && !rc) rc = rcs[i];
  }
  if (rc) { lemsm_node_destroy(nd); return rc; }
  nd->d_scalars.assign(ndev, nullptr); nd->d_points.assign(ndev, nullptr); nd->cap_s.assign(ndev, 0);
  *out = nd;
  return LEMSM_OK;
}

void lemsm_node_destroy(lemsm_node* nd) {
  if (!nd) return;
  for (size_t i = 0; i < nd->ctx.size(); i++) {
    (void)hipSetDevice(nd->ctx[i]->device);
    if (i < nd->d_scalars.size() && nd->d_scalars[i]) (void)hipFree(nd->d_scalars[i]);
    if (i < nd->d_points.size() && nd->d_points[i]) (void)hipFree(nd->d_points[i]);
    (void)lemsm_comm_destroy(nd->ctx[i]);
    lemsm_destroy(nd->ctx[i]);
  }
  delete nd;
}

int lemsm_node_size(const lemsm_node* nd) { return nd ? (int)nd->ctx.size() : 0; }
lemsm_ctx* lemsm_node_ctx(lemsm_node* nd, int i) { return (nd && i >= 0 && i < (int)nd->ctx.size()) ? nd->ctx[i] : nullptr; }
const char* lemsm_node_last_error(const lemsm_node* nd) { return nd ? nd->last_error.c_str() : "null node"; }

// runs fn(rank) on one thread per device and returns the first non-zero status (its message kept in the node)
static int node_parallel(lemsm_node* nd, const std::function<int(int)>& fn) {
  const int G = (int)nd->ctx.size();
  std::vector<int> rcs(G, 0);
  if (G == 1) rcs[0] = fn(0);
  else {
    std::vector<std::thread> th;
    for (int i = 0; i < G; i++) th.emplace_back([&, i] { rcs[i] = fn(i); });
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < G; i++) if (rcs[i]) { nd->last_error = "rank " + std::to_string(i) + ": " + nd->ctx[i]->last_error; return rcs[i]; }
  return LEMSM_OK;
}

// Bases (affine, n x 8 limbs) replicated into every device's HBM once; they stay resident across lemsm_node_* calls.
int lemsm_node_set_bases(lemsm_node* nd, int curve, const uint64_t* points_affine, size_t n) {
  if (!nd || (n && !points_affine)) return LEMSM_ERR_BAD_ARG;
  if (curve != LEMSM_BN254_G1 && curve != LEMSM_GRUMPKIN) return LEMSM_ERR_BAD_CURVE;
  int rc = node_parallel(nd, [&](int i) -> int {
    lemsm_ctx* c = nd->ctx[i];
    HIPCHK(c, hipSetDevice(c->device));
    if (nd->d_points[i]) { HIPCHK(c, hipFree(nd->d_points[i])); nd->d_points[i] = nullptr; }
    hipError_t e = hipMalloc(&nd->d_points[i], n ? n * 64 : 64);
    if (e != hipSuccess) return fail(c, LEMSM_ERR_NOMEM, hipGetErrorString(e));
    if (n) HIPCHK(c, hipMemcpy(nd->d_points[i], points_affine, n * 64, hipMemcpyHostToDevice));
    return LEMSM_OK;
  });
  if (rc) return rc;
  nd->curve = curve; nd->n_bases = n;
  return LEMSM_OK;
}

// Staging of a call's scalars on every GPU, in two steps so that no rank can be left waiting in a collective:
// node_reserve_scalars (no collective: allocation only) and node_stage_scalars.  Every GPU uploads ONE G-th of the vector
// over its own PCIe link and an in-place ncclAllGather over xGMI completes the copies: the host's memory is read once, not
// G times (r02 uploaded the whole vector to every GPU from pageable memory).  The chunk is a whole number of scalars; the
// buffers hold G chunks (>= n scalars; what lies beyond n is never read).
static size_t node_chunk_scalars(const lemsm_node* nd, size_t n) { const size_t G = nd->ctx.size(); return (n + G - 1) / G; }
static int node_reserve_scalars(lemsm_node* nd, int i, size_t n) {
  lemsm_ctx* c = nd->ctx[i];
  HIPCHK(c, hipSetDevice(c->device));
  const size_t need = node_chunk_scalars(nd, n) * nd->ctx.size() * 32 + 64;
  if (nd->cap_s[i] < need) {
    if (nd->d_scalars[i]) { HIPCHK(c, hipFree(nd->d_scalars[i])); nd->d_scalars[i] = nullptr; nd->cap_s[i] = 0; }
    hipError_t e = hipMalloc(&nd->d_scalars[i], need);
    if (e != hipSuccess) return fail(c, LEMSM_ERR_NOMEM, hipGetErrorString(e));
    nd->cap_s[i] = need;
  }
  return LEMSM_OK;
}
static int node_stage_scalars(lemsm_node* nd, int i, const uint8_t* scalars, size_t n) {
  lemsm_ctx* c = nd->ctx[i];
  HIPCHK(c, hipSetDevice(c->device));
  const size_t G = nd->ctx.size(), chunk = node_chunk_scalars(nd, n);
  if (n == 0) return LEMSM_OK;
  const size_t lo = std::min(n, (size_t)i * chunk), hi = std::min(n, lo + chunk);
  char* mine = (char*)nd->d_scalars[i] + (size_t)i * chunk * 32;
  hipError_t e = hi > lo ? hipMemcpy(mine, scalars + lo * 32, (hi - lo) * 32, hipMemcpyHostToDevice) : hipSuccess;
  if (G == 1) { if (e != hipSuccess) return fail(c, LEMSM_ERR_HIP, hipGetErrorString(e)); return LEMSM_OK; }
  if (e != hipSuccess) { (void)fail(c, LEMSM_ERR_HIP, hipGetErrorString(e)); comm_abort(c); return LEMSM_ERR_HIP; }   // peers are (about to be) in the all-gather
  if (!c->comm) return fail(c, LEMSM_ERR_RCCL, "node: no communicator");
  ncclResult_t r_ = Rccl::get().AllGather(mine, nd->d_scalars[i], chunk * 32, ncclUint8, c->comm, c->stream);   // in place: send = recv + rank * count
  if (r_ != ncclSuccess) { c->last_error = std::string("ncclAllGather (scalars): ") + Rccl::get().GetErrorString(r_); comm_abort(c); return LEMSM_ERR_RCCL; }
  return LEMSM_OK;                                 // (the MSM that follows is enqueued on the same queue: stream order)
}

// best_multiexp over the node: scalars (host) are replicated to every GPU, each GPU accumulates its Pippenger windows,
// one RCCL all-gather of the raw window records, every GPU combines; out = rank 0's result (all ranks' agree).
int lemsm_node_msm(lemsm_node* nd, const uint8_t* scalars, size_t n, uint64_t out[12]) {
  if (!nd || !out || (n && !scalars)) return LEMSM_ERR_BAD_ARG;
  if (nd->curve < 0) { nd->last_error = "lemsm_node_set_bases first"; return LEMSM_ERR_BAD_ARG; }
  if (n > nd->n_bases) { nd->last_error = "more scalars than resident bases"; return LEMSM_ERR_LEN_MISMATCH; }
  std::vector<uint64_t> outs((size_t)nd->ctx.size() * 12);
  // phase 1: buffers first (no collective: a failed allocation ends the call before any rank waits), then the staging
  int rc = node_parallel(nd, [&](int i) -> int { return node_reserve_scalars(nd, i, n); });
  if (rc) return rc;
  rc = node_parallel(nd, [&](int i) -> int { return node_stage_scalars(nd, i, scalars, n); });
  if (rc) return rc;
  // phase 2: the collective call (itself collective-safe: sharded_records)
  rc = node_parallel(nd, [&](int i) -> int {
    return lemsm_msm_sharded_device(nd->ctx[i], nd->curve, nd->d_scalars[i], nd->d_points[i], n, outs.data() + 12 * (size_t)i);
  });
  if (rc) return rc;
  memcpy(out, outs.data(), 96);
  return LEMSM_OK;
}

// compute_lhs_witness MSM core over the node, digit positions sharded; bases as set by lemsm_node_set_bases (affine).
int lemsm_node_lhs_msm(lemsm_node* nd, const uint8_t* scalars, size_t n, uint8_t base, uint64_t out_carry[12], uint64_t* out_carries,
                       size_t* bad_index) {
  if (!nd || !out_carry || (n && !scalars)) return LEMSM_ERR_BAD_ARG;
  if (nd->curve < 0) { nd->last_error = "lemsm_node_set_bases first"; return LEMSM_ERR_BAD_ARG; }
  if (n != nd->n_bases) { nd->last_error = "incompatible amount of coefficients"; return LEMSM_ERR_LEN_MISMATCH; }
  uint32_t d = 0; int rc = lemsm_num_digits(nd->curve, base, &d); if (rc || base < 3) return LEMSM_ERR_BAD_BASE;
  const size_t G = nd->ctx.size();
  std::vector<uint64_t> carry(G * 12), carries(out_carries ? G * 12 * d : 0);
  std::vector<size_t> bad(G, 0);
  rc = node_parallel(nd, [&](int i) -> int { return node_reserve_scalars(nd, i, n); });          // phase 1, no collective
  if (rc) return rc;
  rc = node_parallel(nd, [&](int i) -> int { return node_stage_scalars(nd, i, scalars, n); });   // one G-th each over PCIe, all-gather over xGMI
  if (rc) return rc;
  rc = node_parallel(nd, [&](int i) -> int {
    return lemsm_lhs_msm_sharded_device(nd->ctx[i], nd->curve, nd->d_scalars[i], nd->d_points[i], n, base, carry.data() + 12 * (size_t)i,
                                        out_carries ? carries.data() + (size_t)i * 12 * d : nullptr, &bad[i]);
  });
  if (rc) { if (bad_index) *bad_index = bad[0]; return rc; }
  memcpy(out_carry, carry.data(), 96);
  if (out_carries) memcpy(out_carries, carries.data(), (size_t)12 * d * 8);
  return LEMSM_OK;
}

int lemsm_precompute_multiplicities(lemsm_ctx* ctx, int curve, const uint64_t* pts_jac, size_t n, uint8_t base, uint64_t* out) {
  if (!ctx || (n && (!pts_jac || !out))) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (base < 2) return fail(ctx, LEMSM_ERR_BAD_BASE, "base must be >= 2");
  if (n == 0 || base == 1) return LEMSM_OK;
  if (n >= ((size_t)1 << 31)) return fail(ctx, LEMSM_ERR_BAD_ARG, "n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t in_bytes = align_up(n * 96, 256), out_bytes = n * (size_t)(base - 1) * 96;
  rc = reserve(ctx, ctx->ws, in_bytes + out_bytes + 256); if (rc) return rc;
  char* b = (char*)ctx->ws.p;
  HIPCHK(ctx, hipMemcpyAsync(b, pts_jac, n * 96, hipMemcpyHostToDevice, ctx->stream));
  if (curve == LEMSM_BN254_G1)
    hipLaunchKernelGGL((k_precompute_mult<FqDev>), dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4*)b, (u32)n, (u32)base, (uint4*)(b + in_bytes));
  else
    hipLaunchKernelGGL((k_precompute_mult<FrDev>), dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4*)b, (u32)n, (u32)base, (uint4*)(b + in_bytes));
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, b + in_bytes, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LEMSM_OK;
}

int lemsm_precompute_multiplicities_affine(lemsm_ctx* ctx, int curve, const uint64_t* pts_jac, size_t n, uint8_t base, uint64_t* out) {
  if (!ctx || (n && (!pts_jac || !out))) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (base < 2) return fail(ctx, LEMSM_ERR_BAD_BASE, "base must be >= 2");
  if (n == 0 || base == 1) return LEMSM_OK;
  if (n >= ((size_t)1 << 28)) return fail(ctx, LEMSM_ERR_BAD_ARG, "n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t nm = n * (size_t)(base - 1);
  size_t in_bytes = align_up(n * 96, 256), out_bytes = align_up(nm * 64, 256), scr_bytes = nm * 160;
  rc = reserve(ctx, ctx->ws, in_bytes + out_bytes + scr_bytes + 256); if (rc) return rc;
  char* b = (char*)ctx->ws.p;
  HIPCHK(ctx, hipMemcpyAsync(b, pts_jac, n * 96, hipMemcpyHostToDevice, ctx->stream));
  if (curve == LEMSM_BN254_G1)
    hipLaunchKernelGGL((k_precompute_mult_affine<FqDev>), dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4*)b, (u32)n, (u32)base,
                       (uint4*)(b + in_bytes), b + in_bytes + out_bytes);
  else
    hipLaunchKernelGGL((k_precompute_mult_affine<FrDev>), dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4*)b, (u32)n, (u32)base,
                       (uint4*)(b + in_bytes), b + in_bytes + out_bytes);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, b + in_bytes, nm * 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LEMSM_OK;
}

int lemsm_jacobian_to_canonical(int curve, const uint64_t jac[12], uint8_t out[64]) {
  if (!jac || !out) return LEMSM_ERR_BAD_ARG;
  if (curve != LEMSM_BN254_G1 && curve != LEMSM_GRUMPKIN) return LEMSM_ERR_BAD_CURVE;
  u64 aff[8];
  host::fe one_plain = {{1, 0, 0, 0}};
  if (curve == LEMSM_BN254_G1) {
    typedef host::HG<host::FqParams64> G; typedef host::HF<host::FqParams64> F;
    host::pt p = G::from_jacobian(jac); G::to_affine(p, aff);
    if (G::is_identity(p)) { memset(out, 0, 64); return LEMSM_OK; }
    host::fe x, y; memcpy(x.l, aff, 32); memcpy(y.l, aff + 4, 32);
    x = F::mul(x, one_plain); y = F::mul(y, one_plain);
    memcpy(out, x.l, 32); memcpy(out + 32, y.l, 32);
  } else {
    typedef host::HG<host::FrParams64> G; typedef host::HF<host::FrParams64> F;
    host::pt p = G::from_jacobian(jac); G::to_affine(p, aff);
    if (G::is_identity(p)) { memset(out, 0, 64); return LEMSM_OK; }
    host::fe x, y; memcpy(x.l, aff, 32); memcpy(y.l, aff + 4, 32);
    x = F::mul(x, one_plain); y = F::mul(y, one_plain);
    memcpy(out, x.l, 32); memcpy(out + 32, y.l, 32);
  }
  return LEMSM_OK;
}

int lemsm_jacobian_sum(int curve, const uint64_t* jac, size_t count, uint64_t out[12]) {
  if (!out || (count && !jac)) return LEMSM_ERR_BAD_ARG;
  if (curve == LEMSM_BN254_G1) jacobian_sum_t<host::FqParams64>(jac, count, out);
  else if (curve == LEMSM_GRUMPKIN) jacobian_sum_t<host::FrParams64>(jac, count, out);
  else return LEMSM_ERR_BAD_CURVE;
  return LEMSM_OK;
}

int lemsm_device_alloc(lemsm_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return LEMSM_ERR_BAD_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(out, bytes ? bytes : 16);
  if (e != hipSuccess) return fail(ctx, LEMSM_ERR_NOMEM, hipGetErrorString(e));
  return LEMSM_OK;
}
int lemsm_device_free(lemsm_ctx* ctx, void* p) {
  if (!ctx) return LEMSM_ERR_BAD_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipFree(p));
  return LEMSM_OK;
}
int lemsm_device_upload(lemsm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return LEMSM_ERR_BAD_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return LEMSM_OK;
}
int lemsm_device_download(lemsm_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return LEMSM_ERR_BAD_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return LEMSM_OK;
}

int lemsm_device_gen_walk(lemsm_ctx* ctx, int curve, const uint64_t q_affine[8], size_t n, void* d_out) {
  if (!ctx || !q_affine || (n && !d_out)) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (n == 0) return LEMSM_OK;
  if (n >= ((size_t)1 << 31)) return fail(ctx, LEMSM_ERR_BAD_ARG, "n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int KB = 32;
  size_t nthreads = ((n + KB - 1) / KB + 255) / 256 * 256;
  size_t scr = (size_t)KB * nthreads * 160;
  rc = reserve(ctx, ctx->ws, scr + 512); if (rc) return rc;
  char* b = (char*)ctx->ws.p;
  HIPCHK(ctx, hipMemcpyAsync(b, q_affine, 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (curve == LEMSM_BN254_G1)
    hipLaunchKernelGGL((k_gen_walk<FqDev, KB>), dim3((u32)(nthreads / 256)), dim3(256), 0, ctx->stream, (const uint4*)b, (u32)n, (uint4*)d_out, b + 256);
  else
    hipLaunchKernelGGL((k_gen_walk<FrDev, KB>), dim3((u32)(nthreads / 256)), dim3(256), 0, ctx->stream, (const uint4*)b, (u32)n, (uint4*)d_out, b + 256);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LEMSM_OK;
}

// ---- debug hooks ----
static int dbg_run3(lemsm_ctx* ctx, const void* a, size_t abytes, const void* b, size_t bbytes, size_t obytes, char** da, char** db, char** dout) {
  size_t oa = 0, ob = align_up(abytes, 256), oo = ob + align_up(bbytes, 256);
  int rc = reserve(ctx, ctx->ws, oo + obytes + 256); if (rc) return rc;
  char* base = (char*)ctx->ws.p;
  *da = base + oa; *db = base + ob; *dout = base + oo;
  if (abytes) HIPCHK(ctx, hipMemcpyAsync(*da, a, abytes, hipMemcpyHostToDevice, ctx->stream));
  if (bbytes) HIPCHK(ctx, hipMemcpyAsync(*db, b, bbytes, hipMemcpyHostToDevice, ctx->stream));
  return LEMSM_OK;
}

int lemsm_debug_montmul(lemsm_ctx* ctx, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!ctx || !a || !b || !out) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (!n) return LEMSM_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  char *da, *db, *dout;
  rc = dbg_run3(ctx, a, n * 32, b, n * 32, n * 32, &da, &db, &dout); if (rc) return rc;
  dim3 g((u32)((n + 255) / 256)), blk(256);
  if (ctx->opt_field == 0) {   // the hot kernels' default arithmetic
    if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_dbg_fieldop29<Field29<Fq29Params>>), g, blk, 0, ctx->stream, 8, (const uint4*)da, (const uint4*)db, (uint4*)dout, (u32)n);
    else hipLaunchKernelGGL((k_dbg_fieldop29<Field29<Fr29Params>>), g, blk, 0, ctx->stream, 8, (const uint4*)da, (const uint4*)db, (uint4*)dout, (u32)n);
  } else if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_dbg_montmul<FqDev>), g, blk, 0, ctx->stream, (const uint4*)da, (const uint4*)db, (uint4*)dout, (u32)n);
  else hipLaunchKernelGGL((k_dbg_montmul<FrDev>), g, blk, 0, ctx->stream, (const uint4*)da, (const uint4*)db, (uint4*)dout, (u32)n);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LEMSM_OK;
}

int lemsm_debug_fieldop(lemsm_ctx* ctx, int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!ctx || !a || !b || !out) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (!n) return LEMSM_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  char *da, *db, *dout;
  rc = dbg_run3(ctx, a, n * 32, b, n * 32, n * 32, &da, &db, &dout); if (rc) return rc;
  dim3 g((u32)((n + 255) / 256)), blk(256);
  if (ctx->opt_field == 0 && op == 3) {
    if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_dbg_inv_lazy<FqDev>), g, blk, 0, ctx->stream, (const uint4*)da, (uint4*)dout, (u32)n);
    else hipLaunchKernelGGL((k_dbg_inv_lazy<FrDev>), g, blk, 0, ctx->stream, (const uint4*)da, (uint4*)dout, (u32)n);
  } else if (ctx->opt_field == 0) {
    if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_dbg_fieldop29<Field29<Fq29Params>>), g, blk, 0, ctx->stream, op, (const uint4*)da, (const uint4*)db, (uint4*)dout, (u32)n);
    else hipLaunchKernelGGL((k_dbg_fieldop29<Field29<Fr29Params>>), g, blk, 0, ctx->stream, op, (const uint4*)da, (const uint4*)db, (uint4*)dout, (u32)n);
  } else if (op > 4) return fail(ctx, LEMSM_ERR_BAD_ARG, "field ops 5..9 exist in the lazy field only (option field = 0)");
  else if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_dbg_fieldop<FqDev>), g, blk, 0, ctx->stream, op, (const uint4*)da, (const uint4*)db, (uint4*)dout, (u32)n);
  else hipLaunchKernelGGL((k_dbg_fieldop<FrDev>), g, blk, 0, ctx->stream, op, (const uint4*)da, (const uint4*)db, (uint4*)dout, (u32)n);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, n * 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LEMSM_OK;
}

int lemsm_debug_pointop(lemsm_ctx* ctx, int curve, int op, const uint64_t* acc, const uint64_t* q, uint64_t* out, size_t n) {
  if (!ctx || !acc || !q || !out) return LEMSM_ERR_BAD_ARG;
  int rc = check_curve(ctx, curve); if (rc) return rc;
  if (!n) return LEMSM_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (op < 0 || op > 2) return fail(ctx, LEMSM_ERR_BAD_ARG, "point op: 0 madd, 1 add, 2 madd_abi");
  size_t qb = op == 1 ? 128 : 64;
  char *da, *db, *dout;
  rc = dbg_run3(ctx, acc, n * 128, q, n * qb, n * 128, &da, &db, &dout); if (rc) return rc;
  dim3 g((u32)((n + 63) / 64)), blk(64);
  if (ctx->opt_field == 0) {
    if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_dbg_pointop29<GqLazy>), g, blk, 0, ctx->stream, op, da, db, dout, (u32)n);
    else hipLaunchKernelGGL((k_dbg_pointop29<GrLazy>), g, blk, 0, ctx->stream, op, da, db, dout, (u32)n);
  } else if (curve == LEMSM_BN254_G1) hipLaunchKernelGGL((k_dbg_pointop<FqDev>), g, blk, 0, ctx->stream, op == 2 ? 0 : op, da, db, dout, (u32)n);
  else hipLaunchKernelGGL((k_dbg_pointop<FrDev>), g, blk, 0, ctx->stream, op == 2 ? 0 : op, da, db, dout, (u32)n);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, n * 128, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return LEMSM_OK;
}

}  // extern "C"

#include "divisor_abi.inc"
