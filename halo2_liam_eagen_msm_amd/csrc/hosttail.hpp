// Host-side tail of an MSM call, free of any HIP dependency: the conversion of the raw device records, the per-window
// and final Horner recursions, and the builders of the plain-data plans the kernels read (pyramid task tables, merge
// queue layout).  lemsm.hip includes this file; tests/host_tail_check.cpp compiles it alone with
// -fsanitize=address,undefined and drives every function on the CPU (tests/test_host_logic.py), including a simulation of
// the pyramid task tables on integers.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "hostmath.hpp"
#include "plan.h"

namespace lemsm {

typedef uint32_t u32;
typedef uint64_t u64;

// bucket-reduction pyramid: per step a list of pairwise-add tasks (kernels_ec.cuh)
struct PyrPlan {
  std::vector<std::vector<PyrTask>> steps;   // steps[s-1] for s = 1..L
  std::vector<u32> step_max_count;
  CopyTask copy;                              // U_{L-1} = A^{L-1}[1]
};

// ---- pyramid task tables ----

// arena offsets (in points) for one group
struct ArenaLayout {
  u32 bucket_off, apyr_off, rbuf_off, out_off, total_points;
};

inline ArenaLayout make_arena(u32 NBpad, u32 nbp, u32 gw, u32 L, u32 nba = 1 /* bucket areas: one per slab when the slabs of a call share one tail */) {
  ArenaLayout a;
  a.bucket_off = 0;
  a.apyr_off = NBpad * nba;
  a.rbuf_off = a.apyr_off + nbp * gw;
  a.out_off = a.rbuf_off + nbp * gw;
  a.total_points = a.out_off + (L + 1) * gw;
  return a;
}

inline PyrPlan make_pyr_plan(const ArenaLayout& ar, u32 nb, u32 nbw, u32 nbp, u32 L, bool scaled /* bucket_sum[] holds (X, Y, 32 ZZ, 32 ZZZ) */) {
  PyrPlan pp;
  auto offA = [&](u32 l) { return nbp - (nbp >> (l - 1)); };                    // level l >= 1 inside a window's A region
  auto offR = [&](u32 l, u32 j) { u32 rs = nbp >> (l + 1); return (nbp - (nbp >> l)) + (rs - (rs >> (j - 1))); };
  auto srcA = [&](u32 l, PyrTask& t) {    // A^l as a source
    if (l == 0) { t.src_off = ar.bucket_off; t.src_wstride = nbw; t.src_valid = nb; t.src_scaled = scaled ? 1u : 0u; }
    else { t.src_off = ar.apyr_off + offA(l); t.src_wstride = nbp; t.src_valid = nbp >> l; }
  };
  for (u32 s = 1; s <= L; s++) {
    std::vector<PyrTask> tasks;
    PyrTask t; memset(&t, 0, sizeof t);
    srcA(s - 1, t);
    t.stride = 1; t.phase = 0; t.count = nbp >> s;
    if (s == L) { t.dst_off = ar.out_off; t.dst_wstride = L + 1; }
    else { t.dst_off = ar.apyr_off + offA(s); t.dst_wstride = nbp; }
    tasks.push_back(t);
    if (s + 1 <= L) {
      u32 cnt = nbp >> (s + 1);
      for (u32 l = 0; l < s; l++) {
        u32 j = s - l;
        PyrTask r; memset(&r, 0, sizeof r);
        if (j == 1) { srcA(l, r); r.stride = 2; r.phase = 1; }
        else { r.src_off = ar.rbuf_off + offR(l, j - 1); r.src_wstride = nbp; r.src_valid = nbp >> (l + j); r.stride = 1; r.phase = 0; }
        r.count = cnt;
        if (cnt == 1) { r.dst_off = ar.out_off + 1 + l; r.dst_wstride = L + 1; }
        else { r.dst_off = ar.rbuf_off + offR(l, j); r.dst_wstride = nbp; }
        tasks.push_back(r);
      }
    }
    if (L == 1 && scaled) {
      // one level only: U_0 = bucket[1] is read through a task (bucket[1] + nothing) instead of the
      // copy below, because only tasks convert the scaled form on load
      PyrTask r; memset(&r, 0, sizeof r);
      srcA(0, r); r.stride = 1; r.phase = 1; r.count = 1; r.src_valid = std::min(r.src_valid, 2u);
      r.dst_off = ar.out_off + 1; r.dst_wstride = L + 1;
      tasks.push_back(r);
    }
    pp.step_max_count.push_back(nbp >> s);
    pp.steps.push_back(tasks);
  }
  // U_{L-1} = A^{L-1}[1]
  memset(&pp.copy, 0, sizeof pp.copy);
  if (L == 1) { pp.copy.src_off = ar.bucket_off; pp.copy.src_wstride = nbw; pp.copy.src_valid_idx = nb; }
  else { pp.copy.src_off = ar.apyr_off + offA(L - 1); pp.copy.src_wstride = nbp; pp.copy.src_valid_idx = 2; }
  pp.copy.src_idx = 1;
  pp.copy.dst_off = ar.out_off + L; pp.copy.dst_wstride = L + 1;
  return pp;
}

// merge queues of one group (kernels_ec.cuh "edge-record merge").  A bucket of P >= 2 pieces spans P chunks and
// neighbouring buckets share at most one chunk, so at most nthr1 / (P - 1) buckets have P or more pieces.
inline MqLayout make_mq_layout_t(u32 nthr1, u32 slice /* pieces per wave of a long bucket */, u32 wave_th /* 9..32-piece buckets: one wave each up to this many, serial beyond */) {
  MqLayout m; memset(&m, 0, sizeof m);
  m.slice = slice; m.wave_th = wave_th;
  m.capS = nthr1 / 2 + 2; m.capM = nthr1 / 8 + 2;
  m.capF = nthr1 / m.slice + 2;                  // buckets of more than one slice
  m.capL = nthr1 / 32 + nthr1 / m.slice + 4;     // slices: sum ceil(P / slice) over buckets of > 32 pieces
  m.capP = 2 * (nthr1 / m.slice) + 4;            // partial sums (multi-slice buckets only)
  m.offS = 0; m.offM = m.offS + m.capS; m.offL = m.offM + m.capM; m.offF = m.offL + m.capL;
  return m;
}

// Device records -> host XYZZ points (x*2^256 Montgomery, canonical).
// strict arithmetic: records already are 4 x 32-byte canonical x*2^256 values.
// lazy arithmetic: records are 36 raw signed 29-bit limbs of x*2^261 representatives with
// |V| < 8N: add 8N, carry-normalise, reduce mod N, then multiply by 2^-5 (montmul by 2^251).
template <class P64>
host::fe reduce_raw29(const int32_t* l) {
  typedef host::HF<P64> F;
  // N as 29-bit limbs
  int64_t v[9];
  u64 n29[9];
  for (int i = 0; i < 9; i++) {
    int bit = 29 * i, wi = bit >> 6, sh = bit & 63;
    u64 lo = P64::N[wi] >> sh;
    u64 hi = (sh + 29 > 64 && wi + 1 < 4) ? (P64::N[wi + 1] << (64 - sh)) : 0;
    n29[i] = (lo | hi) & ((1ULL << 29) - 1);
  }
  int64_t c = 0;
  for (int i = 0; i < 9; i++) {
    int64_t t = (int64_t)l[i] + (int64_t)(n29[i] << 3) + c;   // + 8N limb-wise (limbs may exceed 29 bits; carried below)
    if (i < 8) { v[i] = t & ((1LL << 29) - 1); c = t >> 29; } else v[i] = t;
  }
  // pack the non-negative value (< 16N < 2^258) into 5 x u64
  u64 w[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 9; i++) {   // limbs are disjoint bit fields now: OR them in
    int bit = 29 * i, wi = bit >> 6, sh = bit & 63;
    w[wi] |= (u64)v[i] << sh;
    if (sh + 29 > 64 && wi + 1 < 5) w[wi + 1] |= (u64)v[i] >> (64 - sh);
  }
  // reduce below N by subtracting N while >= N (at most 16 times)
  for (int it = 0; it < 20; it++) {
    bool ge = w[4] != 0;
    if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (w[i] > P64::N[i]) break; if (w[i] < P64::N[i]) { ge = false; break; } } }
    if (!ge) break;
    u64 bw = 0;
    for (int i = 0; i < 5; i++) { unsigned __int128 d = (unsigned __int128)w[i] - (i < 4 ? P64::N[i] : 0) - bw; w[i] = (u64)d; bw = (u64)(d >> 64) & 1; }
  }
  host::fe r; for (int i = 0; i < 4; i++) r.l[i] = w[i];
  host::fe k251 = {{0, 0, 0, 0x0800000000000000ULL}};   // 2^251 (< N): montmul(a, 2^251) = a * 2^-5
  return F::mul(r, k251);
}

template <class P64, bool LAZY /* 160-byte records of raw 29-bit limbs (XYZZ29) instead of 4 x 32 canonical bytes */, size_t PT_BYTES>
void from_device_records_t(const char* raw, size_t n, host::pt* out) {
  if constexpr (!LAZY) {
    memcpy(out, raw, n * sizeof(host::pt));
  } else {
    for (size_t i = 0; i < n; i++) {
      const int32_t* l = reinterpret_cast<const int32_t*>(raw + i * PT_BYTES);
      bool zz_zero = true;
      for (int k = 0; k < 9; k++) zz_zero &= (l[18 + k] == 0);
      if (zz_zero) { memset(&out[i], 0, sizeof(host::pt)); continue; }
      out[i].x = reduce_raw29<P64>(l); out[i].y = reduce_raw29<P64>(l + 9);
      out[i].zz = reduce_raw29<P64>(l + 18); out[i].zzz = reduce_raw29<P64>(l + 27);
    }
  }
}

// S_w = total + sum_l 2^l U_l  for one window record [total, U_0..U_{L-1}]
template <class P64>
host::pt window_sum(const host::pt* rec, u32 L) {
  typedef host::HG<P64> G;
  host::pt acc = G::identity();
  for (int l = (int)L - 1; l >= 0; l--) { acc = G::dbl(acc); acc = G::add(acc, rec[1 + l]); }
  return G::add(acc, rec[0]);
}

// Horner over the window sums: sum_w 2^(c w) S_w
template <class P64>
void msm_combine_windows(u32 c, u32 W, const host::pt* sums /* W window sums */, u64 out[12]) {
  typedef host::HG<P64> G;
  host::pt acc = G::identity();
  for (int w = (int)W - 1; w >= 0; w--) {
    for (u32 k = 0; k < c; k++) acc = G::dbl(acc);
    acc = G::add(acc, sums[w]);
  }
  G::to_jacobian(acc, out);
}

template <class P64>
void jacobian_sum_t(const uint64_t* jac, size_t count, uint64_t out[12]) {
  typedef host::HG<P64> G;
  host::pt acc = G::identity();
  for (size_t i = 0; i < count; i++) acc = G::add(acc, G::from_jacobian(jac + 12 * i));
  G::to_jacobian(acc, out);
}

// carries: MSB-first Horner with multiplier -base over per-position sums (src/argument_witness_calc.rs:105-127)
template <class P64>
void lhs_combine_positions(u32 base, u32 d, const host::pt* sums /* d per-position sums S_i, LSB-first position order */, u64 out_carry[12],
                           u64* out_carries) {
  typedef host::HG<P64> G;
  host::pt carry = G::identity();
  for (u32 it = 0; it < d; it++) {
    u32 pos = d - 1 - it;
    carry = G::mul_small(G::neg(carry), base);                                     // :118
    carry = G::add(carry, sums[pos]);                                              // :120-125
    if (out_carries) G::to_jacobian(carry, out_carries + 12 * (size_t)it);
  }
  G::to_jacobian(carry, out_carry);
}

// Multi-GPU status agreement: every rank contributes one status word to the exchange (0 = its pipeline ran; otherwise the
// LEMSM_ERR_* it failed with before the exchange) and every rank evaluates the same gathered words, so that all return
// together: a rank that failed returns its own status, the others `peer_failed` (LEMSM_ERR_RCCL); *failed_rank = the
// lowest failing rank.  Returns 0 when every word is 0.
inline int merge_rank_status(const u32* status, int world, int my_rank, int my_rc, int* failed_rank, int peer_failed = 9 /* LEMSM_ERR_RCCL */) {
  int first = -1;
  for (int r = 0; r < world; r++) if (status[r] != 0 && first < 0) first = r;
  if (failed_rank) *failed_rank = first;
  if (first < 0) return my_rc;                    // (a failure this rank could not even report would have aborted the communicator)
  if (my_rc) return my_rc;
  (void)my_rank;
  return peer_failed;
}

// window / digit-position sharding: rank r of `world` owns [W r / world, W (r+1) / world)
inline void shard_range(u32 W, int world, int r, u32& a, u32& b) {
  a = (u32)((u64)W * (u64)r / (u64)world); b = (u32)((u64)W * (u64)(r + 1) / (u64)world);
}

}  // namespace lemsm
