// Heavy templated kernels (group-law arithmetic): level-1 segmented accumulation, edge-record
// reduction, bucket-reduction pyramid, point-domain conversion.  Templates only, so that the
// explicit instantiations can live in their own translation units (inst_*.hip).
#pragma once
#include "plan.h"
#include "xyzz.cuh"
#include "xyzz29.cuh"

namespace lemsm {

static const u32 KEY_NONE = 0xffffffffu;

// meta words written by k_binscan
enum { META_M = 0, META_TILES = 1, META_WORDS = 4, META_CLOCK = 8 /* 4 x u64 written by k_accum1: (shader cycles, 100 MHz ticks) at the start and at the end of one wave's chunk */ };

// ------------------------------------------------------------------------------------
// level 1: segmented accumulation of the bucket-sorted entry list.
// Thread t owns entries [t*L1, (t+1)*L1).  A segment (maximal run of one bucket inside the
// chunk) that covers its whole bucket is stored to bucket_sum[key]; otherwise it is one PIECE of its
// bucket (at most two per thread: the segment that reaches the chunk's end -> slot 2t+1, a segment that
// begins at the chunk's start and ends earlier -> slot 2t) for the merge kernels below; rec_key[t] = the
// bucket this chunk owns (the one that starts here and runs past the chunk's end), or KEY_NONE.
// ------------------------------------------------------------------------------------
template <class G, int WPS /* waves per SIMD the register budget is sized for */,
          bool ABI = false /* points in the C ABI's domain, accumulator and outputs in the scaled form: G::madd_abi */,
          bool RING = false /* entries reach the lanes through a per-wave LDS ring filled by global_load_lds (needs L1 % 16 == 0) */>
__global__ __launch_bounds__(256, WPS) void k_accum1(GroupPlan pl, const u32* __restrict__ sorted,
                                                const u32* __restrict__ bucket_start, u32* __restrict__ meta,
                                                const uint4* __restrict__ points, char* __restrict__ bucket_sum,
                                                u32* __restrict__ rec_key, char* __restrict__ rec_pt) {
  typedef typename G::F_ F;
  typedef typename F::fe fe;
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= pl.nthr1) return;
  const u32 M = meta[META_M];
  // Sustained shader clock of THIS launch (bench.py's roofline.valu_issue.clock_ghz_measured): one wave in the
  // middle of the grid stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its whole chunk;
  // clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, in-kernel clock).  Two scalar reads per launch.
  // (The opening stamps go straight to memory: held in registers across the loop they cost 4 more spilled VGPRs.)
  const bool stamp = blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0;
  if (stamp) {
    u64* ck = reinterpret_cast<u64*>(meta + META_CLOCK);
    ck[0] = __builtin_amdgcn_s_memtime(); ck[1] = __builtin_amdgcn_s_memrealtime();
  }
  const u64 s64 = (u64)t * pl.L1;
  const u32 r0 = 2 * t;
  if (s64 >= M) { rec_key[t] = KEY_NONE; return; }
  const u32 start = (u32)s64;
  const u32 end = (u32)min((u64)M, s64 + pl.L1);

  u32 lo = 0, hi = pl.nbins << pl.LB;   // largest key with bucket_start[key] <= start
  while (hi - lo > 1) {
    u32 mid = (lo + hi) >> 1;
    if (bucket_start[mid] <= start) lo = mid; else hi = mid;
  }
  const u32 nkeys = pl.nbins << pl.LB;   // bucket_start[] has nkeys + 1 entries
  u32 key = lo;
  u32 kbeg = bucket_start[key];
  u32 kend = bucket_start[key + 1];
  u32 kend2 = bucket_start[min(key + 2u, nkeys)], kend3 = bucket_start[min(key + 3u, nkeys)];
  u32 seg_begin = start;

  typename G::pt acc; G::set_identity(acc);
  bool empty = true;          // acc == identity (tracked so that the loop tests a flag, not nine limbs)

  auto flush = [&](u32 seg_end) {
    bool complete = (seg_begin == kbeg) && (seg_end == kend);
    if (complete) {
      G::store(bucket_sum + (size_t)key * G::PT_BYTES, acc);
    } else {
      G::store(rec_pt + (size_t)(r0 + (seg_end == end ? 1u : 0u)) * G::PT_BYTES, acc);
    }
  };

  // Entry stream.  Plain form: every lane walks its own 1-KB chunk with 4-byte loads; 768 lanes per CU keep
  // 98 KB of lines live against a 32 KB L1, so a 128-byte line is re-fetched ~8 times from HBM before its 32
  // entries are used (8 GB per 2^24-point launch).  RING form: a lane's next 16 entries (64 B, half a line) go
  // straight from memory into a per-wave LDS ring (global_load_lds_dwordx4, no VGPRs), two blocks deep; the
  // lane then takes one ds_read_b32 per entry.  A wave only ever reads what it staged itself, so its own
  // s_waitcnt vmcnt(0) orders the DMA before the reads; a block is overwritten 16 iterations after its last read.
  __shared__ u32 ering[RING ? 4 : 1][2][4][64][RING ? 4 : 1];
  const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  auto ring_fill = [&](u32 blk) {       // stage entries [start + 16 blk, start + 16 blk + 16) of every lane of this wave
#pragma unroll
    for (int k = 0; k < 4; k++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sorted + start + 16u * blk + 4u * k),
                                       (__attribute__((address_space(3))) void*)&ering[wv][blk & 1u][k][0][0], 16, 0, 0);
  };
  auto ring_read = [&](u32 rel) -> u32 { return ering[wv][(rel >> 4) & 1u][(rel >> 2) & 3u][lane][rel & 3u]; };
  u32 e_next;
  if constexpr (RING) {
    ring_fill(0);
    if (pl.L1 > 16) ring_fill(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    e_next = ring_read(0);
  } else {
    e_next = sorted[start];
  }
  // The next point travels as its four raw 16-byte words: they are requested a whole mixed addition (~4 us) before
  // they are unpacked, so the gather's latency hides behind the arithmetic.  (Unpacking right behind the loads --
  // what F::load does -- put an s_waitcnt vmcnt(0) directly after them: PMC showed 12 % of all wave cycles in
  // SQ_WAIT_ANY, profiles/r02/pmc_accum1_issue_wait.txt.)
  uint4 nx0, nx1, ny0, ny1;
  { const uint4* p = points + (size_t)(e_next & 0xffffffu) * 4; nx0 = p[0]; nx1 = p[1]; ny0 = p[2]; ny1 = p[3]; }

  // One loop over the chunk with every lane in lockstep (a per-segment inner loop would let the
  // lanes of a wave drift apart: measured 1.5x slower).  The flush is a rare divergent branch.
  for (u32 i = start; i < end; i++) {
    u32 e = e_next; fe px, py;
    F::from_words(px, nx0, nx1); F::from_words(py, ny0, ny1);
    // Bucket boundary first, the next point's loads after it, and NO load inside the boundary branch (some lane
    // takes it in a third of all iterations; a load there ends in an s_waitcnt vmcnt(0) that also waits for the
    // flush's ten stores): the ends of the next two buckets ride along in kend2 / kend3, kend3 re-requested every
    // iteration beside the point gather (one cached 4-byte load) and not looked at before the next iteration.
    if (i >= kend) {
      flush(i);
      key++; kend = kend2; kend2 = kend3;
      while (i >= kend) { key++; kend = kend2; kend2 = bucket_start[min(key + 2u, nkeys)]; }   // empty buckets in between (rare)
      kbeg = i; seg_begin = i;
      G::set_identity(acc); empty = true;
    }
    if (i + 1 < end) {
      if constexpr (RING) {
        const u32 rel1 = i + 1 - start;               // the same in every lane of the wave
        if ((rel1 & 15u) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // block rel1/16, requested 16 iterations ago
        e_next = ring_read(rel1);
        if ((rel1 & 15u) == 0 && rel1 + 16u < pl.L1) ring_fill((rel1 >> 4) + 1u);  // reuse the buffer of the block just finished
      } else {
        e_next = sorted[i + 1];
      }
      const uint4* p = points + (size_t)(e_next & 0xffffffu) * 4;
      nx0 = p[0]; nx1 = p[1]; ny0 = p[2]; ny1 = p[3];
    }
    kend3 = bucket_start[min(key + 3u, nkeys)];
    if (!G::aff_is_identity(px, py)) {
      F::cneg(py, py, (e >> 31) != 0);
      if (ABI) G::madd_abi(acc, px, py, empty); else G::madd(acc, px, py, empty);
    }
  }
  flush(end);
  if (stamp) {
    u64* ck = reinterpret_cast<u64*>(meta + META_CLOCK);
    ck[2] = __builtin_amdgcn_s_memtime(); ck[3] = __builtin_amdgcn_s_memrealtime();
  }
  // this chunk owns the bucket of its last segment if that bucket starts here and continues in the next chunk
  rec_key[t] = (seg_begin == kbeg && end != kend) ? key : KEY_NONE;
}

// ------------------------------------------------------------------------------------
// Edge-record merge.  k_accum1 cuts the bucket-sorted entry list into equal chunks regardless of bucket
// boundaries, so a bucket that straddles chunk ends arrives in PIECES: one partial sum per chunk it touches.
// Where the pieces lie follows from bucket_start[] and the chunk length alone -- chunk t keeps the sum of the
// segment that reaches its END in slot 2t+1 ("last" record) and the sum of a segment that begins at its START
// and ends earlier in slot 2t ("first" record) -- so no keys, fillers or compaction passes are needed: the
// pieces of a bucket that starts in chunk ta and ends in chunk tb are the last records of ta..tb-1 plus
// tb's first record (its last record if the bucket ends exactly where the chunk does).  The chunk a bucket
// STARTS in owns it (rec_key[ta] = key).
//   k_merge_pairs   one thread per chunk: a two-piece bucket (the common case: uniform digits put about one
//                   bucket boundary into every chunk) is added and stored at once; longer ones are queued by size.
//   k_merge_serial  3..8 pieces: one thread per queued bucket, serial (and 9..32 pieces when that queue is long).
//   k_merge_waves   9..32 pieces: one wave per bucket while the queue is short (latency).  More than 32: slices of
//                   <= `slice` pieces, one wave each (lanes stride over the pieces, then a DPP scan); a bucket of
//                   several slices leaves one partial sum per slice.
//   k_merge_final   one wave per multi-slice bucket adds its partials.
// Depth: 1 addition for uniform scalars (was 8 + 4 x 6 through the segmented-scan levels), <= slice/64 + 6 + 6
// for any input; the sum of a bucket is formed in piece order whatever the queue order, so results are
// deterministic.
// ------------------------------------------------------------------------------------
enum { MQ_S = 0, MQ_M = 1, MQ_L = 2, MQ_F = 3, MQ_P = 4, MQ_WORDS = 16 /* [8..15]: debug cycle stamps of the wave kernels */ };   // queue counters: short, medium, long slices, multi-slice buckets, partials
// (struct MqLayout: plan.h)
static const u32 MQ_DST_BUCKET = 0xffffffffu;

template <class G>
__device__ __forceinline__ void merge_load_piece(typename G::pt& q, const char* __restrict__ rec_pt, u32 t, bool first_slot, u32 scaled) {
  G::load(q, rec_pt + (size_t)(2u * t + (first_slot ? 0u : 1u)) * G::PT_BYTES);
  if (scaled) G::unscale(q);                     // k_accum1's ABI form keeps (X, Y, 32 ZZ, 32 ZZZ)
}

template <class G>
__global__ __launch_bounds__(256) void k_merge_pairs(GroupPlan pl, u32 scaled, MqLayout lay, const u32* __restrict__ bucket_start,
                                                     const u32* __restrict__ meta, const u32* __restrict__ rec_key,
                                                     const char* __restrict__ rec_pt, char* __restrict__ bucket_sum,
                                                     u32* __restrict__ mq_cnt, uint4* __restrict__ mq_items) {
  __shared__ u32 lc[5], lbase[5];
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 5) lc[threadIdx.x] = 0;
  __syncthreads();
  const u32 key = t < pl.nthr1 ? rec_key[t] : KEY_NONE;
  u32 P = 0, lastfirst = 0, cls = 0, loff = 0, nsub = 0, poff = 0, foff = 0;
  if (key != KEY_NONE) {
    const u32 e = bucket_start[key + 1], M = meta[META_M];
    const u32 tb = (e - 1u) / pl.L1;
    P = tb - t + 1u;                              // >= 2: the owner's segment ran into its chunk end
    lastfirst = ((u64)(tb + 1u) * pl.L1 == (u64)e || e == M) ? 0u : 1u;
    cls = P == 2 ? 1u : (P <= 8 ? 2u : (P <= 32 ? 3u : 4u));
    if (cls == 2) loff = atomicAdd(&lc[MQ_S], 1u);
    else if (cls == 3) loff = atomicAdd(&lc[MQ_M], 1u);
    else if (cls == 4) {
      nsub = (P + lay.slice - 1u) / lay.slice;
      loff = atomicAdd(&lc[MQ_L], nsub);
      if (nsub > 1) { poff = atomicAdd(&lc[MQ_P], nsub); foff = atomicAdd(&lc[MQ_F], 1u); }
    }
  }
  __syncthreads();
  if (threadIdx.x < 5) lbase[threadIdx.x] = lc[threadIdx.x] ? atomicAdd(&mq_cnt[threadIdx.x], lc[threadIdx.x]) : 0u;
  __syncthreads();
  if (cls == 2) { u32 i = lbase[MQ_S] + loff; if (i < lay.capS) mq_items[lay.offS + i] = make_uint4(key, t, P | (lastfirst << 31), MQ_DST_BUCKET); }
  else if (cls == 3) { u32 i = lbase[MQ_M] + loff; if (i < lay.capM) mq_items[lay.offM + i] = make_uint4(key, t, P | (lastfirst << 31), MQ_DST_BUCKET); }
  else if (cls == 4) {
    const u32 i0 = lbase[MQ_L] + loff, p0 = lbase[MQ_P] + poff;
    for (u32 k = 0; k < nsub; k++) {
      const u32 cnt = min(lay.slice, P - k * lay.slice);
      const u32 lf = (k + 1 == nsub) ? lastfirst : 0u;
      if (i0 + k < lay.capL) mq_items[lay.offL + i0 + k] = make_uint4(key, t + k * lay.slice, cnt | (lf << 31), nsub > 1 ? p0 + k : MQ_DST_BUCKET);
    }
    if (nsub > 1) { u32 f = lbase[MQ_F] + foff; if (f < lay.capF) mq_items[lay.offF + f] = make_uint4(key, p0, nsub, 0); }
  }
  if (cls == 1) {
    typename G::pt acc, q;
    merge_load_piece<G>(acc, rec_pt, t, false, scaled);
    merge_load_piece<G>(q, rec_pt, t + 1u, lastfirst != 0, scaled);
    G::add(acc, q);
    if (scaled) G::scale(acc);                    // bucket_sum[] holds the scaled form throughout (xyzz29.cuh)
    G::store(bucket_sum + (size_t)key * G::PT_BYTES, acc);
  }
}

// One wave adds the pieces [t0, t0 + n) of a bucket (FROM_PARTIALS: the partial sums [t0, t0 + n) of its slices): the lanes
// stride over them, then a DPP scan (G::scan_fetch) leaves the sum in lane `result_lane`.  For n < 64 the pieces are
// replicated with period m = the power of two >= n, so that every lane carries real work (a sparsely populated wave
// runs its additions 1.7-3x slower) and the scan stops at the steps that stay inside one replica.  Serial phase and scan
// share ONE call site of G::add (a second inlined copy of the ~25 KB addition buys nothing and costs instruction cache).
template <class G, bool FROM_PARTIALS>
__device__ __forceinline__ u32 merge_wave_sum(typename G::pt& acc, u32 lane, u32 n, u32 t0, u32 lf, u32 scaled,
                                              const char* __restrict__ src, u32 cap) {
  G::set_identity(acc);
  u32 m = 1; while (m < n && m < 64u) m <<= 1;
  const u32 nser = (n + 63u) >> 6;
  const u32 nscan = m <= 16u ? (u32)__builtin_ctz(m) : (m == 32u ? 5u : 6u);   // row steps with a shift below m, then the row broadcasts
#pragma unroll 1
  for (u32 s = 0; s < nser + nscan; s++) {
    typename G::pt q;
    if (s < nser) {
      const u32 j = (lane & (m - 1u)) + 64u * s;
      if (j < n) {
        if constexpr (FROM_PARTIALS) { if (t0 + j < cap) G::load(q, src + (size_t)(t0 + j) * G::PT_BYTES); else G::set_identity(q); }
        else merge_load_piece<G>(q, src, t0 + j, j + 1 == n && lf, scaled);
      } else G::set_identity(q);
    } else {
      G::scan_fetch(q, acc, (int)(s - nser));
    }
    G::add(acc, q);
  }
  return m - 1u;
}

// One block per CU: the wave kernels below are latency chains (a slice is <= slice/64 + 6 dependent additions), and the
// dispatcher is free to stack several 256-thread blocks on one CU while others idle -- measured: 968 slices on 242
// blocks ran in 198 us where one wave's chain takes 98.  A static LDS footprint above half the CU's 160 KB pins one
// block per CU, i.e. one wave per SIMD; more items than resident waves are taken in turns by the same waves.
static const u32 MERGE_LDS_PAD_WORDS = 21504;    // 84 KB

// 3..8 pieces (and 9..32 when that queue is long): one lane per queued bucket, 64 buckets per wave-sized chunk; the two
// classes start on chunk boundaries so that a wave's trip counts are alike.  One block per CU; chunk c goes to block
// c mod gridDim.x, so a short queue spreads over the CUs one wave each instead of filling a few CUs' wave slots.
template <class G>
__global__ __launch_bounds__(256) void k_merge_serial(u32 scaled, MqLayout lay, const u32* __restrict__ mq_cnt,
                                                      const uint4* __restrict__ mq_items, const char* __restrict__ rec_pt,
                                                      char* __restrict__ bucket_sum) {
  __shared__ u32 pad_[MERGE_LDS_PAD_WORDS];
  if (scaled > 1) pad_[threadIdx.x] = scaled;     // never true: keeps the allocation (one block per CU, above)
  const u32 cS = min(mq_cnt[MQ_S], lay.capS), cMall = min(mq_cnt[MQ_M], lay.capM);
  const u32 cM = cMall > lay.wave_th ? cMall : 0u;
  const u32 chS = (cS + 63u) >> 6, chM = (cM + 63u) >> 6;
  const u32 lane = threadIdx.x & 63u;
#pragma unroll 1
  for (u32 c = blockIdx.x + gridDim.x * (threadIdx.x >> 6); c < chS + chM; c += gridDim.x * 4u) {
    uint4 it;
    if (c < chS) { const u32 v = 64u * c + lane; if (v >= cS) continue; it = mq_items[lay.offS + v]; }
    else { const u32 v = 64u * (c - chS) + lane; if (v >= cM) continue; it = mq_items[lay.offM + v]; }
    const u32 key = it.x, t0 = it.y, n = it.z & 0x7fffffffu, lf = it.z >> 31;
    typename G::pt acc, q; G::set_identity(acc);
#pragma unroll 1
    for (u32 j = 0; j < n; j++) {
      merge_load_piece<G>(q, rec_pt, t0 + j, j + 1 == n && lf, scaled);
      G::add(acc, q);
    }
    if (scaled) G::scale(acc);
    G::store(bucket_sum + (size_t)key * G::PT_BYTES, acc);
  }
}

// 9..32 pieces (while that queue is short) and the slices of longer buckets: one wave per item
template <class G>
__global__ __launch_bounds__(256) void k_merge_waves(u32 scaled, MqLayout lay, const u32* __restrict__ mq_cnt,
                                                     const uint4* __restrict__ mq_items, const char* __restrict__ rec_pt,
                                                     char* __restrict__ partial, char* __restrict__ bucket_sum, u32* __restrict__ dbg) {
  __shared__ u32 pad_[MERGE_LDS_PAD_WORDS];
  if (scaled > 1) pad_[threadIdx.x] = scaled;     // never true: keeps the allocation
  const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
  const u32 cMall = min(mq_cnt[MQ_M], lay.capM), cL = min(mq_cnt[MQ_L], lay.capL);
  const u32 lane = threadIdx.x & 63u;
  const u32 nwaves = gridDim.x * 4u;
  const u32 cM = cMall > lay.wave_th ? 0u : cMall;
#pragma unroll 1
  for (u32 w = blockIdx.x * 4u + (threadIdx.x >> 6); w < cM + cL; w += nwaves) {
    const uint4 it = w < cM ? mq_items[lay.offM + w] : mq_items[lay.offL + (w - cM)];
    const u32 key = it.x, dst = it.w;
    typename G::pt acc;
    const unsigned long long tk1 = __builtin_amdgcn_s_memtime();
    const u32 rl = merge_wave_sum<G, false>(acc, lane, it.z & 0x7fffffffu, it.y, it.z >> 31, scaled, rec_pt, 0);
    const unsigned long long tk2 = __builtin_amdgcn_s_memtime();
    if (lane == rl) {
      if (dst == MQ_DST_BUCKET) { if (scaled) G::scale(acc); G::store(bucket_sum + (size_t)key * G::PT_BYTES, acc); }
      else if (dst < lay.capP) G::store(partial + (size_t)dst * G::PT_BYTES, acc);      // plain form
    }
    if (dbg && w == 0 && lane == 0) { dbg[8] = (u32)(tk1 - tk0); dbg[9] = (u32)(tk2 - tk1); dbg[10] = (u32)(__builtin_amdgcn_s_memtime() - tk2); dbg[11] = it.z; }
  }
}

template <class G>
__global__ __launch_bounds__(256) void k_merge_final(u32 scaled, MqLayout lay, const u32* __restrict__ mq_cnt, const uint4* __restrict__ mq_items,
                                                     const char* __restrict__ partial, char* __restrict__ bucket_sum, u32* __restrict__ dbg) {
  __shared__ u32 pad_[MERGE_LDS_PAD_WORDS];
  if (scaled > 1) pad_[threadIdx.x] = scaled;     // never true: one block per CU (see k_merge_waves)
  const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
  const u32 cF = min(mq_cnt[MQ_F], lay.capF);
  const u32 lane = threadIdx.x & 63u, nwaves = gridDim.x * 4u;
#pragma unroll 1
  for (u32 w = blockIdx.x * 4u + (threadIdx.x >> 6); w < cF; w += nwaves) {
    const uint4 it = mq_items[lay.offF + w];
    typename G::pt acc;
    const unsigned long long tk1 = __builtin_amdgcn_s_memtime();
    const u32 rl = merge_wave_sum<G, true>(acc, lane, it.z, it.y, 0, 0, partial, lay.capP);
    const unsigned long long tk2 = __builtin_amdgcn_s_memtime();
    if (lane == rl) { if (scaled) G::scale(acc); G::store(bucket_sum + (size_t)it.x * G::PT_BYTES, acc); }
    if (dbg && w == 0 && lane == 0) { dbg[12] = (u32)(tk1 - tk0); dbg[13] = (u32)(tk2 - tk1); dbg[14] = (u32)(__builtin_amdgcn_s_memtime() - tk2); dbg[15] = it.z; }
  }
}

// ------------------------------------------------------------------------------------
// bucket reduction: pairwise-add pyramid.  One launch per step; a step is a list of tasks
//   dst[i] = src[(2i)*stride + phase] + src[(2i+1)*stride + phase],   i < count
// applied to every window.  Arena offsets are in points (128 B); src indices >= src_valid
// read as the identity (padding of non-power-of-two bucket counts).
// ------------------------------------------------------------------------------------
// (layout of CopyTask in kernels.cuh)
typedef CopyTask CopyTaskPod;   // (plan.h)

// (struct PyrTask: plan.h)

// one item of one task: dst[i] = src[(2i) stride + phase] + src[(2i+1) stride + phase] for window w
template <class G>
__device__ __forceinline__ void pyr_item(const PyrTask& tk, u32 w, u32 i, char* __restrict__ arena) {
  u32 ia = (2 * i) * tk.stride + tk.phase, ib = (2 * i + 1) * tk.stride + tk.phase;
  const char* src = arena + ((size_t)tk.src_off + (size_t)w * tk.src_wstride) * G::PT_BYTES;
  typename G::pt a, b;
  if (ia < tk.src_valid) G::load(a, src + (size_t)ia * G::PT_BYTES); else G::set_identity(a);
  if (ib < tk.src_valid) G::load(b, src + (size_t)ib * G::PT_BYTES); else G::set_identity(b);
  if (tk.src_scaled) { G::unscale(a); G::unscale(b); }
  G::add(a, b);
  G::store(arena + ((size_t)tk.dst_off + (size_t)w * tk.dst_wstride + i) * G::PT_BYTES, a);
}

// the same with four lanes per item where the arithmetic has that form (XYZZ29::add4_mem): true if the quad has stored the sum
template <class G>
__device__ __forceinline__ bool pyr_try4(const PyrTask& tk, u32 w, u32 i, char* __restrict__ arena, u32 q) {
  if constexpr (G::CONVERTED_DOMAIN) {
    const u32 ia = (2 * i) * tk.stride + tk.phase, ib = (2 * i + 1) * tk.stride + tk.phase;
    if (ia >= tk.src_valid || ib >= tk.src_valid || tk.src_scaled) return false;
    const char* src = arena + ((size_t)tk.src_off + (size_t)w * tk.src_wstride) * G::PT_BYTES;
    return G::add4_mem(src + (size_t)ia * G::PT_BYTES, src + (size_t)ib * G::PT_BYTES,
                       arena + ((size_t)tk.dst_off + (size_t)w * tk.dst_wstride + i) * G::PT_BYTES, q);
  } else return false;
}

template <class G>
__global__ __launch_bounds__(256) void k_pyramid(const PyrTask* __restrict__ tasks, u32 ntasks, u32 nwin,
                                                 u32 max_count, char* __restrict__ arena, u32 quad /* four lanes per item (a narrow step) */) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  const u32 gid = quad ? t >> 2 : t;
  u32 per_task = max_count * nwin;
  u32 ti = gid / per_task;
  if (ti >= ntasks) return;
  u32 rem = gid - ti * per_task;
  u32 w = rem / max_count, i = rem - w * max_count;
  PyrTask tk = tasks[ti];
  if (i >= tk.count) return;
  bool serial = true;
  if (quad) serial = !pyr_try4<G>(tk, w, i, arena, t & 3u) && (t & 3u) == 0u;
  if (serial) pyr_item<G>(tk, w, i, arena);
}

// One tail per call: with several slabs of points every slab sorts and accumulates into a bucket area of its own (the
// sorted entries, records and queues of the workspace are reused slab after slab), and before the ONE pyramid of the call
// this kernel adds the areas bucket by bucket into the first.  `scaled_mask` bit k: slab k ran the ABI form of k_accum1
// (its sums are (X, Y, 32 ZZ, 32 ZZZ)); the result is plain, empty buckets (all-zero slots) stay the identity.
template <class G>
__global__ __launch_bounds__(256) void k_sum_slabs(char* __restrict__ bucket_sum, u32 nbuckets, u32 nslabs, u32 scaled_mask) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nbuckets) return;
  typename G::pt acc, q;
  G::load(acc, bucket_sum + (size_t)i * G::PT_BYTES);
  if (scaled_mask & 1u) G::unscale(acc);
#pragma unroll 1
  for (u32 k = 1; k < nslabs; k++) {
    G::load(q, bucket_sum + ((size_t)k * nbuckets + i) * G::PT_BYTES);
    if ((scaled_mask >> k) & 1u) G::unscale(q);
    G::add(acc, q);
  }
  G::store(bucket_sum + (size_t)i * G::PT_BYTES, acc);
}


// Steps 1 and 2 of the pyramid in one pass over the bucket sums.  A step-per-launch pyramid moves every point of a level
// through HBM twice (written by one launch, read by the next) and its first steps are as much bandwidth- as
// arithmetic-bound (step 1 of a 2^24-point MSM: 354 MB in 111 us); the first two steps are 3/4 of all the additions.
// Here one thread owns 8 consecutive buckets b0..b7 of a window and leaves exactly what steps 1 + 2 of the task tables
// leave -- A^2[2i], A^2[2i+1], stage 2 of U_0's partial sums and stage 1 of U_1's -- without ever storing A^1:
//   a10 = b0+b1  a11 = b2+b3  a12 = b4+b5  a13 = b6+b7      A^2[2i] = a10+a11   A^2[2i+1] = a12+a13
//   U_0: (b1+b3) + (b5+b7)                                    U_1: a11 + a13
// 10 additions for 8 + 3 loads and 4 + 2 stores (the two launches: 10 additions, 14 loads, 7 stores).  The ten additions
// run through ONE call site of G::add: the body is a ten-step program over three named point registers, operands
// fetched and results filed by a switch on the step (an inlined copy per addition would be ~300 KB of code).  Empty buckets are
// recognised from bucket_start[], so bucket_sum[] needs no zero fill.
struct PyrFirst2Args {
  u32 nwin, nb, nbw, nbp;            // windows of the group, buckets per window, key stride of a window, padded bucket count
  u32 bucket_off, a2_off, r02_off, r11_off, wstride;   // arena offsets in points; A^2 / R regions have window stride nbp
  u32 scaled;                        // bucket sums are in k_accum1's (X, Y, 32 ZZ, 32 ZZZ) form
};

template <class G>
__global__ __launch_bounds__(256, 1) void k_pyramid_first2(PyrFirst2Args a, const u32* __restrict__ bucket_start, char* __restrict__ arena, u32* __restrict__ dbg) {
  const unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
  const u32 gid = blockIdx.x * 256 + threadIdx.x;
  const u32 per_win = a.nbp >> 3;
  const u32 w = gid / per_win, i = gid - w * per_win;
  if (w >= a.nwin) return;
  const u32 key0 = w * a.nbw + 8u * i;                     // first of this thread's 8 keys
  // Three named point registers only: with the addition's own ~135 VGPRs a fourth would push the kernel past the 256
  // architectural registers and the allocator then shuffles points through AGPRs around every addition (measured: +40 %
  // instructions).  The two long-lived values -- r = b1+b3 and a11 -- wait in their own output slots (U_0's and U_1's)
  // and are read back when their partners exist: two 160-byte round trips through L2 per thread.
  //   step      0       1        2        3        4       5        6        7       8        9
  //   x      ld b0   ld b2       B        A     ld b4      D    ld r02    ld b6      A     ld r11
  //   y      ld b1   ld b3       D     ld r11   ld b5   ld b7      D      ld b7      B        B
  //   keep   B = b1  D = b3                     D = b5
  //   result A = a10 st r11   st r02   st A2    A = a12  D = r'  st r02   B = a13  st A2'  st r11
  typename G::pt A, B, D, x, y;
  G::set_identity(A); G::set_identity(B); G::set_identity(D);
  const size_t o_a2 = (size_t)a.a2_off + (size_t)w * a.wstride + 2u * i;
  const size_t o_r02 = (size_t)a.r02_off + (size_t)w * a.wstride + i, o_r11 = (size_t)a.r11_off + (size_t)w * a.wstride + i;
  // Memory operands -- a bucket (index < 8; empty / out-of-window ones read as the identity, the ABI form is unscaled) or a
  // point of the arena (this thread's own earlier store) -- are requested one step ahead as raw 16-byte words, so that
  // their latency (~4 us a round trip on cold data, with one wave per SIMD nothing else hides it: measured 38 % of the
  // kernel) passes behind the previous addition.  Bucket loads are unconditional (the address is inside the arena whatever
  // the key) and masked by `have` when consumed: the first operands leave together with the bucket_start[] reads.
  auto operands = [&](u32 step, u32& kx, u32& ky, size_t& tx, size_t& ty) {
    kx = 8; ky = 8; tx = ~(size_t)0; ty = ~(size_t)0;
    switch (step) {
      case 0: kx = 0; ky = 1; break;
      case 1: kx = 2; ky = 3; break;
      case 3: ty = o_r11; break;
      case 4: kx = 4; ky = 5; break;
      case 5: ky = 7; break;
      case 6: tx = o_r02; break;
      case 7: kx = 6; ky = 7; break;
      case 9: tx = o_r11; break;
      default: break;
    }
    if (kx < 8) tx = (size_t)a.bucket_off + key0 + kx;
    if (ky < 8) ty = (size_t)a.bucket_off + key0 + ky;
  };
  uint4 rx[G::RAW_WORDS], ry[G::RAW_WORDS];
  {
    u32 kx, ky; size_t tx, ty; operands(0, kx, ky, tx, ty);
    G::load_raw(rx, arena + tx * G::PT_BYTES); G::load_raw(ry, arena + ty * G::PT_BYTES);
  }
  u32 have = 0;
#pragma unroll
  for (u32 k = 0; k < 8; k++)
    if (8u * i + k < a.nb && bucket_start[key0 + k + 1] != bucket_start[key0 + k]) have |= 1u << k;
#pragma unroll 1
  for (u32 step = 0; step < 10; step++) {
    {
      u32 kx, ky; size_t tx, ty; operands(step, kx, ky, tx, ty);
      if (tx != ~(size_t)0) { G::from_raw(x, rx); if (kx < 8) { if (!((have >> kx) & 1u)) G::set_identity(x); else if (a.scaled) G::unscale(x); } }
      if (ty != ~(size_t)0) { G::from_raw(y, ry); if (ky < 8) { if (!((have >> ky) & 1u)) G::set_identity(y); else if (a.scaled) G::unscale(y); } }
      if (step < 9) {
        operands(step + 1, kx, ky, tx, ty);
        if (tx != ~(size_t)0) G::load_raw(rx, arena + tx * G::PT_BYTES);
        if (ty != ~(size_t)0) G::load_raw(ry, arena + ty * G::PT_BYTES);
      }
    }
    switch (step) {
      case 0: B = y; break;
      case 1: D = y; break;
      case 2: x = B; y = D; break;
      case 3: x = A; break;
      case 4: D = y; break;
      case 5: x = D; break;
      case 6: y = D; break;
      case 8: x = A; y = B; break;
      case 9: y = B; break;
      default: break;
    }
    const unsigned long long ta0 = __builtin_amdgcn_s_memtime();
    G::add(x, y);
    if (dbg && gid == 0) { if (step == 2) dbg[8] = (u32)(__builtin_amdgcn_s_memtime() - ta0); if (step == 0) dbg[9] = (u32)(__builtin_amdgcn_s_memtime() - tk0); }
    size_t dst = ~(size_t)0;
    switch (step) {
      case 0: A = x; break;
      case 1: dst = o_r11; break;
      case 2: dst = o_r02; break;
      case 3: dst = o_a2; break;
      case 4: A = x; break;
      case 5: D = x; break;
      case 6: dst = o_r02; break;
      case 7: B = x; break;
      case 8: dst = o_a2 + 1; break;
      default: dst = o_r11; break;
    }
    if (dst != ~(size_t)0) G::store(arena + dst * G::PT_BYTES, x);
  }
  if (dbg && gid == 0) { dbg[10] = (u32)(__builtin_amdgcn_s_memtime() - tk0); dbg[11] = (u32)(__builtin_amdgcn_s_memrealtime() - tr0); }
}

// The last steps of the pyramid in ONE launch: from the step where a window's whole step fits a few passes of one
// 1024-thread block, the steps are only latency (one XYZZ addition deep each, ~7 us) and a launch per step costs
// 3-4x that.  One block per window walks the remaining steps with a block barrier between them (a step's
// tasks read what the previous step of the SAME window wrote: workgroup-scope visibility suffices), then copies
// U_{L-1} = A^{L-1}[1] into place (what k_copy_points did in a launch of its own).
// steps: [first, last]; task t of step s is tasks[step_off[s - first] + t].
struct PyrTailArgs { u32 first, last, max_count[20], step_off[21]; };
template <class G>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pyramid_tail(const PyrTask* __restrict__ tasks, PyrTailArgs ta, const CopyTaskPod* __restrict__ copy,
                                                       u32 do_copy, char* __restrict__ arena, u32 quad /* four lanes per item */) {
  const u32 w = blockIdx.x, tid = threadIdx.x;
  for (u32 s = ta.first; s <= ta.last; s++) {
    const u32 k = s - ta.first;
    const u32 nt = ta.step_off[k + 1] - ta.step_off[k], mc = ta.max_count[k];
    const u32 lanes = quad ? 4u * nt * mc : nt * mc;
    for (u32 t = tid; t < lanes; t += 1024) {
      const u32 it = quad ? t >> 2 : t;
      const u32 ti = it / mc, i = it - ti * mc;
      PyrTask tk = tasks[ta.step_off[k] + ti];
      bool serial = i < tk.count;
      if (quad && serial) serial = !pyr_try4<G>(tk, w, i, arena, t & 3u) && (t & 3u) == 0u;
      if (serial) pyr_item<G>(tk, w, i, arena);
    }
    __threadfence_block();
    __syncthreads();
  }
  if (do_copy) {
    const u32 wpp = G::PT_BYTES / 16;
    if (tid < wpp) {
      CopyTaskPod tk = *copy;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (tk.src_idx < tk.src_valid_idx)
        v = reinterpret_cast<const uint4*>(arena + ((size_t)tk.src_off + (size_t)w * tk.src_wstride + tk.src_idx) * G::PT_BYTES)[tid];
      reinterpret_cast<uint4*>(arena + ((size_t)tk.dst_off + (size_t)w * tk.dst_wstride) * G::PT_BYTES)[tid] = v;
    }
  }
}

// ------------------------------------------------------------------------------------
// ABI points (x*2^256, canonical) -> the hot kernels' domain (x*2^261, canonical): one pass per slab,
// only when k_accum1 runs in its plain form (short segments; see XYZZ29::madd_abi for the other form).
// (0,0) (the identity) maps to (0,0).
// ------------------------------------------------------------------------------------
template <class F29>
__global__ __launch_bounds__(256) void k_convert_points(const uint4* __restrict__ in, uint4* __restrict__ out, u32 n) {
  u32 i = blockIdx.x * 256 + threadIdx.x;   // one thread per coordinate
  if (i >= 2 * n) return;
  typename F29::fe a, r;
  F29::load(a, in + 2 * (size_t)i);
  F29::from_abi(r, a);
  F29::store(out + 2 * (size_t)i, r);
}

}  // namespace lemsm
