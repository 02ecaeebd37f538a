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
//   k_merge_queues  3..8 pieces: one thread per queued bucket, serial.  9..32 pieces: one wave per bucket while
//                   the queue is short (latency), else serial like the short ones (throughput).  More than 32:
//                   slices of <= `slice` pieces, one wave each (lanes stride over the pieces, then a shuffle
//                   tree); a bucket of several slices leaves one partial sum per slice.
//   k_merge_final   one wave per multi-slice bucket adds its partials.
// Depth: 1 addition for uniform scalars (was 8 + 4 x 6 through the segmented-scan levels), <= slice/64 + 6 + 6
// for any input; the sum of a bucket is formed in piece order whatever the queue order, so results are
// deterministic.
// ------------------------------------------------------------------------------------
enum { MQ_S = 0, MQ_M = 1, MQ_L = 2, MQ_F = 3, MQ_P = 4, MQ_WORDS = 8 };   // queue counters: short, medium, long slices, multi-slice buckets, partials
struct MqLayout { u32 offS, offM, offL, offF, capS, capM, capL, capF, capP, slice, wave_th; };
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

// lane 0 of the wave ends up with the sum of the lanes' points (lanes >= nact hold the identity)
template <class G>
__device__ __forceinline__ void merge_wave_tree(typename G::pt& acc, u32 lane, u32 nact) {
#pragma unroll 1
  for (u32 d = 32; d >= 1; d >>= 1) {
    if (d >= nact) continue;                      // wave-uniform: nothing lives at lane >= d yet
    typename G::pt q; G::shfl(q, acc, (int)((lane + d) & 63u));
    if (lane >= d || lane + d >= nact) G::set_identity(q);
    G::add(acc, q);
  }
}

template <class G>
__global__ __launch_bounds__(256) void k_merge_queues(u32 nblk_short, u32 scaled, MqLayout lay, const u32* __restrict__ mq_cnt,
                                                      const uint4* __restrict__ mq_items, const char* __restrict__ rec_pt,
                                                      char* __restrict__ partial, char* __restrict__ bucket_sum) {
  const u32 cS = min(mq_cnt[MQ_S], lay.capS), cMall = min(mq_cnt[MQ_M], lay.capM), cL = min(mq_cnt[MQ_L], lay.capL);
  const bool m_serial = cMall > lay.wave_th;
  if (blockIdx.x < nblk_short) {
    // one thread per queued bucket; the two classes start on wave boundaries so that a wave's trip counts are alike
    const u32 v = blockIdx.x * 256 + threadIdx.x;
    const u32 rS = (cS + 63u) & ~63u, cM = m_serial ? cMall : 0u;
    uint4 it;
    if (v < rS) { if (v >= cS) return; it = mq_items[lay.offS + v]; }
    else { if (v - rS >= cM) return; it = mq_items[lay.offM + (v - rS)]; }
    const u32 key = it.x, t0 = it.y, n = it.z & 0x7fffffffu, lf = it.z >> 31;
    typename G::pt acc, q;
    merge_load_piece<G>(acc, rec_pt, t0, false, scaled);
#pragma unroll 1
    for (u32 j = 1; j < n; j++) {
      merge_load_piece<G>(q, rec_pt, t0 + j, j + 1 == n && lf, scaled);
      G::add(acc, q);
    }
    if (scaled) G::scale(acc);
    G::store(bucket_sum + (size_t)key * G::PT_BYTES, acc);
    return;
  }
  // one wave per medium bucket / per slice of a long one
  const u32 lane = threadIdx.x & 63u;
  const u32 nwaves = (gridDim.x - nblk_short) * 4u;
  const u32 cM = m_serial ? 0u : cMall;
#pragma unroll 1
  for (u32 w = (blockIdx.x - nblk_short) * 4u + (threadIdx.x >> 6); w < cM + cL; w += nwaves) {
    const uint4 it = w < cM ? mq_items[lay.offM + w] : mq_items[lay.offL + (w - cM)];
    const u32 key = it.x, t0 = it.y, n = it.z & 0x7fffffffu, lf = it.z >> 31, dst = it.w;
    typename G::pt acc, q; G::set_identity(acc);
#pragma unroll 1
    for (u32 j = lane; j < n; j += 64) {
      merge_load_piece<G>(q, rec_pt, t0 + j, j + 1 == n && lf, scaled);
      G::add(acc, q);
    }
    merge_wave_tree<G>(acc, lane, min(n, 64u));
    if (lane == 0) {
      if (dst == MQ_DST_BUCKET) { if (scaled) G::scale(acc); G::store(bucket_sum + (size_t)key * G::PT_BYTES, acc); }
      else if (dst < lay.capP) G::store(partial + (size_t)dst * G::PT_BYTES, acc);      // plain form
    }
  }
}

template <class G>
__global__ __launch_bounds__(256) void k_merge_final(u32 scaled, MqLayout lay, const u32* __restrict__ mq_cnt, const uint4* __restrict__ mq_items,
                                                     const char* __restrict__ partial, char* __restrict__ bucket_sum) {
  const u32 cF = min(mq_cnt[MQ_F], lay.capF);
  const u32 lane = threadIdx.x & 63u, nwaves = gridDim.x * 4u;
#pragma unroll 1
  for (u32 w = blockIdx.x * 4u + (threadIdx.x >> 6); w < cF; w += nwaves) {
    const uint4 it = mq_items[lay.offF + w];
    const u32 key = it.x, p0 = it.y, n = it.z;
    typename G::pt acc, q; G::set_identity(acc);
#pragma unroll 1
    for (u32 j = lane; j < n; j += 64) {
      if (p0 + j < lay.capP) { G::load(q, partial + (size_t)(p0 + j) * G::PT_BYTES); G::add(acc, q); }
    }
    merge_wave_tree<G>(acc, lane, min(n, 64u));
    if (lane == 0) { if (scaled) G::scale(acc); G::store(bucket_sum + (size_t)key * G::PT_BYTES, acc); }
  }
}

// ------------------------------------------------------------------------------------
// bucket reduction: pairwise-add pyramid.  One launch per step; a step is a list of tasks
//   dst[i] = src[(2i)*stride + phase] + src[(2i+1)*stride + phase],   i < count
// applied to every window.  Arena offsets are in points (128 B); src indices >= src_valid
// read as the identity (padding of non-power-of-two bucket counts).
// ------------------------------------------------------------------------------------
// (layout of CopyTask in kernels.cuh)
struct CopyTaskPod { u32 src_off, src_wstride, dst_off, dst_wstride, src_valid_idx, src_idx; };

struct PyrTask {
  u32 src_off, src_wstride;   // per-window base = src_off + w * src_wstride
  u32 dst_off, dst_wstride;
  u32 stride, phase, count, src_valid;
  u32 src_scaled, pad_[3];    // source is bucket_sum[] in the scaled form: convert on load
};

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

template <class G>
__global__ __launch_bounds__(256) void k_pyramid(const PyrTask* __restrict__ tasks, u32 ntasks, u32 nwin,
                                                 u32 max_count, char* __restrict__ arena) {
  u32 gid = blockIdx.x * 256 + threadIdx.x;
  u32 per_task = max_count * nwin;
  u32 ti = gid / per_task;
  if (ti >= ntasks) return;
  u32 rem = gid - ti * per_task;
  u32 w = rem / max_count, i = rem - w * max_count;
  PyrTask tk = tasks[ti];
  if (i >= tk.count) return;
  pyr_item<G>(tk, w, i, arena);
}

// The last steps of the pyramid in ONE launch: from the step where a window's whole step fits a few passes of one
// 1024-thread block, the steps are only latency (one XYZZ addition deep each, ~7 us) and a launch per step costs
// 3-4x that.  One block per window walks the remaining steps with a block barrier between them (a step's
// tasks read what the previous step of the SAME window wrote: workgroup-scope visibility suffices), then copies
// U_{L-1} = A^{L-1}[1] into place (what k_copy_points did in a launch of its own).
// steps: [first, last]; task t of step s is tasks[step_off[s - first] + t].
struct PyrTailArgs { u32 first, last, max_count[20], step_off[21]; };
template <class G>
__global__ __launch_bounds__(1024) void k_pyramid_tail(const PyrTask* __restrict__ tasks, PyrTailArgs ta, const CopyTaskPod* __restrict__ copy,
                                                       u32 do_copy, char* __restrict__ arena) {
  const u32 w = blockIdx.x, tid = threadIdx.x;
  for (u32 s = ta.first; s <= ta.last; s++) {
    const u32 k = s - ta.first;
    const u32 nt = ta.step_off[k + 1] - ta.step_off[k], mc = ta.max_count[k];
    for (u32 it = tid; it < nt * mc; it += 1024) {
      const u32 ti = it / mc, i = it - ti * mc;
      PyrTask tk = tasks[ta.step_off[k] + ti];
      if (i < tk.count) pyr_item<G>(tk, w, i, arena);
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
