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
// chunk) that covers its whole bucket is stored to bucket_sum[key]; otherwise it becomes an
// edge record (at most two per thread: first and last segment) for the next level.
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
  if (s64 >= M) { rec_key[r0] = KEY_NONE; rec_key[r0 + 1] = KEY_NONE; return; }
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
  u32 nrec = 0;
  u32 first_key = KEY_NONE;

  typename G::pt acc; G::set_identity(acc);
  bool empty = true;          // acc == identity (tracked so that the loop tests a flag, not nine limbs)

  auto flush = [&](u32 seg_end) {
    bool complete = (seg_begin == kbeg) && (seg_end == kend);
    if (complete) {
      G::store(bucket_sum + (size_t)key * G::PT_BYTES, acc);
    } else {
      u32 slot = r0 + nrec;
      rec_key[slot] = key;
      G::store(rec_pt + (size_t)slot * G::PT_BYTES, acc);
      if (nrec == 0) first_key = key;
      nrec++;
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
  // Filler rule (DESIGN.md "edge records"): a lone record is followed by an identity record of
  // the same key, so that a bucket's run of records stays contiguous across threads (KEY_NONE is
  // only ever written where no run can pass through).
  if (nrec == 0) { rec_key[r0] = KEY_NONE; rec_key[r0 + 1] = KEY_NONE; }
  else if (nrec == 1) {
    typename G::pt id; G::set_identity(id);
    rec_key[r0 + 1] = first_key;
    G::store(rec_pt + (size_t)(r0 + 1) * G::PT_BYTES, id);
  }
}

// ------------------------------------------------------------------------------------
// level >= 2: segmented reduction of edge records (XYZZ + XYZZ).  Same completeness rule,
// decided from the neighbouring record keys.  R = number of input records.
// ------------------------------------------------------------------------------------
template <class G>
__global__ __launch_bounds__(256) void k_segreduce(u32 R, u32 L, u32 scaled /* k_accum1 ran in its ABI form */,
                                                   const u32* __restrict__ in_key,
                                                   const char* __restrict__ in_pt, char* __restrict__ bucket_sum,
                                                   u32* __restrict__ out_key, char* __restrict__ out_pt) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  const u64 c0_64 = (u64)t * L;
  if (c0_64 >= R) return;
  const u32 c0 = (u32)c0_64;
  const u32 c1 = (u32)min((u64)R, c0_64 + L);
  const u32 r0 = 2 * t;
  const u32 prev_key = c0 > 0 ? in_key[c0 - 1] : KEY_NONE;
  const u32 next_key = c1 < R ? in_key[c1] : KEY_NONE;

  typename G::pt acc; G::set_identity(acc);
  u32 cur = KEY_NONE;         // key of the open segment
  bool open_from_start = false;
  u32 nrec = 0, first_key = KEY_NONE;

  auto flush = [&](bool touches_end) {
    bool complete = !(open_from_start && prev_key == cur) && !(touches_end && next_key == cur);
    if (complete) {
      typename G::pt sc = acc;
      if (scaled) G::scale(sc);                     // bucket_sum[] then holds the scaled form (xyzz29.cuh)
      G::store(bucket_sum + (size_t)cur * G::PT_BYTES, sc);
    } else {
      u32 slot = r0 + nrec;
      out_key[slot] = cur;
      G::store(out_pt + (size_t)slot * G::PT_BYTES, acc);
      if (nrec == 0) first_key = cur;
      nrec++;
    }
  };

  for (u32 i = c0; i < c1; i++) {
    u32 k = in_key[i];
    if (k == KEY_NONE) {
      if (cur != KEY_NONE) { flush(false); cur = KEY_NONE; }
      continue;
    }
    if (k != cur) {
      if (cur != KEY_NONE) flush(false);
      cur = k;
      open_from_start = (i == c0);
      G::set_identity(acc);
    }
    typename G::pt q; G::load(q, in_pt + (size_t)i * G::PT_BYTES);
    if (scaled) G::unscale(q);                     // k_accum1's records are (X, Y, 32 ZZ, 32 ZZZ); ours are plain
    G::add(acc, q);
  }
  if (cur != KEY_NONE) flush(true);
  if (nrec == 0) { out_key[r0] = KEY_NONE; out_key[r0 + 1] = KEY_NONE; }
  else if (nrec == 1) {
    typename G::pt id; G::set_identity(id);
    out_key[r0 + 1] = first_key;
    G::store(out_pt + (size_t)(r0 + 1) * G::PT_BYTES, id);
  }
}

// ------------------------------------------------------------------------------------
// later edge-record levels: one record per lane, wavefront segmented scan (6 shuffle + add
// steps), so a level is 6 additions deep and shrinks the record list 32x.  A run that lies inside
// its wave goes to bucket_sum; per wave at most two edge records survive (its first run if it
// continues from the previous wave, its last run if it continues into the next), written with
// the same filler rule as above.
// ------------------------------------------------------------------------------------
template <class G>
__global__ __launch_bounds__(256) void k_segwave(u32 R, u32 scaled, const u32* __restrict__ in_key, const char* __restrict__ in_pt,
                                                 char* __restrict__ bucket_sum, u32* __restrict__ out_key,
                                                 char* __restrict__ out_pt) {
  const u32 gid = blockIdx.x * 256 + threadIdx.x;
  const u32 lane = threadIdx.x & 63u, wave = gid >> 6;
  const u32 wbase = wave << 6;
  if (wbase >= R) return;
  u32 key = gid < R ? in_key[gid] : KEY_NONE;
  typename G::pt acc;
  if (key != KEY_NONE) G::load(acc, in_pt + (size_t)gid * G::PT_BYTES); else G::set_identity(acc);
  // neighbours across the wave boundary
  u32 prev_glob = wbase > 0 ? in_key[wbase - 1] : KEY_NONE;
  u32 next_glob = wbase + 64 < R ? in_key[wbase + 64] : KEY_NONE;
  u32 knext = __shfl_down(key, 1); if (lane == 63) knext = next_glob;
  const u32 key0 = __shfl(key, 0);
#pragma unroll 1
  for (int d = 1; d < 64; d <<= 1) {
    u32 k2 = __shfl_up(key, d);
    typename G::pt q; G::shfl_up(q, acc, d);
    bool take = lane >= (u32)d && key != KEY_NONE && k2 == key;
    if (!take) G::set_identity(q);
    G::add(acc, q);
  }
  const bool last_of_run = key != KEY_NONE && knext != key;         // run ends inside the wave
  const bool open_end = key != KEY_NONE && lane == 63 && knext == key;   // run continues into the next wave
  const bool from_lane0 = key == key0;                                // run started at lane 0
  const bool before = from_lane0 && prev_glob == key && key != KEY_NONE;
  const bool holder = last_of_run || open_end;                       // lane holding its run's in-wave sum
  const bool edge = holder && (before || open_end);
  if (holder && !edge) { typename G::pt sc = acc; if (scaled) G::scale(sc); G::store(bucket_sum + (size_t)key * G::PT_BYTES, sc); }
  unsigned long long em = __ballot(edge);
  const u32 nedge = __popcll(em);
  const u32 o0 = 2 * wave;
  if (edge) {
    u32 rank = __popcll(em & ((1ull << lane) - 1ull));
    out_key[o0 + rank] = key;
    G::store(out_pt + (size_t)(o0 + rank) * G::PT_BYTES, acc);
    if (nedge == 1) {            // lone record: identity filler of the same key keeps the run contiguous
      typename G::pt id; G::set_identity(id);
      out_key[o0 + 1] = key;
      G::store(out_pt + (size_t)(o0 + 1) * G::PT_BYTES, id);
    }
  }
  if (nedge == 0 && lane == 0) { out_key[o0] = KEY_NONE; out_key[o0 + 1] = KEY_NONE; }
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
