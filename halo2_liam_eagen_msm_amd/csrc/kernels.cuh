// HIP kernels of the MSM witness path (gfx950).  See DESIGN.md for the pipeline:
//
//   digits -> pass-1 partition by coarse bin (LDS histograms) -> pass-2 exact bucket
//   sort inside each bin (LDS histograms) -> level-1 segmented accumulation of the
//   bucket-sorted point list (the dominant kernel: one XYZZ mixed add per entry)
//   -> segmented reduction of the per-thread edge records -> pairwise-add pyramid
//   that turns bucket sums into per-window (total, U_0..U_{L-1}) -> host Horner.
//
// Reference behaviour being replaced (paths relative to /root/reference):
//   src/negbase_utils.rs:20-36        negbase_decompose         -> k_negbase_digits
//   src/negbase_utils.rs:46-51        id_by_digit (bucket = digit-1, 0 skipped) -> NegDec
//   src/argument_witness_calc.rs:97   scalar range assert       -> k_negbase_digits (flag)
//   src/argument_witness_calc.rs:105-127 per-digit sums S_i of the Horner recursion -> buckets
//   halo2 best_multiexp (third party) bucket accumulation       -> k_pip_digits/PipDec + same kernels
#pragma once
#include "kernels_ec.cuh"
#include "inv29.cuh"

namespace lemsm {

// ------------------------------------------------------------------------------------
// window digit matrices and their decoders
// ------------------------------------------------------------------------------------
__device__ __forceinline__ u32 extract_bits(const u32 (&s)[8], u32 bitpos, u32 c) {
  u32 li = bitpos >> 5, sh = bitpos & 31;
  u32 lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    lo = ((u32)i == li) ? s[i] : lo;
    hi = ((u32)i == li + 1) ? s[i] : hi;
  }
  u64 v = ((u64)hi << 32) | lo;
  return (u32)(v >> sh) & ((1u << c) - 1u);
}

struct KAdd { u32 k[8]; u32 order[8]; };   // window offset constant K and the scalar-field order

// Signed-window Pippenger: s' = s + K with K = sum_{w<W-1} 2^(c-1) 2^(cw).  The raw c-bit windows
// of s' are written window-major (dig16[(w-w0)*n + j], one coalesced 2-byte column per window) so
// that the sort passes stream one window at a time.
template <int MODE /* 0: any c <= 16, 1: c == 16, 2: c == 17 (sign bitmap beside the u16 column) */,
          int BS = 256 /* threads per block: 1024 when there are too few ranges to fill the chip with 256 */>
__global__ __launch_bounds__(BS) void k_pip_digits(const uint4* __restrict__ scalars, KAdd kadd, GroupPlan pl,
                                                    uint16_t* __restrict__ dig16, unsigned long long* __restrict__ signbm,
                                                    u32* __restrict__ block_counts,
                                                    u32* __restrict__ bin_total, u32* __restrict__ err) {
  // one block = one pass-1 range of spb scalars, all windows of the group: writes the digit
  // columns AND the per-(window, range, bin) counts the scatter needs (no separate count pass)
  __shared__ u32 hist[MAX_BINS];
  const u32 tid = threadIdx.x, r = blockIdx.x;
  for (u32 i = tid; i < pl.nbins; i += BS) hist[i] = 0;
  __syncthreads();
  const u32 half = 1u << (pl.c - 1);
  u32 j0 = r * pl.spb, j1 = min(j0 + pl.spb, pl.n);
  for (u32 jb = j0 + tid; jb < j1; jb += BS * 4) {
    uint4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {     // four scalars in flight per thread
      u32 j = jb + (u32)BS * u;
      if (j < j1) { a[u] = scalars[2 * (size_t)j]; b[u] = scalars[2 * (size_t)j + 1]; }
      else { a[u] = make_uint4(0, 0, 0, 0); b[u] = a[u]; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      u32 j = jb + (u32)BS * u;
      if (j >= j1) continue;   // wave-uniform for MODE 2's ballot: ranges and n are handled in whole waves (see below)
      u32 s[8] = {a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w};
      // scalars must be canonical (< order), as PrimeField::to_repr() guarantees; anything else is
      // reported (LEMSM_ERR_SCALAR_OUT_OF_RANGE) and contributes nothing
      u32 bw = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) { u32 dmy = __builtin_subc(s[i], kadd.order[i], bw, &bw); (void)dmy; }
      if (!bw) {
        atomicAdd(&err[0], 1u); atomicMax(&err[1], ~j);
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = 0;
      }
      u32 cy = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) s[i] = __builtin_addc(s[i], kadd.k[i], cy, &cy);
      if (MODE == 2) {
        // c = 17: digit d = raw - 2^16 in [-2^16, 2^16) does not fit 16 bits with its sign.  The
        // column stores v = |d| for d >= 0 and |d| - 1 for d < 0; the sign goes to a bitmap (one
        // 64-bit word per wave and window, written from a ballot).  v = 0, sign = 0 is d = 0.
#pragma unroll
        for (int w = 0; w < 15; w++) {
          if ((u32)w < pl.w0 || (u32)w >= pl.w1) continue;
          u32 raw = extract_bits(s, 17u * w, 17u);
          u32 sign = ((u32)w + 1 < pl.W && raw < 65536u) ? 1u : 0u;
          u32 mag = ((u32)w + 1 < pl.W) ? (sign ? 65536u - raw : raw - 65536u) : raw;
          dig16[(size_t)((u32)w - pl.w0) * pl.dstride + j] = (uint16_t)(sign ? mag - 1u : mag);
          unsigned long long bal = __ballot(sign != 0);
          if ((threadIdx.x & 63u) == 0) signbm[(size_t)((u32)w - pl.w0) * ((pl.n + 63) / 64) + (j >> 6)] = bal;
          if (mag) atomicAdd(&hist[((u32)w - pl.w0) * pl.BW + ((mag - 1u) >> pl.LB)], 1u);
        }
      } else if (MODE == 1) {
        // c = 16: window w is the (w & 1)-th half of limb w >> 1 -- static indexing, 2 instructions
        // per window instead of the select chain of extract_bits (this kernel is VALU-heavy)
#pragma unroll
        for (int w = 0; w < 16; w++) {
          if ((u32)w < pl.w0 || (u32)w >= pl.w1) continue;
          u32 raw = (w & 1) ? (s[w >> 1] >> 16) : (s[w >> 1] & 0xffffu);
          dig16[(size_t)((u32)w - pl.w0) * pl.dstride + j] = (uint16_t)raw;
          u32 bucket = ((u32)w + 1 < pl.W) ? (raw < half ? half - raw : raw - half) : raw;
          if (bucket) atomicAdd(&hist[((u32)w - pl.w0) * pl.BW + ((bucket - 1u) >> pl.LB)], 1u);
        }
      } else {
        for (u32 w = pl.w0; w < pl.w1; w++) {
          u32 raw = extract_bits(s, w * pl.c, pl.c);
          dig16[(size_t)(w - pl.w0) * pl.dstride + j] = (uint16_t)raw;
          u32 bucket = (w + 1 < pl.W) ? (raw < half ? half - raw : raw - half) : raw;
          if (bucket) atomicAdd(&hist[(w - pl.w0) * pl.BW + ((bucket - 1u) >> pl.LB)], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (u32 i = tid; i < pl.nbins; i += BS) {
    u32 cnt = hist[i];
    u32 wl = i / pl.BW, bin = i - wl * pl.BW;
    const size_t slot = ((size_t)wl * pl.nblk1 + r) * pl.BW + bin;
    block_counts[slot] = cnt;
    // the total's atomic is also the claim: what it returns is where this block's run starts inside the bin
    // (pass 1 used to claim with a second round of atomics per (block, bin))
    block_counts[(size_t)pl.nblk1 * pl.nbins + slot] = cnt ? atomicAdd(&bin_total[i], cnt) : 0u;
  }
}

// decoders: bucket (0 = skip) and sign of scalar j in window w
struct PipDec {
  static constexpr bool VEC = true;   // pass 1 reads 16 consecutive digits of a column with two 16-byte loads
  const uint16_t* dig16;
  const unsigned long long* signbm;   // c = 17 only (null otherwise)
  __device__ __forceinline__ void get(u32 j, u32 w, const GroupPlan& pl, u32& bucket, u32& sign) const {
    u32 raw = dig16[(size_t)(w - pl.w0) * pl.dstride + j];
    if (signbm) {                      // c = 17 encoding, see k_pip_digits<2>
      sign = (u32)(signbm[(size_t)(w - pl.w0) * ((pl.n + 63) / 64) + (j >> 6)] >> (j & 63u)) & 1u;
      bucket = sign ? raw + 1u : raw;
      return;
    }
    const u32 half = 1u << (pl.c - 1);
    if (w + 1 < pl.W) {          // digit = raw - 2^(c-1), top window unsigned
      sign = raw < half ? 1u : 0u;
      bucket = sign ? half - raw : raw - half;
    } else { sign = 0; bucket = raw; }
  }
};
// negabase digit matrix, position-major; bucket id = digit (id_by_digit: digit-1, 0 skipped;
// src/negbase_utils.rs:46-51)
struct NegDec {
  static constexpr bool VEC = false;
  const uint8_t* digitsT;   // d x nstride, already offset to this slab's first column
  __device__ __forceinline__ void get(u32 j, u32 w, const GroupPlan& pl, u32& bucket, u32& sign) const {
    bucket = digitsT[(size_t)w * pl.nstride + j]; sign = 0;
  }
};

// exclusive scan of one value per thread across a 256-thread block
__device__ __forceinline__ u32 block_excl_scan_256(u32 v, u32* total, u32* wsum /* >= 4 words LDS */) {
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    u32 y = __shfl_up(x, o);
    if (lane >= (u32)o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (u32 w = 0; w < 4; w++) {
    u32 s = wsum[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

// ------------------------------------------------------------------------------------
// pass 1: partition (bucket, point index) pairs by coarse bin.  One block = one window x one
// range of <= STAGE scalars.  Keys: key = (w-w0)*nbw + bucket-1, bin = key >> LB, BW bins per window.
// The scatter stages a block's entries bin-sorted in LDS and writes each (block, bin) run with
// consecutive lanes on consecutive addresses: scattered 4-byte stores cost ~16x write
// amplification in HBM (the first version of this pass ran at 0.5 TB/s).
// ------------------------------------------------------------------------------------
static const u32 BW_MAX = 512;     // coarse bins per window (512 only for 17-bit windows, else <= 256)
static const u32 STAGE = 4096;     // pass-1 range per block (more resident blocks hide its latency chain)
static const u32 STAGE2 = 8192;    // pass-2 tile (longer runs per local bucket)

template <class Dec>
__global__ __launch_bounds__(256) void k_count1(Dec dec, GroupPlan pl, u32* __restrict__ block_counts,
                                                u32* __restrict__ bin_total) {
  __shared__ u32 hist[256];
  const u32 tid = threadIdx.x, wl = blockIdx.x, r = blockIdx.y, w = pl.w0 + wl;
  hist[tid] = 0;
  __syncthreads();
  u32 j0 = r * pl.spb, j1 = min(j0 + pl.spb, pl.n);
  constexpr int PER = STAGE / 256;
  u32 bk[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    u32 j = j0 + tid + 256u * k, sign;
    bk[k] = 0;
    if (j < j1) dec.get(j, w, pl, bk[k], sign);
  }
  // At most 16 bins per window (negabase digits: bins are buckets, base - 1 of them): a block's 4096 atomics on 15 addresses
  // serialise in the LDS (tools/ubench/lds_rank_rates.hip: 2.2 lane-operations per clock at 8 addresses against 6.0 at 512),
  // so lane l counts in copy l % 16 of the histogram and the copies are added up afterwards.
  const bool few = pl.BW <= 16u;
  const u32 rep = few ? (tid & 15u) << 4 : 0u;
#pragma unroll
  for (int k = 0; k < PER; k++)
    if (bk[k]) atomicAdd(&hist[rep + ((bk[k] - 1u) >> pl.LB)], 1u);
  __syncthreads();
  if (few) {
    u32 c = 0;
    if (tid < 16u) {
#pragma unroll
      for (int r = 0; r < 16; r++) c += hist[r * 16 + tid];
    }
    __syncthreads();
    if (tid < 16u) hist[tid] = c;
    __syncthreads();
  }
  if (tid < pl.BW) {
    u32 cnt = hist[tid];
    const size_t slot = ((size_t)wl * pl.nblk1 + r) * pl.BW + tid;
    block_counts[slot] = cnt;
    block_counts[(size_t)pl.nblk1 * pl.nbins + slot] = cnt ? atomicAdd(&bin_total[wl * pl.BW + tid], cnt) : 0u;   // (the claim, as in k_pip_digits)
  }
}

// exclusive scan of one value per thread across a 1024-thread block
__device__ __forceinline__ u32 block_excl_scan_1024(u32 v, u32* total, u32* wsum /* >= 16 words LDS */) {
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    u32 y = __shfl_up(x, o);
    if (lane >= (u32)o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  u32 base = 0, tot = 0;
  for (u32 w = 0; w < 16; w++) {
    u32 s = wsum[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

// bin_start[nbins+1], tile_prefix[nbins+1], meta[META_M], meta[META_TILES]
__global__ __launch_bounds__(1024) void k_binscan(GroupPlan pl, const u32* __restrict__ bin_total,
                                                  u32* __restrict__ bin_start, u32* __restrict__ tile_prefix,
                                                  u32* __restrict__ meta) {
  __shared__ u32 wsum[16];
  const u32 t = threadIdx.x;
  constexpr int PB = MAX_BINS / 1024;   // bins per thread
  u32 cnt[PB], tl[PB], s = 0, st = 0;
#pragma unroll
  for (int k = 0; k < PB; k++) {
    u32 b = t * PB + k;
    cnt[k] = b < pl.nbins ? bin_total[b] : 0;
    tl[k] = (pl.bin_cap && cnt[k] <= pl.bin_cap) ? 0u : (cnt[k] + pl.T2 - 1) / pl.T2;   // tiles only for the bins k_binsort leaves alone
    s += cnt[k]; st += tl[k];
  }
  u32 tot, tott;
  u32 off = block_excl_scan_1024(s, &tot, wsum);
  u32 offt = block_excl_scan_1024(st, &tott, wsum);
#pragma unroll
  for (int k = 0; k < PB; k++) {
    u32 b = t * PB + k;
    if (b < pl.nbins) { bin_start[b] = off; tile_prefix[b] = offt; }
    off += cnt[k]; offt += tl[k];
  }
  if (t == 0) {
    bin_start[pl.nbins] = tot; tile_prefix[pl.nbins] = tott;
    meta[META_M] = tot; meta[META_TILES] = tott;
  }
}

template <class Dec, int STG /* entries staged per block: STAGE, 2*STAGE, or 4*STAGE with 1024 threads */, int BS = 256,
          bool LEAN = false /* 256-thread blocks, an even number of bins per window: a thread owns two ADJACENT bins (8-byte loads of its
          counts / claims / bin starts, ONE block scan instead of two), the staged word keeps local bucket and sign where the entry has
          them (entry = (word & 0xff003fff) + j0), and the store loop is unrolled so that its LDS reads are in flight together */>
__global__ __launch_bounds__(BS) void k_scatter1(Dec dec, GroupPlan pl, const u32* __restrict__ block_counts,
                                                 const u32* __restrict__ bin_start,
                                                 u32* __restrict__ entries, u32 xcd_windows /* 1-D grid, one XCD per window */) {
  static_assert(BS == 256 || BS == 1024, "block size");
  static_assert(STG <= 16384, "jl field is 14 bits");
  static_assert(!LEAN || BS == 256, "the lean form is the 256-thread one");
  __shared__ __align__(8) u32 delta[BW_MAX];        // global position of a bin's run minus its position in the staging buffer
  __shared__ __align__(8) u32 lcur[BW_MAX];
  __shared__ u32 wsum[16];
  __shared__ u32 stage[STG];           // jl:14 | local:7 | sign:1 | bin:9  (jl = index inside the block's range); LEAN: jl:14 | bin:9 | 0 | local:7 | sign:1
  // Block -> (window, range).  With >= 8 windows (1-D grid of 8 * ceil(gw/8) * nblk1 blocks) all blocks of
  // one window carry the same blockIdx.x % 8, i.e. run on one XCD (guide T1: the label groups blocks by XCD):
  // neighbouring (block, bin) runs of a window are then written through ONE L2 and merge into whole lines --
  // with the runs of a window spread over all XCDs, 61 % of this kernel's write requests were 32-byte halves
  // (profiles/r01/x_pmc_sort_kernels_2p24.txt).  Fewer windows: grid = (windows, ranges) as before.
  const u32 tid = threadIdx.x;
  u32 wl, r;
  if (xcd_windows) {
    const u32 wpx = (pl.w1 - pl.w0 + 7u) >> 3, slot = blockIdx.x >> 3;
    wl = (blockIdx.x & 7u) + 8u * (slot % wpx); r = slot / wpx;
    if (wl >= pl.w1 - pl.w0) return;
  } else { wl = blockIdx.x; r = blockIdx.y; }
  const u32 w = pl.w0 + wl;
  const u32 lmask = (1u << pl.LB) - 1u;
  u32 j0 = r * pl.spb, j1 = min(j0 + pl.spb, pl.n);
  // Issue every load of the block up front -- the digits and this block's bin counts and claimed offsets -- so that
  // their latencies overlap (a rolled loop waits out one memory latency per entry: measured 3x slower).
  constexpr int PER = STG / BS;
  u32 bk[PER];   // bucket | sign << 31   (0 = no entry)
  u32 jl0, jls;  // entry k of this thread is scalar j0 + jl0 + jls * k
  if constexpr (Dec::VEC) {
    // PER consecutive digits of the column per thread: PER/8 16-byte loads (columns are 16-byte aligned: pl.dstride)
    // and the PER sign bits in one word, instead of 2 * PER scalar loads with their 64-bit address arithmetic
    static_assert(PER == 16 || PER == 32, "digits per thread");
    jl0 = tid * PER; jls = 1;
    const u32 jb = j0 + jl0;
    uint4 dv[PER / 8];
    u32 sbits = 0;
    if (jb < j1) {
      const uint4* p = (const uint4*)(dec.dig16 + (size_t)wl * pl.dstride + jb);
#pragma unroll
      for (int v = 0; v < PER / 8; v++) dv[v] = p[v];
      if (dec.signbm) {
        const unsigned long long* sw = dec.signbm + (size_t)wl * ((pl.n + 63) / 64);
        if (PER == 16) sbits = ((const uint16_t*)sw)[jb >> 4]; else sbits = ((const u32*)sw)[jb >> 5];
      }
    } else {
#pragma unroll
      for (int v = 0; v < PER / 8; v++) dv[v] = make_uint4(0, 0, 0, 0);
    }
    const u32 half = 1u << (pl.c - 1);
    const bool top = !(w + 1 < pl.W);
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const uint4 q4 = dv[k / 8];
      const u32 word = ((k / 2) % 4 == 0) ? q4.x : ((k / 2) % 4 == 1) ? q4.y : ((k / 2) % 4 == 2) ? q4.z : q4.w;
      const u32 raw = (k & 1) ? (word >> 16) : (word & 0xffffu);
      u32 bucket, sign;
      if (dec.signbm) { sign = (sbits >> k) & 1u; bucket = raw + sign; }                       // c = 17 encoding, see k_pip_digits<2>
      else if (top) { sign = 0; bucket = raw; }
      else { sign = raw < half ? 1u : 0u; bucket = sign ? half - raw : raw - half; }
      bk[k] = (jb + (u32)k < j1 && bucket) ? (bucket | (sign << 31)) : 0u;
    }
  } else {
    jl0 = tid; jls = BS;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      u32 j = j0 + tid + (u32)BS * k, b = 0, sgn = 0;
      if (j < j1) dec.get(j, w, pl, b, sgn);
      bk[k] = b ? (b | (sgn << 31)) : 0u;
    }
  }
  const size_t crow = ((size_t)wl * pl.nblk1 + r) * pl.BW;
  const u32* __restrict__ block_offs = block_counts + (size_t)pl.nblk1 * pl.nbins;   // claimed by the counting kernel's atomics
  u32 total;
  // At most 16 bins per window (negabase digits): 16 cursors per bin, lane l ranks in copy l % 16, so that the block's 4096 returning
  // atomics spread over 16 x BW addresses instead of serialising on BW (see k_count1).  The copies' counts are taken here (one more
  // pass of non-returning atomics) and scanned bin-major: copy r of bin b owns the part [scan(b, r), scan(b, r + 1)) of the bin's run.
  const bool few = !Dec::VEC && BS == 256 && !LEAN && pl.BW <= 16u;
  if (few) {
    lcur[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++)
      if (bk[k]) atomicAdd(&lcur[((((bk[k] & 0x7fffffffu) - 1u) >> pl.LB) << 4) | (tid & 15u)], 1u);
    __syncthreads();
    const u32 v = lcur[tid];
    const u32 off = block_excl_scan_256(v, &total, wsum);
    lcur[tid] = off;
    const u32 b = tid >> 4;
    if ((tid & 15u) == 0 && b < pl.BW) delta[b] = bin_start[wl * pl.BW + b] + block_offs[crow + b] - off;
  } else
  if constexpr (LEAN) {
    // bins 2 tid and 2 tid + 1 (BW is even and <= 512; every table below starts 8-byte aligned and crow, wl * BW are even)
    const u32 b0 = 2u * tid;
    const bool h = b0 < pl.BW;
    const uint2 z2 = make_uint2(0u, 0u);
    const uint2 cnt = h ? *reinterpret_cast<const uint2*>(block_counts + crow + b0) : z2;
    const uint2 clm = h ? *reinterpret_cast<const uint2*>(block_offs + crow + b0) : z2;
    const uint2 bs = h ? *reinterpret_cast<const uint2*>(bin_start + (size_t)wl * pl.BW + b0) : z2;
    const u32 off0 = block_excl_scan_256(cnt.x + cnt.y, &total, wsum), off1 = off0 + cnt.x;
    *reinterpret_cast<uint2*>(&delta[b0]) = make_uint2(bs.x + clm.x - off0, bs.y + clm.y - off1);
    *reinterpret_cast<uint2*>(&lcur[b0]) = make_uint2(off0, off1);
  } else if constexpr (BS == 256) {
    // up to BW_MAX = 512 bins per window: two bins per thread (tid and tid + 256)
    const bool h0 = tid < pl.BW, h1 = tid + 256u < pl.BW;
    u32 cnt0 = h0 ? block_counts[crow + tid] : 0u, cnt1 = h1 ? block_counts[crow + tid + 256u] : 0u;
    u32 g0 = h0 ? bin_start[wl * pl.BW + tid] + block_offs[crow + tid] : 0u;
    u32 g1 = h1 ? bin_start[wl * pl.BW + tid + 256u] + block_offs[crow + tid + 256u] : 0u;
    u32 total0, total1;
    u32 off0 = block_excl_scan_256(cnt0, &total0, wsum);
    u32 off1 = block_excl_scan_256(cnt1, &total1, wsum) + total0;
    total = total0 + total1;
    delta[tid] = g0 - off0; delta[tid + 256u] = g1 - off1;
    lcur[tid] = off0; lcur[tid + 256u] = off1;     // the cursor of a bin starts at its run: one LDS operation ranks and places an entry
  } else {
    const bool h0 = tid < pl.BW;
    u32 cnt0 = h0 ? block_counts[crow + tid] : 0u;
    u32 g0 = h0 ? bin_start[wl * pl.BW + tid] + block_offs[crow + tid] : 0u;
    u32 off0 = block_excl_scan_1024(cnt0, &total, wsum);
    if (tid < BW_MAX) { delta[tid] = g0 - off0; lcur[tid] = off0; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++) {
    if (bk[k]) {
      u32 kk = (bk[k] & 0x7fffffffu) - 1u, b = kk >> pl.LB;
      u32 q = atomicAdd(&lcur[few ? ((b << 4) | (tid & 15u)) : b], 1u);
      if constexpr (LEAN) stage[q] = (bk[k] & 0x80000000u) | (jl0 + jls * (u32)k) | (b << 14) | ((kk & lmask) << 24);
      else stage[q] = (jl0 + jls * (u32)k) | ((kk & lmask) << 14) | ((bk[k] >> 31) << 21) | (b << 22);
    }
  }
  __syncthreads();
  if constexpr (LEAN) {
    u32 v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) { const u32 q = tid + (u32)BS * k; v[k] = q < total ? stage[q] : 0u; }
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const u32 q = tid + (u32)BS * k;
      if (q < total) entries[delta[(v[k] >> 14) & 511u] + q] = (v[k] & 0xff003fffu) + j0;
    }
  } else {
    for (u32 q = tid; q < total; q += BS) {
      u32 v = stage[q];
      entries[delta[v >> 22] + q] = (j0 + (v & 0x3fffu)) | (((v >> 14) & 127u) << 24) | (((v >> 21) & 1u) << 31);
    }
  }
}

// ------------------------------------------------------------------------------------
// pass 2: exact bucket sort inside each bin, one tile (<= T2 <= STAGE entries of one bin) per block
// ------------------------------------------------------------------------------------
// tile table: one thread per tile finds its bin once (binary search over tile_prefix), so that
// the two pass-2 kernels start with one 16-byte load instead of a dependent-load chain per block
__global__ __launch_bounds__(256) void k_tilemap(GroupPlan pl, const u32* __restrict__ bin_start,
                                                 const u32* __restrict__ tile_prefix, const u32* __restrict__ meta,
                                                 uint4* __restrict__ tile_info) {
  u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= pl.max_tiles) return;
  if (t >= meta[META_TILES]) { tile_info[t] = make_uint4(0, 0, 0, 0); return; }
  u32 lo = 0, hi = pl.nbins;   // largest b with tile_prefix[b] <= t
  while (hi - lo > 1) {
    u32 mid = (lo + hi) >> 1;
    if (tile_prefix[mid] <= t) lo = mid; else hi = mid;
  }
  u32 off = bin_start[lo] + (t - tile_prefix[lo]) * pl.T2;
  u32 end = min(off + pl.T2, bin_start[lo + 1]);
  tile_info[t] = make_uint4(lo, off, end, 1u);
}
__device__ __forceinline__ bool locate_tile(const uint4* __restrict__ tile_info, u32 t, u32& bin, u32& off, u32& end) {
  uint4 ti = tile_info[t];
  bin = ti.x; off = ti.y; end = ti.z;
  return ti.w != 0;
}

__global__ __launch_bounds__(256) void k_count2(GroupPlan pl, const u32* __restrict__ entries,
                                                const uint4* __restrict__ tile_info, u32* __restrict__ bucket_count) {
  __shared__ u32 hist[1u << MAX_LB];
  const u32 tid = threadIdx.x;
  // tiles are a prefix of the table; the grid may be smaller than the table (k_binsort takes most bins: few tiles)
  for (u32 t = blockIdx.x; t < pl.max_tiles; t += gridDim.x) {
    u32 bin, off, end;
    if (!locate_tile(tile_info, t, bin, off, end)) return;
    if (tid < (1u << MAX_LB)) hist[tid] = 0;
    __syncthreads();
    constexpr int PER = STAGE2 / 256;    // T2 <= STAGE2
    u32 e[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) { u32 i = off + tid + 256u * k; e[k] = i < end ? entries[i] : 0xffffffffu; }
#pragma unroll
    for (int k = 0; k < PER; k++) { u32 i = off + tid + 256u * k; if (i < end) atomicAdd(&hist[(e[k] >> 24) & 127u], 1u); }
    __syncthreads();
    if (tid < (1u << pl.LB)) {
      u32 cnt = hist[tid];
      if (cnt) atomicAdd(&bucket_count[(bin << pl.LB) + tid], cnt);
    }
    __syncthreads();
  }
}

// bucket_start[(nbins << LB) + 1]: one thread per bucket, one block = 256 >> LB whole bins.
// (A thread per bin walking its 2^LB counts serially was a 50 us latency chain with strided loads.)
__global__ __launch_bounds__(256) void k_bucketscan(GroupPlan pl, const u32* __restrict__ bin_start,
                                                    const u32* __restrict__ bucket_count,
                                                    u32* __restrict__ bucket_start) {
  __shared__ u32 ex[256];
  __shared__ u32 wsum[4];
  const u32 tid = threadIdx.x;
  const u32 key = blockIdx.x * 256 + tid;
  const u32 nkeys = pl.nbins << pl.LB;
  u32 cnt = key < nkeys ? bucket_count[key] : 0u;
  u32 total;
  u32 e = block_excl_scan_256(cnt, &total, wsum);
  ex[tid] = e;
  __syncthreads();
  if (key < nkeys) {
    u32 first = tid & ~((1u << pl.LB) - 1u);          // first bucket of this thread's bin inside the block
    const u32 b0 = bin_start[key >> pl.LB];
    if (!(pl.bin_cap && bin_start[(key >> pl.LB) + 1] - b0 <= pl.bin_cap))   // (k_binsort wrote the starts of the bins it sorted)
      bucket_start[key] = b0 + (e - ex[first]);
  }
  if (key == 0) bucket_start[nkeys] = bin_start[pl.nbins];
}

__global__ __launch_bounds__(256) void k_scatter2(GroupPlan pl, const u32* __restrict__ entries,
                                                  const uint4* __restrict__ tile_info, const u32* __restrict__ bucket_start,
                                                  u32* __restrict__ bucket_cursor, u32* __restrict__ sorted) {
  __shared__ u32 hist[256];
  __shared__ u32 lstart[257];
  __shared__ u32 delta[256];          // global position of a bucket's run minus its position in the staging buffer
  __shared__ u32 wsum[4];
  __shared__ u32 stage[STAGE2];       // the pass-1 entry itself: its local-bucket bits say which run it belongs to
  const u32 tid = threadIdx.x;
  for (u32 t = blockIdx.x; t < pl.max_tiles; t += gridDim.x) {     // (see k_count2)
    u32 bin, off, end;
    if (!locate_tile(tile_info, t, bin, off, end)) return;
    hist[tid] = 0;
    __syncthreads();
    constexpr int PER = STAGE2 / 256;    // T2 <= STAGE2
    u32 e[PER], rk[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) { u32 i = off + tid + 256u * k; e[k] = i < end ? entries[i] : 0xffffffffu; }
#pragma unroll
    for (int k = 0; k < PER; k++) {      // the histogram atomic also hands out the rank inside (tile, bucket)
      u32 i = off + tid + 256u * k;
      rk[k] = i < end ? atomicAdd(&hist[(e[k] >> 24) & 127u], 1u) : 0u;
    }
    __syncthreads();
    u32 cnt = tid < (1u << pl.LB) ? hist[tid] : 0u;
    u32 total;
    u32 o = block_excl_scan_256(cnt, &total, wsum);
    lstart[tid] = o;
    if (tid == 255) lstart[256] = total;
    u32 key = (bin << pl.LB) + tid;
    delta[tid] = cnt ? bucket_start[key] + atomicAdd(&bucket_cursor[key], cnt) - o : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
      u32 i = off + tid + 256u * k;
      if (i < end) {
        u32 l = (e[k] >> 24) & 127u;
        stage[lstart[l] + rk[k]] = e[k];
      }
    }
    __syncthreads();
    for (u32 q = tid; q < total; q += 256) {
      u32 v = stage[q];
      sorted[delta[(v >> 24) & 127u] + q] = v & 0x80ffffffu;
    }
    __syncthreads();
  }
}

// pass 2 in one launch for every bin that fits LDS (the common case: 2^15 entries per bin at 2^24 points and 17-bit
// windows): ONE 1024-thread block loads the whole bin into registers, ranks every entry inside its bucket with the LDS
// histogram atomic, scans the 2^LB counts, stages the bin bucket-sorted in LDS and streams it out -- no count pass, no
// global atomics, no tile table, and the bucket starts fall out of the block's own scan.  Bins over pl.bin_cap entries
// (skewed scalars) are left to the tiled kernels above, whose tile table then lists those bins only.
static const u32 BIN_CAP = 36864;    // 144 KiB of the CU's 160 KiB LDS
static const u32 BIN_CAP_SMALL = 8192;   // the variant for short bins (below 2^22 points): 32 KiB, four blocks per CU, 8 entries per thread instead of 36 mostly idle slots
template <u32 CAP>
__global__ __launch_bounds__(1024) void k_binsort(GroupPlan pl, const u32* __restrict__ entries, const u32* __restrict__ bin_start,
                                                  u32* __restrict__ sorted, u32* __restrict__ bucket_start) {
  __shared__ u32 stage[CAP];
  __shared__ u32 hist[256];
  __shared__ u32 lstart[256];
  __shared__ u32 wsum[4];
  const u32 tid = threadIdx.x, bin = blockIdx.x;
  const u32 off = bin_start[bin], end = bin_start[bin + 1], cnt = end - off;
  if (cnt > pl.bin_cap) return;
  const u32 nl = 1u << pl.LB, lmask = nl - 1u;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  constexpr int PER = CAP / 1024;
  u32 e[PER], rk[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) { u32 i = off + tid + 1024u * k; e[k] = i < end ? entries[i] : 0u; }
#pragma unroll
  for (int k = 0; k < PER; k++) {
    u32 i = off + tid + 1024u * k;
    rk[k] = i < end ? atomicAdd(&hist[(e[k] >> 24) & lmask], 1u) : 0u;
  }
  __syncthreads();
  if (tid < 256) {     // (whole waves: block_excl_scan_256 has barriers, so the other waves wait below)
    u32 c = tid < nl ? hist[tid] : 0u, total;
    // scan across the first four waves only: shuffles inside a wave, wave totals through LDS with a flag-free
    // hand-off (the fifth barrier below orders it for everyone else)
    const u32 lane = tid & 63, wave = tid >> 6;
    u32 x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 y = __shfl_up(x, o); if (lane >= (u32)o) x += y; }
    if (lane == 63) wsum[wave] = x;
    lstart[tid] = x - c;     // exclusive inside the wave; the wave base is added by the readers
    (void)total;
  }
  __syncthreads();
  const u32 wb1 = wsum[0], wb2 = wb1 + wsum[1], wb3 = wb2 + wsum[2];
  if (tid < nl) {
    u32 l = tid, base = l < 64 ? 0u : (l < 128 ? wb1 : (l < 192 ? wb2 : wb3));
    bucket_start[(bin << pl.LB) + l] = off + base + lstart[l];
  }
#pragma unroll
  for (int k = 0; k < PER; k++) {
    u32 i = off + tid + 1024u * k;
    if (i < end) {
      u32 l = (e[k] >> 24) & lmask, base = l < 64 ? 0u : (l < 128 ? wb1 : (l < 192 ? wb2 : wb3));
      stage[base + lstart[l] + rk[k]] = e[k] & 0x80ffffffu;
    }
  }
  __syncthreads();
  for (u32 q = tid; q < cnt; q += 1024) sorted[off + q] = stage[q];
}

// copy single points (task results that are already final, e.g. U_{L-1} = A^{L-1}[1])
// (struct CopyTask: plan.h)
__global__ void k_copy_points(const CopyTask* __restrict__ tasks, u32 ntasks, u32 nwin, u32 pt_bytes, char* __restrict__ arena) {
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (task, window, 16-byte word)
  u32 wpp = pt_bytes / 16;
  u32 word = gid % wpp; u32 r = gid / wpp;
  u32 ti = r / nwin, w = r - ti * nwin;
  if (ti >= ntasks) return;
  CopyTask tk = tasks[ti];
  uint4 v = make_uint4(0, 0, 0, 0);
  if (tk.src_idx < tk.src_valid_idx)
    v = reinterpret_cast<const uint4*>(arena + ((size_t)tk.src_off + (size_t)w * tk.src_wstride + tk.src_idx) * pt_bytes)[word];
  reinterpret_cast<uint4*>(arena + ((size_t)tk.dst_off + (size_t)w * tk.dst_wstride) * pt_bytes)[word] = v;
}

// ------------------------------------------------------------------------------------
// negabase decomposition (src/negbase_utils.rs:20-36) of scalars < isqrt(order)+2
// (range assert of src/argument_witness_calc.rs:97: first offending index -> err[0] via atomicMin).
// x is held as sign + 128-bit magnitude in 8 x 16-bit half limbs; one step is
//   digit = x mod B (non-negative), x <- -((x - digit)/B).
// Output: digits[j*d + i] (scalar-major, the C ABI layout) and/or digitsT[i*n + j].
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_negbase_digits(const uint4* __restrict__ scalars, u32 n, u32 base, u32 d,
                                                        const u32* __restrict__ bound /*8 limbs*/, int check_range,
                                                        uint8_t* __restrict__ digits, uint8_t* __restrict__ digitsT,
                                                        u32* __restrict__ err, u32 row_begin, u32 row_end /* rows of digitsT to write:
                                                        a digit-position-sharded rank stores only the rows it will sort */,
                                                        const uint8_t* __restrict__ negative = nullptr /* optional: scalar j is -|s_j| (negbase_decompose
                                                        takes a signed BigInt; prepare_scalar_witness passes it through) */,
                                                        uint8_t* __restrict__ trunc = nullptr /* optional: 1 where the expansion needed more than d digits */) {
  u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint4 a = scalars[2 * (size_t)j], b = scalars[2 * (size_t)j + 1];
  u32 s[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  // s < bound ?
  u32 bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { u32 dummy = __builtin_subc(s[i], bound[i], bw, &bw); (void)dummy; }
  bool in_range = bw != 0;
  if (check_range && !in_range) { atomicMin(&err[0], j); }
  // magnitude in 16-bit halves (values beyond 128 bits only occur for out-of-range input; the
  // generic API path (check_range = 0) handles full 256-bit values with 16 halves)
  u32 h[16];
#pragma unroll
  for (int i = 0; i < 8; i++) { h[2 * i] = s[i] & 0xffffu; h[2 * i + 1] = s[i] >> 16; }
  bool neg = negative ? negative[j] != 0 : false;
  {   // -0 is 0
    u32 nz0 = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) nz0 |= h[k];
    if (!nz0) neg = false;
  }
  const float rb = 1.0f / (float)base;
  const u32 shift = (base & (base - 1u)) == 0 ? (u32)__builtin_ctz(base) : 0u;
  // Power-of-two base and a magnitude below 2^128 (every in-range scalar; B = 16 is the bench configuration): the same
  // recurrence on four 32-bit words -- a funnel shift per word instead of sixteen 16-bit halves, ~4x fewer instructions
  // (the kernel is ALU-bound: 75 -> ~20 us per 2^20 scalars).  The generic loop below then has nothing left to do.
  u32 i_begin = 0;
  if (shift && !(s[4] | s[5] | s[6] | s[7])) {
    u32 m0 = s[0], m1 = s[1], m2 = s[2], m3 = s[3];
    for (u32 i = 0; i < d; i++) {
      const u32 rem = m0 & (base - 1u);
      m0 = __builtin_amdgcn_alignbit(m1, m0, shift); m1 = __builtin_amdgcn_alignbit(m2, m1, shift);
      m2 = __builtin_amdgcn_alignbit(m3, m2, shift); m3 >>= shift;
      u32 digit;
      if (!neg) { digit = rem; neg = true; }
      else {
        digit = rem ? base - rem : 0u;
        if (rem) {   // |x| <- q + 1
          m0 += 1u; const u32 c0 = m0 == 0u; m1 += c0; const u32 c1 = c0 & (m1 == 0u); m2 += c1; const u32 c2 = c1 & (m2 == 0u); m3 += c2;
        }
        neg = false;
      }
      if (!(m0 | m1 | m2 | m3)) neg = false;
      if (digits) digits[(size_t)j * d + i] = (uint8_t)digit;
      if (digitsT && i >= row_begin && i < row_end) digitsT[(size_t)i * n + j] = (uint8_t)digit;
    }
    h[0] = m0 & 0xffffu; h[1] = m0 >> 16; h[2] = m1 & 0xffffu; h[3] = m1 >> 16; h[4] = m2 & 0xffffu; h[5] = m2 >> 16; h[6] = m3 & 0xffffu; h[7] = m3 >> 16;
    i_begin = d;
  }
  for (u32 i = i_begin; i < d; i++) {
    // (q, rem) = divmod(|x|, base)
    u32 rem = 0;
    if (shift) {            // power-of-two base (the bench configuration, B = 16): a 2^shift-bit right shift
      rem = h[0] & (base - 1u);
#pragma unroll
      for (int k = 0; k < 16; k++) {
        u32 nxt = k < 15 ? h[k + 1] : 0u;
        h[k] = ((h[k] | (nxt << 16)) >> shift) & 0xffffu;
      }
    } else {                // most significant half first; partial dividends < 2^24 are exact in fp32
#pragma unroll
      for (int k = 15; k >= 0; k--) {
        u32 cur = (rem << 16) | h[k];
        u32 q = (u32)((float)cur * rb);
        int r = (int)cur - (int)(q * base);
        if (r < 0) { q--; r += (int)base; }
        if (r >= (int)base) { q++; r -= (int)base; }
        h[k] = q; rem = (u32)r;
      }
    }
    u32 digit;
    if (!neg) { digit = rem; neg = true; }
    else {
      digit = rem ? base - rem : 0u;
      if (rem) {   // |x| <- q + 1
        u32 cy = 1;
#pragma unroll
        for (int k = 0; k < 16; k++) { u32 v = h[k] + cy; h[k] = v & 0xffffu; cy = v >> 16; }
      }
      neg = false;
    }
    u32 nz = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) nz |= h[k];
    if (!nz) neg = false;
    if (digits) digits[(size_t)j * d + i] = (uint8_t)digit;
    if (digitsT && i >= row_begin && i < row_end) digitsT[(size_t)i * n + j] = (uint8_t)digit;
  }
  // digits beyond d are truncated exactly like chain(repeat(0)).take(d) at
  // src/argument_witness_calc.rs:99; err[1] counts scalars whose expansion did not fit
  u32 nz = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) nz |= h[k];
  if (nz) { atomicAdd(&err[1], 1u); atomicMin(&err[2], j); }
  if (trunc) trunc[j] = nz ? 1 : 0;
}

// ------------------------------------------------------------------------------------
// prepare_scalar_witness (src/negbase_utils.rs:79-124), one thread per scalar, EXACTLY in the reference's order of
// operations so that the first statement that would panic there (debug build: assert :81, index out of bounds
// :98-101, i128 / u32 arithmetic overflow) is the one reported here.  Output per scalar: base x (num_limbs+1) entries
// of 24 bytes {i128 value (two's complement LE), u32 mask, u32 kind}; kind 0 = Entry::Scalar (value = the scalar's low
// 128 bits, sign applied), 1 = Entry::Bucket(value), 2 = Entry::Limb(value, mask).
// powtab[e] = (-base)^e as {lo, hi (i128 halves), overflow flag, pad}: 24 bytes, e < max(d, logtable).
// fail word: atomicMin of (j << 4 | code), code 1 = too many digits (:81), 2 = index out of bounds, 3 = overflow.
// ------------------------------------------------------------------------------------
struct WitnessPow { unsigned long long lo; long long hi; u32 ovf; u32 pad; };

__device__ __forceinline__ bool add_i128(unsigned long long& lo, long long& hi, unsigned long long blo, long long bhi) {
  unsigned long long rlo = lo + blo;
  long long rhi = (long long)((unsigned long long)hi + (unsigned long long)bhi + (rlo < lo ? 1ull : 0ull));
  bool ovf = ((hi ^ rhi) & (bhi ^ rhi)) < 0;     // operands of one sign, result of the other
  lo = rlo; hi = rhi;
  return ovf;
}

__global__ __launch_bounds__(256) void k_scalar_witness(const uint4* __restrict__ scalars, const uint8_t* __restrict__ negative,
                                                        const uint8_t* __restrict__ digits /* n x d, LSB first */,
                                                        const uint8_t* __restrict__ trunc, u32 n, u32 base, u32 d, u32 logtable,
                                                        u32 num_limbs, const WitnessPow* __restrict__ powtab,
                                                        char* __restrict__ out, unsigned long long* __restrict__ fail) {
  const u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const u32 cols = num_limbs + 1;
  char* cells = out + (size_t)j * base * cols * 24;
  auto cell = [&](u32 r, u32 c) -> char* { return cells + ((size_t)r * cols + c) * 24; };
  for (u32 r = 0; r < base; r++)
    for (u32 c = 0; c < cols; c++) {
      u32* w = reinterpret_cast<u32*>(cell(r, c));
      w[0] = w[1] = w[2] = w[3] = w[4] = 0;
      w[5] = (c == 0) ? (r == 0 ? 0u : 1u) : 2u;
    }
  auto report = [&](u32 code) { atomicMin(fail, ((unsigned long long)j << 4) | code); };
  if (trunc[j]) { report(1); return; }                                    // assert!(digits.len() <= num_digits)  :81
  auto add_value = [&](char* c, const WitnessPow& pw) -> bool {
    unsigned long long* v = reinterpret_cast<unsigned long long*>(c);
    unsigned long long lo = v[0]; long long hi = (long long)v[1];
    if (pw.ovf) return true;                                              // pow(-(base as i128), e) itself overflows
    bool o = add_i128(lo, hi, pw.lo, pw.hi);
    v[0] = lo; v[1] = (unsigned long long)hi;
    return o;
  };
  auto add_mask = [&](char* c, u32 k) -> bool {
    if (k >= 32) return true;                                             // pow(2 as u32, k) overflows
    u32* m = reinterpret_cast<u32*>(c + 16);
    u32 a = *m, r = a + (1u << k);
    *m = r;
    return r < a;
  };
  for (u32 i = 0; i < d; i++) {
    u32 dig = digits[(size_t)j * d + i];
    if (!dig) continue;                                                   // id_by_digit: None  :95
    const u32 id = dig - 1, k = i % logtable;
    // (Rust evaluates the right operand of a primitive `+=` before the place expression: pow(..) panics before the index does)
    if (add_value(cell(id + 1, 0), powtab[i])) { report(3); return; }     // ret[id+1][0].0 += pow(-(base), i)        :97
    if (powtab[k].ovf) { report(3); return; }                             // pow(-(base), i%logtable)                 :98
    if (k + 1 > num_limbs) { report(2); return; }                         // ret[id+1][i%logtable + 1]                :98
    if (add_value(cell(id + 1, k + 1), powtab[k])) { report(3); return; } //   .0 += ..                               :98
    if (add_mask(cell(id + 1, k + 1), k)) { report(3); return; }          //   .1 += pow(2u32, i%logtable)            :99
    if (add_value(cell(0, k + 1), powtab[k])) { report(3); return; }      // ret[0][i%logtable+1].0 += ...            :100
    if (add_mask(cell(0, k + 1), k)) { report(3); return; }               //   .1 += ...                              :101
  }
  // Entry::Scalar(sc): the scalar's low 128 bits with its sign (informational; the caller owns the scalar)
  {
    uint4 a = scalars[2 * (size_t)j];
    unsigned long long lo = ((unsigned long long)a.y << 32) | a.x, hi = ((unsigned long long)a.w << 32) | a.z;
    if (negative && negative[j]) { lo = ~lo + 1ull; hi = ~hi + (lo == 0 ? 1ull : 0ull); }
    unsigned long long* v = reinterpret_cast<unsigned long long*>(cell(0, 0));
    v[0] = lo; v[1] = hi;
  }
}

// table_entry_by_id (src/negbase_utils.rs:58-77) for ids [id0, id0 + count): with b = -base in the field,
// acc = 0; for the bits of id, most significant first: { if bit: acc += 1; acc *= b }  -- i.e. sum over the set bits k of
// id of (-base)^(k+1) (the reference multiplies once more than a plain Horner would, and that is what is computed).
template <class F>
__global__ __launch_bounds__(256) void k_table_entries(u32 base, unsigned long long id0, u32 count, uint4* __restrict__ out) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= count) return;
  typedef typename F::fe fe;
  unsigned long long id = id0 + t;
  fe acc, one, b, bm;
  F::set_zero(acc); F::set_one(one);
  // b = -base in Montgomery form: base * one summed by doubling, then negated
  F::set_zero(bm);
  for (int bit = 7; bit >= 0; bit--) { F::add(bm, bm, bm); if ((base >> bit) & 1u) F::add(bm, bm, one); }
  F::neg(b, bm);
  int l = 64 - (id ? __clzll((long long)id) : 64);
  for (int i = l - 1; i >= 0; i--) {
    if ((id >> i) & 1ull) F::add(acc, acc, one);
    F::mul(acc, acc, b);
  }
  F::store(out + 2 * (size_t)t, acc);
}

// ------------------------------------------------------------------------------------
// Jacobian (X,Y,Z) -> affine, batched inversion (Montgomery's trick) over KB points per thread,
// prefix products parked in a global scratch laid out [k][thread].  Z == 0 -> (0,0).
// The reference hands arbitrary-Z projective points to compute_lhs_witness
// (src/regular_functions_utils.rs:447-451 gen_random_pt; src/argument_witness_calc.rs:87).
// ------------------------------------------------------------------------------------
template <class F, int KB>
__global__ __launch_bounds__(256) void k_jac_to_affine(const uint4* __restrict__ jac, u32 n,
                                                       uint4* __restrict__ aff, char* __restrict__ scratch) {
  typedef typename F::fe fe;
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  const u32 nthreads = gridDim.x * 256;
  const u64 j0 = (u64)t * KB;
  if (j0 >= n) return;
  const u32 cnt = (u32)min((u64)KB, (u64)n - j0);
  fe acc; F::set_one(acc);
  for (u32 k = 0; k < cnt; k++) {
    fe z; F::load(z, jac + (j0 + k) * 6 + 4);
    F::store(scratch + ((size_t)k * nthreads + t) * 32, acc);
    if (!F::is_zero(z)) F::mul(acc, acc, z);
  }
  fe inv; inv_via_lazy<F>(inv, acc);
  for (u32 k = cnt; k-- > 0;) {
    fe x, y, z, pref;
    const uint4* p = jac + (j0 + k) * 6;
    F::load(z, p + 4);
    uint4* o = aff + (j0 + k) * 4;
    if (F::is_zero(z)) {
      o[0] = make_uint4(0, 0, 0, 0); o[1] = o[0]; o[2] = o[0]; o[3] = o[0];
      continue;
    }
    F::load(x, p); F::load(y, p + 2);
    F::load(pref, scratch + ((size_t)k * nthreads + t) * 32);
    fe zi, zi2, zi3;
    F::mul(zi, inv, pref);
    F::mul(inv, inv, z);
    F::sqr(zi2, zi); F::mul(zi3, zi2, zi);
    F::mul(x, x, zi2); F::mul(y, y, zi3);
    F::store(o, x); F::store(o + 2, y);
  }
}

// ------------------------------------------------------------------------------------
// debug / KAT kernels (exposed through the C ABI for the parity tests)
// ------------------------------------------------------------------------------------
template <class F>
__global__ void k_dbg_montmul(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename F::fe x, y, r;
  F::load(x, a + 2 * (size_t)i); F::load(y, b + 2 * (size_t)i);
  F::mul(r, x, y);
  F::store(out + 2 * (size_t)i, r);
}
// op 0: add, 1: sub, 2: neg(a), 3: inverse(a), 4: sqr(a)
template <class F>
__global__ void k_dbg_fieldop(int op, const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ out, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename F::fe x, y, r;
  F::load(x, a + 2 * (size_t)i); F::load(y, b + 2 * (size_t)i);
  if (op == 0) F::add(r, x, y);
  else if (op == 1) F::sub(r, x, y);
  else if (op == 2) F::neg(r, x);
  else if (op == 3) F::inv(r, x);
  else F::sqr(r, x);
  F::store(out + 2 * (size_t)i, r);
}
// out_xyzz[i] = (acc_xyzz[i] + sign*aff[i]) via madd (op 0) or acc_xyzz[i] + q_xyzz[i] via add (op 1)
template <class F>
__global__ void k_dbg_pointop(int op, const char* __restrict__ acc_in, const char* __restrict__ q_in, char* __restrict__ out, u32 n) {
  typedef XYZZ<F> G;
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typename G::pt acc; G::load(acc, acc_in + (size_t)i * 128);
  if (op == 0) {
    typename F::fe x, y; F::load(x, q_in + (size_t)i * 64); F::load(y, q_in + (size_t)i * 64 + 32);
    if (!(F::is_zero(x) && F::is_zero(y))) G::madd(acc, x, y);
  } else {
    typename G::pt q; G::load(q, q_in + (size_t)i * 128);
    G::add(acc, q);
  }
  G::store(out + (size_t)i * 128, acc);
}

}  // namespace lemsm
