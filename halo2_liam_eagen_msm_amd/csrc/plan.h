// Launch plan shared by host code and kernels (plain C structs, passed by value).
#pragma once
#include <stdint.h>

namespace lemsm {

// Maximum coarse bins per pass-1 launch (LDS histogram: 2 x 4096 x 4 B = 32 KiB).
static const uint32_t MAX_BINS = 8192;
// Local (within-bin) bucket bits carried in a pass-1 entry.
static const uint32_t MAX_LB = 7;
// Points per slab: a pass-1 entry packs idx(24) | local(7) | sign(1).
static const uint32_t MAX_SLAB_LOG = 24;

// One "window group": the windows [w0,w1) of one MSM that are sorted and
// accumulated together.  "Window" is a Pippenger window for the full-width
// path and a negabase digit position for the lhs path.
struct GroupPlan {
  uint32_t n;        // points in this slab
  uint32_t c;        // window bits (Pippenger path; 0 on the negabase path)
  uint32_t nb;       // buckets per window: 2^(c-1) or B-1
  uint32_t nbw;      // key stride of a window: nb rounded up to a multiple of 2^LB  (= BW << LB)
  uint32_t BW;       // coarse bins per window (<= 256)
  uint32_t w0, w1;   // windows of this group
  uint32_t W;        // total windows of the MSM (top window is unsigned)
  uint32_t NB;       // (w1-w0)*nbw keys in this group
  uint32_t LB;       // local bucket bits
  uint32_t nbins;    // (w1-w0)*BW <= MAX_BINS
  uint32_t spb;      // scalars per pass-1 block (<= STAGE)
  uint32_t nblk1;    // pass-1 blocks = ceil(n / spb)
  uint32_t T2;       // entries per pass-2 tile
  uint32_t max_tiles;// upper bound on pass-2 tiles
  uint32_t bin_cap;  // pass 2: bins of at most this many entries are sorted whole by ONE block (k_binsort); 0 = every bin goes through the tiled path
  uint32_t L1;       // entries per thread in the accumulate kernel
  uint32_t nthr1;    // upper bound on accumulate threads = ceil(n*(w1-w0) / L1)
  uint32_t d;        // negabase: digits per scalar (number of rows of the position-major digit matrix)
  uint32_t dstride;  // Pippenger: stride of a u16 digit column = n rounded up to 64 (columns start 16-byte aligned for the vector loads of pass 1)
  uint32_t nstride;  // negabase: row stride of the digit matrix = points of the WHOLE call (a slab sees a column range of it)
};

// Bucket-reduction pyramid: one task of one step (kernels_ec.cuh: k_pyramid), plain data shared by the host's plan
// builder (hosttail.hpp) and the kernels.  dst[i] = src[(2i) stride + phase] + src[(2i+1) stride + phase], i < count, for
// every window; arena offsets are in points; src indices >= src_valid read as the identity.
struct PyrTask {
  uint32_t src_off, src_wstride;   // per-window base = src_off + w * src_wstride
  uint32_t dst_off, dst_wstride;
  uint32_t stride, phase, count, src_valid;
  uint32_t src_scaled, pad_[3];    // source is bucket_sum[] in the scaled form: convert on load
};
// copy of a single point per window (task results that are already final, e.g. U_{L-1} = A^{L-1}[1])
struct CopyTask { uint32_t src_off, src_wstride, dst_off, dst_wstride, src_valid_idx, src_idx; };

// Edge-record merge queues (kernels_ec.cuh): item offsets and capacities of one window group
struct MqLayout { uint32_t offS, offM, offL, offF, capS, capM, capL, capF, capP, slice, wave_th; };

}  // namespace lemsm
