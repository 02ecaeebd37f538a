// Divisor witness on the GPU: compute_divisor_witness / Propagation::group_merge of the reference
// (/root/reference/src/regular_functions_utils.rs:222-273 RegularFunction, :285-303 linefunc, :311-405 Propagation,
// :453-480 compute_divisor_witness{_partial}; the polynomial products of :54-62 / :102-129 / :209-216).
//
// Shape of the computation.  The reference pairs the points up (from_pair, :457-463) and merges neighbours level by
// level (group_merge, :380-405): node k of level l+1 = merge(node 2k, node 2k+1 of level l), a lone last node passes
// through.  That is a perfect binary tree over the index range with a ragged right edge, so every level is ONE batch:
//   * the nodes' outputs (points) are added in XYZZ and brought to affine with one batched inversion per level;
//   * merge (:333-360) is, per node, numerator = L.w * (R.w * line(-L.out, -R.out)) divided twice by (x - Lx), (x - Rx)
//     (kate_div: exact synthetic division), or the plain product L.w * R.w when an output is the identity.
//     Both are POINTWISE in an evaluation domain: with a(x) + y b(x) evaluated at x_i = omega^i,
//       (a1 + y b1)(a2 + y b2) = a1 a2 + b1 b2 (x^3 + B) + y (a1 b2 + b1 a2)          (:266-273, y^2 = x^3 + B)
//     and the division by (x - c) is a division of both parts by (x_i - c).  So one level = 4 forward NTTs per node
//     (L.a, L.b, R.a, R.b), one pointwise kernel (with a batched inversion of the denominators), 2 inverse NTTs.
//     The transform size N only has to hold the QUOTIENT (the numerator is never interpolated).  The domain is a COSET
//     g omega^i (g = a power of 7, the field's multiplicative generator): x_i - c = 0 would need c to be g times a
//     root of unity -- with g = 1 the Grumpkin generator itself (x = 1) would hit it; a zero denominator is detected
//     and the level redone with the next power of 7.  Exact arithmetic:
//     the coefficients are those of the reference's mul_naive / mul_fft / kate_div, whatever the algorithm.
//   * polynomial LENGTHS follow the reference's bookkeeping exactly (trailing zeros it carries are carried here):
//     len(p*q) = len p + len q - 1 (:55, :104), len(p+q) = max (:183), kate_div: len - 1; two empty operands are the
//     reference's usize underflow at :55 (a panic) and are reported, not computed.
//
// A witness is defined up to a non-zero scalar: linefunc (:285-303) builds its line from PROJECTIVE coordinates of
// whatever Jacobian representative a point has (:426-431); here lines are built from affine coordinates.  The C ABI
// offers the normalised form (leading coefficient by pole order = 1), which is representation-independent.
//
// Field: bn256::Fr = the base field of Grumpkin, the only field the reference implements FftPrecomp for
// (src/precomputed_fft_data.rs:3; compute_lhs_witness is bounded by it, src/argument_witness_calc.rs:87).  Strict
// 8 x 32-bit Montgomery arithmetic (field32.cuh), 32-byte canonical storage = the C ABI's raw Montgomery limbs.
#pragma once
#include "xyzz.cuh"
#include "field29.cuh"
#include "inv29.cuh"

namespace lemsm {
namespace dw {

typedef Field32<FrParams> F;
typedef XYZZ<F> G;
typedef F::fe fe;

enum { MODE_PASS = 0, MODE_PRODUCT = 1, MODE_DIVIDE = 2 };
enum { STAT_MAXLEN = 0, STAT_PANIC = 1, STAT_ZERODEN = 2, STAT_MAXCHILD = 3, STAT_ZERO0 = 4, STAT_MAXPASS = 5, STAT_WORDS = 8 };

// what k_plan decides for one node of the next level
struct Plan {
  u32 mode, la, lb, child0;  // child0: global index (level below) of the left child; the right one is child0 + 1
  u32 evk[2];                // reuse mode (child-evaluation reuse, below): child 0 / 1's values on the even half of this level's domain
                             // come from the level below's transform buffer (0: sequence = the child's node index) or from the
                             // side buffer of passed-through children (1: sequence pair of this node's tree)
  u32 tree;                  // tree index of this node
  u32 c0[8], c1[8], d0[8];   // line(-L.out, -R.out) = (c0 + c1 x) + y d0      (from_line(lx, ly, lz): a = [lz, lx], b = [ly], :244-246)
  u32 lX[8], lZZ[8], rX[8], rZZ[8];   // L.out, R.out as X / ZZ: the two kate_div points (:351-357) divide by (x - X/ZZ); here by
                                      // (ZZ x - X), a scalar multiple -- the witness is only defined up to a scalar anyway
};

__device__ __forceinline__ void ld(fe& r, const u32* p) {
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = p[i];
}
__device__ __forceinline__ void st(u32* p, const fe& a) {
#pragma unroll
  for (int i = 0; i < 8; i++) p[i] = a.v[i];
}
__device__ __forceinline__ bool aff_id(const fe& x, const fe& y) { return F::is_zero(x) && F::is_zero(y); }

// a^-1 by Fermat in the lazy 29-bit field (inv29.cuh)
__device__ __forceinline__ void inv_fast(fe& r, const fe& a) { inv_lazy<FrParams, Fr29Params>(r, a); }

// line through two affine, non-identity points P = (x1,y1), Q = (x2,y2): lx x + ly y + lz with
// lx = y1 - y2, ly = x2 - x1, lz = x1 y2 - y1 x2  (linefunc :290-292 with z = 1); all zero iff P == Q.
__device__ __forceinline__ bool line_through(fe& lx, fe& ly, fe& lz, const fe& x1, const fe& y1, const fe& x2, const fe& y2) {
  fe t, u;
  F::sub(lx, y1, y2); F::sub(ly, x2, x1);
  F::mul(t, x1, y2); F::mul(u, y1, x2); F::sub(lz, t, u);
  return !(F::is_zero(lx) && F::is_zero(ly) && F::is_zero(lz));
}

// the same through two points in homogeneous coordinates (X : Y : Z) -- linefunc :290-292 literally
struct Hom { fe x, y, z; };
__device__ __forceinline__ void hom_of(Hom& h, const G::pt& p) {          // x = X/ZZ, y = Y/ZZZ  ->  (X ZZZ : Y ZZ : ZZ ZZZ)
  F::mul(h.x, p.x, p.zzz); F::mul(h.y, p.y, p.zz); F::mul(h.z, p.zz, p.zzz);
}
__device__ __forceinline__ bool line_through_hom(fe& lx, fe& ly, fe& lz, const Hom& a, const Hom& b) {
  fe t, u;
  F::mul(t, a.x, b.y); F::mul(u, a.y, b.x); F::sub(lz, t, u);
  F::mul(t, a.y, b.z); F::mul(u, a.z, b.y); F::sub(lx, t, u);
  F::mul(t, a.z, b.x); F::mul(u, a.x, b.z); F::sub(ly, t, u);
  return !(F::is_zero(lx) && F::is_zero(ly) && F::is_zero(lz));
}

// ---------------------------------------------------------------------------------------------------------
// A FOREST of T independent trees runs as one batch (compute_lhs_witness builds d of them, one per digit position:
// one forest costs the launches of one tree).  Tree t owns nodes [off[t], off[t+1]) of a level (points of the
// concatenated input at the leaves); node k of a tree merges its children 2k and 2k+1 of the level below, a lone
// last child passes through, and a tree that is finished (one node) passes through until the tallest one is.
// ---------------------------------------------------------------------------------------------------------
struct Forest {
  const u32* off_child;   // T + 1 offsets of the level below (points, for the leaves)
  const u32* off_node;    // T + 1 offsets of this level
  u32 T;
};
// global node g -> (first child c0 as a global index of the level below, number of children 1 or 2 [0: none])
__device__ __forceinline__ void locate(const Forest& f, u32 g, u32& c0, u32& nch, u32* tree = nullptr) {
  u32 lo = 0, hi = f.T;                      // largest t with off_node[t] <= g
  while (hi - lo > 1) { u32 mid = (lo + hi) >> 1; if (f.off_node[mid] <= g) lo = mid; else hi = mid; }
  if (tree) *tree = lo;
  const u32 k = g - f.off_node[lo];
  const u32 cb = f.off_child[lo], cn = f.off_child[lo + 1] - cb;
  c0 = cb + 2 * k;
  nch = (2 * k + 1 < cn) ? 2u : (2 * k < cn ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------------------
// points: node outputs stay in XYZZ from level to level (no inversion anywhere in the tree)
// ---------------------------------------------------------------------------------------------------------
// leaves: sum[k] = -(pts[2k] + pts[2k+1]) (a lone last point: -pts[2k]) as XYZZ               (:321, :330)
__global__ __launch_bounds__(256) void k_leaf_sum(const uint4* __restrict__ pts, Forest f, u32 nleaf, char* __restrict__ out_xyzz) {
  u32 g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nleaf) return;
  u32 c0, nch; locate(f, g, c0, nch);
  G::pt acc; G::set_identity(acc);
  for (u32 t = 0; t < nch; t++) {
    fe x, y; F::load(x, pts + (size_t)(c0 + t) * 4); F::load(y, pts + (size_t)(c0 + t) * 4 + 2);
    if (!aff_id(x, y)) G::madd(acc, x, y);
  }
  F::neg(acc.y, acc.y);
  G::store(out_xyzz + (size_t)g * 128, acc);
}

// inner nodes: out[k] = child[2k] + child[2k+1] (a lone child passes through)   (:335)
__global__ __launch_bounds__(256) void k_merge_sum(const char* __restrict__ child_xyzz, Forest f, u32 nnodes, char* __restrict__ out_xyzz) {
  u32 g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nnodes) return;
  u32 c0, nch; locate(f, g, c0, nch);
  G::pt acc; G::load(acc, child_xyzz + (size_t)c0 * 128);
  if (nch == 2) { G::pt q; G::load(q, child_xyzz + (size_t)(c0 + 1) * 128); G::add(acc, q); }
  G::store(out_xyzz + (size_t)g * 128, acc);
}

// ---------------------------------------------------------------------------------------------------------
// leaves: the line of every pair (from_pair :328-331, from_point :319-322, empty :324-326)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_leaf_lines(const uint4* __restrict__ pts, Forest f, u32 nleaf, const char* __restrict__ out_xyzz,
                                                    u32* __restrict__ A, u32* __restrict__ B, u32 capA, u32 capB, uint2* __restrict__ lens) {
  u32 k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nleaf) return;
  u32 c0, nch; locate(f, k, c0, nch);
  fe x1, y1, x2, y2, zero, one;
  F::set_zero(zero); F::set_one(one);
  if (nch == 0) {                         // the single leaf of an EMPTY list: (RegularFunction::from_const(ONE), identity)  :455
    st(A + (size_t)k * capA * 8, one); lens[k] = make_uint2(1, 0);
    return;
  }
  F::load(x1, pts + (size_t)c0 * 4); F::load(y1, pts + (size_t)c0 * 4 + 2);
  const bool lone = nch < 2;
  if (lone) { x2 = zero; y2 = zero; } else { F::load(x2, pts + (size_t)(c0 + 1) * 4); F::load(y2, pts + (size_t)(c0 + 1) * 4 + 2); }
  const bool id1 = aff_id(x1, y1), id2 = aff_id(x2, y2);
  u32* a = A + (size_t)k * capA * 8; u32* b = B + (size_t)k * capB * 8;
  fe lx, ly, lz;
  if (id1 && id2) {                       // empty(): wtns = 1, no y part
    st(a, one); lens[k] = make_uint2(1, 0);
    return;
  }
  if (id1 || id2 || lone) {
    // one real point q.  from_point(q) = linefunc(q, -q) (:321): lx = 2 qy, ly = 0, lz = -2 qx qy;
    // from_pair(q, O) = linefunc(q, O) (:330): the vertical line through q as well (lx = -1, ly = 0, lz = qx with O = (0,1,0)).
    // Either way a scalar multiple of (x - qx); the first form is used for both.
    const fe& qx = id1 ? x2 : x1; const fe& qy = id1 ? y2 : y1;
    fe t;
    F::dbl(lx, qy); F::mul(t, qx, lx); F::neg(lz, t); ly = zero;
  } else if (!line_through(lx, ly, lz, x1, y1, x2, y2)) {
    // p1 == p2: the tangent, as the line through p1 and c = -(p1 + p2) = this leaf's output (:298-302)
    G::pt c; G::load(c, out_xyzz + (size_t)k * 128);
    Hom ha, hc; ha.x = x1; ha.y = y1; F::set_one(ha.z); hom_of(hc, c);
    line_through_hom(lx, ly, lz, ha, hc);
  }
  st(a, lz); st(a + 8, lx); st(b, ly);    // from_line(lx, ly, lz): a = [lz, lx], b = [ly]
  lens[k] = make_uint2(2, 1);
}

// ---------------------------------------------------------------------------------------------------------
// plan of one level: mode, line, lengths (the reference's length bookkeeping, see the header comment)
// ---------------------------------------------------------------------------------------------------------
// len(p * q) per mul_naive :55 / mul_fft :104; both empty = usize underflow (panic)
__device__ __forceinline__ u32 plen(u32 a, u32 b, bool& panic) {
  if (a + b == 0) { panic = true; return 0; }
  return a + b - 1;
}
// lengths of (a1 + y b1)(a2 + y b2)  (:266-273: subst_y2 has 4 coefficients)
__device__ __forceinline__ void rf_len(u32& la, u32& lb, u32 a1, u32 b1, u32 a2, u32 b2, bool& panic) {
  u32 aa = plen(a1, a2, panic), bb = plen(b1, b2, panic);
  u32 bbs = plen(bb, 4, panic);
  u32 ab = plen(a1, b2, panic), ba = plen(b1, a2, panic);
  la = max(aa, bbs); lb = max(ab, ba);
}

__global__ __launch_bounds__(256) void k_plan(const char* __restrict__ child_xyzz, const uint2* __restrict__ child_lens, Forest f, u32 nnodes,
                                              const char* __restrict__ node_xyzz, Plan* __restrict__ plan, u32* __restrict__ stats,
                                              const Plan* __restrict__ prevplan /* the level below's plans (null at the first level): a child that was
                                              passed through there has no evaluations on this level's even half */) {
  u32 k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nnodes) return;
  Plan pl;
  u32 c0, nch, tree; locate(f, k, c0, nch, &tree);
  pl.child0 = c0; pl.tree = tree;
  pl.evk[0] = (prevplan && nch >= 1 && prevplan[c0].mode != MODE_PASS) ? 0u : 1u;
  pl.evk[1] = (prevplan && nch >= 2 && prevplan[c0 + 1].mode != MODE_PASS) ? 0u : 1u;
  const u32 L = c0, R = c0 + 1;
  uint2 ll = child_lens[L];
  if (nch < 2) {                                      // MaybePair::Unit: passes through unchanged (:363-366)
    pl.mode = MODE_PASS; pl.la = ll.x; pl.lb = ll.y;
    atomicMax(&stats[STAT_MAXPASS], max(ll.x, ll.y));
    plan[k] = pl;
    return;
  }
  uint2 rl = child_lens[R];
  G::pt lo, ro;
  G::load(lo, child_xyzz + (size_t)L * 128); G::load(ro, child_xyzz + (size_t)R * 128);
  bool panic = false;
  if (G::is_identity(lo) || G::is_identity(ro)) {     // :340-342
    pl.mode = MODE_PRODUCT;
    rf_len(pl.la, pl.lb, ll.x, ll.y, rl.x, rl.y, panic);
  } else {
    pl.mode = MODE_DIVIDE;
    // line through -L.out and -R.out (:344); equal points: through -L.out and c = -((-L.out) + (-R.out)) = this node's output
    Hom ha, hb;
    hom_of(ha, lo); hom_of(hb, ro);
    F::neg(ha.y, ha.y); F::neg(hb.y, hb.y);
    fe lx, ly, lz;
    if (!line_through_hom(lx, ly, lz, ha, hb)) {
      G::pt c; G::load(c, node_xyzz + (size_t)k * 128);
      Hom hc; hom_of(hc, c);
      line_through_hom(lx, ly, lz, ha, hc);
    }
    st(pl.c0, lz); st(pl.c1, lx); st(pl.d0, ly);
    st(pl.lX, lo.x); st(pl.lZZ, lo.zz); st(pl.rX, ro.x); st(pl.rZZ, ro.zz);
    u32 ta, tb, na, nb;
    rf_len(ta, tb, rl.x, rl.y, 2, 1, panic);          // b.wtns * linefunc
    rf_len(na, nb, ll.x, ll.y, ta, tb, panic);        // a.wtns * (..)
    if (na < 2 || nb < 2) panic = true;               // kate_division of an empty vector (len - 1 underflows)
    pl.la = na - 2; pl.lb = nb - 2;                    // two kate_div each (:357)
  }
  plan[k] = pl;
  if (panic) atomicOr(&stats[STAT_PANIC], 1u);
  else { atomicMax(&stats[STAT_MAXLEN], max(pl.la, pl.lb)); atomicMax(&stats[STAT_MAXCHILD], max(max(ll.x, ll.y), max(rl.x, rl.y))); }
}

// ---------------------------------------------------------------------------------------------------------
// NTT over Fr, batched: nseq sequences of N = 2^logN contiguous elements (32 B each).
// W[t] = omega_Nmax^t for t < Nmax/2 (Montgomery form); ws = Nmax / N.
// forward: decimation in frequency, natural order in -> bit-reversed order out;
// inverse: decimation in time with omega^-1, bit-reversed in -> natural out (not scaled by 1/N: the pointwise kernel does).
// One launch per stage, one butterfly per thread (HBM-bound: 128 B moved per butterfly and stage).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void twiddle(fe& w, const u32* __restrict__ W, u32 half_max, u32 t, bool inverse) {
  // omega^t (t < Nmax/2) or omega^-t = -omega^(Nmax/2 - t)
  if (!inverse || t == 0) { ld(w, W + (size_t)t * 8); return; }
  fe p; ld(p, W + (size_t)(half_max - t) * 8); F::neg(w, p);
}

__global__ __launch_bounds__(256) void k_ntt_stage(u32* __restrict__ buf, u32 nseq, u32 logN, u32 logm /* span m = 2^logm */,
                                                   const u32* __restrict__ W, u32 log_half_max, u32 inverse) {
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  const u32 half = 1u << (logN - 1);
  if (gid >= (u64)nseq * half) return;
  const u32 s = (u32)(gid >> (logN - 1)), bfly = (u32)gid & (half - 1);
  const u32 m = 1u << logm, r = bfly & (m - 1);
  const u32 i = ((bfly >> logm) << (logm + 1)) + r, j = i + m;
  u32* base = buf + ((size_t)s << logN) * 8;
  // omega_N^(r * N/(2m)) = omega_Nmax^(r * (N/2m) * (Nmax/N)) = W[r << (log_half_max - logm)]
  const u32 t = r << (log_half_max - logm);
  fe w; twiddle(w, W, 1u << log_half_max, t, inverse != 0);
  fe u, v; ld(u, base + (size_t)i * 8); ld(v, base + (size_t)j * 8);
  fe x, y;
  if (!inverse) { F::add(x, u, v); F::sub(y, u, v); F::mul(y, y, w); }
  else { F::mul(v, v, w); F::add(x, u, v); F::sub(y, u, v); }
  st(base + (size_t)i * 8, x); st(base + (size_t)j * 8, y);
}

// LDS-tiled form: one launch runs S consecutive stages on tiles of 1024 elements held in LDS (32 KB, limb-major so
// that lanes touch consecutive banks), instead of one launch -- one trip through HBM -- per stage.
//   lo == 0 : the tile is 1024 CONTIGUOUS elements and the S = min(logN, 10) stages with spans 2^(S-1)..1 run inside
//             every aligned 2^S chunk of it (for logN < 10 a tile holds several whole sequences);
//   lo  > 0 : stages with spans 2^(lo+S-1)..2^lo: a tile is 2^S positions 2^lo apart, TW = 1024 >> S neighbouring
//             tiles side by side so that every access is a run of TW consecutive elements.
// Forward (DIF) runs the strided passes from the top spans down, then the contiguous pass; inverse (DIT) the reverse.
// What the first forward pass reads and the last inverse pass writes when the level's load / store steps are fused into
// the transforms (k_load / k_store below are the unfused forms and say what each field means).
struct TileIO {
  const u32* cA; const u32* cB; const uint2* child_lens; u32 ccapA, ccapB; const Plan* plan; u32 nnodes; const u32* GP; u32* c0in;      // load side
  u32* nA; u32* nB; u32 capA, capB; uint2* lens; const u32* GI; const u32* c0out; const u32* consts;                                      // store side
  const u32* GO;       // reuse mode: (g omega_N)^i, i < N/2: the coset factors of the ODD half of this level's domain
  const u32* exc;      // reuse mode: per tree, the child (global index of the level below) that was passed through there and is merged here, or ~0
  u32 T;
};
// element g = (q * nnodes + k) * N + i of the forward input: coefficient i of child part q of node k, times g^i (k_load)
__device__ __forceinline__ void tile_load_coeff(fe& v, const TileIO& io, u64 g, u32 logN) {
  const u64 per = (u64)io.nnodes << logN;
  const u32 q = (u32)(g / per); const u64 rem = g - (u64)q * per;
  const u32 k = (u32)(rem >> logN), i = (u32)rem & ((1u << logN) - 1);
  F::set_zero(v);
  if (io.plan[k].mode != MODE_PASS) {
    const u32 c = io.plan[k].child0 + (q >> 1);
    const uint2 cl = io.child_lens[c];
    bool have = false;
    if (q & 1) { if (i < cl.y) { ld(v, io.cB + ((size_t)c * io.ccapB + i) * 8); have = true; } }
    else { if (i < cl.x) { ld(v, io.cA + ((size_t)c * io.ccapA + i) * 8); have = true; } }
    if (have && i) { fe gp; ld(gp, io.GP + (size_t)i * 8); F::mul(v, v, gp); }
  }
  if (io.c0in && i == 0) st(io.c0in + ((size_t)q * io.nnodes + k) * 8, v);
}
// ---- child-evaluation reuse -------------------------------------------------------------------------------------------
// A node's forward transforms evaluate its children's polynomials on the coset g omega_N^j.  The even j are the
// children's OWN domain (g omega_{N/2}^j'), where the level below has just evaluated exactly these polynomials -- its
// quotient values sit in its transform buffer, in the same (bit-reversed) order as the first half of this level's slots.
// So a level only transforms onto the ODD half of its domain: a size-N/2 transform of a_i (g omega_N)^i (a decimation in
// frequency step with a zero upper half: the even half is DIF_{N/2}(a_i g^i), the odd half DIF_{N/2}(a_i (g omega_N)^i)),
// half the butterflies and half the HBM passes of the forward transforms, a third of all transform work of a level.
// A coefficient at index N/2 (wrap-mode children have N/2 + 1) folds onto index 0: +a_{N/2} g^{N/2} on the even half,
// -a_{N/2} g^{N/2} on the odd.  A child that was passed through at the level below (the lone last node of a tree) has no
// values there: its even half is transformed into a side buffer (two sequences per tree, tile_load_even_exc).
// element g = (q * nnodes + k) * (N/2) + i of the odd-half input; logNh = log2(N/2)
__device__ __forceinline__ void tile_load_odd(fe& v, const TileIO& io, u64 g, u32 logNh) {
  const u64 per = (u64)io.nnodes << logNh;
  const u32 q = (u32)(g / per); const u64 rem = g - (u64)q * per;
  const u32 k = (u32)(rem >> logNh), i = (u32)rem & ((1u << logNh) - 1);
  const u32 Nh = 1u << logNh;
  F::set_zero(v);
  if (io.plan[k].mode != MODE_PASS) {
    const u32 c = io.plan[k].child0 + (q >> 1);
    const uint2 cl = io.child_lens[c];
    const u32 len = (q & 1) ? cl.y : cl.x;
    const u32* src = (q & 1) ? io.cB + (size_t)c * io.ccapB * 8 : io.cA + (size_t)c * io.ccapA * 8;
    if (i < len) ld(v, src + (size_t)i * 8);
    if (io.c0in && i == 0) st(io.c0in + ((size_t)q * io.nnodes + k) * 8, v);          // p(0): the true constant term
    if (i == 0) {
      if (len > Nh) { fe top, gn; ld(top, src + (size_t)Nh * 8); ld(gn, io.consts + 56); F::mul(top, top, gn); F::sub(v, v, top); }   // a_0 - a_{N/2} g^{N/2}
    } else if (i < len) { fe go; ld(go, io.GO + (size_t)i * 8); F::mul(v, v, go); }
  } else if (io.c0in && i == 0) st(io.c0in + ((size_t)q * io.nnodes + k) * 8, v);
}
// element g = (2 t + part) * (N/2) + i of the side buffer: the passed-through child of tree t on the EVEN half
__device__ __forceinline__ void tile_load_even_exc(fe& v, const TileIO& io, u64 g, u32 logNh) {
  const u32 s = (u32)(g >> logNh), i = (u32)g & ((1u << logNh) - 1);
  const u32 t = s >> 1, part = s & 1u, Nh = 1u << logNh;
  F::set_zero(v);
  if (t >= io.T) return;
  const u32 c = io.exc[t];
  if (c == 0xffffffffu) return;
  const uint2 cl = io.child_lens[c];
  const u32 len = part ? cl.y : cl.x;
  const u32* src = part ? io.cB + (size_t)c * io.ccapB * 8 : io.cA + (size_t)c * io.ccapA * 8;
  if (i < len) { ld(v, src + (size_t)i * 8); if (i) { fe gp; ld(gp, io.GP + (size_t)i * 8); F::mul(v, v, gp); } }
  if (i == 0 && len > Nh) { fe top, gn; ld(top, src + (size_t)Nh * 8); ld(gn, io.consts + 56); F::mul(top, top, gn); F::add(v, v, top); }
}
// GO[i] = (g omega_N)^i = GP[i] * omega_N^i, i < N/2
__global__ __launch_bounds__(256) void k_odd_factors(u32 logN, const u32* __restrict__ W, u32 log_half_max, const u32* __restrict__ GP, u32* __restrict__ GO) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (1u << (logN - 1))) return;
  fe w, g; ld(w, W + ((size_t)i << (log_half_max + 1 - logN)) * 8); ld(g, GP + (size_t)i * 8);
  F::mul(g, g, w);
  st(GO + (size_t)i * 8, g);
}

// element g = (which * nnodes + k) * N + i of the inverse output -> coefficient i of part `which` of node k (k_store);
// requires every length of the level <= N (N + 1 in wrap mode): the host checks STAT_MAXLEN / STAT_MAXPASS
__device__ __forceinline__ void tile_store_coeff(const fe& t, const TileIO& io, u64 g, u32 logN) {
  const u32 N = 1u << logN;
  const u64 per = (u64)io.nnodes << logN;
  const u32 which = (u32)(g / per); const u64 rem = g - (u64)which * per;
  const u32 k = (u32)(rem >> logN), i = (u32)rem & (N - 1);
  const Plan& pl = io.plan[k];
  if (which == 0 && i == 0) io.lens[k] = make_uint2(pl.la, pl.lb);
  const u32 len = which ? pl.lb : pl.la;
  u32* dst = (which ? io.nB : io.nA) + (size_t)k * (which ? io.capB : io.capA) * 8;
  if (pl.mode == MODE_PASS) {
    if (i < len) { fe v; ld(v, (which ? io.cB + (size_t)pl.child0 * io.ccapB * 8 : io.cA + (size_t)pl.child0 * io.ccapA * 8) + (size_t)i * 8); st(dst + (size_t)i * 8, v); }
    return;
  }
  if (io.c0out && len == N + 1 && i == 0) {          // wrap mode: c_0 = N * (value at 0), c_N = (t_0 - c_0) g^-N
    fe e, nn, c0v, gi, tn; ld(e, io.c0out + ((size_t)which * io.nnodes + k) * 8); ld(nn, io.consts + 24);
    F::mul(c0v, e, nn);
    st(dst, c0v);
    ld(gi, io.consts + 32); F::sub(tn, t, c0v); F::mul(tn, tn, gi);
    st(dst + (size_t)N * 8, tn);
    return;
  }
  if (i < len) {
    fe v = t;
    if (i) { fe gm; ld(gm, io.GI + (size_t)i * 8); F::mul(v, v, gm); }
    st(dst + (size_t)i * 8, v);
  }
}

template <bool INV, int IO /* 0: src -> buf; 1: the input is gathered from the coefficient arrays (fused k_load); 2: the output goes to the coefficient
                              arrays (fused k_store); 3 / 4: reuse mode, the odd half / a passed-through child's even half gathered from the coefficient arrays */>
__global__ __launch_bounds__(256) void k_ntt_tile(u32* __restrict__ buf, u64 total, u32 logN, u32 lo, u32 S,
                                                  const u32* __restrict__ W, u32 log_half_max, TileIO io,
                                                  const u32* __restrict__ src /* what a pass that does not gather reads: buf itself, or -- first pass of an
                                                  out-of-place inverse transform -- the buffer whose values must survive */) {
  __shared__ u32 sm[8][1024];
  const u32 tid = threadIdx.x;
  const u32 TW = lo ? (1024u >> S) : 1u;            // lo > 0: TW tiles side by side; lo == 0: chunks of J inside 1024 contiguous
  const u32 logTW = lo ? (10u - S) : 0u;
  // element e of the tile -> global index
  u64 base; u32 Lfull0 = 0;
  if (lo == 0) base = (u64)blockIdx.x * 1024u;
  else {
    const u32 hi1 = lo + S;                          // a tile spans 2^hi1 positions
    const u32 lgroups = 1u << (lo - logTW);          // groups of TW neighbouring low offsets
    const u64 b = blockIdx.x;
    const u32 Lg = (u32)(b % lgroups); const u64 rest = b / lgroups;   // rest = seq * (N >> hi1) + H
    base = (rest << hi1) + ((u64)Lg << logTW);       // (seq * N + H * 2^hi1) + Lg * TW   [N is a multiple of 2^hi1]
    Lfull0 = Lg << logTW;
  }
  auto gidx = [&](u32 e) -> u64 { return lo == 0 ? base + e : base + ((u64)(e >> logTW) << lo) + (e & (TW - 1)); };
  // load
#pragma unroll
  for (u32 q = 0; q < 4; q++) {
    const u32 e = tid + 256u * q;
    const u64 g = gidx(e);
    fe v; F::set_zero(v);
    if (g < total) {
      if (IO == 1) tile_load_coeff(v, io, g, logN);
      else if (IO == 3) tile_load_odd(v, io, g, logN);
      else if (IO == 4) tile_load_even_exc(v, io, g, logN);
      else ld(v, src + g * 8);
    }
#pragma unroll
    for (int l = 0; l < 8; l++) sm[l][e] = v.v[l];
  }
  __syncthreads();
  // twiddle of the butterfly whose lower element sits at tile position j (sub-tile tw), local span 2^ls
  auto tw_of = [&](fe& w, u32 j, u32 tw, u32 ls) {
    const u32 logm = lo + ls;                        // global span
    const u32 r = lo == 0 ? (j & ((1u << ls) - 1)) : (((j & ((1u << ls) - 1)) << lo) + Lfull0 + tw);   // global index of the element mod 2^logm
    twiddle(w, W, 1u << log_half_max, r << (log_half_max - logm), INV);
  };
  auto lds_get = [&](fe& v, u32 e) {
#pragma unroll
    for (int l = 0; l < 8; l++) v.v[l] = sm[l][e];
  };
  auto lds_put = [&](u32 e, const fe& v) {
#pragma unroll
    for (int l = 0; l < 8; l++) sm[l][e] = v.v[l];
  };
  auto bfly = [&](fe& u, fe& v, const fe& w) {
    fe x, y;
    if (!INV) { F::add(x, u, v); F::sub(y, u, v); F::mul(y, y, w); }
    else { fe t; F::mul(t, v, w); F::add(x, u, t); F::sub(y, u, t); }
    u = x; v = y;
  };
  // one stage: two butterflies per thread, through LDS
  auto single = [&](u32 ls) {
#pragma unroll
    for (u32 q = 0; q < 2; q++) {
      const u32 bb = tid + 256u * q;                 // butterfly 0..511
      // bb -> (pair index p over positions, tw): with TW side-by-side tiles the low logTW bits select the tile
      const u32 tw = bb & (TW - 1), p = bb >> logTW;                 // p in [0, 512 / TW): over (chunk, j-pairs)
      const u32 j0 = ((p >> ls) << (ls + 1)) + (p & ((1u << ls) - 1));   // position (chunk bits included for lo == 0)
      const u32 e0 = (j0 << logTW) + tw, e1 = e0 + ((1u << ls) << logTW);
      fe w; tw_of(w, j0, tw, ls);
      fe u, v; lds_get(u, e0); lds_get(v, e1);
      bfly(u, v, w);
      lds_put(e0, u); lds_put(e1, v);
    }
    __syncthreads();
  };
  // two stages (spans 2^b and 2^(b+1)) on four elements held in registers: half the LDS traffic and barriers of two
  // single stages (the four products are the same: a prime field has no free multiplication by the fourth root of unity)
  auto pair = [&](u32 b) {
    const u32 tw = tid & (TW - 1), p = tid >> logTW;               // p in [0, 256 / TW)
    const u32 j00 = ((p >> b) << (b + 2)) | (p & ((1u << b) - 1));
    const u32 j01 = j00 + (1u << b), j10 = j00 + (2u << b), j11 = j00 + (3u << b);
    const u32 e0 = (j00 << logTW) + tw, e1 = (j01 << logTW) + tw, e2 = (j10 << logTW) + tw, e3 = (j11 << logTW) + tw;
    fe x0, x1, x2, x3, wl, wh0, wh1;
    lds_get(x0, e0); lds_get(x1, e1); lds_get(x2, e2); lds_get(x3, e3);
    tw_of(wl, j00, tw, b); tw_of(wh0, j00, tw, b + 1); tw_of(wh1, j01, tw, b + 1);
    if (!INV) { bfly(x0, x2, wh0); bfly(x1, x3, wh1); bfly(x0, x1, wl); bfly(x2, x3, wl); }
    else { bfly(x0, x1, wl); bfly(x2, x3, wl); bfly(x0, x2, wh0); bfly(x1, x3, wh1); }
    lds_put(e0, x0); lds_put(e1, x1); lds_put(e2, x2); lds_put(e3, x3);
    __syncthreads();
  };
  if (!INV) {            // spans from 2^(S-1) down
    int ls = (int)S - 1;
    for (; ls >= 1; ls -= 2) pair((u32)ls - 1);
    if (ls == 0) single(0);
  } else {               // spans from 2^0 up
    u32 ls = 0;
    for (; ls + 1 < S; ls += 2) pair(ls);
    if (ls < S) single(ls);
  }
#pragma unroll
  for (u32 q = 0; q < 4; q++) {
    const u32 e = tid + 256u * q;
    const u64 g = gidx(e);
    if (g < total) {
      fe v;
#pragma unroll
      for (int l = 0; l < 8; l++) v.v[l] = sm[l][e];
      if (IO == 2) tile_store_coeff(v, io, g, logN); else st(buf + g * 8, v);
    }
  }
}

// A/B variant (option dw_ntt_lazy = 1; measured 1 % SLOWER than k_ntt_tile, profiles/r03/l_ntt_lazy_field_ab.txt): the same pass
// with the butterflies in the lazy 9 x 29-bit field (field29.cuh).  The transforms are bound by VALU issue
// (profiles/r03/g_pmc_lhs_witness_2p18_summary.txt: 85 % of the issue peak), but this file's strict product is already 128
// 64-bit multiply-adds + ~120 adds (disassembly: 768 v_mad_u64_u32 for the 6 products of k_ntt_tile<false, 0>) against the
// lazy one's 162 + ~50, and the lazy butterfly pays two carry passes and a twiddle unpack on top: 3073 VALU instructions
// against 2315 in the same kernel body.  Kept because it pins the strict kernel's buffers bit for bit (tests).
// The elements stay in the ABI's domain (v 2^256) because the twiddles come from a second table in the lazy field's own
// domain (W32 = 32 W: (a 2^256)(w 2^261) / 2^261 = a w 2^256).  A tile is unpacked into 9 signed limbs when it is loaded
// (36 KB of LDS instead of 32) and canonicalised and packed when it is stored, so every pass reads and writes the same
// 32-byte canonical elements as the strict kernel: bit-identical buffers.
// Ranges.  Everything parked in LDS has normalised limbs.  Forward (decimation in frequency) x = u + v doubles the
// bound on |V| per stage and the product's operand is the difference u - v: with a canonical twiddle the product
// tolerates |u - v| < 128N, so the sums are pulled back below 2N (reduce_small, ~40 instructions) before a pair of
// stages whenever the bound has passed 32N -- once in a ten-stage pass.  Inverse (decimation in time) x, y = u +- v w grow
// by 2N per stage: 21N after ten.  Before canon() (|V| < 8N) the store reduces once more where the bound is above 4N.
typedef Field29<Fr29Params> NL29;
template <bool INV, int IO>
__global__ __launch_bounds__(256) void k_ntt_tile_lz(u32* __restrict__ buf, u64 total, u32 logN, u32 lo, u32 S,
                                                     const u32* __restrict__ W32, u32 log_half_max, TileIO io, const u32* __restrict__ src) {
  typedef NL29::fe lf;
  __shared__ i32 sm[9][1024];
  const u32 tid = threadIdx.x;
  const u32 TW = lo ? (1024u >> S) : 1u;
  const u32 logTW = lo ? (10u - S) : 0u;
  u64 base; u32 Lfull0 = 0;
  if (lo == 0) base = (u64)blockIdx.x * 1024u;
  else {
    const u32 hi1 = lo + S;
    const u32 lgroups = 1u << (lo - logTW);
    const u64 b = blockIdx.x;
    const u32 Lg = (u32)(b % lgroups); const u64 rest = b / lgroups;
    base = (rest << hi1) + ((u64)Lg << logTW);
    Lfull0 = Lg << logTW;
  }
  auto gidx = [&](u32 e) -> u64 { return lo == 0 ? base + e : base + ((u64)(e >> logTW) << lo) + (e & (TW - 1)); };
#pragma unroll
  for (u32 q = 0; q < 4; q++) {
    const u32 e = tid + 256u * q;
    const u64 g = gidx(e);
    fe v; F::set_zero(v);
    if (g < total) {
      if (IO == 1) tile_load_coeff(v, io, g, logN);
      else if (IO == 3) tile_load_odd(v, io, g, logN);
      else if (IO == 4) tile_load_even_exc(v, io, g, logN);
      else ld(v, src + g * 8);
    }
    lf t; NL29::unpack(t, v.v);
#pragma unroll
    for (int l = 0; l < 9; l++) sm[l][e] = t.l[l];
  }
  __syncthreads();
  auto tw_of = [&](lf& w, u32 j, u32 tw, u32 ls) {
    const u32 logm = lo + ls;
    const u32 r = lo == 0 ? (j & ((1u << ls) - 1)) : (((j & ((1u << ls) - 1)) << lo) + Lfull0 + tw);
    const u32 t = r << (log_half_max - logm), half_max = 1u << log_half_max;
    fe p;
    if (!INV || t == 0) { ld(p, W32 + (size_t)t * 8); NL29::unpack(w, p.v); }
    else { ld(p, W32 + (size_t)(half_max - t) * 8); lf q; NL29::unpack(q, p.v); NL29::neg(w, q); }   // omega^-t = -omega^(Nmax/2 - t)
  };
  auto lds_get = [&](lf& v, u32 e) {
#pragma unroll
    for (int l = 0; l < 9; l++) v.l[l] = sm[l][e];
  };
  auto lds_put = [&](u32 e, const lf& v) {
#pragma unroll
    for (int l = 0; l < 9; l++) sm[l][e] = v.l[l];
  };
  auto bfly = [&](lf& u, lf& v, const lf& w) {
    lf x, y;
    if (!INV) { NL29::add(x, u, v); NL29::wnorm(x); NL29::sub(y, u, v); NL29::mul(v, y, w); u = x; }
    else { lf t; NL29::mul(t, v, w); NL29::add(x, u, t); NL29::wnorm(x); NL29::sub(y, u, t); NL29::wnorm(y); u = x; v = y; }
  };
  u32 bound = 1;                                     // |V| < bound N for everything in LDS (uniform)
  auto single = [&](u32 ls) {
    const bool red = !INV && bound > 64;
#pragma unroll
    for (u32 q = 0; q < 2; q++) {
      const u32 bb = tid + 256u * q;
      const u32 tw = bb & (TW - 1), p = bb >> logTW;
      const u32 j0 = ((p >> ls) << (ls + 1)) + (p & ((1u << ls) - 1));
      const u32 e0 = (j0 << logTW) + tw, e1 = e0 + ((1u << ls) << logTW);
      lf w; tw_of(w, j0, tw, ls);
      lf u, v; lds_get(u, e0); lds_get(v, e1);
      if (red) { lf t; NL29::reduce_small(t, u); u = t; NL29::reduce_small(t, v); v = t; }
      bfly(u, v, w);
      lds_put(e0, u); lds_put(e1, v);
    }
    if (red) bound = 2;
    bound = INV ? bound + 2 : 2 * bound;
    __syncthreads();
  };
  auto pair = [&](u32 b) {
    const u32 tw = tid & (TW - 1), p = tid >> logTW;
    const u32 j00 = ((p >> b) << (b + 2)) | (p & ((1u << b) - 1));
    const u32 j01 = j00 + (1u << b), j10 = j00 + (2u << b), j11 = j00 + (3u << b);
    const u32 e0 = (j00 << logTW) + tw, e1 = (j01 << logTW) + tw, e2 = (j10 << logTW) + tw, e3 = (j11 << logTW) + tw;
    lf x0, x1, x2, x3, wl, wh0, wh1;
    lds_get(x0, e0); lds_get(x1, e1); lds_get(x2, e2); lds_get(x3, e3);
    if (!INV && bound > 32) {
      lf t; NL29::reduce_small(t, x0); x0 = t; NL29::reduce_small(t, x1); x1 = t; NL29::reduce_small(t, x2); x2 = t; NL29::reduce_small(t, x3); x3 = t;
      bound = 2;
    }
    tw_of(wl, j00, tw, b); tw_of(wh0, j00, tw, b + 1); tw_of(wh1, j01, tw, b + 1);
    if (!INV) { bfly(x0, x2, wh0); bfly(x1, x3, wh1); bfly(x0, x1, wl); bfly(x2, x3, wl); }
    else { bfly(x0, x1, wl); bfly(x2, x3, wl); bfly(x0, x2, wh0); bfly(x1, x3, wh1); }
    lds_put(e0, x0); lds_put(e1, x1); lds_put(e2, x2); lds_put(e3, x3);
    bound = INV ? bound + 4 : 4 * bound;
    __syncthreads();
  };
  if (!INV) {
    int ls = (int)S - 1;
    for (; ls >= 1; ls -= 2) pair((u32)ls - 1);
    if (ls == 0) single(0);
  } else {
    u32 ls = 0;
    for (; ls + 1 < S; ls += 2) pair(ls);
    if (ls < S) single(ls);
  }
#pragma unroll
  for (u32 q = 0; q < 4; q++) {
    const u32 e = tid + 256u * q;
    const u64 g = gidx(e);
    if (g < total) {
      lf t;
#pragma unroll
      for (int l = 0; l < 9; l++) t.l[l] = sm[l][e];
      if (bound > 4) { lf r; NL29::reduce_small(r, t); t = r; }
      NL29::canon(t);
      fe v; NL29::pack(v.v, t);
      if (IO == 2) tile_store_coeff(v, io, g, logN); else st(buf + g * 8, v);
    }
  }
}

// W32[t] = 32 W[t] (the twiddles in the lazy field's own domain, canonical)
__global__ __launch_bounds__(256) void k_times32(const u32* __restrict__ W, u32 n, u32* __restrict__ W32) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  fe w; ld(w, W + (size_t)i * 8);
#pragma unroll
  for (int d = 0; d < 5; d++) F::add(w, w, w);
  st(W32 + (size_t)i * 8, w);
}

// ---------------------------------------------------------------------------------------------------------
// one level: load (zero-padded children into the transform buffer), pointwise, store
// buffer layout: seq (q * nnodes + k), q = 0: L.a, 1: L.b, 2: R.a, 3: R.b; results overwrite q = 0 (a) and 1 (b)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_load(const u32* __restrict__ cA, const u32* __restrict__ cB, const uint2* __restrict__ child_lens,
                                              u32 ccapA, u32 ccapB, const Plan* __restrict__ plan, u32 nnodes, u32 logN,
                                              const u32* __restrict__ GP /* g^i */, u32* __restrict__ buf,
                                              u32* __restrict__ c0in /* wrap mode: the four constant terms per node, [q][node]; else null */) {
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 per = (u64)nnodes << logN;
  if (gid >= 4 * per) return;
  const u32 q = (u32)(gid / per); const u64 rem = gid - (u64)q * per;
  const u32 k = (u32)(rem >> logN), i = (u32)rem & ((1u << logN) - 1);
  fe v; F::set_zero(v);
  if (plan[k].mode != MODE_PASS) {
    const u32 c = plan[k].child0 + (q >> 1);
    const uint2 cl = child_lens[c];
    bool have = false;
    if (q & 1) { if (i < cl.y) { ld(v, cB + ((size_t)c * ccapB + i) * 8); have = true; } }
    else { if (i < cl.x) { ld(v, cA + ((size_t)c * ccapA + i) * 8); have = true; } }
    if (have && i) { fe g; ld(g, GP + (size_t)i * 8); F::mul(v, v, g); }   // p(g x): coefficient i times g^i (coset evaluation)
  }
  st(buf + gid * 8, v);
  if (c0in && i == 0) st(c0in + ((size_t)q * nnodes + k) * 8, v);          // p(0)
}

// x_i of the (bit-reversed) slot i of a size-N transform: omega_N^rev(i)
__device__ __forceinline__ void eval_point(fe& x, const u32* __restrict__ W, u32 log_half_max, u32 logN, u32 i) {
  const u32 j = __brev(i) >> (32 - logN);
  const u32 halfN = 1u << (logN - 1);
  if (j < halfN) ld(x, W + ((size_t)j << (log_half_max + 1 - logN)) * 8);
  else { fe p; ld(p, W + ((size_t)(j - halfN) << (log_half_max + 1 - logN)) * 8); F::neg(x, p); }
}

// the evaluation domain of one level, once for all nodes: XS[2i] = x_i = g omega_N^rev(i), XS[2i+1] = x_i^3 + B
__global__ __launch_bounds__(256) void k_domain(u32 logN, const u32* __restrict__ W, u32 log_half_max, const u32* __restrict__ consts, u32* __restrict__ XS) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (1u << logN)) return;
  fe cb, cg, x, t, s; ld(cb, consts); ld(cg, consts + 16);
  eval_point(x, W, log_half_max, logN, i); F::mul(x, x, cg);            // the coset keeps x_i - c != 0
  F::sqr(t, x); F::mul(t, t, x); F::add(s, t, cb);
  st(XS + (size_t)i * 16, x); st(XS + (size_t)i * 16 + 8, s);
  // the same two in the lazy field's own domain (x 2^261 = 32 x 2^256), canonical: operands of pw_numerators29
  fe x5 = x, s5 = s;
#pragma unroll
  for (int d = 0; d < 5; d++) { F::add(x5, x5, x5); F::add(s5, s5, s5); }
  u32* XS32 = XS + ((size_t)16 << logN);            // a second table behind the first (the strict path's table keeps its 64-byte stride)
  st(XS32 + (size_t)i * 16, x5); st(XS32 + (size_t)i * 16 + 8, s5);
}

// The pointwise step of a level, in two kernels: k_pw_num (numerators and denominators, a thread per slot) and k_pw_inv
// (a thread per share of slots: one inversion for all of them; the host picks the share: 64 slots when there are
// millions of elements, fewer when a level is short on threads).
//
// Wrap mode (c0in != null).  A node over m = 2^k points has an a-part of m/2 + 1 coefficients: ONE more than a
// transform of m/2 holds.  Evaluating on the m/2-point coset anyway folds the top coefficient onto the constant term
// (x^N = g^N on the domain: the inverse transform returns c_0 + g^N c_N at index 0).  The constant term itself is the
// value at x = 0, which needs only the four constant terms of the children: thread 0 of every node carries that extra
// "slot" through the same formulas (and the same shared inversion), and k_store unfolds c_N = (t_0 - c_0) g^-N.  Every
// transform of such a level is half as long.  (No Grumpkin point has x = 0 -- -17 is a non-residue -- so the extra
// denominator cannot vanish for points on the curve; if it does, STAT_ZERO0 sends the level back to the full size.)
__device__ __forceinline__ void pw_numerators(fe& A, fe& Bv, fe& den, bool divide, const fe& x, const fe& s,
                                              const fe& La, const fe& Lb, const fe& Ra, const fe& Rb,
                                              const fe& c0, const fe& c1, const fe& d0, const fe& lX, const fe& lZZ, const fe& rX, const fe& rZZ, const fe& ninv) {
  fe t, u;
  // (p + y q)(r + y w) with y^2 = s, by three products and one more for s (Karatsuba): p r + s q w,  (p + q)(r + w) - p r - q w
  auto rfmul = [&](fe& oa, fe& ob, const fe& pp, const fe& qq, const fe& rr, const fe& ww) {
    fe m1, m2, m3, e, f;
    F::mul(m1, pp, rr); F::mul(m2, qq, ww);
    F::add(e, pp, qq); F::add(f, rr, ww); F::mul(m3, e, f);
    F::sub(m3, m3, m1); F::sub(ob, m3, m2);
    F::mul(e, m2, s); F::add(oa, m1, e);
  };
  if (divide) {
    fe l, tA, tB;
    F::mul(l, c1, x); F::add(l, l, c0);                               // c0 + c1 x
    rfmul(tA, tB, Ra, Rb, l, d0);                                     // R.w * line:  R.a l + s R.b d0,  R.a d0 + R.b l
    rfmul(A, Bv, La, Lb, tA, tB);                                     // L.w * (..)
    F::mul(t, lZZ, x); F::sub(t, t, lX); F::mul(u, rZZ, x); F::sub(u, u, rX); F::mul(den, t, u);   // (ZZ_L x - X_L)(ZZ_R x - X_R)
  } else {
    rfmul(A, Bv, La, Lb, Ra, Rb);
    F::mul(A, A, ninv); F::mul(Bv, Bv, ninv);                          // 1/N of the inverse transform
    F::set_one(den);
  }
}

// The same in the lazy 9 x 29-bit field (field29.cuh): a product is ~200 instructions there against ~370 in the strict field,
// and k_pw_num is bound by VALU issue (6.4e9 instructions per 2^18-point call at the issue peak:
// profiles/r03/g_pmc_lhs_witness_2p18_summary.txt).  Values stay in the ABI's domain (v 2^256) as long as every product has
// ONE factor in the lazy field's own domain (v 2^261): (a 2^256)(b 2^261) / 2^261 = ab 2^256.  The domain points arrive in
// both forms (k_domain), the per-node constants and the intermediate tA, tB are lifted with mul32 (32 v - q N, ~50
// instructions, no Montgomery product).  Sums are carry-normalised before they enter a product (limbs < 2^29; the value may be
// any representative with |V| < 8N); the three results are canonicalised and packed back into 32-byte elements.
typedef Field29<Fr29Params> L29;
typedef L29::fe lfe;
__device__ __forceinline__ void to_l(lfe& r, const fe& a) { L29::unpack(r, a.v); }
// (p + y q)(r + y w) with y^2 = s; p, q in the ABI domain, r, w, s in the 2^261 domain: oa + y ob in the ABI domain
__device__ __forceinline__ void rfmul29(lfe& oa, lfe& ob, const lfe& pp, const lfe& qq, const lfe& rr, const lfe& ww, const lfe& s) {
  lfe m1, m2, m3, e, f;
  L29::mul(m1, pp, rr); L29::mul(m2, qq, ww);
  L29::add(e, pp, qq); L29::wnorm(e); L29::add(f, rr, ww); L29::wnorm(f); L29::mul(m3, e, f);
  L29::sub(m3, m3, m1); L29::sub(ob, m3, m2); L29::wnorm(ob);
  L29::mul(e, m2, s); L29::add(oa, m1, e); L29::wnorm(oa);
}
__device__ __forceinline__ void pw_numerators29(fe& A, fe& Bv, fe& den, bool& den_zero, bool divide, const fe& x32, const fe& s32,
                                                const fe& La, const fe& Lb, const fe& Ra, const fe& Rb,
                                                const fe& c0, const fe& c1, const fe& d0, const fe& lX, const fe& lZZ, const fe& rX, const fe& rZZ, const fe& scale) {
  lfe xq, sq, la, lb, ra, rb, t;
  to_l(xq, x32); to_l(sq, s32); to_l(la, La); to_l(lb, Lb); to_l(ra, Ra); to_l(rb, Rb);
  lfe a, b, dn;
  den_zero = false;
  if (divide) {
    lfe k0, k1, kd, l, tA, tB, tA2, tB2, u, zl, xl, zr, xr;
    to_l(t, c0); L29::mul32(k0, t); to_l(t, c1); L29::mul32(k1, t); to_l(t, d0); L29::mul32(kd, t);     // the scaled line, 2^261 domain
    L29::mul(l, k1, xq); L29::add(l, l, k0); L29::wnorm(l);               // c0 + c1 x
    rfmul29(tA, tB, ra, rb, l, kd, sq);                                   // R.w * line (ABI domain)
    L29::mul32(tA2, tA); L29::mul32(tB2, tB);
    rfmul29(a, b, la, lb, tA2, tB2, sq);                                  // L.w * (..)
    to_l(zl, lZZ); to_l(xl, lX); L29::mul(t, zl, xq); L29::sub(t, t, xl); // ZZ_L x - X_L (ABI domain)
    to_l(u, rZZ); L29::mul32(zr, u); to_l(u, rX); L29::mul32(xr, u);
    L29::mul(u, zr, xq); L29::sub(u, u, xr);                              // ZZ_R x - X_R (2^261 domain)
    L29::mul(dn, t, u);
    L29::canon(dn);
    den_zero = L29::limbs_zero(dn);
    L29::pack(den.v, dn);
  } else {
    lfe ra2, rb2, sc, tq;
    L29::mul32(ra2, ra); L29::mul32(rb2, rb);
    rfmul29(a, b, la, lb, ra2, rb2, sq);
    to_l(tq, scale); L29::mul32(sc, tq);
    L29::mul(a, a, sc); L29::mul(b, b, sc);                               // 1/N of the inverse transform (and the reuse scale)
    F::set_one(den);
  }
  L29::canon(a); L29::pack(A.v, a);
  L29::canon(b); L29::pack(Bv.v, b);
}

// pass A: one thread per slot// pass A: one thread per slot (and one per node for the extra slot x = 0 of wrap mode, gid >= nnodes * N): numerators
// into the L.a / L.b slots, the denominator into the R.b slot.  Nothing here depends on a neighbouring slot, so the loads
// of a whole wave are in flight together (the one-kernel version walked a thread's slots one after the other at 190
// VGPRs, two waves per SIMD).
template <bool LAZY /* the products in the lazy 29-bit field (pw_numerators29: A/B variant, no faster) instead of the strict field; a template
                       parameter: with both paths in one kernel the strict one lost 4 % to the larger register footprint */>
__global__ __launch_bounds__(256) void k_pw_num(u32* __restrict__ buf, const Plan* __restrict__ plan, u32 nnodes, u32 logN,
                                                const u32* __restrict__ XS, const u32* __restrict__ consts /* [0]: curve b, [8]: 1/N, [16]: g */,
                                                u32* __restrict__ stats, u32* __restrict__ c0in /* wrap mode, else null; [3][k] receives the extra slot's denominator */,
                                                u32* __restrict__ c0out,
                                                const u32* __restrict__ odd /* reuse mode (else null): the children on the odd half, [q][node][N/2] */,
                                                const u32* __restrict__ evprev /* the level below's transform buffer: quotient values (times 1/(N/2)) on the even half */,
                                                const u32* __restrict__ evexc /* passed-through children on the even half, [2 tree + part][N/2] */, u32 nn_prev,
                                                u32 k0, u32 kn /* this launch covers the nodes [k0, k0 + kn) (a level runs as two halves on two queues) */) {
  constexpr bool lazy = LAZY;
  const u32 N = 1u << logN;
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 per = (u64)nnodes << logN;
  const u64 perl = (u64)kn << logN;
  const bool extra = gid >= perl;
  if (extra && (c0in == nullptr || gid >= perl + kn)) return;
  const u32 k = k0 + (extra ? (u32)(gid - perl) : (u32)(gid >> logN)), i = extra ? 0u : ((u32)gid & (N - 1));
  const Plan& pl = plan[k];
  if (pl.mode == MODE_PASS) return;
  u32* sLa = buf + (((size_t)k << logN) + i) * 8; u32* sLb = sLa + per * 8; u32* sRa = sLb + per * 8; u32* sRb = sRa + per * 8;
  if (extra) { sLa = c0in + ((size_t)0 * nnodes + k) * 8; sLb = c0in + ((size_t)1 * nnodes + k) * 8; sRa = c0in + ((size_t)2 * nnodes + k) * 8; sRb = c0in + ((size_t)3 * nnodes + k) * 8; }
  fe ninv; ld(ninv, consts + 8);
  fe c0, c1, d0, lX, lZZ, rX, rZZ;
  const bool divide = pl.mode == MODE_DIVIDE;
  // reuse mode, even half of the domain: values kept from the level below carry its 1/(N/2) (the scale of its inverse
  // transform rode on its numerators); it is undone through the scale of this level's numerators: (N/2)^kept
  const bool even_half = odd != nullptr && !extra && i < (N >> 1);
  const u32 kept = even_half ? (pl.evk[0] == 0 ? 1u : 0u) + (pl.evk[1] == 0 ? 1u : 0u) : 0u;
  fe scale = ninv;                                                        // 1/N of the inverse transform (x (N/2)^kept)
  // The scaled line coefficients are the same for every slot of a node's half: where a whole block lies inside one
  // (N >= 512), three lanes compute them once for the block instead of every lane for itself (6 of ~18 products per slot)
  __shared__ u32 shc[3][8];
  const bool uni = logN >= 9 && ((u64)blockIdx.x + 1) * 256 <= perl;
  if (divide) {
    ld(lX, pl.lX); ld(lZZ, pl.lZZ); ld(rX, pl.rX); ld(rZZ, pl.rZZ);
    if (uni) {
      if (threadIdx.x < 3) {
        fe c; ld(c, threadIdx.x == 0 ? pl.c0 : (threadIdx.x == 1 ? pl.c1 : pl.d0));
        F::mul(c, c, ninv);
        if (kept) { fe sc; ld(sc, consts + (kept == 2 ? 48 : 40)); F::mul(c, c, sc); }
        st(shc[threadIdx.x], c);
      }
      __syncthreads();
      ld(c0, shc[0]); ld(c1, shc[1]); ld(d0, shc[2]);
    } else {
      ld(c0, pl.c0); ld(c1, pl.c1); ld(d0, pl.d0);
      if (kept) { fe sc; ld(sc, consts + (kept == 2 ? 48 : 40)); F::mul(scale, scale, sc); }
      F::mul(c0, c0, scale); F::mul(c1, c1, scale); F::mul(d0, d0, scale);   // the numerator is linear in the line: the scale rides on it
    }
  } else if (kept) { fe sc; ld(sc, consts + (kept == 2 ? 48 : 40)); F::mul(scale, scale, sc); }
  fe x, sv, La, Lb, Ra, Rb, A, Bv, den;
  if (extra) {                                                           // x = 0: 0^3 + b
    F::set_zero(x); ld(sv, consts);
    if (lazy) { fe b5 = sv; for (int d = 0; d < 5; d++) F::add(b5, b5, b5); sv = b5; }
  } else if (lazy) { const u32* XS32 = XS + ((size_t)16 << logN); ld(x, XS32 + (size_t)i * 16); ld(sv, XS32 + (size_t)i * 16 + 8); }   // 32 x_i, 32 (x_i^3 + b): the 2^261 domain
  else { ld(x, XS + (size_t)i * 16); ld(sv, XS + (size_t)i * 16 + 8); }  // x_i, x_i^3 + b (= y^2)
  if (odd == nullptr || extra) { ld(La, sLa); ld(Lb, sLb); ld(Ra, sRa); ld(Rb, sRb); }
  else {
    const u32 logNh = logN - 1, Nh = N >> 1;
    if (i >= Nh) {
      const size_t perh = (size_t)nnodes << logNh;
      const u32* o = odd + ((((size_t)k << logNh) + (i - Nh)) * 8);
      ld(La, o); ld(Lb, o + perh * 8); ld(Ra, o + 2 * perh * 8); ld(Rb, o + 3 * perh * 8);
    } else {
      // even half: the children's own domain -- the level below's transform buffer, or the side buffer of a passed-through child
#pragma unroll
      for (int side = 0; side < 2; side++) {
        const u32* pa; const u32* pb;
        if (pl.evk[side] == 0) { const u32 c = pl.child0 + side; pa = evprev + ((((size_t)c << logNh) + i) * 8); pb = pa + ((size_t)nn_prev << logNh) * 8; }
        else { pa = evexc + (((size_t)(2 * pl.tree) << logNh) + i) * 8; pb = pa + ((size_t)1 << logNh) * 8; }
        if (side == 0) { ld(La, pa); ld(Lb, pb); } else { ld(Ra, pa); ld(Rb, pb); }
      }
    }
  }
  bool den_zero = false;
  if constexpr (LAZY) pw_numerators29(A, Bv, den, den_zero, divide, x, sv, La, Lb, Ra, Rb, c0, c1, d0, lX, lZZ, rX, rZZ, scale);
  else { pw_numerators(A, Bv, den, divide, x, sv, La, Lb, Ra, Rb, c0, c1, d0, lX, lZZ, rX, rZZ, scale); den_zero = divide && F::is_zero(den); }
  if (divide && den_zero) { atomicOr(&stats[extra ? STAT_ZERO0 : STAT_ZERODEN], 1u); F::set_one(den); }
  if (extra) {
    st(c0out + ((size_t)0 * nnodes + k) * 8, A); st(c0out + ((size_t)1 * nnodes + k) * 8, Bv);
    if (divide) st(sRb, den);
  } else {
    st(sLa, A); st(sLb, Bv);
    if (divide) st(sRb, den);
  }
}

// pass B, three launches.  Thread c of a node owns the slots i = c + q * stride (consecutive lanes on consecutive elements
// in every trip).  k_pw_prefix: prefix products of the thread's denominators forwards (parked in the R.a slots), the
// product of all of them to roots[thread] (1 for threads of nodes without divisions).  k_pw_rootinv: the roots are
// inverted by Montgomery's trick once more, RK of them per thread, so that a level pays one Fermat inversion per
// RK * (slots per thread) elements instead of one per thread.  k_pw_apply: backwards over the same slots, the inverse of
// every denominator onto its numerators.  Thread 0 of a node also owns the node's extra slot (wrap mode).
__global__ __launch_bounds__(256) void k_pw_prefix(u32* __restrict__ buf, const Plan* __restrict__ plan, u32 nnodes, u32 logN, u32 stride /* threads per node */,
                                                   const u32* __restrict__ c0in, u32* __restrict__ roots /* of this launch's nodes */, u32 k0, u32 kn) {
  const u32 N = 1u << logN;
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (u64)kn * stride) return;
  const u32 kl = (u32)(gid / stride), k = k0 + kl, c = (u32)(gid - (u64)kl * stride);
  fe run; F::set_one(run);
  if (plan[k].mode == MODE_DIVIDE) {
    const size_t per = (size_t)nnodes << logN;
    u32* sRa = buf + (2 * per + ((size_t)k << logN)) * 8; const u32* sRb = sRa + per * 8;
    for (u32 i = c; i < N; i += stride) {
      fe den; ld(den, sRb + (size_t)i * 8);
      st(sRa + (size_t)i * 8, run);                                        // prefix product before this element
      F::mul(run, run, den);
    }
    if (c0in != nullptr && c == 0) { fe eden; ld(eden, c0in + ((size_t)3 * nnodes + k) * 8); F::mul(run, run, eden); }   // (its prefix is the product of the thread's regular slots: recomputed in k_pw_apply)
  }
  st(roots + gid * 8, run);
}

static const u32 DW_RK = 32;     // most roots per thread of k_pw_rootinv (fewer when a level has few roots: a short level is latency, not work)
__global__ __launch_bounds__(256) void k_pw_rootinv(u32* __restrict__ roots, u32* __restrict__ rpre /* scratch, as long as roots */, u64 count, u32 rk /* roots per thread */) {
  const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 r0 = t * rk;
  if (r0 >= count) return;
  const u32 m = (u32)min((u64)rk, count - r0);
  fe run; F::set_one(run);
  for (u32 j = 0; j < m; j++) {
    fe v; ld(v, roots + (r0 + j) * 8);
    st(rpre + (r0 + j) * 8, run);
    F::mul(run, run, v);
  }
  fe inv; inv_fast(inv, run);
  for (u32 j = m; j-- > 0;) {
    fe v, pj, o; ld(v, roots + (r0 + j) * 8); ld(pj, rpre + (r0 + j) * 8);
    F::mul(o, inv, pj); F::mul(inv, inv, v);
    st(roots + (r0 + j) * 8, o);
  }
}

__global__ __launch_bounds__(256) void k_pw_apply(u32* __restrict__ buf, const Plan* __restrict__ plan, u32 nnodes, u32 logN, u32 stride /* threads per node */,
                                                  const u32* __restrict__ c0in, u32* __restrict__ c0out, const u32* __restrict__ rootinv /* of this launch's nodes */,
                                                  u32 k0, u32 kn) {
  const u32 N = 1u << logN;
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (u64)kn * stride) return;
  const u32 kl = (u32)(gid / stride), k = k0 + kl, c = (u32)(gid - (u64)kl * stride);
  if (plan[k].mode != MODE_DIVIDE) return;
  const size_t per = (size_t)nnodes << logN;
  u32* sLa = buf + (((size_t)k << logN)) * 8; u32* sLb = sLa + per * 8; u32* sRa = sLb + per * 8; u32* sRb = sRa + per * 8;
  fe inv; ld(inv, rootinv + gid * 8);
  const u32 cnt = c < N ? (N - c + stride - 1) / stride : 0u;      // slots of this thread
  if (c0in != nullptr && c == 0) {
    // the extra slot came last in the product: its prefix = product of the regular slots = prefix of the last slot * its denominator
    fe epref, eden, di, A, Bv;
    if (cnt) { fe lp, ld_; const u32 il = c + (cnt - 1) * stride; ld(lp, sRa + (size_t)il * 8); ld(ld_, sRb + (size_t)il * 8); F::mul(epref, lp, ld_); }
    else F::set_one(epref);
    ld(eden, c0in + ((size_t)3 * nnodes + k) * 8);
    F::mul(di, inv, epref); F::mul(inv, inv, eden);
    ld(A, c0out + ((size_t)0 * nnodes + k) * 8); ld(Bv, c0out + ((size_t)1 * nnodes + k) * 8);
    F::mul(A, A, di); F::mul(Bv, Bv, di);
    st(c0out + ((size_t)0 * nnodes + k) * 8, A); st(c0out + ((size_t)1 * nnodes + k) * 8, Bv);
  }
  for (u32 q = cnt; q-- > 0;) {
    const u32 i = c + q * stride;
    fe pref, den, di, A, Bv;
    ld(pref, sRa + (size_t)i * 8); ld(den, sRb + (size_t)i * 8);
    F::mul(di, inv, pref); F::mul(inv, inv, den);
    ld(A, sLa + (size_t)i * 8); ld(Bv, sLb + (size_t)i * 8);
    F::mul(A, A, di); F::mul(Bv, Bv, di);
    st(sLa + (size_t)i * 8, A); st(sLb + (size_t)i * 8, Bv);
  }
}

__global__ __launch_bounds__(256) void k_store(const u32* __restrict__ buf, const Plan* __restrict__ plan, u32 nnodes, u32 logN,
                                               const u32* __restrict__ cA, const u32* __restrict__ cB, u32 ccapA, u32 ccapB,
                                               u32* __restrict__ nA, u32* __restrict__ nB, u32 capA, u32 capB, uint2* __restrict__ lens,
                                               const u32* __restrict__ GI /* g^-i */,
                                               const u32* __restrict__ c0out /* wrap mode: (a(0), b(0)) / N per node, [part][node]; else null */,
                                               const u32* __restrict__ consts /* [24]: N, [32]: g^-N */) {
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  const u32 cap = max(capA, capB);
  if (gid >= (u64)nnodes * cap) return;
  const u32 k = (u32)(gid / cap), i = (u32)(gid - (u64)k * cap);
  const Plan& pl = plan[k];
  if (i == 0) lens[k] = make_uint2(pl.la, pl.lb);
  const size_t per = (size_t)nnodes << logN;
  const u32 N = 1u << logN;
  fe v;
  // part `which` (0: a, 1: b) of a merged node, coefficient i: straight from the inverse transform (times g^-i), except
  // in wrap mode for a part of N + 1 coefficients: c_0 = N * (value at 0 from the extra slot), c_N = (t_0 - c_0) g^-N
  auto merged = [&](u32 which, u32 len) {
    const u32* seq = buf + ((size_t)which * per + ((size_t)k << logN)) * 8;
    if (c0out && len == N + 1 && (i == 0 || i == N)) {
      fe e, nn, c0v; ld(e, c0out + ((size_t)which * nnodes + k) * 8); ld(nn, consts + 24);
      F::mul(c0v, e, nn);
      if (i == 0) { v = c0v; return; }
      fe t0, gi; ld(t0, seq); ld(gi, consts + 32);
      F::sub(t0, t0, c0v); F::mul(v, t0, gi);
      return;
    }
    ld(v, seq + (size_t)i * 8);
    if (i) { fe g; ld(g, GI + (size_t)i * 8); F::mul(v, v, g); }
  };
  if (i < pl.la) {
    if (pl.mode == MODE_PASS) ld(v, cA + ((size_t)pl.child0 * ccapA + i) * 8);
    else merged(0, pl.la);
    st(nA + ((size_t)k * capA + i) * 8, v);
  }
  if (i < pl.lb) {
    if (pl.mode == MODE_PASS) ld(v, cB + ((size_t)pl.child0 * ccapB + i) * 8);
    else merged(1, pl.lb);
    st(nB + ((size_t)k * capB + i) * 8, v);
  }
}

// W[t] = omega^t, t < count: omega^t = prod over the set bits b of t of P2[b] = omega^(2^b)
__global__ __launch_bounds__(256) void k_twiddles(const u32* __restrict__ P2, u32 nbits, u32 count, u32* __restrict__ W) {
  u32 t = blockIdx.x * 256 + threadIdx.x;
  if (t >= count) return;
  fe acc; F::set_one(acc);
  for (u32 b = 0; b < nbits; b++)
    if ((t >> b) & 1u) { fe p; ld(p, P2 + (size_t)b * 8); F::mul(acc, acc, p); }
  st(W + (size_t)t * 8, acc);
}

// normalise: every coefficient times `scale` (the inverse of the leading coefficient by pole order)
__global__ __launch_bounds__(256) void k_scale(u32* __restrict__ p, u32 cnt, const u32* __restrict__ scale) {
  u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  fe s, v; ld(s, scale); ld(v, p + (size_t)i * 8); F::mul(v, v, s); st(p + (size_t)i * 8, v);
}
// the same for the T roots of a forest: node t's part (la or lb coefficients at stride cap) times scale[t]
__global__ __launch_bounds__(256) void k_scale_roots(u32* __restrict__ p, u32 cap, const uint2* __restrict__ lens, u32 which, u32 T, const u32* __restrict__ scale) {
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (u64)T * cap) return;
  const u32 t = (u32)(gid / cap), i = (u32)(gid - (u64)t * cap);
  const uint2 l = lens[t];
  if (i >= (which ? l.y : l.x)) return;
  fe s, v; ld(s, scale + (size_t)t * 8); ld(v, p + ((size_t)t * cap + i) * 8); F::mul(v, v, s); st(p + ((size_t)t * cap + i) * 8, v);
}

// debug / KAT: plain forward or inverse transform of nseq sequences in natural order (bit reversal applied on the way in/out)
__global__ __launch_bounds__(256) void k_bitrev_copy(const u32* __restrict__ in, u32* __restrict__ out, u32 nseq, u32 logN) {
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  if (gid >= ((u64)nseq << logN)) return;
  const u32 s = (u32)(gid >> logN), i = (u32)gid & ((1u << logN) - 1);
  const u32 j = logN ? (__brev(i) >> (32 - logN)) : 0;
  fe v; ld(v, in + gid * 8); st(out + (((size_t)s << logN) + j) * 8, v);
}

// tmp lists of compute_lhs_witness (src/argument_witness_calc.rs:110-127): flags and gather
__global__ __launch_bounds__(256) void k_lhs_flags(const uint8_t* __restrict__ digitsT /* d x n position-major, LSB-first positions */, u32 n, u32 pos, u32* __restrict__ flags) {
  u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  flags[j] = digitsT[(size_t)pos * n + j] ? 1u : 0u;
}
// nnz[pos] = number of scalars whose digit at position pos is non-zero (grid: (blocks of 4096 scalars, d)): a block counts
// its 4096 digits in registers and LDS and issues ONE atomic (a ballot + atomic per wave was 4.8 ms at 2^20 points: 135 000
// atomics on 33 addresses)
__global__ __launch_bounds__(256) void k_lhs_count(const uint8_t* __restrict__ digitsT, u32 n, u32* __restrict__ nnz) {
  __shared__ u32 wsum[4];
  const u32 pos = blockIdx.y, j0 = blockIdx.x * 4096u;
  const uint8_t* row = digitsT + (size_t)pos * n;
  u32 c = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const u32 j = j0 + threadIdx.x + 256u * k;
    if (j < n) c += row[j] != 0 ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { const u32 t = wsum[0] + wsum[1] + wsum[2] + wsum[3]; if (t) atomicAdd(&nnz[pos], t); }
}
__global__ __launch_bounds__(256) void k_lhs_gather(const uint8_t* __restrict__ digitsT, u32 n, u32 pos, u32 base, const u32* __restrict__ offs,
                                                    const uint4* __restrict__ table /* n x (base-1) affine */, u32 lead, uint4* __restrict__ out) {
  u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  u32 dg = digitsT[(size_t)pos * n + j];
  if (!dg) return;
  const uint4* s = table + ((size_t)j * (base - 1) + (dg - 1)) * 4;      // precomputed_points[j][id_by_digit(digit)]  :123
  uint4* o = out + ((size_t)lead + offs[j]) * 4;
  o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3];
}
// affine rows (64 B, (0,0) = identity) -> Jacobian rows (96 B, z = 1 in Montgomery form, identity = zeros)
__global__ __launch_bounds__(256) void k_aff_to_jac(const uint4* __restrict__ aff, u32 n, uint4* __restrict__ jac) {
  u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint4 a0 = aff[(size_t)i * 4], a1 = aff[(size_t)i * 4 + 1], a2 = aff[(size_t)i * 4 + 2], a3 = aff[(size_t)i * 4 + 3];
  const bool id = (a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w | a2.x | a2.y | a2.z | a2.w | a3.x | a3.y | a3.z | a3.w) == 0;
  fe one; F::set_one(one);
  uint4 z0 = id ? make_uint4(0, 0, 0, 0) : make_uint4(one.v[0], one.v[1], one.v[2], one.v[3]);
  uint4 z1 = id ? make_uint4(0, 0, 0, 0) : make_uint4(one.v[4], one.v[5], one.v[6], one.v[7]);
  uint4* o = jac + (size_t)i * 6;
  o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = z0; o[5] = z1;
}
// out[i] = pt for i < count (the `base` copies of -carry, :112-116) ; and single-slot writes
__global__ __launch_bounds__(256) void k_fill_points(const uint4* __restrict__ pt, u32 count, uint4* __restrict__ out) {
  u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  out[(size_t)i * 4] = pt[0]; out[(size_t)i * 4 + 1] = pt[1]; out[(size_t)i * 4 + 2] = pt[2]; out[(size_t)i * 4 + 3] = pt[3];
}

}  // namespace dw
}  // namespace lemsm
