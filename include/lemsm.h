/*
 * lemsm -- MI355X (gfx950) MSM witness path for Liam Eagen's MSM argument.
 *
 * C ABI that the reference crate's Rust code would bind (see INTEGRATION.md for the
 * Rust `extern "C"` block and the shim that keeps the reference's entry signatures).
 * The reference has no FFI today; each entry below names the Rust function it stands
 * behind (paths relative to the reference repository root):
 *
 *   lemsm_msm_*                  halo2::arithmetic::best_multiexp(coeffs, bases)
 *                                (third-party; imported src/argument_witness_calc.rs:20,
 *                                called :144 and src/regular_functions_utils.rs:655,679,694,726)
 *   lemsm_lhs_msm_*              compute_lhs_witness MSM core, src/argument_witness_calc.rs:87-127,132-134
 *                                (the per-digit divisor witnesses of :129 stay in Rust; they are
 *                                built from the per-digit carries returned here)
 *   lemsm_negbase_decompose_batch negbase_decompose, src/negbase_utils.rs:20-36, over a slice
 *   lemsm_num_digits             d = logb_ceil(isqrt(order)+2, base)+1, src/argument_witness_calc.rs:89-91
 *   lemsm_precompute_multiplicities precompute_multiplicities, src/argument_witness_calc.rs:43-51
 *
 * Conventions
 *   scalars   n x 32 bytes, canonical little-endian integers < scalar-field order
 *             (== PrimeField::to_repr(), what best_multiexp and compute_lhs_witness read:
 *             src/argument_witness_calc.rs:93)
 *   field elt 4 x uint64 little-endian limbs in Montgomery form, R = 2^256
 *             (== SerdeObject::to_raw_bytes(), src/scripts.rs:44, src/precomputed_fft_data.rs:72)
 *   affine    (x[4], y[4]); the identity is (0,0)
 *   jacobian  (x[4], y[4], z[4]), x_aff = X/Z^2, y_aff = Y/Z^3; the identity has z == 0.
 *             Outputs are valid Jacobian points but their coordinates are NOT canonical
 *             (they depend on summation order); compare as group elements or through
 *             lemsm_jacobian_to_canonical.
 *   curves    LEMSM_BN254_G1: y^2 = x^3 + 3 over p, scalars mod r
 *             LEMSM_GRUMPKIN:  y^2 = x^3 - 17 over r, scalars mod p
 *
 * Ownership: the caller owns every buffer; the library neither keeps nor frees them.
 * Errors: non-zero lemsm_status; the reference's assert!/panic! sites map to
 *   LEMSM_ERR_LEN_MISMATCH (:88), LEMSM_ERR_SCALAR_OUT_OF_RANGE (:97, index in *bad_index),
 *   LEMSM_ERR_BAD_BASE (base < 3: the reference silently truncates for base 2).
 * Threading: a context is one GPU + one HIP stream + its workspace.  Calls on one context
 *   are serialised by the caller; different contexts may be used concurrently.  Every entry
 *   is synchronous: results are final on return.
 * There is no CPU fallback: without a usable gfx950 device lemsm_create fails.
 */
#ifndef LEMSM_H
#define LEMSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lemsm_ctx lemsm_ctx;

enum lemsm_curve { LEMSM_BN254_G1 = 0, LEMSM_GRUMPKIN = 1 };

enum lemsm_status {
  LEMSM_OK = 0,
  LEMSM_ERR_LEN_MISMATCH = 1,
  LEMSM_ERR_SCALAR_OUT_OF_RANGE = 2,
  LEMSM_ERR_BAD_BASE = 3,
  LEMSM_ERR_BAD_CURVE = 4,
  LEMSM_ERR_HIP = 5,
  LEMSM_ERR_BAD_ARG = 6,
  LEMSM_ERR_NOMEM = 7,
  LEMSM_ERR_TOO_MANY_DIGITS = 8,
  LEMSM_ERR_RCCL = 9,
  LEMSM_ERR_INDEX_OUT_OF_BOUNDS = 10,
  LEMSM_ERR_ARITH_OVERFLOW = 11,
  LEMSM_ERR_SUM_NOT_IDENTITY = 12,
  LEMSM_ERR_WOULD_NOT_TERMINATE = 13,
  LEMSM_ERR_DIVISION_BY_ZERO = 14
};

/* ---- context ------------------------------------------------------------------------- */
int lemsm_create(int device, lemsm_ctx** out);
void lemsm_destroy(lemsm_ctx* ctx);
const char* lemsm_strerror(int status);
const char* lemsm_last_error(const lemsm_ctx* ctx);
/* index of the offending scalar after a LEMSM_ERR_SCALAR_OUT_OF_RANGE from an entry that has no
   bad_index parameter (lemsm_msm*: a scalar that is not a canonical field element) */
size_t lemsm_last_bad_index(const lemsm_ctx* ctx);
/* Number of scalars of the last negabase pass (lemsm_negbase_decompose_batch, lemsm_lhs_*) whose
   expansion needed more than d digits and was truncated exactly like the reference's
   `.chain(repeat(0)).take(d)` (src/argument_witness_calc.rs:99).  In-range scalars never truncate for
   base >= 3 with d = lemsm_num_digits; a non-zero count after a call with a smaller d
   means the digits no longer recompose the scalar. */
size_t lemsm_last_truncated_count(const lemsm_ctx* ctx);
/* Tuning / test knobs: "window_bits" (0 = auto), "chunk" (entries per accumulate thread,
   0 = auto), "tile" (pass-2 tile entries, 0 = auto), "binsort" (pass 2: 0/1 = bins of up to 36 864
   entries are bucket-sorted whole by one block, larger ones by the tiled kernels; 2 = tiled kernels only (A/B knob);
   > 2 = that capacity instead of 36 864, for tests that want both paths in one call), "field" (0 = lazy radix-2^29 arithmetic, the default;
   1 = strict 32-bit-limb arithmetic, kept for A/B and as an in-library cross-check),
   "accum_waves" (2..4 waves per SIMD of the accumulate kernel, 0 = auto), "groups" (window groups
   pipelined over three queues, device-pointer entries only; 0 = one group),
   "host_slab_bits" (host-pointer entries: log2 of the slab of pairs uploaded while the previous
   slab is being accumulated; 0 = auto = 21), "slab_bits" (device-pointer entries: log2 of the
   slab of pairs one pass of the pipeline covers; 0 = auto = 24, smaller values are a test knob),
   "merge_slice" (edge-record merge: pieces of a long bucket one wave adds, 0 = auto = 512; small values are a
   test knob), "merge_wave_th" (buckets of 9..32 pieces get one wave each while there are fewer than this many,
   0 = auto = 2049; 1 = always serial),
   "abi_points" (lazy arithmetic: 1 = convert the points to the kernels' domain in a pass of their
   own, 2 = let the accumulation consume them as passed in, 0 = choose by segment length and window
   count), "pyr_fuse" (bucket-reduction pyramid: 0 = one launch per step while a window's step has more than 256 items, the
   rest in one launch of one block per window; 1 = one launch per step throughout; 2 = fuse from 2048 items: A/B knob), "host_threads" (host tail: 0 = up to 8 threads, 1 = serial), "pyr_first2" (1 = the first two pyramid steps in one pass over the bucket sums where the window has
   >= 32 buckets -- half the HBM traffic of the two launches, measured no faster: A/B knob; 0 = one launch per step), "ntt_tiled" (divisor witness: 0 = LDS-tiled transforms, up to 10 stages per launch; 2 = one launch per
   stage: A/B knob), "dw_kb" (divisor witness: slots per thread of the batched-inversion kernels, 8..64; 0 = auto),
   "dw_fuse" (divisor witness: 0 = the first forward transform pass of a level gathers its
   input from the coefficient arrays and the last inverse pass scatters into them; 2 = separate load / store kernels:
   A/B knob), "dw_pw_lazy" (divisor witness: 1 = the pointwise numerators in the lazy 29-bit field -- 12 cheaper products, paid back by the
   conversions around them: measured no faster; 0 = strict field: A/B knob), "scatter_lean" (pass 1 of the bucket sort: 1 = a thread of k_scatter1 owns two adjacent bins, one block scan, unrolled store loop -- measured no faster, profiles/r03/s_scatter_lean_and_clear_beside_ab.txt; 0 = bins tid and tid + 256: A/B knob), "pyr_quad" (bucket-reduction pyramid, lazy arithmetic: 0 = steps that leave most of the chip without a wave -- and the fused last steps -- run four lanes per addition (one product per lane and stage, DPP quad exchanges: ~2.6x shallower); 2 = one lane per addition: A/B knob), "slab_tail" (calls of more than one slab of points -- more than 2^24, or option slab_bits: 0 = every slab accumulates into a bucket area of its own and the call runs one bucket reduction and leaves one block of records (up to 8 slabs); 2 = a reduction per slab, the records added on the host: A/B knob), "dw_halves" (divisor witness: 1 = the pointwise chain of a level runs as two halves of its nodes on two queues, one half's batched inversion beside the other half's numerators: measured no faster, profiles/r03/k_halves_trace.txt; 0 = one chain: A/B knob), "dw_ntt_lazy" (divisor witness: 1 = the butterflies of the LDS-tiled transforms in the lazy 29-bit field, twiddles from a second table -- measured 1 % slower: the strict product is 128 multiply-adds here, the lazy one 162 plus carry passes; 0 = strict field: A/B knob), "dw_reuse" (divisor witness: 0 = a level transforms its children onto the odd half of its domain only and reads the even half from the level
   below's evaluations, 2 = whole transforms: A/B knob), "dw_wrap" (divisor witness: 0 = levels whose longest part
   has 2^k + 1 coefficients run on 2^k-point transforms, the folded top coefficient recovered from the value at x = 0;
   2 = always the next power of two: A/B knob), "ws_canary" (1 = debug: every sub-buffer of the MSM workspace is followed by a 256-byte guard that is
   filled before and verified after every window group; an overrun returns LEMSM_ERR_HIP naming the guard; the fuzz
   soak turns it on at random), "validate_points" (1: lemsm_msm*, lemsm_lhs_msm* check y^2 = x^3 + b for every non-identity input point
   before any work and return LEMSM_ERR_BAD_ARG with the index in lemsm_last_bad_index; 0, the default, trusts the
   caller like the reference's from_raw_bytes_unchecked does -- the hot kernel tests y alone for the identity,
   so an invalid (x != 0, y == 0) input would otherwise be skipped silently). */
int lemsm_set_option(lemsm_ctx* ctx, const char* name, long value);
/* Device-time (ms, from HIP events on the context's stream) of the last MSM call: whole
   pipeline in [0], the dominant accumulate kernel in [1], its launch count in [2]. */
int lemsm_last_timing(const lemsm_ctx* ctx, double out[3]);
/* Shader clock (MHz) the accumulate kernel of the last MSM call sustained, from in-kernel stamps
   (s_memtime / s_memrealtime around one mid-grid wave's whole chunk); 0 if no launch stamped. */
int lemsm_last_accum_clock_mhz(const lemsm_ctx* ctx, double* out);
/* Test / profiling aid: how the edge-record merge of the last MSM call classified the buckets that straddle
   accumulate chunks, summed over window groups and slabs: [0] 3..8 pieces (serial), [1] 9..32 pieces,
   [2] slices of buckets of more than 32 pieces, [3] buckets of several slices.  Two-piece buckets are not counted. */
int lemsm_debug_last_merge_counts(const lemsm_ctx* ctx, uint64_t out[4]);

/* ---- best_multiexp ------------------------------------------------------------------- */
/* sum_i scalars[i] * points[i]; host buffers. */
int lemsm_msm(lemsm_ctx* ctx, int curve, const uint8_t* scalars, const uint64_t* points_affine,
              size_t n, uint64_t out_jacobian[12]);
/* `batch` MSMs over the same points: sum_i scalars_k[i] * points[i] for k < batch; d_scalars[k]: device pointer to n x 32 B,
   outs: batch x 12 limbs.  The calls are pipelined over two lanes of queues and workspaces: the host tail of call k - 1 and
   its latency-bound bucket-reduction tail overlap call k's passes (a prover calls best_multiexp many times over one SRS:
   /root/reference/src/argument_witness_calc.rs:144, src/regular_functions_utils.rs:655-726).  Results equal `batch`
   separate lemsm_msm_device calls; the first failing call's status is returned. */
int lemsm_msm_batch_device(lemsm_ctx* ctx, int curve, const void* const* d_scalars, size_t batch, const void* d_points_affine,
                           size_t n, uint64_t* outs);
int lemsm_msm_bn254_g1(lemsm_ctx* ctx, const uint8_t* scalars, const uint64_t* points_affine,
                       size_t n, uint64_t out_jacobian[12]);
int lemsm_msm_grumpkin(lemsm_ctx* ctx, const uint8_t* scalars, const uint64_t* points_affine,
                       size_t n, uint64_t out_jacobian[12]);
/* Same with inputs already resident in this context's GPU memory (device pointers). */
int lemsm_msm_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine,
                     size_t n, uint64_t out_jacobian[12]);

/* Window-sharded form (multi-GPU: each rank computes a range of Pippenger windows, ranks
   exchange the small partial records, everyone combines).  lemsm_msm_plan reports the number
   of windows and the byte size of one window's partial record for an n-point MSM: the record is
   the window's sum S_w = sum_k k * Bucket_{w,k} as one XYZZ point (x, y, zz, zzz; 4 x 32 bytes raw
   Montgomery; zz == 0 is the identity). */
int lemsm_msm_plan(const lemsm_ctx* ctx, int curve, size_t n, uint32_t* num_windows,
                   size_t* partial_bytes_per_window);
int lemsm_msm_partial_device(lemsm_ctx* ctx, int curve, const void* d_scalars,
                             const void* d_points_affine, size_t n, uint32_t win_begin,
                             uint32_t win_end, uint8_t* out_partials /* (win_end-win_begin) records */);
int lemsm_msm_combine(const lemsm_ctx* ctx, int curve, size_t n, const uint8_t* partials_all_windows,
                      uint64_t out_jacobian[12]);

/* ---- multi-GPU: Pippenger-window sharding over the GPUs of one node, RCCL over xGMI ------ */
/* The reference has no distributed code (its parallelism is Rayon inside best_multiexp, called at
   src/argument_witness_calc.rs:144); these entries are what its Rust host binds to use more than one GPU.
   Rank r of G owns windows [W r / G, W (r+1) / G) (digit positions on the lhs path) over ALL points, which every rank
   holds in its own HBM; the ranks exchange the raw per-window records with ONE ncclAllGather straight from device
   memory (EC addition is not an RCCL reduction operator) and every rank combines, so every rank returns the
   same group element.  RCCL is bound at run time (dlopen of librccl.so.1; override with LEMSM_RCCL_LIB): the
   single-GPU entries never need it, a failure to load it is LEMSM_ERR_RCCL here.
   Two ways in:
     one process per GPU   lemsm_comm_unique_id on rank 0, the 128 bytes sent to the other ranks by the host's own
                           means, lemsm_comm_init on every rank (collective), then lemsm_*_sharded_device (collective);
     one process, N GPUs   lemsm_node_*: contexts, communicators and one host thread per GPU inside the library. */
#define LEMSM_COMM_ID_BYTES 128
int lemsm_comm_unique_id(uint8_t id[LEMSM_COMM_ID_BYTES]);
int lemsm_comm_init(lemsm_ctx* ctx, const uint8_t id[LEMSM_COMM_ID_BYTES], int nranks, int rank);
int lemsm_comm_destroy(lemsm_ctx* ctx);
int lemsm_comm_info(const lemsm_ctx* ctx, int* nranks, int* rank);   /* nranks = 0: no communicator */
/* Collective over the context's communicator; inputs resident on every rank.  With automatic window width a
   sharded call uses 16-bit windows from 2^24 points (16 windows split evenly over 2/4/8 ranks). */
int lemsm_msm_sharded_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine, size_t n,
                             uint64_t out_jacobian[12]);
int lemsm_lhs_msm_sharded_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine, size_t n,
                                 uint8_t base, uint64_t out_carry[12], uint64_t* out_carries, size_t* bad_index);

typedef struct lemsm_node lemsm_node;
/* devices == NULL: devices 0..ndev-1 */
int lemsm_node_create(const int* devices, int ndev, lemsm_node** out);
void lemsm_node_destroy(lemsm_node* node);
int lemsm_node_size(const lemsm_node* node);
lemsm_ctx* lemsm_node_ctx(lemsm_node* node, int i);
const char* lemsm_node_last_error(const lemsm_node* node);
/* bases: n affine points (host), replicated into every GPU once and kept resident (halo2's bases are a fixed SRS) */
int lemsm_node_set_bases(lemsm_node* node, int curve, const uint64_t* points_affine, size_t n);
/* best_multiexp(scalars, bases[..n]) over the node.  Every GPU uploads one G-th of the scalar vector over its own PCIe link and
   an in-place all-gather over xGMI completes the copies (the host's memory is read once, not G times). */
int lemsm_node_msm(lemsm_node* node, const uint8_t* scalars, size_t n, uint64_t out_jacobian[12]);
/* compute_lhs_witness MSM core over the node (n must equal the number of resident bases, :88) */
int lemsm_node_lhs_msm(lemsm_node* node, const uint8_t* scalars, size_t n, uint8_t base, uint64_t out_carry[12],
                       uint64_t* out_carries, size_t* bad_index);

/* ---- resident bases on one GPU ------------------------------------------------------- */
/* best_multiexp(&scalars, &pts_aff) (src/argument_witness_calc.rs:144) with pts_aff uploaded once: later calls move
   only the scalars over PCIe, slab by slab under the kernels of the previous slab. */
typedef struct lemsm_bases lemsm_bases;
int lemsm_bases_upload(lemsm_ctx* ctx, int curve, const uint64_t* points_affine, size_t n, lemsm_bases** out);
void lemsm_bases_free(lemsm_bases* bases);
const void* lemsm_bases_device_ptr(const lemsm_bases* bases);
int lemsm_msm_with_bases(lemsm_ctx* ctx, const lemsm_bases* bases, const uint8_t* scalars, size_t n, uint64_t out_jacobian[12]);
/* `batch` MSMs over the same resident bases with the scalar vectors in host memory (scalars[k]: n x 32 B; outs: batch x 12 limbs):
   the upload of call k's scalars and the host fold of call k - 1 run while the GPU works on the neighbouring call, so a call costs
   the longer of upload and compute, not their sum (how a prover calls best_multiexp over one SRS:
   /root/reference/src/argument_witness_calc.rs:144).  Results equal `batch` lemsm_msm_with_bases calls; the first failing
   call's status is returned. */
int lemsm_msm_batch_with_bases(lemsm_ctx* ctx, const lemsm_bases* bases, const uint8_t* const* scalars, size_t batch, size_t n, uint64_t* outs);

/* ---- negabase decomposition ---------------------------------------------------------- */
int lemsm_num_digits(int curve, uint8_t base, uint32_t* d);
/* digits[i*d + k] = k-th negabase digit (LSB first, in [0,base)) of scalar i, zero padded /
   truncated to d exactly like `.chain(repeat(0)).take(d)` (src/argument_witness_calc.rs:99). */
int lemsm_negbase_decompose_batch(lemsm_ctx* ctx, const uint8_t* scalars, size_t n, uint8_t base,
                                  uint32_t d, uint8_t* digits);

/* ---- prepare_scalar_witness / table_entry_by_id -------------------------------------- */
/* prepare_scalar_witness(sc, base, num_digits, logtable), src/negbase_utils.rs:79-124, for each of n scalars, computed
   exactly as the reference's code does (limb index i % logtable + 1, :98-101 -- not upstream's presumable intent).
   scalars: n x 32 bytes little-endian magnitudes; negative: optional n flags (the reference takes a signed BigInt).
   out_entries: n x base x (num_limbs+1) entries, num_limbs = ceil(num_digits / logtable), row-major like the returned
   Vec<Vec<Entry>>; one entry = 24 bytes {int128 value (two's complement LE), uint32 mask, uint32 kind} with
   kind 0 = Entry::Scalar (value: the scalar's low 128 bits), 1 = Entry::Bucket(value), 2 = Entry::Limb(value, mask).
   Where the reference would panic the first offending scalar's index goes to *bad_index and the status says why:
   LEMSM_ERR_TOO_MANY_DIGITS (assert :81), LEMSM_ERR_INDEX_OUT_OF_BOUNDS (:98-101 when i % logtable + 1 > num_limbs),
   LEMSM_ERR_ARITH_OVERFLOW (pow / += overflow of i128 or u32, :97-101: a panic in the reference's debug build). */
int lemsm_prepare_scalar_witness_batch(lemsm_ctx* ctx, const uint8_t* scalars, const uint8_t* negative, size_t n,
                                       uint8_t base, uint32_t num_digits, uint32_t logtable, uint8_t* out_entries,
                                       size_t* bad_index);
/* table_entry_by_id(base, id), src/negbase_utils.rs:58-77, for id in [id_begin, id_begin + count), in the BASE field of
   `curve` (the circuit's native field: C::Base, src/config.rs:486); out: count x 4 limbs raw Montgomery.
   As the reference computes it: sum over the set bits k of id of (-base)^(k+1). */
int lemsm_table_entries(lemsm_ctx* ctx, int curve, uint8_t base, uint64_t id_begin, size_t count, uint64_t* out);

/* ---- compute_lhs_witness MSM core ---------------------------------------------------- */
/* carry = sum_j scalars[j]*pts[j] via the Horner-in-(-base) recursion over negabase digits.
   out_carries (optional, d x 12 limbs): carry after each digit position, MSB first -- the
   value the reference negates and pushes at :127.  Scalars must be < isqrt(order)+2 (:97). */
int lemsm_lhs_msm(lemsm_ctx* ctx, int curve, const uint8_t* scalars, const uint64_t* pts_jacobian,
                  size_t n, uint8_t base, uint64_t out_carry[12], uint64_t* out_carries,
                  size_t* bad_index);
int lemsm_lhs_msm_grumpkin(lemsm_ctx* ctx, const uint8_t* scalars, const uint64_t* pts_jacobian,
                           size_t n, uint8_t base, uint64_t out_carry[12], uint64_t* out_carries,
                           size_t* bad_index);
int lemsm_lhs_msm_bn254_g1(lemsm_ctx* ctx, const uint8_t* scalars, const uint64_t* pts_jacobian,
                           size_t n, uint8_t base, uint64_t out_carry[12], uint64_t* out_carries,
                           size_t* bad_index);
/* Device-resident form; points already affine (n x 8 limbs). */
int lemsm_lhs_msm_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine,
                         size_t n, uint8_t base, uint64_t out_carry[12], uint64_t* out_carries,
                         size_t* bad_index);
/* Digit-position-sharded form for multi-GPU. */
int lemsm_lhs_plan(int curve, uint8_t base, uint32_t* num_positions, size_t* partial_bytes_per_position);
int lemsm_lhs_partial_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine,
                             size_t n, uint8_t base, uint32_t pos_begin, uint32_t pos_end,
                             uint8_t* out_partials, size_t* bad_index);
int lemsm_lhs_combine(int curve, uint8_t base, const uint8_t* partials_all_positions,
                      uint64_t out_carry[12], uint64_t* out_carries);

/* ---- divisor witness (the second return value of compute_lhs_witness) ----------------- */
/* compute_divisor_witness_partial / compute_divisor_witness (src/regular_functions_utils.rs:453-480): the regular function
   a(x) + y b(x) vanishing on the n points and on minus their sum, built by the reference's pairwise merge tree
   (Propagation::group_merge :380-405) -- every level one GPU batch: NTTs over bn256::Fr, pointwise products with
   y^2 = x^3 - 17 substituted (:266-273), the two exact divisions of merge (:357) done in the evaluation domain.
   Grumpkin only (C::Base: FftPrecomp, src/precomputed_fft_data.rs:3).  points: n affine points, identity = (0,0).
   out_a / out_b: coefficients, 4 raw-Montgomery limbs each, constant term first; *len_a / *len_b: the reference's vector
   lengths exactly (trailing zero coefficients it carries included); n + 2 coefficients per part always suffice.
   normalise != 0: scaled so that the coefficient of highest pole order (x^i: 2i, y x^i: 2i + 3) is 1.  A witness is only
   defined up to a scalar (linefunc works on whatever projective representative a point has, :285-303, :426-431): the
   normalised form is the representation-independent one.
   require_zero_sum != 0: LEMSM_ERR_SUM_NOT_IDENTITY when the points do not sum to the identity (panic at :478).
   out_point_affine (optional): the tree's output = minus the sum of the points (the `.1` of _partial).
   LEMSM_ERR_ARITH_OVERFLOW: two empty polynomials meet in a product (usize underflow at :55, a panic in the reference:
   four identity points in an aligned group of four). */
int lemsm_divisor_witness(lemsm_ctx* ctx, int curve, const uint64_t* points_affine, size_t n, int require_zero_sum,
                          int normalise, uint64_t* out_a, size_t cap_a, size_t* len_a, uint64_t* out_b, size_t cap_b,
                          size_t* len_b, uint64_t out_point_affine[8]);
int lemsm_divisor_witness_device(lemsm_ctx* ctx, int curve, const void* d_points_affine, size_t n, int require_zero_sum,
                                 int normalise, uint64_t* out_a, size_t cap_a, size_t* len_a, uint64_t* out_b,
                                 size_t cap_b, size_t* len_b, uint64_t out_point_affine[8]);
/* T independent point lists in ONE batch (the merge trees of all lists advance level by level together: the launch
   count of a single tree).  list t = counts[t] affine points, lists concatenated.  out_index: T x 4 entries
   {offset_a, len_a, offset_b, len_b} in elements of 4 limbs into out_coeffs (a list of n points gives n + 1 coefficients in
   all unless identities or P / -P pairs make the reference carry zero padding: 2 sum(counts) + 4 T elements always suffice);
   out_points_affine (optional): T x 8 limbs.  With require_zero_sum the first offending list is in lemsm_last_bad_index. */
int lemsm_divisor_witness_batch(lemsm_ctx* ctx, int curve, const uint64_t* points_affine, const size_t* counts, size_t T,
                                int require_zero_sum, int normalise, uint64_t* out_coeffs, size_t cap_coeffs,
                                size_t* out_index, uint64_t* out_points_affine);
/* Test / profiling aid: levels of the last divisor-witness forest whose forward transforms covered only the odd half of the
   level's domain, the even half being the level below's own evaluations (child-evaluation reuse; option "dw_reuse" 2 = off). */
int lemsm_debug_divisor_last_reuse_levels(const lemsm_ctx* ctx, uint32_t* levels);
/* Device time (ms) of the transform launches of the last divisor-witness call (with option dw_fuse at its default the
   first forward and the last inverse pass of a level also gather from / scatter into the coefficient arrays), their algorithmic bytes (every element
   read once and written once per pass over HBM: 1 pass up to 2^10 elements, 2 up to 2^18, 3 beyond) and their butterfly
   count (one field multiplication each): the figures the HBM and VALU rooflines of the transforms are priced with. */
int lemsm_divisor_last_ntt(const lemsm_ctx* ctx, double* ms, uint64_t* algorithmic_bytes, uint64_t* butterflies);
/* compute_lhs_witness in full (src/argument_witness_calc.rs:87-136): carry = sum_j scalars[j] pts[j] AND the d divisor
   witnesses of :129 (function f = digit iteration d - 1 - f, the reference's `ret.reverse()` order).
   out_coeffs: cap_coeffs field elements (4 limbs each); out_index: d x 4 entries {offset_a, len_a, offset_b, len_b} in
   elements; a function over a list of c points has c + 1 coefficients in all (up to 2 c + 2 when identities or
   opposite points make the reference carry zero padding): 2 d (n + base + 3) elements always suffice. */
int lemsm_lhs_witness(lemsm_ctx* ctx, int curve, const uint8_t* scalars, const uint64_t* pts_jacobian, size_t n,
                      uint8_t base, uint64_t out_carry[12], uint64_t* out_coeffs, size_t cap_coeffs, size_t* out_index,
                      int normalise, size_t* bad_index);
/* The same with the scalars and the AFFINE points (64-byte rows, as the device-pointer MSM entries take them) already in
   HBM and the coefficients left in HBM (d_out_coeffs: device buffer of cap_coeffs 32-byte elements); carry and index
   come back to the host.  For a prover that goes on to commit to the witness on the GPU, and the form bench.py times
   (--workload lhs_witness: inputs and outputs resident). */
int lemsm_lhs_witness_device(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine, size_t n,
                             uint8_t base, uint64_t out_carry[12], void* d_out_coeffs, size_t cap_coeffs, size_t* out_index,
                             int normalise, size_t* bad_index);
/* A rank's share of it: only the functions [f_begin, f_end) of the d of the call are computed (out_index rows of the
   others read length 0).  The d merge trees are independent, so a node shards compute_lhs_witness by digit position with
   no exchange (halo2_liam_eagen_msm_amd.dist.window_range gives the split); every rank runs the MSM core for the carries. */
int lemsm_lhs_witness_device_range(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine, size_t n,
                                   uint8_t base, uint32_t f_begin, uint32_t f_end, uint64_t out_carry[12], void* d_out_coeffs,
                                   size_t cap_coeffs, size_t* out_index, int normalise, size_t* bad_index);
/* Host-clock milliseconds of the four phases of the last lemsm_lhs_witness call (each ends on a stream synchronise):
   [0] the MSM core (upload of scalars and points, digits, buckets, carries), [1] the table of multiples and the d point
   lists, [2] the merge forest (every field operation of the divisor witnesses), [3] the download of the coefficients
   into the caller's buffer (32 B each over PCIe; pageable memory is paged in by this copy). */
int lemsm_lhs_witness_last_phases(const lemsm_ctx* ctx, double out_ms[4]);

/* ---- challenge post-processing helpers (src/config.rs:166-187) ------------------------ */
/* Host-side (a handful of field operations each); field elements are raw Montgomery limbs of the BASE field of `curve`.
   to_curve_x (:166-175): returns c itself when c^3 + b is a square; the reference's loop never changes x (:170-173), so
   for a non-residue it spins forever: LEMSM_ERR_WOULD_NOT_TERMINATE.
   y_from_x (:177-182): the second component of sqrt_alt(x^3 + b): a square root when there is one (*is_square = 1), else
   sqrt(ROOT_OF_UNITY * (x^3 + b)) (*is_square = 0), as ff::Field::sqrt_alt specifies.  Which of the two roots: the one
   Tonelli-Shanks seeded with ROOT_OF_UNITY produces (a^((p+1)/4) for p = 3 mod 4); the upstream source is not in the
   reference tree, so the sign convention is parity-unpinned -- the other root is the negation.
   slope (:184-187): 3 x^2 / (2 y); y == 0 is the reference's invert().unwrap() panic: LEMSM_ERR_DIVISION_BY_ZERO. */
int lemsm_to_curve_x(int curve, const uint64_t c[4], uint64_t out_x[4]);
int lemsm_y_from_x(int curve, const uint64_t x[4], uint64_t out_y[4], int* is_square);
int lemsm_slope(int curve, const uint64_t xy[8], uint64_t out[4]);

/* ---- precompute_multiplicities ------------------------------------------------------- */
/* out[(k-1)] = k * pt for k = 1..base-1 (Jacobian), for each of n points:
   out_jacobian[(j*(base-1) + (k-1))*12 .. +12]. */
int lemsm_precompute_multiplicities(lemsm_ctx* ctx, int curve, const uint64_t* pts_jacobian, size_t n,
                                    uint8_t base, uint64_t* out_jacobian);

/* Same multiples as affine (x,y) pairs, out_affine[(j*(base-1) + (k-1))*8 .. +8]: the form
   src/config.rs:542-560 writes into the fixed table column (the reference inverts once per multiple,
   :550-554; here one batched inversion per input point). */
int lemsm_precompute_multiplicities_affine(lemsm_ctx* ctx, int curve, const uint64_t* pts_jacobian, size_t n,
                                           uint8_t base, uint64_t* out_affine);

/* ---- helpers ------------------------------------------------------------------------- */
/* Jacobian -> canonical comparison form: affine x||y, 32-byte little-endian canonical
   (non-Montgomery) integers; identity = 64 zero bytes. */
int lemsm_jacobian_to_canonical(int curve, const uint64_t jacobian[12], uint8_t out[64]);
/* Sum of `count` Jacobian points on the host (identity for count == 0).  Combines the partial
   results of an MSM whose POINTS were split across callers / GPUs -- the step halo2's
   best_multiexp performs over its per-thread chunk results (`results.iter().fold(identity, a + b)`
   in the halo2_proofs dependency the reference calls at src/argument_witness_calc.rs:144). */
int lemsm_jacobian_sum(int curve, const uint64_t* jacobian, size_t count, uint64_t out[12]);
/* Device memory helpers so that non-HIP hosts can stage resident inputs. */
int lemsm_device_alloc(lemsm_ctx* ctx, size_t bytes, void** out);
int lemsm_device_free(lemsm_ctx* ctx, void* p);
int lemsm_device_upload(lemsm_ctx* ctx, void* dst, const void* src, size_t bytes);
int lemsm_device_download(lemsm_ctx* ctx, void* dst, const void* src, size_t bytes);
/* P_i = (i+1) * Q on the device (affine, n x 8 limbs): synthetic bench/test inputs with a
   known discrete-log relation (sum s_i P_i == (sum s_i (i+1)) Q). */
int lemsm_device_gen_walk(lemsm_ctx* ctx, int curve, const uint64_t q_affine[8], size_t n, void* d_points_out);

/* ---- debug / known-answer hooks used by the parity tests ----------------------------- */
/* Raw Montgomery limbs (x * 2^256, canonical) in and out.  montmul: out = a*b*R^-1.  fieldop: 0 add, 1 sub, 2 neg(a),
   3 inverse(a) (Montgomery inverse), 4 sqr(a).  pointop: 0 = madd(acc XYZZ, q affine), 1 = add(acc XYZZ, q XYZZ),
   2 = madd_abi (the accumulate kernel's second form; an alias of 0 in the strict arithmetic).
   They run the arithmetic option "field" selects: 0, the default, is the lazy radix-2^29 field of the hot kernels
   (Field29 / XYZZ29; values enter through from_abi and leave through div32 + canon; op 3 is inv_lazy, the inversion of
   every batched-inversion kernel); 1 is the strict 32-bit-limb field.  Lazy field only: fieldop 5 = mul2(a, b, b, a) =
   2ab/R, 6 = sqr_addhi = a^2/R + b, 7 = mul_addhi = ab/R + b, 9 = mul32 round trip = a. */
int lemsm_debug_montmul(lemsm_ctx* ctx, int curve, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int lemsm_debug_fieldop(lemsm_ctx* ctx, int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int lemsm_debug_pointop(lemsm_ctx* ctx, int curve, int op, const uint64_t* acc_xyzz, const uint64_t* q,
                        uint64_t* out_xyzz, size_t n);
/* Plain transform over bn256::Fr of nseq sequences of 2^logn elements, natural order in and out, with the reference's
   omega = FftPrecomp::omega_pow(S - logn) (src/regular_functions_utils.rs:111-124); the inverse is scaled by 1/N. */
int lemsm_debug_ntt(lemsm_ctx* ctx, const uint64_t* in, uint64_t* out, size_t nseq, uint32_t logn, int inverse);
/* One-GPU rehearsal of the sharded entries: the pipelines of all `world` ranks run one after the other on this
   context and their record areas are placed where the all-gather would put them; everything but the ncclAllGather
   call itself is the code of lemsm_*_sharded_device. */
int lemsm_debug_msm_sharded_sim(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine, size_t n,
                                int world, uint64_t out_jacobian[12]);
int lemsm_debug_lhs_sharded_sim(lemsm_ctx* ctx, int curve, const void* d_scalars, const void* d_points_affine, size_t n,
                                uint8_t base, int world, uint64_t out_carry[12], uint64_t* out_carries, size_t* bad_index);

#ifdef __cplusplus
}
#endif
#endif /* LEMSM_H */
