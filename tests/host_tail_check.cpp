// CPU-only check of the product's host tail (halo2_liam_eagen_msm_amd/csrc/hosttail.hpp, hostmath.hpp, hostpool.hpp),
// built with -fsanitize=address,undefined by tests/test_host_logic.py.  No GPU, no HIP: everything the library does on the
// host after the kernels have run is driven here on synthetic data, and the pyramid task tables the kernels execute are
// simulated on integers (a "point" is its discrete logarithm), so that an indexing error in the plan shows up without a GPU.
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <numeric>
#include <random>

#include "../halo2_liam_eagen_msm_amd/csrc/hostpool.hpp"
#include "../halo2_liam_eagen_msm_amd/csrc/hosttail.hpp"

using namespace lemsm;
typedef unsigned __int128 u128;

#define CHECK(cond)                                                                 \
  do {                                                                              \
    if (!(cond)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); exit(1); } \
  } while (0)

static std::mt19937_64 rng(12345);

template <class P>
host::fe rand_fe() {
  host::fe a; for (int i = 0; i < 4; i++) a.l[i] = rng();
  a.l[3] &= 0x0fffffffffffffffULL;        // < 2^252 < N
  return a;
}

// ---- field and group identities -------------------------------------------------------------------
template <class P>
void field_checks() {
  typedef host::HF<P> F;
  for (int it = 0; it < 200; it++) {
    host::fe a = rand_fe<P>(), b = rand_fe<P>(), c = rand_fe<P>();
    CHECK(F::eq(F::sub(F::add(a, b), b), a));
    CHECK(F::eq(F::mul(a, F::add(b, c)), F::add(F::mul(a, b), F::mul(a, c))));
    CHECK(F::eq(F::mul(a, b), F::mul(b, a)));
    CHECK(F::eq(F::add(a, F::neg(a)), F::zero()));
    if (!F::is_zero(a)) CHECK(F::eq(F::mul(a, F::inv(a)), F::one()));
    CHECK(F::eq(F::mul(a, F::one()), a));
  }
}

// a point with known discrete log: k * G, G = (1, 2) on y^2 = x^3 + 3 (BN254) or (1, sqrt(-16)) on Grumpkin
template <class P>
host::pt generator(const u64 gy[4]) {
  typedef host::HF<P> F;
  host::pt g; g.x = F::one();
  host::fe y; memcpy(y.l, gy, 32);
  host::fe r2; memcpy(r2.l, P::R2, 32);
  g.y = F::mul(y, r2);                    // canonical -> Montgomery
  g.zz = F::one(); g.zzz = F::one();
  return g;
}

template <class P>
bool same_point(const host::pt& a, const host::pt& b) {
  typedef host::HG<P> G;
  u64 ja[8], jb[8]; G::to_affine(a, ja); G::to_affine(b, jb);
  return memcmp(ja, jb, 64) == 0;
}

template <class P>
void group_checks(const u64 gy[4]) {
  typedef host::HG<P> G;
  host::pt g = generator<P>(gy);
  // on-curve sanity through the group law: 2G by doubling == G + G by the addition's doubling branch
  CHECK(same_point<P>(G::dbl(g), G::add(g, g)));
  CHECK(G::is_identity(G::add(g, G::neg(g))));
  host::pt acc = G::identity();
  for (u32 k = 1; k <= 40; k++) { acc = G::add(acc, g); CHECK(same_point<P>(acc, G::mul_small(g, k))); }
  // Jacobian round trip
  u64 jac[12]; G::to_jacobian(acc, jac);
  CHECK(same_point<P>(G::from_jacobian(jac), acc));
  u64 two[24]; G::to_jacobian(g, two); G::to_jacobian(G::dbl(g), two + 12);
  u64 sum[12]; jacobian_sum_t<P>(two, 2, sum);
  CHECK(same_point<P>(G::from_jacobian(sum), G::mul_small(g, 3)));
  // window_sum: total + sum_l 2^l U_l with U_l = u_l G, total = t G
  const u32 L = 7;
  std::vector<host::pt> rec(L + 1);
  u64 expect = 5; rec[0] = G::mul_small(g, 5);
  for (u32 l = 0; l < L; l++) { u32 u = 3 + 2 * l; rec[1 + l] = G::mul_small(g, u); expect += (u64)u << l; }
  CHECK(same_point<P>(window_sum<P>(rec.data(), L), G::mul_small(g, (uint32_t)expect)));
  // msm_combine_windows: sum_w 2^(c w) S_w with small c
  std::vector<host::pt> sums(4);
  u64 e2 = 0; for (u32 w = 0; w < 4; w++) { sums[w] = G::mul_small(g, 1 + w); e2 += (u64)(1 + w) << (3 * w); }
  u64 out[12]; msm_combine_windows<P>(3, 4, sums.data(), out);
  CHECK(same_point<P>(G::from_jacobian(out), G::mul_small(g, (uint32_t)e2)));
  // lhs_combine_positions: Horner in -base; sum_i s_i (-B)^i for small values, compared through signed arithmetic
  const u32 base = 5, d = 6; std::vector<host::pt> pos(d); long long ev = 0, pw = 1;
  for (u32 i = 0; i < d; i++) { pos[i] = G::mul_small(g, 2 + i); ev += (long long)(2 + i) * pw; pw *= -(long long)base; }
  u64 carry[12]; std::vector<u64> carries(12 * d);
  lhs_combine_positions<P>(base, d, pos.data(), carry, carries.data());
  host::pt want = ev >= 0 ? G::mul_small(g, (uint32_t)ev) : G::neg(G::mul_small(g, (uint32_t)(-ev)));
  CHECK(same_point<P>(G::from_jacobian(carry), want));
  CHECK(memcmp(carries.data() + 12 * (d - 1), carry, 96) == 0);
}

// ---- raw 29-bit records -----------------------------------------------------------------------------
// a value v (< N) as a LAZY record coordinate: limbs of v * 2^5 ... the device keeps x*2^261 while the host wants x*2^256,
// so a record limb vector encodes 32 * a (mod N) plus a multiple of N, limbs 0..7 in [0, 2^29), limb 8 signed
template <class P>
void raw29_checks() {
  typedef host::HF<P> F;
  for (int it = 0; it < 300; it++) {
    host::fe a = rand_fe<P>();
    // t = 32 * a + k * N as an integer, k in [-7, 7] kept within |V| < 8N, written as 9 signed 29-bit limbs
    host::fe a32 = a; for (int i = 0; i < 5; i++) a32 = F::dbl(a32);   // 32 a mod N, canonical
    int k = (int)(rng() % 14) - 7;
    // integer arithmetic on 5 x 64-bit words, two's complement
    long long w[6] = {0};
    __int128 cy = 0;
    for (int i = 0; i < 4; i++) { __int128 t = (__int128)(u128)a32.l[i] + (__int128)k * (__int128)(u128)P::N[i] + cy; w[i] = (long long)(u64)t; cy = t >> 64; }
    w[4] = (long long)cy;
    int32_t limbs[9];
    // extract 29-bit limbs of the signed 320-bit integer (arithmetic shift)
    for (int i = 0; i < 9; i++) {
      int bit = 29 * i, wi = bit >> 6, sh = bit & 63;
      u128 lo = (u128)(u64)w[wi] | ((u128)(u64)w[wi + 1] << 64);
      u64 v = (u64)(lo >> sh);
      if (i < 8) limbs[i] = (int32_t)(v & ((1u << 29) - 1));
      else { __int128 full = (__int128)(((u128)(u64)w[4] << 64) | (u128)(u64)w[3]); limbs[i] = (int32_t)(long long)(full >> (29 * 8 - 192)); }   // sign-extended top limb
    }
    host::fe got = reduce_raw29<P>(limbs);
    CHECK(F::eq(got, a));
  }
  // identity record: all-zero limbs convert to the identity point
  std::vector<char> raw(160 * 2, 0);
  host::pt out[2];
  from_device_records_t<P, true, 160>(raw.data(), 2, out);
  CHECK(host::HG<P>::is_identity(out[0]) && host::HG<P>::is_identity(out[1]));
}

// ---- pyramid task tables simulated on integers -----------------------------------------------------------
// arena of u64 "points" (discrete logs mod 2^64); executes every step exactly as k_pyramid / k_copy_points do
void pyramid_sim(u32 nb, u32 gw, bool scaled) {
  u32 nbp = 1; while (nbp < nb) nbp <<= 1; if (nbp < 2) nbp = 2;
  u32 L = 0; while ((1u << L) < nbp) L++;
  const u32 nbw = nb + (rng() % 3);       // key stride >= nb
  const u32 NBpad = nbw * gw + 5;
  ArenaLayout ar = make_arena(NBpad, nbp, gw, L);
  PyrPlan pp = make_pyr_plan(ar, nb, nbw, nbp, L, scaled);
  std::vector<u64> arena(ar.total_points, 0xdeadbeefcafef00dULL);
  std::vector<u64> want(gw, 0);
  for (u32 w = 0; w < gw; w++)
    for (u32 k = 0; k < nb; k++) { u64 v = rng() >> 20; arena[ar.bucket_off + w * nbw + k] = v; want[w] += (u64)(k + 1) * v; }
  CHECK(pp.steps.size() == L);
  for (u32 s = 1; s <= L; s++) {
    std::vector<u64> next = arena;       // a step reads the previous state only where it does not write: emulate a launch
    for (const PyrTask& tk : pp.steps[s - 1]) {
      CHECK(tk.count <= pp.step_max_count[s - 1] || tk.count == 1);
      for (u32 w = 0; w < gw; w++)
        for (u32 i = 0; i < tk.count; i++) {
          u32 ia = (2 * i) * tk.stride + tk.phase, ib = (2 * i + 1) * tk.stride + tk.phase;
          size_t sa = (size_t)tk.src_off + (size_t)w * tk.src_wstride + ia, sb = (size_t)tk.src_off + (size_t)w * tk.src_wstride + ib;
          u64 a = 0, b = 0;
          if (ia < tk.src_valid) { CHECK(sa < arena.size()); a = arena[sa]; }
          if (ib < tk.src_valid) { CHECK(sb < arena.size()); b = arena[sb]; }
          size_t d = (size_t)tk.dst_off + (size_t)w * tk.dst_wstride + i;
          CHECK(d < arena.size());
          CHECK(d >= ar.apyr_off);         // never into the bucket sums
          next[d] = a + b;
        }
    }
    arena.swap(next);
  }
  if (!(L == 1 && scaled))
    for (u32 w = 0; w < gw; w++) {
      u64 v = 0;
      if (pp.copy.src_idx < pp.copy.src_valid_idx) v = arena[(size_t)pp.copy.src_off + (size_t)w * pp.copy.src_wstride + pp.copy.src_idx];
      arena[(size_t)pp.copy.dst_off + (size_t)w * pp.copy.dst_wstride] = v;
    }
  // the records [total, U_0..U_{L-1}] of every window fold to sum_k (k+1) B_k
  for (u32 w = 0; w < gw; w++) {
    const u64* rec = arena.data() + ar.out_off + (size_t)w * (L + 1);
    u64 acc = 0;
    for (int l = (int)L - 1; l >= 0; l--) acc = 2 * acc + rec[1 + l];
    acc += rec[0];
    CHECK(acc == want[w]);
  }
}

void plan_checks() {
  for (u32 nb : {1u, 2u, 3u, 4u, 5u, 7u, 8u, 15u, 16u, 17u, 31u, 64u, 100u, 254u, 1024u, 4096u})
    for (u32 gw : {1u, 3u}) { pyramid_sim(nb, gw, false); pyramid_sim(nb, gw, true); }
  // merge queue layout: regions disjoint and inside the stated capacities
  for (u32 nthr1 : {1u, 7u, 64u, 1000u, 393216u, 983040u})
    for (u32 slice : {33u, 512u, 65536u}) {
      MqLayout m = make_mq_layout_t(nthr1, slice, 2048);
      CHECK(m.offS == 0 && m.offM == m.offS + m.capS && m.offL == m.offM + m.capM && m.offF == m.offL + m.capL);
      CHECK(m.capS >= nthr1 / 2 && m.capM >= nthr1 / 8 && m.capL >= nthr1 / 32 + nthr1 / slice && m.capF >= nthr1 / slice && m.capP >= 2 * (nthr1 / slice));
    }
  // multi-GPU status agreement: all ranks derive the same verdict from the same gathered words
  {
    u32 ok4[4] = {0, 0, 0, 0}, one_bad[4] = {0, 7, 0, 0}, two_bad[4] = {0, 7, 0, 5};
    int fr = 99;
    CHECK(merge_rank_status(ok4, 4, 2, 0, &fr) == 0 && fr == -1);
    for (int me = 0; me < 4; me++) {
      int rc = merge_rank_status(one_bad, 4, me, me == 1 ? 7 : 0, &fr);
      CHECK(fr == 1 && rc == (me == 1 ? 7 : 9));
      rc = merge_rank_status(two_bad, 4, me, (int)two_bad[me], &fr);
      CHECK(fr == 1 && rc == (two_bad[me] ? (int)two_bad[me] : 9));
    }
  }
  for (u32 W : {1u, 15u, 16u, 33u})
    for (int world : {1, 2, 3, 8, 40}) {
      u32 prev = 0;
      for (int r = 0; r < world; r++) { u32 a, b; shard_range(W, world, r, a, b); CHECK(a == prev && b >= a); prev = b; }
      CHECK(prev == W);
    }
}

// ---- worker pool ---------------------------------------------------------------------------------
void pool_checks() {
  for (int workers : {0, 1, 3, 7}) {
    host::Pool pool(workers);
    for (int round = 0; round < 200; round++) {
      const int n = 1 + (int)(rng() % 40);
      std::vector<int> hits(n, 0);
      std::atomic<long> total{0};
      pool.run(n, [&](int i) { hits[i]++; total += i; });
      for (int i = 0; i < n; i++) CHECK(hits[i] == 1);
      CHECK(total == (long)n * (n - 1) / 2);
    }
  }
}

int main() {
  static const u64 GY_BN254[4] = {2, 0, 0, 0};
  // Grumpkin generator y: 0x2cf135e7506a45d632d270d45f1181294833fc48d823f272c
  static const u64 GY_GRUMPKIN[4] = {0x833fc48d823f272cULL, 0x2d270d45f1181294ULL, 0xcf135e7506a45d63ULL, 0x0000000000000002ULL};
  field_checks<host::FqParams64>(); field_checks<host::FrParams64>();
  group_checks<host::FqParams64>(GY_BN254); group_checks<host::FrParams64>(GY_GRUMPKIN);
  raw29_checks<host::FqParams64>(); raw29_checks<host::FrParams64>();
  plan_checks();
  pool_checks();
  printf("host tail ok\n");
  return 0;
}
