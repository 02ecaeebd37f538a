// Straight-line body of one mixed addition (XYZZ29 common path) for the ISA budget table of DESIGN.md 4.2:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o madd_body.s madd_body.hip && python ../isa_budget.py madd_body.s k_body --whole
#include "../../halo2_liam_eagen_msm_amd/csrc/xyzz29.cuh"
using namespace lemsm;
typedef Field29<Fq29Params> F; typedef XYZZ29<F> G;
// straight-line body of one mixed addition in k_accum1's ABI form (common path only)
__global__ void k_body(const int* in, int* out, int n) {
  G::pt acc; F::fe x, y;
  const int* p = in + threadIdx.x * 54;
  for (int i = 0; i < 9; i++) { acc.x.l[i] = p[i]; acc.y.l[i] = p[9+i]; acc.zz.l[i] = p[18+i]; acc.zzz.l[i] = p[27+i]; x.l[i] = p[36+i]; y.l[i] = p[45+i]; }
  asm volatile("; BODY_BEGIN");
  {
    F::fe U2, S2, P, R, PP;
    F::mul(U2, x, acc.zz);
    F::mul(S2, y, acc.zzz);
    F::sub(P, U2, acc.x);
    F::sub(R, S2, acc.y);
    F::sqr(PP, P);
    F::fe PPP, Q, t, nY;
    F::mul(PPP, P, PP);
    F::mul(Q, acc.x, PP);
    G::hi_term(t, PPP, Q);
    F::neg(nY, acc.y);
    F::sqr_addhi(acc.x, R, t);
    F::sub(t, Q, acc.x);
    F::mul2(acc.y, R, t, nY, PPP);
    F::mul(acc.zz, acc.zz, PP);
    F::mul(acc.zzz, acc.zzz, PPP);
  }
  asm volatile("; BODY_END");
  int* o = out + threadIdx.x * 36;
  for (int i = 0; i < 9; i++) { o[i] = acc.x.l[i]; o[9+i] = acc.y.l[i]; o[18+i] = acc.zz.l[i]; o[27+i] = acc.zzz.l[i]; }
}
