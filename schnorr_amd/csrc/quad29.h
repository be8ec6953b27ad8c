// quad29.h — JubJub group law with FOUR lanes per point (small batches: latency, not throughput).
//
// A batch of 1024 signatures is 16 waves on a chip with 1024 SIMDs: with one lane per signature the
// call takes one lane's whole serial instruction stream (~0.5 M instructions, ~1.1 ms) while 98 %
// of the machine idles.  Parallelism INSIDE one field multiplication does not pay (a
// limbs-across-lanes multiplier is 2.2x slower per multiplication, tools/microbench/coop_mul.hip);
// parallelism ACROSS the multiplications of one group operation does: a doubling is 4 independent
// products followed by 4, a (niels) addition 4 followed by 4.  So a quad of lanes (a DPP quad: lanes
// 4k .. 4k+3) owns one point; every lane holds the whole point (u, v, z, t = u*v/z), each runs ONE
// of the four products of a step on its own operands, and the four results are broadcast through
// the quad with v_mov_b32_dpp quad_perm (measured 4.1 cycles each, profiles/r02/valu_rates2.txt;
// 36 per step against ~780 cycles for the multiplication).  The cheap limb-wise additions are done
// redundantly by all four lanes.  Two multiplication-times per operation instead of 7 / 8.
//
// Same formulas, same lazy-reduction bounds as jubjub29.h (every product here is one of the
// products there, with operands taken from the same expressions); the extended coordinate is
// carried as the single product t = t1*t2, computed by the lane that would otherwise idle.
#pragma once
#include "jubjub29.h"

namespace dsv {

struct QExt {  // replicated in the four lanes of a quad
  Fe u, v, z, t;
};

// value of lane k (0..3) of this lane's quad, for every limb
template <int K>
DSV_DEV Fe quad_bcast(const Fe& x) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++)
    r.l[i] = (u32)__builtin_amdgcn_mov_dpp((int)x.l[i], K * 0x55, 0xf, 0xf, true);
  return r;
}
// operand of lane q: x0 in lane 0, x1 in lane 1, ...
DSV_DEV Fe quad_pick(int q, const Fe& x0, const Fe& x1, const Fe& x2, const Fe& x3) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 lo = (q & 1) ? x1.l[i] : x0.l[i];
    const u32 hi = (q & 1) ? x3.l[i] : x2.l[i];
    r.l[i] = (q & 2) ? hi : lo;
  }
  return r;
}

DSV_DEV QExt qext_identity() {
  QExt r;
  r.u = fe_zero();
  r.v = fe_one();
  r.z = fe_one();
  r.t = fe_zero();
  return r;
}

// p <- 2p.  Products of ext_double (jubjub29.h): uu, vv, zz, uv | u' = cu*ct, v' = vpu*vmu,
// z' = vmu*ct, t' = cu*vpu.  WANT_T = false inside a run of doublings (nobody reads t).
template <bool WANT_T>
DSV_DEV void qext_double(QExt& p, int q) {
  // lanes 0..2 square u, v, z; lane 3 multiplies u * v  (a general multiplication everywhere: the
  // instruction stream is shared)
  const Fe x = quad_pick(q, p.u, p.v, p.z, p.u);
  const Fe y = quad_pick(q, p.u, p.v, p.z, p.v);
  const Fe r = fe_mul(x, y);
  const Fe uu = quad_bcast<0>(r), vv = quad_bcast<1>(r), zz = quad_bcast<2>(r), uv = quad_bcast<3>(r);
  const Fe zz2 = fe_dbl(zz);
  const Fe cu = fe_dbl(uv);
  const Fe vpu = fe_add(vv, uu);
  const Fe vmu = fe_sub2_raw(vv, uu);
  const Fe ct = fe_sub4w(zz2, vmu);
  const Fe x2 = quad_pick(q, cu, vpu, vmu, cu);
  const Fe y2 = quad_pick(q, ct, vmu, ct, vpu);
  const Fe r2 = fe_mul(x2, y2);
  p.u = quad_bcast<0>(r2);
  p.v = quad_bcast<1>(r2);
  p.z = quad_bcast<2>(r2);
  if (WANT_T) p.t = quad_bcast<3>(r2);
}

// shared tail of the additions: a, b, c, d as in ext_add_tail
DSV_DEV void qext_add_tail(QExt& p, int q, const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
  const Fe cu = fe_sub2_raw(b, a);
  const Fe cv = fe_add(b, a);
  const Fe cz = fe_add(d, c);
  const Fe ct = fe_sub2(d, c);
  const Fe x = quad_pick(q, cu, cv, cz, cu);
  const Fe y = quad_pick(q, ct, cz, ct, cv);
  const Fe r = fe_mul(x, y);
  p.u = quad_bcast<0>(r);
  p.v = quad_bcast<1>(r);
  p.z = quad_bcast<2>(r);
  p.t = quad_bcast<3>(r);
}
// p <- p + n (extended niels): a = (v-u)*n.vmu, b = (v+u)*n.vpu, c = t*n.t2d, d = 2*z*n.z
DSV_DEV void qext_add_niels(QExt& p, int q, const Niels& n) {
  const Fe x = quad_pick(q, fe_sub2_raw(p.v, p.u), fe_add(p.v, p.u), p.t, p.z);
  const Fe y = quad_pick(q, n.vmu, n.vpu, n.t2d, n.z);
  const Fe r = fe_mul(x, y);
  const Fe a = quad_bcast<0>(r), b = quad_bcast<1>(r), c = quad_bcast<2>(r);
  const Fe d = fe_dbl(quad_bcast<3>(r));
  qext_add_tail(p, q, a, b, c, d);
}
// p <- p + n (affine niels, z = 1): lane 3 has nothing to multiply in the first step
DSV_DEV void qext_add_aniels(QExt& p, int q, const ANiels& n) {
  const Fe x = quad_pick(q, fe_sub2_raw(p.v, p.u), fe_add(p.v, p.u), p.t, p.t);
  const Fe y = quad_pick(q, n.vmu, n.vpu, n.t2d, n.t2d);
  const Fe r = fe_mul(x, y);
  const Fe a = quad_bcast<0>(r), b = quad_bcast<1>(r), c = quad_bcast<2>(r);
  const Fe d = fe_dbl(p.z);
  qext_add_tail(p, q, a, b, c, d);
}

}  // namespace dsv
