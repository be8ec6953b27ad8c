// jubjub29.h — JubJub (twisted Edwards a = -1 over Fq) group law on fe29 lanes.
//
// Formulas are the extended-coordinate ones dusk-jubjub's operators implement (SURVEY.md
// Appendix A.3), i.e. what `GENERATOR_EXTENDED * u`, `pk * c`, `+` and `==` evaluate at
// /root/reference/src/keys/public.rs:127-129, :236-243, :411-414:
//   double : 4S + 3M (2uv as (u+v)^2 - u^2 - v^2)
//   add (extended niels) : 8M     add (affine niels, z = 1) : 7M
// They are complete on the whole curve (a = -1 is a square, d is not), so identity, small-order
// and repeated points need no special case — same as the reference.
//
// Lazy-reduction bookkeeping (bounds in units of q, "N" = output of fe_mul, < 1.5 q,
// limbs < 2^29): every fe_add result feeds a multiply directly (limbs < 2^30); every
// subtraction goes through a biased subtract + one carry pass, except the ones marked "raw" whose
// every consumer is a multiplication by a carried operand (or fe_sub4w, whose bias is lifted to
// dominate them).  The worst case of each formula is
// annotated inline, PROVED for all inputs by the interval prover tests/fe29_bounds.py and
// re-checked numerically by tests/test_fe29_model.py.
#pragma once
#include "fe29.h"

namespace dsv {

struct Ext {  // (u, v, z, t1, t2): u = U/Z, v = V/Z, t1*t2 = UV/Z
  Fe u, v, z, t1, t2;
};
struct Niels {  // (v+u, v-u, z, 2d*t)
  Fe vpu, vmu, z, t2d;
};
struct ANiels {  // affine niels: z = 1
  Fe vpu, vmu, t2d;
};

DSV_DEV Ext ext_identity() {
  Ext r;
  r.u = fe_zero();
  r.v = fe_one();
  r.z = fe_one();
  r.t1 = fe_zero();
  r.t2 = fe_zero();
  return r;
}
DSV_DEV Ext ext_from_affine(const Fe& u, const Fe& v) {
  Ext r;
  r.u = u;
  r.v = v;
  r.z = fe_one();
  r.t1 = u;
  r.t2 = v;
  return r;
}
DSV_DEV Niels niels_identity() {
  Niels n;
  n.vpu = fe_one();
  n.vmu = fe_one();
  n.z = fe_one();
  n.t2d = fe_zero();
  return n;
}

// in : u, v, z  N (< 1.5q)            (t1, t2 unused)
// out: u, v, z  N (u < 1.44q);  t1 < 5.1q carried, t2 < 2.1q (limbs < 2^30)
// 2uv as (u+v)^2 - (u^2 + v^2): 117 MADs + an addition and a biased subtraction instead of a
// multiplication's 153 (+0.26 % single, +0.9 % double, profiles/r02/ab_double_sqr.txt; bounds:
// tests/fe29_bounds.py)
DSV_DEV Fe ext_two_uv(const Fe& u, const Fe& v, const Fe& vpu) {
  return fe_sub4w(fe_sqr(fe_add(u, v)), vpu);  // < 5.1, carried
}
DSV_DEV Ext ext_double(const Ext& p) {
  Fe uu = fe_sqr(p.u);                      // < 1.04
  Fe vv = fe_sqr(p.v);                      // < 1.04
  Fe zz2 = fe_dbl(fe_sqr(p.z));             // < 2.1, limbs < 2^30
  Fe vpu = fe_add(vv, uu);                  // < 2.1, limbs < 2^30
  Fe cu = ext_two_uv(p.u, p.v, vpu);        // 2uv
  Fe vmu = fe_sub2_raw(vv, uu);             // < 3.1, limbs < 2^31 (partners vpu < 2^30, ct carried)
  Fe ct = fe_sub4w(zz2, vmu);               // < 6.1, carried
  Ext r;
  r.u = fe_mul(cu, ct);                     // 5.1*6.1*0.01414+1 = 1.44
  r.v = fe_mul(vpu, vmu);                   // < 1.1
  r.z = fe_mul(vmu, ct);                    // < 1.3
  r.t1 = cu;
  r.t2 = vpu;
  return r;
}

// 2 * (u, v) for an affine point (z = 1): 2 z^2 is the constant 2
DSV_DEV Ext ext_double_affine(const Fe& u, const Fe& v) {
  Fe uu = fe_sqr(u);
  Fe vv = fe_sqr(v);
  Fe zz2 = fe_dbl(fe_one());
  Fe vpu = fe_add(vv, uu);
  Fe cu = ext_two_uv(u, v, vpu);
  Fe vmu = fe_sub2_raw(vv, uu);
  Fe ct = fe_sub4w(zz2, vmu);
  Ext r;
  r.u = fe_mul(cu, ct);
  r.v = fe_mul(vpu, vmu);
  r.z = fe_mul(vmu, ct);
  r.t1 = cu;
  r.t2 = vpu;
  return r;
}
// doubling that keeps only (u, v, z): inside a run of doublings nobody reads t1/t2
DSV_DEV void ext_double_uvz(Fe& u, Fe& v, Fe& z) {
  Fe uu = fe_sqr(u);
  Fe vv = fe_sqr(v);
  Fe zz2 = fe_dbl(fe_sqr(z));
  Fe vpu = fe_add(vv, uu);
  Fe cu = ext_two_uv(u, v, vpu);
  Fe vmu = fe_sub2_raw(vv, uu);
  Fe ct = fe_sub4w(zz2, vmu);
  u = fe_mul(cu, ct);
  v = fe_mul(vpu, vmu);
  z = fe_mul(vmu, ct);
}

// shared tail of both additions.  a, b, c < 1.2 (N);  d < 3.0 with limbs < 2^30
DSV_DEV Ext ext_add_tail(const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
  Fe cu = fe_sub2_raw(b, a);                // < 3.2, limbs < 1.5 * 2^30 (partners cz, ct, t2 < 2^30)
  Fe cv = fe_add(b, a);                     // < 2.4, limbs < 2^30
  Fe cz = fe_add(d, c);                     // < 4.2, limbs < 1.5 * 2^30 (r02: no carry pass, proved)
  Fe ct = fe_sub2(d, c);                    // < 5.0
  Ext r;
  r.u = fe_mul(cu, ct);                     // < 1.23
  r.v = fe_mul(cv, cz);                     // < 1.15
  r.z = fe_mul(cz, ct);                     // < 1.3
  r.t1 = cu;
  r.t2 = cv;
  return r;
}
// p: u,v,z N; t1 < 3.2 (limbs < 1.5 * 2^30 when it comes from an addition), t2 < 2.4 limbs < 2^30.
// n: all N / carried, limbs < 2^29 + 3
DSV_DEV Ext ext_add_niels(const Ext& p, const Niels& n) {
  Fe a = fe_mul(fe_sub2_raw(p.v, p.u), n.vmu);  // 3.5*1.5 -> < 1.08; raw limbs x table limbs < 2^29 + 3
  Fe b = fe_mul(fe_add(p.v, p.u), n.vpu);   // 3.0*1.5 -> < 1.07
  Fe c = fe_mul(fe_mul(p.t1, p.t2), n.t2d); // (5.2*2.4 -> 1.18) * 1.5 -> < 1.03
  Fe d = fe_dbl(fe_mul(p.z, n.z));          // < 2.1, limbs < 2^30
  return ext_add_tail(a, b, c, d);
}
DSV_DEV Ext ext_add_aniels(const Ext& p, const ANiels& n) {
  Fe a = fe_mul(fe_sub2_raw(p.v, p.u), n.vmu);
  Fe b = fe_mul(fe_add(p.v, p.u), n.vpu);
  Fe c = fe_mul(fe_mul(p.t1, p.t2), n.t2d);
  Fe d = fe_dbl(p.z);                       // < 3.0, limbs < 2^30
  return ext_add_tail(a, b, c, d);
}
// [ p + n == O ] without finishing the addition: the sum is ((b-a)(d-c), (b+a)(d+c), (d+c)(d-c))
// and z3 != 0 for points of the curve (the formulas are complete), so
//   u3 == 0 and v3 == z3   <=>   b == a  and  b + a == d - c
// — the last addition of a verification costs 4 multiplications instead of 7.
DSV_DEV bool ext_add_aniels_is_identity(const Ext& p, const ANiels& n) {
  const Fe a = fe_mul(fe_sub2_raw(p.v, p.u), n.vmu);
  const Fe b = fe_mul(fe_add(p.v, p.u), n.vpu);
  const Fe c = fe_mul(fe_mul(p.t1, p.t2), n.t2d);
  const Fe d = fe_dbl(p.z);
  return (bool)((int)fe_equal(b, a) & (int)fe_equal(fe_add(b, a), fe_sub2(d, c)));
}
// O + n as an extended point: the addition formulas with p = (0, 1, 1, 0, 0) — a = n.vmu,
// b = n.vpu, c = 0, d = 2 n.z — i.e. three multiplications instead of eight
DSV_DEV Ext ext_from_niels(const Niels& n) {
  Fe cu = fe_sub4(n.vpu, n.vmu);            // 2u; a table value v-u can reach 3.5 q: 4q bias, carried
  Fe cv = fe_add(n.vpu, n.vmu);             // 2v: limbs < 2^30
  Fe d = fe_dbl(n.z);                       // cz = ct = 2z
  Ext r;
  r.u = fe_mul(cu, d);
  r.v = fe_mul(cv, d);
  r.z = fe_sqr(d);
  r.t1 = cu;
  r.t2 = cv;
  return r;
}
// the same two with the product tt = p.t1 * p.t2 handed in: a table build needs it twice per
// entry (once for the entry's own 2d*t, once inside the addition that produces the next entry)
DSV_DEV Ext ext_add_aniels_t(const Ext& p, const Fe& tt, const ANiels& n) {
  Fe a = fe_mul(fe_sub2_raw(p.v, p.u), n.vmu);
  Fe b = fe_mul(fe_add(p.v, p.u), n.vpu);
  Fe c = fe_mul(tt, n.t2d);
  Fe d = fe_dbl(p.z);
  return ext_add_tail(a, b, c, d);
}
// sum = p + n AND diff = p - n for an affine niels n, tt = p.t1 * p.t2 handed in: -n swaps n.vpu /
// n.vmu and negates c, which only swaps the roles of cz = d + c and ct = d - c in the tail — the two
// results share c, d and z = cz * ct: 10 multiplications instead of 12 (the joint table build has
// three such pairs).  Bounds as ext_add_tail; the carried copy of cz takes ct's place next to a raw
// cu (tests/fe29_bounds.py: ext_add_sub_aniels).
DSV_DEV void ext_add_sub_aniels_t(Ext& sum, Ext& diff, const Ext& p, const Fe& tt, const ANiels& n) {
  const Fe pm = fe_sub2_raw(p.v, p.u), pp = fe_add(p.v, p.u);
  const Fe a = fe_mul(pm, n.vmu), b = fe_mul(pp, n.vpu);
  const Fe a2 = fe_mul(pm, n.vpu), b2 = fe_mul(pp, n.vmu);
  const Fe c = fe_mul(tt, n.t2d);
  const Fe d = fe_dbl(p.z);
  const Fe cu = fe_sub2_raw(b, a), cv = fe_add(b, a), cz = fe_add(d, c), ct = fe_sub2(d, c);
  sum.u = fe_mul(cu, ct);
  sum.v = fe_mul(cv, cz);
  sum.z = fe_mul(cz, ct);
  sum.t1 = cu;
  sum.t2 = cv;
  const Fe cu2 = fe_sub2_raw(b2, a2), cv2 = fe_add(b2, a2);
  diff.u = fe_mul(cu2, fe_carry(cz));
  diff.v = fe_mul(cv2, ct);
  diff.z = sum.z;
  diff.t1 = cu2;
  diff.t2 = cv2;
}
DSV_DEV Niels ext_to_niels_t(const Ext& p, const Fe& tt) {
  Niels n;
  n.vpu = fe_carry(fe_add(p.v, p.u));
  n.vmu = fe_sub2(p.v, p.u);
  n.z = p.z;
  n.t2d = fe_mul(tt, fe_const(kD2));
  return n;
}
// to_niels: (v+u, v-u, z, t1*t2*2d), all brought to N / carried form for table storage
DSV_DEV Niels ext_to_niels(const Ext& p) {
  Niels n;
  n.vpu = fe_carry(fe_add(p.v, p.u));       // < 3.0, limbs < 2^29 + 8
  n.vmu = fe_sub2(p.v, p.u);                // < 3.5
  n.z = p.z;
  n.t2d = fe_mul(fe_mul(p.t1, p.t2), fe_const(kD2));
  return n;
}

// projective equality against an affine point (Ru, Rv), i.e. `point_1.eq(&sig.R())` with
// R.z = 1: u1 * 1 == Ru * z1  and  v1 * 1 == Rv * z1
DSV_DEV bool ext_eq_affine(const Ext& p, const Fe& ru, const Fe& rv) {
  bool e1 = fe_equal(p.u, fe_mul(ru, p.z));
  bool e2 = fe_equal(p.v, fe_mul(rv, p.z));
  return e1 & e2;
}

}  // namespace dsv
