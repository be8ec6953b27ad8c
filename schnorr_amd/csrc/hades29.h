// hades29.h — Hades permutation (width 5, x^5 S-box, 4 + 59 + 4 rounds) and the Poseidon
// sponge + 250-bit truncation that dusk-poseidon's `sponge::truncated::hash` evaluates for
// `challenge_hash` / `challenge_hash_double` (/root/reference/src/signatures.rs:127-134,
// :275-290; semantics SURVEY.md Appendix A.4).  One lane = one hash.
//
// Round constants and the MDS matrix are wave-uniform: they sit in __constant__ memory and are
// fetched with scalar loads, so the MADs take them as SGPR operands (no VGPR cost).
// The MDS layer is 5 dot products of 5 terms, each with ONE Montgomery reduction (fe_dot5):
// 25 full multiplications become 25 x 81 MADs + 5 reductions.
//
// Code-size note: the round body is kept as a rolled loop and the five S-boxes / five dot
// products of a full round are executed by rotating the state through one inlined copy, so a
// round is ~1.2k instructions of code instead of ~6k (the instruction cache is shared by two CUs).
#pragma once
#include "fe29.h"

namespace dsv {

__constant__ u32 c_hades_rc[(DSV_HADES_FULL + DSV_HADES_PARTIAL) * DSV_HADES_WIDTH][NL];
__constant__ u32 c_hades_mds[DSV_HADES_WIDTH * DSV_HADES_WIDTH][NL];

DSV_DEV Fe fe_load_const(const u32* p) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = p[i];
  return r;
}

// x^5.  x limbs < 2^30, x < 3q  ->  N
DSV_DEV Fe hades_sbox(const Fe& x) {
  Fe x2 = fe_sqr(x);
  Fe x4 = fe_sqr(x2);
  return fe_mul(x4, x);
}

// state' = MDS * state, one output word per iteration; state words: limbs < 2^30, < 3q
DSV_DEV void hades_mds(Fe (&s)[5]) {
  Fe out[5];
#pragma unroll 1
  for (int k = 0; k < 5; k++) {
    Fe m[5];
#pragma unroll
    for (int j = 0; j < 5; j++) m[j] = fe_load_const(c_hades_mds[k * 5 + j]);
    Fe r = fe_dot5(s, m);
    // rotate `out` so that after 5 iterations out[k] holds row k (no dynamic register index)
    out[0] = out[1];
    out[1] = out[2];
    out[2] = out[3];
    out[3] = out[4];
    out[4] = r;
  }
#pragma unroll
  for (int k = 0; k < 5; k++) s[k] = out[k];
}

DSV_DEV void hades_permute(Fe (&s)[5]) {
  int ci = 0;
#pragma unroll 1
  for (int round = 0; round < DSV_HADES_FULL + DSV_HADES_PARTIAL; round++) {
    const bool full = round < DSV_HADES_FULL / 2 || round >= DSV_HADES_FULL / 2 + DSV_HADES_PARTIAL;
#pragma unroll
    for (int k = 0; k < 5; k++) s[k] = fe_add(s[k], fe_load_const(c_hades_rc[ci + k]));
    ci += 5;
    if (full) {
      // S-box on words 0..3 through one inlined copy: apply to s[0], rotate left, 4 times,
      // then rotate once more so every word is back in place with s[4] still pending.
#pragma unroll 1
      for (int k = 0; k < 4; k++) {
        Fe t = hades_sbox(s[0]);
        s[0] = s[1];
        s[1] = s[2];
        s[2] = s[3];
        s[3] = t;
      }
    }
    s[4] = hades_sbox(s[4]);  // partial rounds: last word only
    hades_mds(s);
  }
}

// sponge::hash over 3 inputs: state = [0, a, b, c, 1] -> one permutation -> state[1]
DSV_DEV Fe poseidon_hash3(const Fe& a, const Fe& b, const Fe& c) {
  Fe s[5] = {fe_zero(), a, b, c, fe_one()};
  hades_permute(s);
  return s[1];
}
// 5 inputs: [0,a,b,c,d] -> perm -> s[1] += e, s[2] += 1 -> perm -> state[1]
DSV_DEV Fe poseidon_hash5(const Fe& a, const Fe& b, const Fe& c, const Fe& d, const Fe& e) {
  Fe s[5] = {fe_zero(), a, b, c, d};
  hades_permute(s);
  s[1] = fe_add(s[1], e);
  s[2] = fe_add(s[2], fe_one());
  hades_permute(s);
  return s[1];
}

// truncation: canonical(h) & (2^250 - 1), returned as 8 LE words (a JubJubScalar < r)
DSV_DEV void poseidon_truncate(u32 (&c)[8], const Fe& h_mont) {
  Fe h = fe_from_mont(h_mont);
  fe_to_words_plain(c, h);
  c[7] &= 0x03ffffffu;
}

}  // namespace dsv
