// hades29.h — Hades permutation (width 5, x^5 S-box, 4 + 59 + 4 rounds) and the Poseidon
// sponge + 250-bit truncation that dusk-poseidon's `sponge::truncated::hash` evaluates for
// `challenge_hash` / `challenge_hash_double` (/root/reference/src/signatures.rs:127-134,
// :275-290; semantics SURVEY.md Appendix A.4).  One lane = one hash; the 64 hashes of a wave
// cooperate through the matrix cores.
//
// Every product with a CONSTANT field element — the dense layer of the full rounds, the recurrence
// of the partial rounds, its start-up rows and the state rebuild — runs as an exact int8 product on
// the matrix cores (hades_mfma.h); the VALU keeps the S-boxes and the additions of round
// constants.  Consequences for callers: every lane of a wave has to run the permutation (spare
// lanes of a ragged batch clamp their index instead of leaving) and the workgroup calls
// hades_mfma_load_table() first.
// (The all-VALU forms of r01 / r02 — dense MDS by 5-term limb dot products, the blocked sparse
// partial rounds, the scalar recurrence on limb products — are no longer in the product source:
// +12 % for the matrix-core form, profiles/r02/ab_hades_mfma.txt; tools/variants/ has them.)
//
// The 59 partial rounds as ONE scalar recurrence: only the S-box inputs a_r and outputs
// z_r = a_r^5 are carried; a_{r+5} is a fixed 10-term combination of (a, z)_{r..r+4} plus a round
// constant (Cayley-Hamilton on the 5x5 matrix; derivation and self-test against the dense rounds
// in gen_constants.py: arma_partial_rounds).  a_1..a_4 come from the state that enters the partial
// rounds, the state that leaves them is rebuilt from (a, z)_{54..58}.
//
// Code-size note: round bodies are rolled loops and the five S-boxes / rows of a full round rotate
// the state through ONE inlined copy (the instruction cache is shared by two CUs).
#pragma once
#include "fe29.h"
#include "hades_mfma.h"

namespace dsv {

__constant__ u32 c_hades_rc[(DSV_HADES_FULL + DSV_HADES_PARTIAL) * DSV_HADES_WIDTH][NL];
__constant__ u32 c_hades_k0[NL];  // constant of the first partial round's S-box input

DSV_DEV Fe fe_load_const(const u32* p) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = p[i];
  return r;
}

// x^5.  x limbs < 2^30, x < 5q  ->  N
DSV_DEV Fe hades_sbox(const Fe& x) {
  Fe x2 = fe_sqr(x);
  Fe x4 = fe_sqr(x2);
  return fe_mul(x4, x);
}

// state' = MDS * state on the matrix cores, rows first_row .. first_row + rows - 1 (the last one
// lands in s[4]).  s[j] = S-box outputs as fe_mul returns them (limb 0 in [1, 2^29], limbs 1..7
// < 2^29), or constants given one below their value (bit j of one_below_mask).
DSV_DEV void hades_mds_mfma(Fe (&s)[5], int first_row, int rows, unsigned one_below_mask = 0) {
  const MfmaTable tab = hades_mfma_table();
  Dig d[5];
#pragma unroll
  for (int j = 0; j < 5; j++) {
    Fe t = s[j];
    if (!((one_below_mask >> j) & 1)) t.l[0] -= 1;
    d[j] = mfma_digits(t);
  }
  Fe out[5] = {fe_zero(), fe_zero(), fe_zero(), fe_zero(), fe_zero()};
#pragma unroll 1
  for (int k = first_row; k < first_row + rows; k++) {
    const Fe r = hades_mfma_mds_row(d, k, tab);
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

// x^5 on all five words through one inlined copy: apply to s[0], rotate, 5 times
DSV_DEV void hades_sbox_all(Fe (&s)[5]) {
#pragma unroll 1
  for (int k = 0; k < 5; k++) {
    Fe t = hades_sbox(s[0]);
    s[0] = s[1];
    s[1] = s[2];
    s[2] = s[3];
    s[3] = s[4];
    s[4] = t;
  }
}
// one full round: add constants, x^5 on every word, dense matrix.  word1_only (wave-uniform): only
// row 1 of the matrix — the last round of a permutation of which only word 1 is read (the sponge's
// output); the other words are then unspecified
DSV_DEV void hades_full_round(Fe (&s)[5], const u32 (*rc)[NL], bool word1_only) {
#pragma unroll
  for (int k = 0; k < 5; k++) s[k] = fe_add(s[k], fe_load_const(rc[k]));
  hades_sbox_all(s);
  hades_mds_mfma(s, word1_only ? 1 : 0, word1_only ? 1 : 5);  // a single row lands in s[4]
  if (word1_only) s[1] = s[4];
}

__device__ constexpr u32 kSbox0Cap[NL] = DSV_HADES_SBOX0_CAP_M1;  // one below: operand convention
__device__ constexpr u32 kSbox0Pad[NL] = DSV_HADES_SBOX0_PAD_M1;

// round 0 of a permutation whose word 0 (and, with PAD, word 4) enters as the constant 0 (1): the
// S-boxes of those words are compile-time constants — x^5 runs on words 1..3 (1..4) only
template <bool PAD>
DSV_DEV void hades_first_round_const(Fe (&s)[5], const u32 (*rc)[NL]) {
  constexpr int LAST = PAD ? 3 : 4;
#pragma unroll
  for (int k = 1; k <= LAST; k++) s[k] = fe_add(s[k], fe_load_const(rc[k]));
#pragma unroll 1
  for (int k = 1; k <= LAST; k++) {  // one inlined copy, words 1..LAST rotating through it
    Fe t = hades_sbox(s[1]);
#pragma unroll
    for (int j = 1; j < LAST; j++) s[j] = s[j + 1];
    s[LAST] = t;
  }
  s[0] = fe_const(kSbox0Cap);
  if (PAD) s[4] = fe_const(kSbox0Pad);
  hades_mds_mfma(s, 0, 5, PAD ? 0x11u : 0x01u);
}

DSV_DEV void hades_partial_rounds(Fe (&s)[5]) {
  static_assert(DSV_HADES_PARTIAL == 59, "window bookkeeping below is written for 59 rounds");
  static_assert(DSV_HADES_MFMA_ROUNDS == DSV_HADES_PARTIAL - 5, "one start row per recurrence round");
  // window of the last five (a, z) pairs as MFMA operands; a z entry is stored one below its value
  // (fe_mul returns limb 0 in [1, 2^29]; the generator folds the missing 1 x multiplier into the
  // start limbs)
  Dig win[kMfmaTerms];
  const MfmaTable tab = hades_mfma_table();
  {
    // start-up rows: operands are the entering state and the (a, z) pairs so far
    const v4i* edge = reinterpret_cast<const v4i*>(g_hades_mfma_edge) + tab.lane;
    Dig ds[5];
#pragma unroll
    for (int i = 0; i < 5; i++) ds[i] = mfma_digits(s[i]);
    // a_0 = s[4] + k0 is the one operand that is not the output of a row: s[4] < 2.0001 q (dense
    // layer on the matrix cores), k0 < q, so the sum can pass 2^256 = 2.2 q — one conditional
    // subtraction of q brings it back under 2.0001 q (32 digits)
    Fe a = fe_cond_sub(fe_ripple(fe_add(s[4], fe_load_const(c_hades_k0))), kQx1);
    auto push = [&](int i) {  // S-box of a_i, both into the window
      Fe z = hades_sbox(a);
      z.l[0] -= 1;
      win[i] = mfma_digits(a);
      win[5 + i] = mfma_digits(z);
    };
    push(0);
    { const Dig t[7] = {ds[0], ds[1], ds[2], ds[3], ds[4], win[0], win[5]};
      a = mfma_dot<7>(t, edge + kMfmaEdgeInitOff[0], g_hades_mfma_edge_start[0]); push(1); }
    { const Dig t[9] = {ds[0], ds[1], ds[2], ds[3], ds[4], win[0], win[5], win[1], win[6]};
      a = mfma_dot<9>(t, edge + kMfmaEdgeInitOff[1], g_hades_mfma_edge_start[1]); push(2); }
    { const Dig t[11] = {ds[0], ds[1], ds[2], ds[3], ds[4], win[0], win[5], win[1], win[6], win[2], win[7]};
      a = mfma_dot<11>(t, edge + kMfmaEdgeInitOff[2], g_hades_mfma_edge_start[2]); push(3); }
    { const Dig t[13] = {ds[0], ds[1], ds[2], ds[3], ds[4], win[0], win[5], win[1], win[6], win[2], win[7],
                         win[3], win[8]};
      a = mfma_dot<13>(t, edge + kMfmaEdgeInitOff[3], g_hades_mfma_edge_start[3]); push(4); }
  }
  {
    const v4i* rbase = tab.a + tab.lane;
    // Software pipeline: of the ten terms of round r+1 eight are known before round r has produced
    // anything, so their 16 MFMAs are issued as soon as round r's accumulators have been read out
    // and run under the reduction and the S-box of round r; a_r adds its term before the S-box,
    // z_r at the top of the next round — only those 4 MFMAs are ever waited for.
    // OLD(t) / NEW: window slot of the pair that is t rounds from the oldest / that is replaced.
#define DSV_MFMA_ROUND(OLD, NEW, START)                                                        \
  {                                                                                            \
    mfma_term(acc, 9, win[5 + OLD(4)], rbase); /* z_{r-1} */                                   \
    const MfmaGroups grp = mfma_collect(acc);                                                  \
    mfma_clear(acc); /* round r+1: terms 0..3 = a_{r-4..r-1}, 5..8 = z_{r-4..r-1} */           \
    mfma_term(acc, 0, win[OLD(1)], rbase);                                                     \
    mfma_term(acc, 5, win[5 + OLD(1)], rbase);                                                 \
    mfma_term(acc, 1, win[OLD(2)], rbase);                                                     \
    mfma_term(acc, 6, win[5 + OLD(2)], rbase);                                                 \
    mfma_term(acc, 2, win[OLD(3)], rbase);                                                     \
    mfma_term(acc, 7, win[5 + OLD(3)], rbase);                                                 \
    mfma_term(acc, 3, win[OLD(4)], rbase);                                                     \
    mfma_term(acc, 8, win[5 + OLD(4)], rbase);                                                 \
    u32 aw[8];                                                                                 \
    mfma_finish(aw, grp, START);                                                               \
    const Fe an = fe_from_words_plain(aw);                                                     \
    const Dig da = mfma_digits_words(aw);                                                      \
    mfma_term(acc, 4, da, rbase); /* a_r */                                                    \
    Fe zn = hades_sbox(an);                                                                    \
    zn.l[0] -= 1;                                                                              \
    NEW(da, mfma_digits(zn))                                                                   \
  }
    MfmaAcc acc;
    mfma_clear(acc);
#pragma unroll
    for (int j = 0; j < kMfmaTerms - 1; j++) mfma_term(acc, j, win[j], rbase);
    constexpr int kBlocks = (DSV_HADES_PARTIAL - 5) / 5;
    // five rounds per iteration, the window slots rotating by NAME: slot i holds the oldest pair in
    // round i of a block and takes the new one, so nothing is moved
#define DSV_ROT0(t) (t)
#define DSV_ROT1(t) ((1 + (t)) % 5)
#define DSV_ROT2(t) ((2 + (t)) % 5)
#define DSV_ROT3(t) ((3 + (t)) % 5)
#define DSV_ROT4(t) ((4 + (t)) % 5)
#define DSV_PUT0(a, z) { win[0] = a; win[5] = z; }
#define DSV_PUT1(a, z) { win[1] = a; win[6] = z; }
#define DSV_PUT2(a, z) { win[2] = a; win[7] = z; }
#define DSV_PUT3(a, z) { win[3] = a; win[8] = z; }
#define DSV_PUT4(a, z) { win[4] = a; win[9] = z; }
#pragma unroll 1
    for (int blk = 0; blk < kBlocks; blk++) {
      const u32(*st)[8] = g_hades_mfma_start + 5 * blk;
      DSV_MFMA_ROUND(DSV_ROT0, DSV_PUT0, st[0])
      DSV_MFMA_ROUND(DSV_ROT1, DSV_PUT1, st[1])
      DSV_MFMA_ROUND(DSV_ROT2, DSV_PUT2, st[2])
      DSV_MFMA_ROUND(DSV_ROT3, DSV_PUT3, st[3])
      DSV_MFMA_ROUND(DSV_ROT4, DSV_PUT4, st[4])
    }
    // rolled copy for the last four rounds (54 = 10 * 5 + 4): the window is shifted by moves
#define DSV_SHIFT_PUT(a, z)           \
  {                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) { \
      win[i_] = win[i_ + 1];          \
      win[5 + i_] = win[6 + i_];      \
    }                                 \
    win[4] = a;                       \
    win[9] = z;                       \
  }
#pragma unroll 1
    for (int r = 5 + 5 * kBlocks; r < DSV_HADES_PARTIAL; r++) {
      DSV_MFMA_ROUND(DSV_ROT0, DSV_SHIFT_PUT, g_hades_mfma_start[r - 5])
    }
#undef DSV_MFMA_ROUND
  }
  // state rebuild: five 10-term rows straight from the window
  const v4i* fin = reinterpret_cast<const v4i*>(g_hades_mfma_edge) + kMfmaEdgeFinalOff + tab.lane;
#pragma unroll 1
  for (int j = 0; j < 5; j++) {  // one inlined copy: rotate the output through s[]
    const Fe r = mfma_dot<kMfmaTerms>(win, fin + j * (kMfmaTerms * 64), g_hades_mfma_edge_start[4 + j]);
    s[0] = s[1];
    s[1] = s[2];
    s[2] = s[3];
    s[3] = s[4];
    s[4] = r;
  }
}

// The permutation.  FIRST: 0 = generic input, 1 = word 0 is the constant 0, 2 = additionally word 4
// is the constant 1 (state of the 3-input hash).  word1_only (wave-uniform): the caller reads s[1]
// only (last permutation of a hash); the other words are then left unspecified.
template <int FIRST>
DSV_DEV void hades_permute(Fe (&s)[5], bool word1_only) {
  constexpr int HALF = DSV_HADES_FULL / 2;
  if (FIRST == 1) hades_first_round_const<false>(s, c_hades_rc);
  if (FIRST == 2) hades_first_round_const<true>(s, c_hades_rc);
#pragma unroll 1
  for (int r = FIRST ? 1 : 0; r < HALF; r++) hades_full_round(s, c_hades_rc + 5 * r, false);
  hades_partial_rounds(s);
#pragma unroll 1
  for (int r = 0; r < HALF; r++)
    hades_full_round(s, c_hades_rc + 5 * (HALF + DSV_HADES_PARTIAL + r), word1_only && r == HALF - 1);
}

// sponge::hash over 3 inputs: state = [0, a, b, c, 1] -> one permutation -> state[1]
DSV_DEV Fe poseidon_hash3(const Fe& a, const Fe& b, const Fe& c) {
  Fe s[5] = {fe_zero(), a, b, c, fe_one()};
  hades_permute<2>(s, true);
  return s[1];
}

// truncation: canonical(h) & (2^250 - 1), returned as 8 LE words (a JubJubScalar < r)
DSV_DEV void poseidon_truncate(u32 (&c)[8], const Fe& h_mont) {
  Fe h = fe_from_mont(h_mont);
  fe_to_words_plain(c, h);
  c[7] &= 0x03ffffffu;
}

}  // namespace dsv
