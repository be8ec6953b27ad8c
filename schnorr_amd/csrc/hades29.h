// hades29.h — Hades permutation (width 5, x^5 S-box, 4 + 59 + 4 rounds) and the Poseidon
// sponge + 250-bit truncation that dusk-poseidon's `sponge::truncated::hash` evaluates for
// `challenge_hash` / `challenge_hash_double` (/root/reference/src/signatures.rs:127-134,
// :275-290; semantics SURVEY.md Appendix A.4).  One lane = one hash.
//
// Shipped form: every product with a CONSTANT field element — the dense layer of the full rounds,
// the recurrence of the partial rounds, its start-up rows and the state rebuild — runs as an int8
// product on the matrix cores over the 64 hashes of a wave (hades_mfma.h); the VALU keeps the
// S-boxes and the additions of round constants.  The all-VALU form (-DDSV_HADES_MFMA=0, the A/B
// baseline) is kept below: round constants and the MDS matrix are wave-uniform, sit in
// __constant__ memory and are fetched with scalar loads, so the MADs take them as SGPR operands;
// the MDS layer is 5 dot products of 5 terms, each with ONE Montgomery reduction (fe_dot5).
//
// Code-size note: the round body is kept as a rolled loop and the five S-boxes / five dot
// products of a full round are executed by rotating the state through one inlined copy, so a
// round is ~1.2k instructions of code instead of ~6k (the instruction cache is shared by two CUs).
#pragma once
#include "fe29.h"
// DSV_HADES_MFMA (shipped: 1): the recurrence of the partial rounds runs on the matrix cores
// (hades_mfma.h).  The hashes of a wave then cooperate: every lane of the wave has to run the
// permutation (callers clamp the index instead of leaving) and the workgroup calls
// hades_mfma_load_table() first.  -DDSV_HADES_MFMA=0: all-VALU form (A/B).
#ifndef DSV_HADES_MFMA
#define DSV_HADES_MFMA 1
#endif
#ifndef DSV_HADES_MFMA_UNROLL5
#define DSV_HADES_MFMA_UNROLL5 1  /* rotating window slots, five rounds per loop iteration */
#endif
#if DSV_HADES_MFMA
#include "hades_mfma.h"
#endif

namespace dsv {

__constant__ u32 c_hades_rc[(DSV_HADES_FULL + DSV_HADES_PARTIAL) * DSV_HADES_WIDTH][NL];
__constant__ u32 c_hades_mds[DSV_HADES_WIDTH * DSV_HADES_WIDTH][NL];
// sparse form of the 59 partial rounds (schnorr_amd/csrc/gen_constants.py: sparse_partial_rounds)
__constant__ u32 c_hades_pre_mds[DSV_HADES_WIDTH * DSV_HADES_WIDTH][NL];
__constant__ u32 c_hades_kappa0[DSV_HADES_WIDTH][NL];
__constant__ u32 c_hades_kfinal[DSV_HADES_WIDTH - 1][NL];
__constant__ u32 c_hades_blocks[sizeof(DSV_HADES_BLOCKS_HOST) / sizeof(DSV_HADES_BLOCKS_HOST[0])][NL];
// the partial rounds as one scalar recurrence (gen_constants.py: arma_partial_rounds)
#ifndef DSV_HADES_ARMA
#define DSV_HADES_ARMA 1  /* 0: the r01 blocked sparse form (A/B) */
#endif
__constant__ u32 c_hades_arma[sizeof(DSV_HADES_ARMA_HOST) / sizeof(DSV_HADES_ARMA_HOST[0])][NL];

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

// state' = MDS * state, one output word per iteration; state words: limbs < 2^29 + 8
DSV_DEV void hades_mds(Fe (&s)[5], const u32 (*mat)[NL]) {
  Fe out[5] = {fe_zero(), fe_zero(), fe_zero(), fe_zero(), fe_zero()};
#pragma unroll 1
  for (int k = 0; k < 5; k++) {
    Fe m[5];
#pragma unroll
    for (int j = 0; j < 5; j++) m[j] = fe_load_const(mat[k * 5 + j]);
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

#if DSV_HADES_MFMA && DSV_HADES_MFMA_MDS
// state' = MDS * state on the matrix cores.  s[j] = S-box outputs as fe_mul returns them (limb 0
// in [1, 2^29], limbs 1..7 < 2^29), or constants given one below their value (ONE_BELOW).
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
    out[0] = out[1];
    out[1] = out[2];
    out[2] = out[3];
    out[3] = out[4];
    out[4] = r;
  }
#pragma unroll
  for (int k = 0; k < 5; k++) s[k] = out[k];
}
#endif

// one full round: add constants, x^5 on every word, dense matrix
DSV_DEV void hades_full_round(Fe (&s)[5], const u32 (*rc)[NL], const u32 (*mat)[NL]) {
#pragma unroll
  for (int k = 0; k < 5; k++) s[k] = fe_add(s[k], fe_load_const(rc[k]));
  // S-box through one inlined copy: apply to s[0], rotate, 5 times
#pragma unroll 1
  for (int k = 0; k < 5; k++) {
    Fe t = hades_sbox(s[0]);
    s[0] = s[1];
    s[1] = s[2];
    s[2] = s[3];
    s[3] = s[4];
    s[4] = t;
  }
#if DSV_HADES_MFMA && DSV_HADES_MFMA_MDS
  (void)mat;
  hades_mds_mfma(s, 0, 5);
#else
  hades_mds(s, mat);
#endif
}

#if DSV_HADES_MFMA && DSV_HADES_MFMA_MDS
__device__ constexpr u32 kSbox0Cap[NL] = DSV_HADES_SBOX0_CAP_M1;  // one below: operand convention
__device__ constexpr u32 kSbox0Pad[NL] = DSV_HADES_SBOX0_PAD_M1;
#else
__device__ constexpr u32 kSbox0Cap[NL] = DSV_HADES_SBOX0_CAP;
__device__ constexpr u32 kSbox0Pad[NL] = DSV_HADES_SBOX0_PAD;
#endif

// round 0 of a permutation whose word 0 (and, with PAD, word 4) enters as the constant 0 (1): the
// S-boxes of those words are compile-time constants — x^5 runs on words 1..3 (1..4) only
template <bool PAD>
DSV_DEV void hades_first_round_const(Fe (&s)[5], const u32 (*rc)[NL], const u32 (*mat)[NL]) {
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
#if DSV_HADES_MFMA && DSV_HADES_MFMA_MDS
  (void)mat;
  hades_mds_mfma(s, 0, 5, PAD ? 0x11u : 0x01u);
#else
  hades_mds(s, mat);
#endif
}
// last round of a permutation of which only word 1 is read (the sponge's output): one matrix row
DSV_DEV void hades_last_round_word1(Fe (&s)[5], const u32 (*rc)[NL], const u32 (*mat)[NL]) {
#pragma unroll
  for (int k = 0; k < 5; k++) s[k] = fe_add(s[k], fe_load_const(rc[k]));
#pragma unroll 1
  for (int k = 0; k < 5; k++) {
    Fe t = hades_sbox(s[0]);
    s[0] = s[1];
    s[1] = s[2];
    s[2] = s[3];
    s[3] = s[4];
    s[4] = t;
  }
#if DSV_HADES_MFMA && DSV_HADES_MFMA_MDS
  (void)mat;
  hades_mds_mfma(s, 1, 1);  // row 1 lands in s[4]
  s[1] = s[4];
#else
  Fe m[5];
#pragma unroll
  for (int j = 0; j < 5; j++) m[j] = fe_load_const(mat[5 + j]);
  s[1] = fe_dot5(s, m);
#endif
}

// The 59 partial rounds as ONE scalar recurrence.  Only the S-box inputs a_r and outputs
// z_r = a_r^5 are carried: a_{r+5} is a fixed 10-term combination of (a, z)_{r..r+4} plus a round
// constant (Cayley-Hamilton on the 5x5 matrix; derivation and self-test against the dense rounds
// in gen_constants.py: arma_partial_rounds) — ten products and ONE reduction per round, the
// constant riding in the start values of the column sums.  a_1..a_4 come from the state that
// enters the partial rounds, the state that leaves them is rebuilt from (a, z)_{54..58}.
DSV_DEV void hades_partial_rounds_arma(Fe (&s)[5]) {
  static_assert(DSV_HADES_PARTIAL == 59, "window bookkeeping below is written for 59 rounds");
  const u32(*k)[NL] = c_hades_arma;
  Fe A[5], Z[5];
#if DSV_HADES_MFMA && DSV_HADES_MFMA_EDGE
  // start-up rows on the matrix cores as well: operands are the entering state and the (a, z)
  // pairs so far, as MFMA digits (a z entry is stored one below its value, see below)
  Dig win[kMfmaTerms];
  const MfmaTable tab = hades_mfma_table();
  {
    const v4i* edge = reinterpret_cast<const v4i*>(g_hades_mfma_edge) + tab.lane;
    Dig ds[5];
#pragma unroll
    for (int i = 0; i < 5; i++) ds[i] = mfma_digits(s[i]);
    // A[0] = s[4] + k0 is the one operand that is not the output of a row: s[4] < 2.0001 q (dense
    // layer on the matrix cores), k0 < q, so the sum can pass 2^256 = 2.2 q — one conditional
    // subtraction of q brings it back under 2.0001 q (32 digits)
    A[0] = fe_cond_sub(fe_ripple(fe_add(s[4], fe_load_const(k[0]))), kQx1);
    auto push = [&](int i) {  // S-box of a_i, both into the window
      Fe z = hades_sbox(A[i]);
      z.l[0] -= 1;
      win[i] = mfma_digits(A[i]);
      win[5 + i] = mfma_digits(z);
    };
    push(0);
    { const Dig t[7] = {ds[0], ds[1], ds[2], ds[3], ds[4], win[0], win[5]};
      A[1] = mfma_dot<7>(t, edge + kMfmaEdgeInitOff[0], g_hades_mfma_edge_start[0]); push(1); }
    { const Dig t[9] = {ds[0], ds[1], ds[2], ds[3], ds[4], win[0], win[5], win[1], win[6]};
      A[2] = mfma_dot<9>(t, edge + kMfmaEdgeInitOff[1], g_hades_mfma_edge_start[1]); push(2); }
    { const Dig t[11] = {ds[0], ds[1], ds[2], ds[3], ds[4], win[0], win[5], win[1], win[6], win[2], win[7]};
      A[3] = mfma_dot<11>(t, edge + kMfmaEdgeInitOff[2], g_hades_mfma_edge_start[2]); push(3); }
    { const Dig t[13] = {ds[0], ds[1], ds[2], ds[3], ds[4], win[0], win[5], win[1], win[6], win[2], win[7],
                         win[3], win[8]};
      A[4] = mfma_dot<13>(t, edge + kMfmaEdgeInitOff[3], g_hades_mfma_edge_start[3]); push(4); }
  }
#else
  A[0] = fe_carry(fe_add(s[4], fe_load_const(k[0])));
  k += 1;
  asm volatile("" : "+s"(k));  // (see the loop below: keeps the scalar loads of each dot product local)
  Z[0] = hades_sbox(A[0]);
  { const Fe t[7] = {s[0], s[1], s[2], s[3], s[4], A[0], Z[0]};
    A[1] = fe_dot_const_plus<7>(t, k, k[7]); k += 8; asm volatile("" : "+s"(k)); Z[1] = hades_sbox(A[1]); }
  { const Fe t[9] = {s[0], s[1], s[2], s[3], s[4], A[0], Z[0], A[1], Z[1]};
    A[2] = fe_dot_const_plus<9>(t, k, k[9]); k += 10; asm volatile("" : "+s"(k)); Z[2] = hades_sbox(A[2]); }
  { const Fe t[11] = {s[0], s[1], s[2], s[3], s[4], A[0], Z[0], A[1], Z[1], A[2], Z[2]};
    A[3] = fe_dot_const_plus<11>(t, k, k[11]); k += 12; asm volatile("" : "+s"(k)); Z[3] = hades_sbox(A[3]); }
  { const Fe t[13] = {s[0], s[1], s[2], s[3], s[4], A[0], Z[0], A[1], Z[1], A[2], Z[2], A[3], Z[3]};
    A[4] = fe_dot_const_plus<13>(t, k, k[13]); Z[4] = hades_sbox(A[4]); }
#endif
#if DSV_HADES_MFMA
  {
    static_assert(DSV_HADES_MFMA_ROUNDS == DSV_HADES_PARTIAL - 5, "one start row per recurrence round");
    // window as MFMA operands; a z entry is stored one below its value (fe_mul returns limb 0 in
    // [1, 2^29]; the generator folds the missing 1 x multiplier into the start limbs)
#if !DSV_HADES_MFMA_EDGE
    const MfmaTable tab = hades_mfma_table();
    Dig win[kMfmaTerms];
    A[0] = fe_cond_sub(fe_ripple(A[0]), kQx1);  // (s[4] + k0 can pass 2^256, see above)
#pragma unroll
    for (int i = 0; i < 5; i++) {
      win[i] = mfma_digits(A[i]);
      Fe zd = Z[i];
      zd.l[0] -= 1;
      win[5 + i] = mfma_digits(zd);
    }
#endif
    const v4i* rbase = tab.a + tab.lane;
    // Software pipeline: of the ten terms of round r+1 eight are known before round r has produced
    // anything, so their 32 MFMAs are issued as soon as round r's accumulators have been read out
    // and run under the reduction and the S-box of round r; a_r adds its term before the S-box,
    // z_r at the top of the next round — only those 4 MFMAs are ever waited for.
    // OLD(t) / NEW: window slot of the pair that is t rounds from the oldest / that is replaced.
#define DSV_MFMA_ROUND(OLD, NEW, START)                                                        \
  {                                                                                            \
    mfma_term(acc, 9, win[5 + OLD(4)], rbase); /* z_{r-1} */                                     \
    const MfmaGroups grp = mfma_collect(acc);                                                  \
    mfma_clear(acc); /* round r+1: terms 0..3 = a_{r-4..r-1}, 5..8 = z_{r-4..r-1} */           \
    mfma_term(acc, 0, win[OLD(1)], rbase);                                                       \
    mfma_term(acc, 5, win[5 + OLD(1)], rbase);                                                   \
    mfma_term(acc, 1, win[OLD(2)], rbase);                                                       \
    mfma_term(acc, 6, win[5 + OLD(2)], rbase);                                                   \
    mfma_term(acc, 2, win[OLD(3)], rbase);                                                       \
    mfma_term(acc, 7, win[5 + OLD(3)], rbase);                                                   \
    mfma_term(acc, 3, win[OLD(4)], rbase);                                                       \
    mfma_term(acc, 8, win[5 + OLD(4)], rbase);                                                   \
    u32 aw[8];                                                                                 \
    mfma_finish(aw, grp, START);                                                               \
    const Fe an = fe_from_words_plain(aw);                                                     \
    const Dig da = mfma_digits_words(aw);                                                      \
    mfma_term(acc, 4, da, rbase); /* a_r */                                                      \
    Fe zn = hades_sbox(an);                                                                    \
    zn.l[0] -= 1;                                                                              \
    NEW(da, mfma_digits(zn))                                                                   \
  }
#define DSV_ROT0(t) (t)
    MfmaAcc acc;
    mfma_clear(acc);
#pragma unroll
    for (int j = 0; j < kMfmaTerms - 1; j++) mfma_term(acc, j, win[j], rbase);
    constexpr int kBlocks = DSV_HADES_MFMA_UNROLL5 ? (DSV_HADES_PARTIAL - 5) / 5 : 0;
#if DSV_HADES_MFMA_UNROLL5
    // five rounds per iteration, the window slots rotating by NAME: slot i holds the oldest pair in
    // round i of a block and takes the new one, so nothing is moved
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
#endif
    // rolled copy (all 54 rounds, or the last four after the blocks: 54 = 10 * 5 + 4): the window
    // is shifted by register moves
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
#if DSV_HADES_MFMA_EDGE
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
    return;
#else
#pragma unroll
    for (int i = 0; i < 5; i++) {
      A[i] = mfma_undigits(win[i]);
      Z[i] = mfma_undigits(win[5 + i]);
      Z[i].l[0] += 1;
    }
#endif
  }
#else
  const u32(*rec)[NL] = c_hades_arma + DSV_HADES_ARMA_REC;
  const u32(*gam)[NL] = c_hades_arma + DSV_HADES_ARMA_GAMMA;
  // One round per loop iteration, the window (oldest -> newest) shifted by register moves (72
  // v_mov, ~3 % of a round): an unrolled-by-5 body with rotating slot names spills 136-163 VGPRs
  // whatever is pinned or fenced (A/B: 4 % slower than the blocked form).
  // The ten multipliers are the same in every round; left to itself the compiler loads all 90
  // words once and keeps them in SGPRs, which spills (421 SGPR spills, 8 k v_readlane, hash 37 %
  // SLOWER).  Laundering the table pointer after every dot product makes each round re-load its
  // constants — issued under the S-box that follows, nine SGPRs at a time.
#pragma unroll 1
  for (int r = 5; r < DSV_HADES_PARTIAL; r++) {
    const Fe t[10] = {A[0], A[1], A[2], A[3], A[4], Z[0], Z[1], Z[2], Z[3], Z[4]};
    const Fe an = fe_dot_const_plus<10>(t, rec, gam[0]);
    gam += 1;
    asm volatile("" : "+s"(rec));
    const Fe zn = hades_sbox(an);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      A[i] = A[i + 1];
      Z[i] = Z[i + 1];
    }
    A[4] = an;
    Z[4] = zn;
  }
#endif
  const u32(*fin)[NL] = c_hades_arma + DSV_HADES_ARMA_FINAL;
  const Fe t[10] = {A[0], A[1], A[2], A[3], A[4], Z[0], Z[1], Z[2], Z[3], Z[4]};
#pragma unroll 1
  for (int j = 0; j < 5; j++) {  // one inlined copy: rotate the output through s[]
    const Fe r = fe_dot_const_plus<10>(t, fin + 11 * j, fin[11 * j + 10]);
    s[0] = s[1];
    s[1] = s[2];
    s[2] = s[3];
    s[3] = s[4];
    s[4] = r;
  }
}

// The permutation.  The 59 partial rounds run in their sparse-matrix form (per round one S-box,
// one dot product for the new last word, four multiply-accumulates for words 0..3 instead of
// five dense dot products), and the four multiply-accumulates are deferred block-wise (below).
// FIRST: 0 = generic input, 1 = word 0 is the constant 0, 2 = additionally word 4 is the constant
// 1 (state of the 3-input hash).  WORD1_ONLY: the caller reads s[1] only (last permutation of a
// hash); the other words are then left unspecified.
template <int FIRST, bool WORD1_ONLY>
DSV_DEV void hades_permute(Fe (&s)[5]) {
  constexpr int HALF = DSV_HADES_FULL / 2;
  if (FIRST == 1) hades_first_round_const<false>(s, c_hades_rc, c_hades_mds);
  if (FIRST == 2) hades_first_round_const<true>(s, c_hades_rc, c_hades_mds);
#pragma unroll 1
  for (int r = FIRST ? 1 : 0; r < HALF; r++)
    hades_full_round(s, c_hades_rc + 5 * r,
                     (!DSV_HADES_ARMA && r == HALF - 1) ? c_hades_pre_mds : c_hades_mds);
  if (DSV_HADES_ARMA) {
    hades_partial_rounds_arma(s);
  } else {
  // words 0..3 carry NO round constants inside the loop: their running sum K_i is folded into the
  // constant of the last word (kappa4'_i = kappa4_{i+1} + c_i . K_i) and added back once at the end.
  // Rounds run in blocks of 4 (gen_constants.py): inside a block words 0..3 stay untouched and
  // their pending updates reach the later rounds through extra dot-product terms, so a round is
  // one S-box + ONE reduction, and each block ends with four 5-term dot products.
  s[4] = fe_add(s[4], fe_load_const(c_hades_kappa0[4]));
  const u32(*k)[NL] = c_hades_blocks;
#pragma unroll 1
  for (int blk = 0; blk < DSV_HADES_PARTIAL / 4; blk++) {
    Fe z0 = hades_sbox(s[4]);
    { const Fe a[5] = {s[0], s[1], s[2], s[3], z0};
      s[4] = fe_add(fe_dot_const<5>(a, k), fe_load_const(k[5])); }
    Fe z1 = hades_sbox(s[4]);
    { const Fe a[6] = {s[0], s[1], s[2], s[3], z0, z1};
      s[4] = fe_add(fe_dot_const<6>(a, k + 6), fe_load_const(k[12])); }
    Fe z2 = hades_sbox(s[4]);
    { const Fe a[7] = {s[0], s[1], s[2], s[3], z0, z1, z2};
      s[4] = fe_add(fe_dot_const<7>(a, k + 13), fe_load_const(k[20])); }
    Fe z3 = hades_sbox(s[4]);
    { const Fe a[8] = {s[0], s[1], s[2], s[3], z0, z1, z2, z3};
      s[4] = fe_add(fe_dot_const<8>(a, k + 21), fe_load_const(k[29])); }
    // (written out: a rolled loop here would index s[] dynamically, i.e. through scratch)
    { const Fe a[5] = {s[0], z0, z1, z2, z3}; s[0] = fe_dot_const<5>(a, k + 30); }
    { const Fe a[5] = {s[1], z0, z1, z2, z3}; s[1] = fe_dot_const<5>(a, k + 35); }
    { const Fe a[5] = {s[2], z0, z1, z2, z3}; s[2] = fe_dot_const<5>(a, k + 40); }
    { const Fe a[5] = {s[3], z0, z1, z2, z3}; s[3] = fe_dot_const<5>(a, k + 45); }
    k += 50;
  }
  {  // last block: 59 = 14 * 4 + 3 rounds
    static_assert(DSV_HADES_PARTIAL % 4 == 3 && DSV_HADES_BLOCK == 4, "block layout");
    Fe z0 = hades_sbox(s[4]);
    { const Fe a[5] = {s[0], s[1], s[2], s[3], z0};
      s[4] = fe_add(fe_dot_const<5>(a, k), fe_load_const(k[5])); }
    Fe z1 = hades_sbox(s[4]);
    { const Fe a[6] = {s[0], s[1], s[2], s[3], z0, z1};
      s[4] = fe_add(fe_dot_const<6>(a, k + 6), fe_load_const(k[12])); }
    Fe z2 = hades_sbox(s[4]);
    { const Fe a[7] = {s[0], s[1], s[2], s[3], z0, z1, z2};
      s[4] = fe_add(fe_dot_const<7>(a, k + 13), fe_load_const(k[20])); }
    { const Fe a[4] = {s[0], z0, z1, z2}; s[0] = fe_dot_const<4>(a, k + 21); }
    { const Fe a[4] = {s[1], z0, z1, z2}; s[1] = fe_dot_const<4>(a, k + 25); }
    { const Fe a[4] = {s[2], z0, z1, z2}; s[2] = fe_dot_const<4>(a, k + 29); }
    { const Fe a[4] = {s[3], z0, z1, z2}; s[3] = fe_dot_const<4>(a, k + 33); }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) s[j] = fe_carry(fe_add(s[j], fe_load_const(c_hades_kfinal[j])));
  }
#pragma unroll 1
  for (int r = 0; r < HALF - (WORD1_ONLY ? 1 : 0); r++)
    hades_full_round(s, c_hades_rc + 5 * (HALF + DSV_HADES_PARTIAL + r), c_hades_mds);
  if (WORD1_ONLY)
    hades_last_round_word1(s, c_hades_rc + 5 * (HALF + DSV_HADES_PARTIAL + HALF - 1), c_hades_mds);
}

// sponge::hash over 3 inputs: state = [0, a, b, c, 1] -> one permutation -> state[1]
DSV_DEV Fe poseidon_hash3(const Fe& a, const Fe& b, const Fe& c) {
  Fe s[5] = {fe_zero(), a, b, c, fe_one()};
  hades_permute<2, true>(s);
  return s[1];
}
// 5 inputs: [0,a,b,c,d] -> perm -> s[1] += e, s[2] += 1 -> perm -> state[1]
DSV_DEV Fe poseidon_hash5(const Fe& a, const Fe& b, const Fe& c, const Fe& d, const Fe& e) {
  Fe s[5] = {fe_zero(), a, b, c, d};
  hades_permute<1, false>(s);
  s[1] = fe_add(s[1], e);
  s[2] = fe_add(s[2], fe_one());
  hades_permute<0, true>(s);
  return s[1];
}

// truncation: canonical(h) & (2^250 - 1), returned as 8 LE words (a JubJubScalar < r)
DSV_DEV void poseidon_truncate(u32 (&c)[8], const Fe& h_mont) {
  Fe h = fe_from_mont(h_mont);
  fe_to_words_plain(c, h);
  c[7] &= 0x03ffffffu;
}

}  // namespace dsv
