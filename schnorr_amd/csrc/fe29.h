// fe29.h — BLS12-381 scalar field (= JubJub base field Fq) arithmetic for gfx950 lanes.
//
// Representation.  One field element lives in ONE lane as 9 limbs of 29 bits in 9 VGPRs
// (value = sum l[i] 2^(29 i)), Montgomery form with R = 2^261.  Why not 8 x 32 saturated
// limbs: measured on MI355X (tools/microbench/valu_rates.hip, profiles/r01_valu_rates.txt)
// v_mad_u64_u32 issues at the SAME ~4 cycles per wave64 as v_add_co_u32 / v_addc_co_u32,
// so what costs is the carry bookkeeping, not the multiplier.  With 29-bit limbs a whole
// 9x9 schoolbook product plus the interleaved Montgomery reduction accumulates in 64-bit
// column registers with NO carry instruction at all: each step is one in-place
// v_mad_u64_u32 (D = S0*S1 + D).  A multiply is 81 + 72 MADs + 36 digit/shift ops = 189 VALU
// instructions (a squaring 162), see fe_mont_fips below.
// R = 2^261 leaves 6 spare bits over q (255 bits), so products of lazily-added operands
// never need a conditional subtraction: values stay below ~8q between reductions.
//
// Contracts (checked by tests/test_fe29_model.py on a bit-exact Python model of this file):
//   fe_mul/fe_sqr(a, b): max_limb(a) * max_limb(b) <= 1.5 * 2^60  and  a*b < 2^261 * q * 0.5
//                        -> result limbs < 2^29 (limb 0 <= 2^29, limb 8 < 2^25), value <= a*b/2^261 + q
//   fe_add(a, b)       : limb-wise, no carry; caller keeps limbs < 2^31
//   fe_sub2/4/8(a, b)  : a + k*q - b with a redundant-limb k*q whose limbs dominate b's
//                        (b limbs <= 2^30 - 2, b < (k - 0.01) q), then one parallel carry pass
//                        -> limbs < 2^29 + 8
//
// Reference semantics being reproduced: dusk-bls12_381 `BlsScalar` mul/add/sub/square as used
// by the verify path (/root/reference/src/keys/public.rs:121-130 via dusk-jubjub operators);
// only values mod q are observable, never the representation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsv_constants.h"

namespace dsv {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int NL = 9;
constexpr u32 M29 = (1u << 29) - 1;

struct Fe {
  u32 l[NL];
};

#define DSV_DEV __device__ __forceinline__

__device__ constexpr u32 kQ29[NL] = DSV_Q29;
__device__ constexpr u32 kBias2[NL] = DSV_BIAS2;
__device__ constexpr u32 kBias4[NL] = DSV_BIAS4;
__device__ constexpr u32 kBias8[NL] = DSV_BIAS8;
__device__ constexpr u32 kBias4W[NL] = DSV_BIAS4W;
__device__ constexpr u32 kQx1[NL] = DSV_Q29_X1;
__device__ constexpr u32 kQx2[NL] = DSV_Q29_X2;
__device__ constexpr u32 kQx4[NL] = DSV_Q29_X4;
__device__ constexpr u32 kQx8[NL] = DSV_Q29_X8;
__device__ constexpr u32 kR2[NL] = DSV_R2;
__device__ constexpr u32 kOne[NL] = DSV_ONE;
__device__ constexpr u32 kD2[NL] = DSV_D2;
__device__ constexpr u32 kQ32[8] = DSV_Q32;
__device__ constexpr u32 kR32[8] = DSV_R32;

DSV_DEV Fe fe_const(const u32 (&c)[NL]) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = c[i];
  return r;
}
DSV_DEV Fe fe_zero() {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = 0;
  return r;
}
DSV_DEV Fe fe_one() { return fe_const(kOne); }

// ---- Montgomery multiplication, product scanning ("FIPS"), R = 2^261 ------------------------
// One 64-bit accumulator walks the 17 columns of a*b + M*q.  The carry out of a column is the
// ADDEND of the first v_mad_u64_u32 of the next column, so a column costs its MADs plus exactly
// two other instructions (digit, 64-bit shift) — no separate 64-bit carry addition.
//
// hipcc would not emit that on its own: LLVM's reassociation pass sorts the terms of every column
// sum by rank and leaves the carry in the middle of the chain, which costs a v_lshl_add_u64 per
// column (r01: 211 instructions per multiplication, now 190).  mad_pin() therefore gives every
// partial sum a second use — the operand of an llvm.assume — because a value with two uses is a
// leaf for the reassociation pass, so the chain stays in source order and instruction selection
// folds each `partial + x*y` into one v_mad_u64_u32.  The assumed fact must be TRUE (else it is
// undefined behaviour) and UNPROVABLE (else the optimiser deletes it and the pin is gone — a range
// fact like `acc != ~0` is provable for most columns): `(acc & tok) == 0` with tok the output of
// an `s_mov_b32 tok, 0` asm statement, i.e. really zero at run time but opaque to the compiler.
// Every pin renews tok from the previous one, so each token has ONE user (one shared token makes
// every known-bits query walk thousands of assumptions: 20 min of InstCombine instead of 3).
// Assumptions are dropped before instruction selection, the asm statements (not volatile, outputs
// then unused) die with them: the pins cost no instruction, no register and no scheduling edge.
// (Tried and rejected: inline-asm MADs — the hazard recogniser puts an s_nop after every asm that
// defines a VGPR its neighbour reads; empty asm "sinks" as second use — the scheduler parks them
// at the end of the multiplication and all 153 partial sums stay live, i.e. spill.)
//
// q = 1 (mod 2^29)  =>  -q^-1 = -1 (mod 2^29): the quotient digit of a column with true sum A is
// m = -A mod 2^29.  The accumulator deliberately runs ONE BEHIND the true column sum (acc = A - 1)
// from column 1 on, which makes both the digit and the carry free of any bias:
//   column 0 : A_0 = a0*b0,  m_0 = 2^29 - (A_0 mod 2^29) in [1, 2^29],
//              true carry (A_0 + m_0) / 2^29 = (A_0 >> 29) + 1, kept as acc = A_0 >> 29
//   column k : acc = A_k - 1,  m_k = ~acc mod 2^29 (= -A_k),  (A_k + m_k) / 2^29 = (acc >> 29) + 1
//   column 9 : the missing 1 is returned through result limb 0 (which may therefore equal 2^29).
// m_0 >= 1 is what keeps every later true column sum >= 1, i.e. acc >= 0 with unsigned shifts.
// Result = (a*b + M*q) / 2^261 with M <= 2^261, limbs 1..7 < 2^29, limb 0 <= 2^29.
// One multiplication is one serial MAD chain; keeping the scheduler from interleaving several of
// them (to hide a latency that a dependent v_mad_u64_u32 chain does not have: tools/microbench
// mad_u64_u32_chain) keeps the live set to one accumulator and one digit vector.  The fence lets
// scalar and memory instructions cross, vector ALU instructions not.
#define DSV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0x0f4) /* may cross: SALU | all VMEM | all DS */
DSV_DEV void mad_pin(u64& acc, u32& tok, u32 x, u32 y) {
  acc += (u64)x * y;
  asm("s_mov_b32 %0, 0" : "+s"(tok));
  __builtin_assume(((u32)acc & tok) == 0);
}

// `products(k, acc, tok)` adds the operand products of column k (k = 0..16) with mad_pin
template <class P>
DSV_DEV Fe fe_mont_fips(P&& products) {
  u32 m[NL];
  Fe r;
  u32 tok;
  DSV_SCHED_FENCE();
  asm("s_mov_b32 %0, 0" : "=s"(tok));
  u64 acc = 0;
  products(0, acc, tok);
  m[0] = ((~(u32)acc) & M29) + 1;
  acc >>= 29;
#pragma unroll
  for (int k = 1; k < 2 * NL - 1; k++) {
    products(k, acc, tok);
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (i < k && j >= 1 && j < NL) mad_pin(acc, tok, m[i], kQ29[j]);
    }
    if (k < NL)
      m[k] = (~(u32)acc) & M29;
    else
      r.l[k - NL] = (u32)acc & M29;
    acc >>= 29;
  }
  r.l[0] += 1;
  r.l[NL - 1] = (u32)acc;
  DSV_SCHED_FENCE();
  return r;
}

DSV_DEV Fe fe_mul(const Fe& a, const Fe& b) {
  return fe_mont_fips([&](int k, u64& acc, u32& tok) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (j >= 0 && j < NL) mad_pin(acc, tok, a.l[i], b.l[j]);
    }
  });
}

// squaring: cross terms once, against a pre-doubled operand (45 MADs instead of 81)
DSV_DEV Fe fe_sqr(const Fe& a) {
  u32 d[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) d[i] = a.l[i] << 1;
  return fe_mont_fips([&](int k, u64& acc, u32& tok) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int j = k - i;
      if (j > i && j < NL) mad_pin(acc, tok, d[i], a.l[j]);
    }
    if ((k & 1) == 0) mad_pin(acc, tok, a.l[k / 2], a.l[k / 2]);
  });
}

DSV_DEV Fe fe_add(const Fe& a, const Fe& b) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
  return r;
}
DSV_DEV Fe fe_dbl(const Fe& a) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] << 1;
  return r;
}
// one parallel carry pass: limbs < 2^32 in -> limbs 0..7 < 2^29 + 8 out (limb 8 absorbs)
DSV_DEV Fe fe_carry(const Fe& a) {
  Fe r;
  r.l[0] = a.l[0] & M29;
#pragma unroll
  for (int i = 1; i < NL - 1; i++) r.l[i] = (a.l[i] & M29) + (a.l[i - 1] >> 29);
  r.l[NL - 1] = a.l[NL - 1] + (a.l[NL - 2] >> 29);
  return r;
}
template <int K>
DSV_DEV Fe fe_sub_bias_raw(const Fe& a, const Fe& b) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 bias = (K == 2) ? kBias2[i] : (K == 4) ? kBias4[i] : kBias8[i];
    r.l[i] = a.l[i] + (bias - b.l[i]);
  }
  return r;
}
template <int K>
DSV_DEV Fe fe_sub_bias(const Fe& a, const Fe& b) {
  return fe_carry(fe_sub_bias_raw<K>(a, b));
}
DSV_DEV Fe fe_sub2(const Fe& a, const Fe& b) { return fe_sub_bias<2>(a, b); }
// a + 2q - b WITHOUT the carry pass: limbs up to a.l + bias2 limb (< 2^31).  Only for results
// whose every use is a multiplication by an operand with limbs < 2^30 (or fe_sub4w): the column
// sums of that product still fit 64 bits — proved for the worst case by tests/fe29_bounds.py
// (the blanket 1.5 * 2^60 per-product contract above is sufficient, not necessary).
DSV_DEV Fe fe_sub2_raw(const Fe& a, const Fe& b) { return fe_sub_bias_raw<2>(a, b); }
DSV_DEV Fe fe_sub4(const Fe& a, const Fe& b) { return fe_sub_bias<4>(a, b); }
// a + 4q - b for an un-carried subtrahend (a fe_sub2_raw result, limbs < 2^31): the bias limbs
// are lifted to >= 2^31 - 4.  a limbs < 2^30.  One carry pass.
DSV_DEV Fe fe_sub4w(const Fe& a, const Fe& b) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + (kBias4W[i] - b.l[i]);
  return fe_carry(r);
}
DSV_DEV Fe fe_sub8(const Fe& a, const Fe& b) { return fe_sub_bias<8>(a, b); }
DSV_DEV Fe fe_neg2(const Fe& b) { return fe_sub_bias<2>(fe_zero(), b); }

DSV_DEV Fe fe_select(bool c, const Fe& a, const Fe& b) {  // c ? a : b
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// ---- canonical form ------------------------------------------------------------------------
// full ripple carry; value unchanged, limbs 0..7 < 2^29
DSV_DEV Fe fe_ripple(const Fe& a) {
  Fe r;
  u32 k = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; i++) {
    u32 s = a.l[i] + k;  // callers keep a.l[i] < 2^31
    r.l[i] = s & M29;
    k = s >> 29;
  }
  r.l[NL - 1] = a.l[NL - 1] + k;
  return r;
}
// r = a - m if a >= m else a   (both ripple-normalised)
DSV_DEV Fe fe_cond_sub(const Fe& a, const u32 (&m)[NL]) {
  Fe d;
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u32 s = a.l[i] - m[i] - borrow;
    borrow = s >> 31;  // limbs < 2^30 so bit 31 set iff negative
    d.l[i] = (i < NL - 1) ? (s & M29) : s;
  }
  return fe_select(borrow != 0, a, d);
}
// unique representative in [0, q) of a value < 16 q (plain integer, not a Montgomery op)
DSV_DEV Fe fe_canon(const Fe& a) {
  Fe r = fe_ripple(a);
  r = fe_cond_sub(r, kQx8);
  r = fe_cond_sub(r, kQx4);
  r = fe_cond_sub(r, kQx2);
  r = fe_cond_sub(r, kQx1);
  return r;
}
DSV_DEV bool fe_is_zero_canon(const Fe& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) o |= a.l[i];
  return o == 0;
}
// a == b (mod q) for lazily reduced a, b (limbs <= 2^30 - 2, b < 7.9 q)
DSV_DEV bool fe_equal(const Fe& a, const Fe& b) {
  return fe_is_zero_canon(fe_canon(fe_sub8(a, b)));
}

// ---- conversions: 8 x u32 little-endian words (canonical integer) <-> fe29 -----------------
DSV_DEV Fe fe_from_words_plain(const u32 (&w)[8]) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = 29 * i;
    const int wi = bit >> 5, sh = bit & 31;
    u32 lo = w[wi] >> sh;
    if (sh > 3 && wi + 1 < 8) lo |= w[wi + 1] << (32 - sh);
    r.l[i] = lo & M29;
  }
  return r;
}
DSV_DEV void fe_to_words_plain(u32 (&w)[8], const Fe& a) {  // a canonical (limbs < 2^29)
#pragma unroll
  for (int k = 0; k < 8; k++) {
    // word k covers bits [32k, 32k+32)
    const int lo_limb = (32 * k) / 29, off = (32 * k) % 29;
    u32 v = a.l[lo_limb] >> off;
    int have = 29 - off;
    if (lo_limb + 1 < NL) v |= a.l[lo_limb + 1] << have;
    have += 29;
    if (have < 32 && lo_limb + 2 < NL) v |= a.l[lo_limb + 2] << have;
    w[k] = v;
  }
}
DSV_DEV Fe fe_to_mont(const Fe& plain) { return fe_mul(plain, fe_const(kR2)); }
// Montgomery -> canonical plain integer limbs
DSV_DEV Fe fe_from_mont(const Fe& a) {
  Fe one = fe_zero();
  one.l[0] = 1;
  return fe_canon(fe_mul(a, one));
}

// words < modulus ?  (8 x u32 LE)
DSV_DEV bool words_lt(const u32 (&w)[8], const u32 (&m)[8]) {
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 d = (u64)w[i] - m[i] - borrow;
    borrow = (u32)(d >> 63);
  }
  return borrow != 0;
}

}  // namespace dsv
