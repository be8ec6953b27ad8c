// halfgcd.h — half-size scalars for the verification equation, exact on the WHOLE curve group.
//
// The reference checks  Q := u*G + c*PK - R == O   (/root/reference/src/keys/public.rs:127-129)
// with a 250-bit challenge c on the variable base PK: ~250 doublings.  For any integers (a, b)
// with  a = b*c (mod 8r),  b odd,  0 < |b| < r:
//        b*Q = (b*u mod r)*G + a*PK - b*R        and        b*Q == O  <=>  Q == O.
// Proof.  E(Fq) = Z_r x E[8] (order 8r, r prime).  G has order exactly r, so b*(u*G) =
// (b*u mod r)*G.  Every point is killed by 8r, so (b*c)*PK = a*PK whenever a = b*c (mod 8r) —
// this covers a small-order component of PK, which the reference's types can hold.  b*R uses the
// integer b itself.  Finally multiplication by b is injective: on Z_r because 0 < |b| < r, on the
// 2-group E[8] because b is odd.  Hence the verdict is identical to the reference's for every
// on-curve input, with or without torsion (tests: test_half_scalar_check_is_exact_with_torsion,
// GPU: test_identity_small_order_and_default_signature).
//
// (a, b) comes from the extended Euclidean algorithm on (8r, c): remainders r_i = t_i*c (mod 8r);
// stop at the first r_i < 2^128, where |t_i| < 2^127.  Consecutive cofactors are coprime, so if
// t_i is even, t_{i-1} is odd and a balanced combination (r_{i-1} - k r_i, t_{i-1} + k t_i) is
// used instead.  Sizes are ~128 bits: the variable-base part of a verification becomes a two-base
// Straus chain of ~33 signed 4-bit windows instead of 63.
//
// Lanes diverge in iteration count only; every lane exits after at most kHalfGcdMaxIter
// iterations (the pair it holds then is still a valid one, just longer).
#pragma once
#include "fe29.h"

namespace dsv {

// 8 * r  (255 bits)
__device__ constexpr u32 kN8R[8] = {0xb7b965b8u, 0x84b872f6u, 0x66408416u, 0x3341049eu,
                                    0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
constexpr int kHalfGcdMaxIter = 512;

DSV_DEV int bitlen8(const u32 (&x)[8]) {
  int len = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (x[i] != 0) len = 32 * i + (32 - __clz(x[i]));
  return len;
}
// d = a - b, returns true if a < b (borrow out)
DSV_DEV bool sub8(u32 (&d)[8], const u32 (&a)[8], const u32 (&b)[8]) {
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 t = (u64)a[i] - b[i] - borrow;
    d[i] = (u32)t;
    borrow = (u32)(t >> 63);
  }
  return borrow != 0;
}
DSV_DEV bool below_2_128(const u32 (&x)[8]) { return (x[4] | x[5] | x[6] | x[7]) == 0; }

// value of an 8-word integer as a double (relative error < 2^-51: plenty for a quotient estimate)
DSV_DEV double to_double8(const u32 (&x)[8]) {
  double d = (double)x[7];
#pragma unroll
  for (int i = 6; i >= 0; i--) d = __builtin_fma(d, 4294967296.0, (double)x[i]);
  return d;
}

// out: a (magnitude, 8 words), b (magnitude, 8 words, only 5 can be non-zero), b_neg.
// a = (b_neg ? -b : b) * c  (mod 8r),  b odd.   c < 2^250.
//
// Euclid on (8r, c) with quotient ESTIMATES and ALTERNATING roles: a half-step reduces X by Y with
// qe = floor(X/Y * (1 - 2^-30)) taken from double-precision images (qe <= true quotient, clamped
// below 2^31; qe = 0 when X < Y), X -= qe*Y, tX += qe*tY; first (X, Y) = (A, B), then (B, A), and
// so on — no trial subtraction, no conditional swap of the 13 words, and only the operand that
// changed is converted to a double again (r01 compared, swapped and converted both operands in
// every iteration: 208 instructions per Euclidean step as compiled; same-box A/B +0.95 % on the
// whole verification step, profiles/r02/ab_alternating_halfgcd.txt).  An under-estimate leaves X >= Y;
// the following half-step then has quotient 0 and changes nothing, and the one after it finishes
// the job (with an exact comparison when X / Y is within 2^-28 of 1, where the estimate alone
// would say 0 again), so every value that drops below the other is a true Euclidean remainder and the
// (remainder, cofactor) pairs are those of the exact algorithm (tests/pymodel.py: half_scalars).
// The loop ends at the first remainder below 2^128 (its partner is still >= 2^128).  Cofactors of
// the A side are <= 0, those of the B side >= 0 (magnitudes are stored), so the sign of the final
// cofactor is known from the side it is on.
DSV_DEV void half_step(u32 (&X)[8], u32 (&tX)[5], const u32 (&Y)[8], const u32 (&tY)[5], double dX,
                       double dY) {
  const double qd = dX / dY * (1.0 - 0x1p-30);
  u32 qe = qd >= 2147483647.0 ? 2147483647u : (u32)qd;
  // X / Y in [1, 1 + 2^-30) rounds to quotient 0 in BOTH roles: without help the loop would spin on
  // that pair until kHalfGcdMaxIter (r02 did: ~2^-24 per signature, correct verdict, one wave 13x
  // slower).  When the images are too close to call, compare exactly.
  if (qe == 0 && dX >= dY * (1.0 - 0x1p-28)) {
    u32 d[8];
    if (!sub8(d, X, Y)) qe = 1;
  }
  u32 mc = 0, borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 p = (u64)qe * Y[i] + mc;
    mc = (u32)(p >> 32);
    const u64 d = (u64)X[i] - (u32)p - borrow;
    X[i] = (u32)d;
    borrow = (u32)(d >> 63);
  }
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const u64 p = (u64)qe * tY[i] + tX[i] + carry;
    tX[i] = (u32)p;
    carry = (u32)(p >> 32);
  }
}
DSV_DEV void half_scalars(u32 (&a)[8], u32 (&b)[8], bool& b_neg, const u32 (&c)[8]) {
  u32 A[8], B[8], tA[5] = {0, 0, 0, 0, 0}, tB[5] = {1, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    A[i] = kN8R[i];
    B[i] = c[i];
  }
  bool done = below_2_128(B);
  bool final_is_A = false;
  double dA = to_double8(A), dB = to_double8(B);
#pragma unroll 1
  for (int it = 0; it < kHalfGcdMaxIter && !done; it++) {
    half_step(A, tA, B, tB, dA, dB);
    dA = to_double8(A);
    if (below_2_128(A)) {
      done = true;
      final_is_A = true;
    } else {
      half_step(B, tB, A, tA, dB, dA);
      dB = to_double8(B);
      done = below_2_128(B);
    }
  }
  // from here on (B, tB) is the final pair and (A, tA) the one before it
  if (final_is_A) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32 x = A[i];
      A[i] = B[i];
      B[i] = x;
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const u32 x = tA[i];
      tA[i] = tB[i];
      tB[i] = x;
    }
  }
  const bool neg = final_is_A;  // sign of the cofactor paired with B
  const bool use_b = (tB[0] & 1) != 0;
  if (use_b) {
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = B[i];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = i < 5 ? tB[i] : 0u;
    b_neg = neg;
    return;
  }
  // tB is even, so tA (coprime to it) is odd, and so is tA + k*tB for every k: all the vectors
  // (A - k*B, tA + k*tB), 0 <= k <= A/B, qualify.  Take k near (A - tA) / (B + tB), where the two
  // components balance, instead of k = 0 (whose first component can be many bits longer): the
  // longest lane of a wave sets the length of the Straus chain for all 64.
  u32 bestA[8], bestT[5];
#pragma unroll
  for (int i = 0; i < 8; i++) bestA[i] = A[i];
#pragma unroll
  for (int i = 0; i < 5; i++) bestT[i] = tA[i];
  u32 tA8[8], tB8[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    tA8[i] = i < 5 ? tA[i] : 0u;
    tB8[i] = i < 5 ? tB[i] : 0u;
  }
  int best = bitlen8(A);  // >= 129 > bitlen(tA)
  const double kd = (to_double8(A) - to_double8(tA8)) / (to_double8(B) + to_double8(tB8));
  if (kd >= 1.0 && kd < 2147483000.0) {
    const u32 k0 = (u32)kd;
#pragma unroll 1
    for (u32 kk = k0; kk <= k0 + 1; kk++) {
      u32 ca[8], ct[8];
      u32 mc = 0, borrow = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const u64 p = (u64)kk * B[i] + mc;
        mc = (u32)(p >> 32);
        const u64 d = (u64)A[i] - (u32)p - borrow;
        ca[i] = (u32)d;
        borrow = (u32)(d >> 63);
      }
      u32 carry = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const u64 p = (u64)kk * tB8[i] + tA8[i] + carry;
        ct[i] = (u32)p;
        carry = (u32)(p >> 32);
      }
      const bool valid = borrow == 0 && mc == 0 && (ct[5] | ct[6] | ct[7]) == 0;
      const int la = bitlen8(ca), lt = bitlen8(ct);
      const int size = la > lt ? la : lt;
      if (valid && size < best) {
        best = size;
#pragma unroll
        for (int i = 0; i < 8; i++) bestA[i] = ca[i];
#pragma unroll
        for (int i = 0; i < 5; i++) bestT[i] = ct[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = bestA[i];
#pragma unroll
  for (int i = 0; i < 8; i++) b[i] = i < 5 ? bestT[i] : 0u;
  b_neg = !neg;  // sign of t_{i-1}; adding multiples of t_i (opposite sign, subtracted) keeps it
}

}  // namespace dsv
