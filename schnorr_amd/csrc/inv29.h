// inv29.h — field inversion by the extended Euclidean algorithm (r04).
//
// `JubJubExtended::to_hash_inputs` (call sites /root/reference/src/signatures.rs:131, :280-281) inverts
// z; k_normalize_uvz shares ONE inversion among the 16 - 128 points of a lane, and that inversion is
// the lane's dependent chain: by Fermat (decode29.h: fe_invert, 255 squarings + 91 multiplications
// = 66 k instructions) it kept a compute lane of the host pipeline busy for 0.8 ms per 2^18-item
// chunk with 512 waves (profiles/r04/host_timeline_affine_ext_final.txt).  Euclid on (q, x) with the
// float-guided half-steps of halfgcd.h — quotient estimate from double-precision images, never
// above the true quotient, alternating roles — needs ~150 half-steps of ~150 instructions.
//
//   remainders  r_0 = q, r_1 = x, r_{i+1} = r_{i-1} - q_i r_i        (A holds the even, B the odd ones)
//   cofactors   t_0 = 0, t_1 = 1, t_{i+1} = t_{i-1} - q_i t_i,  r_i = t_i x (mod q),  |t_i| <= q / r_{i-1}
// Magnitudes are stored; cofactors on the A side are <= 0, on the B side >= 0.  The loop ends when
// one side is 0; the other then holds gcd(q, x) = 1 (q is prime, x != 0) and its cofactor is +-1/x.
// Exactness does not rest on the floating-point arithmetic: an estimate is only ever too SMALL, a
// too-small quotient leaves a valid (longer) Euclidean pair, and the result is checked —
// r == 1 — before it is used; x = 1 is answered at once, anything else (x = 0, a quotient above
// 2^31 - 1 such as for x = 2, the iteration cap) falls back to Fermat.  Tests: test_inversion_edge_values_through_to_hash_inputs
// (tests/test_gpu_r04.py) against Python integers; every projective / limb parity test and soak runs
// through it.
#pragma once
#include "decode29.h"
#include "halfgcd.h"

namespace dsv {

constexpr int kInvMaxIter = 1024;  // half-steps per role pair; random inputs need ~75

// X -= qe * Y, tX += qe * tY with qe = floor(X / Y * (1 - 2^-30)) from the images (halfgcd.h:
// half_step, with 8-word cofactors); returns false if the quotient does not fit 31 bits
DSV_DEV bool inv_step(u32 (&X)[8], u32 (&tX)[8], const u32 (&Y)[8], const u32 (&tY)[8], double dX, double dY) {
  const double qd = dX / dY * (1.0 - 0x1p-30);
  if (!(qd < 2147483647.0)) return false;  // (also NaN / infinity: dY == 0 never gets here)
  u32 qe = (u32)qd;
  if (qe == 0 && dX >= dY * (1.0 - 0x1p-28)) {  // too close to call from the images: compare exactly
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
  for (int i = 0; i < 8; i++) {
    const u64 p = (u64)qe * tY[i] + tX[i] + carry;
    tX[i] = (u32)p;
    carry = (u32)(p >> 32);
  }
  return true;
}
DSV_DEV bool is_zero8(const u32 (&x)[8]) { return (x[0] | x[1] | x[2] | x[3] | x[4] | x[5] | x[6] | x[7]) == 0; }
DSV_DEV bool is_one8(const u32 (&x)[8]) { return x[0] == 1 && (x[1] | x[2] | x[3] | x[4] | x[5] | x[6] | x[7]) == 0; }

// 1/z in Montgomery form for a Montgomery-form z (any lazily reduced representative); 0 for z = 0
DSV_DEV Fe fe_invert_euclid(const Fe& z) {
  u32 A[8], B[8], tA[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tB[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  fe_to_words_plain(B, fe_from_mont(z));  // canonical plain value, < q
  // x = 1: the product of a lane's z's whenever every point of the lane has z = 1 — the most common
  // real input (deserialised keys and signatures, this library's own sign / keygen outputs).  The
  // first quotient would be q itself (> 31 bits) and the lane would pay a failed attempt PLUS the whole
  // Fermat chain; 1/1 needs neither.
  if (is_one8(B)) return fe_one();
#pragma unroll
  for (int i = 0; i < 8; i++) A[i] = kQ32[i];
  double dA = to_double8(A), dB = to_double8(B);
  bool ok = true;
  int it = 0;
#pragma unroll 1
  for (; it < kInvMaxIter && ok && dB != 0.0; it++) {
    ok = inv_step(A, tA, B, tB, dA, dB);   // A -= q B
    dA = to_double8(A);
    if (dA == 0.0 || !ok) break;
    ok = inv_step(B, tB, A, tA, dB, dA);   // B -= q A
    dB = to_double8(B);
  }
  // the side that is not 0 holds the gcd: 1 for every x != 0
  const bool on_b = is_zero8(A);
  u32 t[8];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = on_b ? tB[i] : tA[i];
  const bool good = ok && it < kInvMaxIter && (on_b ? is_one8(B) : (is_zero8(B) && is_one8(A))) && words_lt(t, kQ32);
  if (!good) return fe_invert(z);  // x = 0 (-> 0), an oversize quotient (x = 1, 2, ...), the iteration cap
  // 1/x = +t on the B side, -t on the A side; back to Montgomery form
  const Fe m = fe_to_mont(fe_from_words_plain(t));
  return fe_select(on_b, m, fe_neg2(m));
}

}  // namespace dsv
