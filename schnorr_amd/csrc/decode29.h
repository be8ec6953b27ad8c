// decode29.h — wire-format decoding on the device: JubJub point decompression
// (`JubJubAffine::from_bytes`, reached from `Signature::from_bytes` / `PublicKey::from_bytes`,
// /root/reference/src/signatures.rs:117-122, src/keys/public.rs:94-100) — the step in front of
// verify for callers that hold serialized signatures and keys.
//
// Encoding: 32 bytes = canonical v (255 bits) with bit 255 = lowest bit
// of canonical u.  Decode: v < q required; u^2 = n / d with n = v^2 - 1, d = 1 + d_curve v^2;
// reject if not a square; pick the root whose parity matches the sign bit.  (Like the oracle, and
// like the jubjub lineage dusk forked, no further canonicity or subgroup check is applied.)
//
// No inversion: u = n * (n d)^(-1/2).  q - 1 = 2^32 * t; Tonelli-Shanks is run on z = n d while
// tracking the INVERSE square root y (invariant y^2 z = b, b = g^s in the 2^32-torsion of Fq*,
// g = 7^t): y = z^((t-1)/2), b = z^t, then s is cancelled in four 8-bit windows (fe_inv_sqrt).
// z^((t-1)/2) uses sliding 4-bit windows over the constant exponent (its bits are wave-uniform).
// No lane-dependent control flow at all: every lane runs the same ~350 field operations.
#pragma once
#include "fe29.h"

namespace dsv {

__device__ constexpr u32 kSqrtE[7] = DSV_SQRT_E_WORDS;  // (t - 1) / 2, DSV_SQRT_E_BITS bits
__device__ constexpr u32 kD[NL] = DSV_D;

// a == 1 (Montgomery one) for a multiplication output (limbs < 2^29, value < 2q)
DSV_DEV bool fe_is_one(const Fe& a) { return fe_is_zero_canon(fe_canon(fe_sub2(a, fe_one()))); }

// z^e for a wave-uniform constant exponent of exactly NBITS bits: SLIDING windows of up to 4 bits
// over the odd powers z, z^3, .., z^15 (1 squaring + 7 multiplications), scanned from the top —
// every branch is on bits of the constant, i.e. scalar.  NBITS - 1 squarings (+ 1) and one
// multiplication per window: 45 + 7 for (t-1)/2 (r02: 74 + 5 with fixed 3-bit windows), 53 + 7
// for q - 2 (r02: 85 + 5).  One copy of the multiplication: the window's operand is selected first.
template <int NWORDS, int NBITS>
DSV_DEV Fe fe_pow_const(const Fe& z, const u32 (&e)[NWORDS]) {
  const Fe z2 = fe_sqr(z);
  const Fe t1 = z, t3 = fe_mul(t1, z2), t5 = fe_mul(t3, z2), t7 = fe_mul(t5, z2), t9 = fe_mul(t7, z2),
           t11 = fe_mul(t9, z2), t13 = fe_mul(t11, z2), t15 = fe_mul(t13, z2);
  auto bit = [&](int k) -> u32 { return (e[k >> 5] >> (k & 31)) & 1u; };
  Fe acc = fe_one();
  bool first = true;
  int i = NBITS - 1;
#pragma unroll 1
  while (i >= 0) {
    if (bit(i) == 0) {  // (never before the first window: bit NBITS - 1 is set)
      acc = fe_sqr(acc);
      i--;
      continue;
    }
    int j = i >= 3 ? i - 3 : 0;  // the window is bits i .. j, j the lowest SET bit within 4 of i
    while (bit(j) == 0) j++;
    u32 val = 0;
    for (int k = i; k >= j; k--) val = 2 * val + bit(k);
    Fe m;
    switch (val >> 1) {  // val is odd: 1, 3, .., 15
      case 0: m = t1; break;
      case 1: m = t3; break;
      case 2: m = t5; break;
      case 3: m = t7; break;
      case 4: m = t9; break;
      case 5: m = t11; break;
      case 6: m = t13; break;
      default: m = t15; break;
    }
    if (first) {
      acc = m;
      first = false;
    } else {
#pragma unroll 1
      for (int k = i - j + 1; k > 0; k--) acc = fe_sqr(acc);
      acc = fe_mul(acc, m);
    }
    i = j - 1;
  }
  return acc;
}
// z^((t-1)/2)
DSV_DEV Fe fe_pow_sqrt_exp(const Fe& z) { return fe_pow_const<7, DSV_SQRT_E_BITS>(z, kSqrtE); }
// z^(q-2) = 1/z  (0 for z = 0)
DSV_DEV Fe fe_invert(const Fe& z) {
  // q - 2: the low word of q is 1, so the subtraction borrows from word 1
  const u32 e[8] = {0xffffffffu, kQ32[1] - 1, kQ32[2], kQ32[3], kQ32[4], kQ32[5], kQ32[6], kQ32[7]};
  return fe_pow_const<8, 255>(z, e);
}

// tables for the windowed discrete log (device global memory, filled at dsv_init)
struct TsTables {
  const u32* cancel;      // [4][256][9]: g^(-d 2^(8i-1)) (i = 0: g^(-(d >> 1)))
  const uint8_t* hash;    // [2^DSV_TS_HASH_BITS]: perfect hash of <h>, h = g^(2^24), to its log
};

// log_h(w) for w in the order-256 subgroup (anything in [0, 255] otherwise)
DSV_DEV u32 ts_digit(const Fe& w, const TsTables& t) {
  const Fe c = fe_canon(w);
  const u32 idx = (c.l[0] * DSV_TS_HASH_K1 + c.l[1] * DSV_TS_HASH_K2) >> (32 - DSV_TS_HASH_BITS);
  return t.hash[idx];
}
DSV_DEV Fe ts_cancel(const TsTables& t, int window, u32 d) {
  const u32* p = t.cancel + ((size_t)window * 256 + d) * NL;
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = p[i];
  return r;
}

// y = z^(-1/2) for a non-zero square z (any root); unspecified otherwise — callers validate.
// Invariant y^2 z = b with b = g^s in the 2^32-torsion (s even for a square z).  s is read off in
// four 8-bit digits, least significant first: b^(2^(24-8i)) lies in <h> and equals h^(digit i);
// each digit is cancelled from b (A^2) and half of it from y (A).  48 + 3 squarings and 7
// multiplications instead of the ~250 squarings of the bit-by-bit order search.
DSV_DEV Fe fe_inv_sqrt(const Fe& z, const TsTables& t) {
  Fe y = fe_pow_sqrt_exp(z);
  Fe b = fe_mul(fe_mul(z, y), y);  // z^t
#pragma unroll 1
  for (int w = 0; w < 3; w++) {
    Fe p = b;
#pragma unroll 1
    for (int j = 0; j < 24 - 8 * w; j++) p = fe_sqr(p);
    const Fe a = ts_cancel(t, w, ts_digit(p, t));
    y = fe_mul(y, a);
    b = fe_mul(b, fe_sqr(a));
  }
  return fe_mul(y, ts_cancel(t, 3, ts_digit(b, t)));
}

}  // namespace dsv
