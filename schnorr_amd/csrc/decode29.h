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
// tracking the INVERSE square root y (invariant y^2 z = b, b in the 2^32-torsion of Fq*):
//   y = z^((t-1)/2),  b = z^t;  while b != 1: k = log2(order of b), y *= c^(2^(31-k)),
//   b *= c^(2^(32-k))          (c = 7^t, powers from a 33-entry constant table)
// z^((t-1)/2) uses fixed 3-bit windows over the constant exponent (digits are wave-uniform).
// Lanes diverge only in how many squarings the order search needs; every loop is bounded.
#pragma once
#include "fe29.h"

namespace dsv {

__device__ constexpr u32 kSqrtE[7] = DSV_SQRT_E_WORDS;  // (t - 1) / 2, DSV_SQRT_E_BITS bits
__device__ constexpr u32 kD[NL] = DSV_D;
__constant__ u32 c_root_powers[33][NL];  // c^(2^j), j = 0..32

// a == 1 (Montgomery one) for a multiplication output (limbs < 2^29, value < 2q)
DSV_DEV bool fe_is_one(const Fe& a) { return fe_is_zero_canon(fe_canon(fe_sub2(a, fe_one()))); }

// z^((t-1)/2)
DSV_DEV Fe fe_pow_sqrt_exp(const Fe& z) {
  Fe t2 = fe_sqr(z), t3 = fe_mul(t2, z), t4 = fe_sqr(t2), t5 = fe_mul(t4, z), t6 = fe_sqr(t3),
     t7 = fe_mul(t6, z);
  static_assert(DSV_SQRT_E_BITS % 3 == 0, "3-bit windows");
  Fe acc = fe_one();
#pragma unroll 1
  for (int w = DSV_SQRT_E_BITS / 3 - 1; w >= 0; w--) {
    acc = fe_sqr(fe_sqr(fe_sqr(acc)));
    const int pos = 3 * w;
    u32 lo = kSqrtE[pos >> 5] >> (pos & 31);
    if ((pos & 31) > 29 && (pos >> 5) + 1 < 7) lo |= kSqrtE[(pos >> 5) + 1] << (32 - (pos & 31));
    switch (lo & 7u) {  // wave-uniform
      case 1: acc = fe_mul(acc, z); break;
      case 2: acc = fe_mul(acc, t2); break;
      case 3: acc = fe_mul(acc, t3); break;
      case 4: acc = fe_mul(acc, t4); break;
      case 5: acc = fe_mul(acc, t5); break;
      case 6: acc = fe_mul(acc, t6); break;
      case 7: acc = fe_mul(acc, t7); break;
      default: break;
    }
  }
  return acc;
}

// y = z^(-1/2) for a non-zero square z (any root); unspecified otherwise — callers validate.
DSV_DEV Fe fe_inv_sqrt(const Fe& z) {
  Fe y = fe_pow_sqrt_exp(z);
  Fe b = fe_mul(fe_mul(z, y), y);  // z^t
  int v = 32;
#pragma unroll 1
  for (int round = 0; round < 32; round++) {
    if (fe_is_one(b)) break;
    int k = 0;
    Fe bb = b;
#pragma unroll 1
    while (k < v - 1) {
      bb = fe_sqr(bb);
      k++;
      if (fe_is_one(bb)) break;
    }
    if (!fe_is_one(bb)) break;  // order of b is 2^v or b == 0: z is not a non-zero square
    Fe cc, c2;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      cc.l[i] = c_root_powers[31 - k][i];
      c2.l[i] = c_root_powers[32 - k][i];
    }
    y = fe_mul(y, cc);
    b = fe_mul(b, c2);
    v = k;
  }
  return y;
}

}  // namespace dsv
