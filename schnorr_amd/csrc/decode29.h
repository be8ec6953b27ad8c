// decode29.h — wire-format decoding on the device: JubJub point decompression
// (`JubJubAffine::from_bytes`, reached from `Signature::from_bytes` / `PublicKey::from_bytes`,
// /root/reference/src/signatures.rs:117-122, src/keys/public.rs:94-100) — the step in front of
// verify for callers that hold serialized signatures and keys.
//
// Encoding: 32 bytes = canonical v (255 bits) with bit 255 = lowest bit
// of canonical u.  Decode: v < q required; u^2 = (v^2 - 1) / (1 + d v^2); reject if not a square;
// pick the root whose parity matches the sign bit.  (Like the oracle, and like the jubjub
// lineage dusk forked, no further canonicity or subgroup check is applied.)
//
// q - 1 = 2^32 * t: square roots by Tonelli-Shanks.  Lanes diverge only in how many squarings
// the discrete-log step needs; every loop is bounded (<= 32 outer x 32 inner iterations).
#pragma once
#include "fe29.h"

namespace dsv {

__device__ constexpr u32 kSqrtE[7] = DSV_SQRT_E_WORDS;   // (t - 1) / 2
__device__ constexpr u32 kRootOfUnity[NL] = DSV_ROOT_OF_UNITY;
__device__ constexpr u32 kD[NL] = DSV_D;

// a == 1 (Montgomery one) for a multiplication output (limbs < 2^29, value < 2q)
DSV_DEV bool fe_is_one(const Fe& a) { return fe_is_zero_canon(fe_canon(fe_sub2(a, fe_one()))); }

// square root in Fq.  Returns false when a is a non-residue.  a: multiplication output (N).
DSV_DEV bool fe_sqrt(Fe& root, const Fe& a) {
  // w = a^((t-1)/2)
  Fe w = fe_one();
#pragma unroll 1
  for (int bit = DSV_SQRT_E_BITS - 1; bit >= 0; bit--) {
    w = fe_sqr(w);
    if ((kSqrtE[bit >> 5] >> (bit & 31)) & 1) w = fe_mul(w, a);
  }
  Fe x = fe_mul(a, w);  // a^((t+1)/2)
  Fe b = fe_mul(x, w);  // a^t
  Fe c = fe_const(kRootOfUnity);
  int v = 32;
  bool ok = true;
#pragma unroll 1
  for (int round = 0; round < 32; round++) {
    if (fe_is_one(b) || !ok) break;
    // least k with b^(2^k) == 1
    int k = 0;
    Fe bb = b;
#pragma unroll 1
    while (k < v) {
      bb = fe_sqr(bb);
      k++;
      if (fe_is_one(bb)) break;
    }
    if (k >= v) {  // order of b does not divide 2^(v-1): non-residue (or a == 0)
      ok = false;
      break;
    }
    Fe cc = c;
#pragma unroll 1
    for (int j = 0; j < v - k - 1; j++) cc = fe_sqr(cc);
    x = fe_mul(x, cc);
    c = fe_sqr(cc);
    b = fe_mul(b, c);
    v = k;
  }
  root = x;
  // a == 0: x == 0 is the root; otherwise confirm x^2 == a
  return fe_equal(fe_sqr(x), a);
}

}  // namespace dsv
