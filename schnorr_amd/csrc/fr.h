// fr.h — JubJub scalar field Fr (order r of the prime subgroup) on gfx950 lanes.
//
// Only signing needs Fr arithmetic: u = r - c * sk  (/root/reference/src/keys/secret.rs:165,
// :237, :448); verification uses scalars as bit strings only.  One multiply + one subtract per
// signature, i.e. < 0.1 % of a signature's work, so this is a plain 8 x 32-bit-limb CIOS
// Montgomery multiply (R = 2^256) with no tuning.
#pragma once
#include "fe29.h"

namespace dsv {

__device__ constexpr u32 kFrR2[8] = {0x95e57731u, 0x67719aa4u, 0x9ce3fc26u, 0x51b0cef0u,
                                     0xc026e9a5u, 0x69dab7fau, 0x8d127688u, 0x04f6547bu};
constexpr u32 kFrInv32 = 0xef788ef9u;  // -r^-1 mod 2^32

// t = a * b * 2^-256 mod r, a, b < r
DSV_DEV void fr_mont_mul(u32 (&out)[8], const u32 (&a)[8], const u32 (&b)[8]) {
  u32 t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll 1
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      u64 x = (u64)a[j] * b[i] + t[j] + c;
      t[j] = (u32)x;
      c = x >> 32;
    }
    u64 x = (u64)t[8] + c;
    t[8] = (u32)x;
    t[9] = (u32)(x >> 32);
    u32 m = t[0] * kFrInv32;
    c = ((u64)m * kR32[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      u64 y = (u64)m * kR32[j] + t[j] + c;
      t[j - 1] = (u32)y;
      c = y >> 32;
    }
    x = (u64)t[8] + c;
    t[7] = (u32)x;
    t[8] = t[9] + (u32)(x >> 32);
  }
  // conditional subtract
  u32 d[8];
  u32 borrow = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u64 y = (u64)t[j] - kR32[j] - borrow;
    d[j] = (u32)y;
    borrow = (u32)(y >> 63);
  }
  const bool ge = (t[8] != 0) | (borrow == 0);
#pragma unroll
  for (int j = 0; j < 8; j++) out[j] = ge ? d[j] : t[j];
}
// a * b mod r
DSV_DEV void fr_mul(u32 (&out)[8], const u32 (&a)[8], const u32 (&b)[8]) {
  u32 t[8];
  fr_mont_mul(t, a, b);
  fr_mont_mul(out, t, kFrR2);
}
// a - b mod r  (a, b < r)
DSV_DEV void fr_sub(u32 (&out)[8], const u32 (&a)[8], const u32 (&b)[8]) {
  u32 d[8];
  u32 borrow = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u64 y = (u64)a[j] - b[j] - borrow;
    d[j] = (u32)y;
    borrow = (u32)(y >> 63);
  }
  u32 carry = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    u64 y = (u64)d[j] + (borrow ? kR32[j] : 0u) + carry;
    out[j] = (u32)y;
    carry = (u32)(y >> 32);
  }
}

}  // namespace dsv
