// rlc.h — geometry and launcher of k_rlc.hip (random-linear-combination fast accept, SURVEY.md
// §8(f)-4), shared with the host side (dsv_rlc.hip).  Kept out of launch.h: that header is part of the
// dominant kernel's translation unit, whose sources are what the recorded roofline evidence is
// valid for (schnorr_amd/build.py: unit_sources_sha256).
#pragma once
#include "launch.h"

namespace dsv {

// One group of n items, c-bit unsigned windows: wpk windows over the 252-bit scalars z_i c_i of the
// keys, wr windows over the 128-bit z_i of the nonce points (separate buckets: k_rlc.hip says why).
// A window's 2^c buckets are a 2^half x 2^half matrix; row / column sums run in nseg segments, the
// per-bit subset sums over them in nseg2 segments, so that no serial chain exceeds ~32 additions.
struct RlcPlan {
  uint32_t n;             // items of this bucket pass ...
  uint32_t first, total;  // ... which are items first .. first + n of a group of `total` (a host call may run its
                          // bucket pass in two ranges: the first while the second is still on the bus)
  int lpts, spts, fixed;  // per item: points with 252-bit scalars, points with the z themselves, fixed-base terms
  int c, half, wpk, wr, windows, nseg, nseg2, key_bits;
  uint32_t kmul;   // floor(2^(wpk c) / r): the keys' scalars get a random multiple of r below it added
  size_t entries;  // n * (wpk * lpts + wr * spts) (key, index) pairs
  size_t buckets;  // windows << c
};
constexpr size_t kRlcMaxGroup = (size_t)1 << 22;  // the pair count (at most 48 n) must fit 32 bits with room to spare
// below this an aggregate is slower than the per-signature kernels (its tail of ~0.9 ms does not shrink with
// the batch: 2^16 items 0.76 x, 2^18 items 1.4 x): with automatic window bits such groups skip it
constexpr size_t kRlcMinAuto = (size_t)1 << 17;
constexpr int kRlcFsumBlocks = 64;
enum : uint32_t { kRlcOffCurve = 1, kRlcTorsion = 2, kRlcSum = 4 };  // flags[0]; flags[1] = 1: chain complete
inline int rlc_default_bits(size_t n) {
  return n >= ((size_t)1 << 19) ? 16 : n >= ((size_t)1 << 17) ? 14 : n >= ((size_t)1 << 14) ? 12 : 8;
}
// even, and the keys' windows cover 252 or 256 bits exactly (10 would need 260: a ninth scalar word)
// the plan of a range [first, first + cnt) of a group planned as `whole`: same bucket geometry, fewer pairs
inline RlcPlan rlc_range(const RlcPlan& whole, size_t first, size_t cnt) {
  RlcPlan p = whole;
  p.n = (uint32_t)cnt;
  p.first = (uint32_t)first;
  p.entries = cnt * ((size_t)p.wpk * p.lpts + (size_t)p.wr * p.spts);
  return p;
}
inline bool rlc_bits_ok(int c) { return c == 4 || c == 6 || c == 8 || c == 12 || c == 14 || c == 16; }
// scheme: 0 single, 1 double, 2 var-generator (k_rlc.hip: k_rlc_prep says which point gets which scalar)
inline RlcPlan rlc_plan(int scheme, size_t n, int c) {
  RlcPlan p;
  p.n = (uint32_t)n;
  p.first = 0;
  p.total = (uint32_t)n;
  p.lpts = scheme == 0 ? 1 : 2;
  p.spts = scheme == 1 ? 2 : 1;
  p.fixed = scheme == 0 ? 1 : (scheme == 1 ? 2 : 0);
  p.c = c;
  p.half = c / 2;
  p.wpk = (252 + c - 1) / c;
  p.wr = (128 + c - 1) / c;
  p.windows = p.wpk + p.wr;
  const int side = 1 << p.half;
  p.nseg = side >= 32 ? side / 16 : 1;
  p.nseg2 = side >= 64 ? side / 32 : 1;
  p.kmul = p.wpk * c == 256 ? 17u : 1u;  // floor(2^256 / r) = 17, floor(2^252 / r) = 1
  p.key_bits = c;
  for (int w = p.windows; w; w >>= 1) p.key_bits++;
  p.entries = n * ((size_t)p.wpk * p.lpts + (size_t)p.wr * p.spts);
  p.buckets = (size_t)p.windows << c;
  return p;
}
struct RlcInputs {
  const uint8_t* u;
  const uint8_t* c;      // k_challenge's output for the group
  const uint8_t* valid;  // ... and its validity bytes
  const uint8_t* pk[2];  // PK, PK' (double)
  const uint8_t* r[2];   // R, R' (double)
  const uint8_t* gen;    // Gen (var-generator)
};
struct RlcBuffers {
  uint32_t* pts;      // (lpts + spts) n x 32 words: the long points, then the negated short ones, as affine niels
  uint32_t* fsc;      // fixed x n x 8 words: z_i u_i mod r (z'_i u_i)
  uint32_t* fpart;    // kRlcFsumBlocks x 8
  uint32_t* fsum;     // 2 x 8
  uint32_t* keys[2];  // entries each (unsorted / sorted)
  uint32_t* vals[2];
  uint32_t* start;    // buckets + 1: first sorted pair of every bucket
  uint32_t* cnt[2];   // buckets each: run lengths (unsorted / sorted, longest first)
  uint32_t* order[2]; // buckets each: bucket numbers (identity / in the order of the sorted lengths)
  uint32_t* buckets;  // buckets x 36 words (extended niels)
  uint32_t* buckets2; // the same for a second range (null: device-pointer calls run one range)
  uint32_t* tmp[2];   // rlc_tmp_points(p, k) x 36 words
  uint32_t* flags;    // 4 words
  void* sort_temp;
  size_t sort_temp_bytes;
};
inline size_t rlc_tmp_points(const RlcPlan& p, int k) {
  const size_t side = (size_t)1 << p.half, w = (size_t)p.windows;
  if (k == 0) {
    size_t a = w * 2 * side * p.nseg, b = w * 2 * p.half * p.nseg2, c = w * p.c + 1;
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
  }
  const size_t a = w * 2 * side, b = w * p.c;
  return a > b ? a : b;
}
size_t rlc_sort_temp_bytes(const RlcPlan& p);
// hash output c / valid of the group in, ok[i] = "item i is well-formed" and flags out; never synchronises;
// returns the first error of a launch or of the sorts
hipError_t launch_rlc(int scheme, const RlcPlan& p, const RlcBuffers& b, const RlcInputs& in, ChaChaKey key,
                const uint32_t* tableG, const uint32_t* tableG2, uint8_t* ok, hipStream_t s);
// the same in pieces: begin (flags), the bucket pass of one range into b.buckets (second = false) or
// b.buckets2, and the rest over the whole group (`merged`: the two bucket arrays are added first)
hipError_t launch_rlc_begin(const RlcBuffers& b, hipStream_t s);
hipError_t launch_rlc_buckets(int scheme, const RlcPlan& range, const RlcBuffers& b, const RlcInputs& in,
                              ChaChaKey key, uint8_t* ok, bool second, hipStream_t s);
hipError_t launch_rlc_finish(const RlcPlan& whole, const RlcBuffers& b, const uint32_t* tableG,
                             const uint32_t* tableG2, bool merged, hipStream_t s);

}  // namespace dsv
