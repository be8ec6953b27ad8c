// rlc.h — geometry and launchers of k_rlc.hip (random-linear-combination fast accept, SURVEY.md
// §8(f)-4), shared with the host side (dsv_rlc.hip).  Kept out of launch.h: that header is part of the
// dominant kernel's translation unit, whose sources are what the recorded roofline evidence is
// valid for (schnorr_amd/build.py: unit_sources_sha256).
#pragma once
#include "launch.h"

namespace dsv {

// One GROUP of at most kRlcMaxGroup items is decided by `groups` independent aggregates over SUB-GROUPS of
// `sub` consecutive items each (the last one may be shorter).  groups == 1 is the steady state of a caller
// whose batches are valid; after a rejected group the host cuts the next ones into several sub-groups
// (dsv_rlc.hip), so that one wrong signature sends only its own sub-group to the per-signature kernels.
// All sub-groups run through ONE set of launches (blockIdx.y = sub-group): the latency-bound tail of an
// aggregate (~250 dependent point operations on 13 workgroups) is paid once, not per sub-group.
//
// Per sub-group, c-bit unsigned windows: wpk windows over the 252-bit scalars z_i c_i of the keys, wr
// windows over the 128-bit z_i of the nonce points (separate buckets: k_rlc.hip says why).  A window's
// 2^c buckets are a 2^half x 2^half matrix; row / column sums run in nseg segments, the per-bit subset
// sums over them in nseg2 segments, so that no serial chain exceeds ~32 additions.
//
// Buckets are filled from a two-pass partition of the digits (k_rlc_part1 / k_rlc_part2): every (window,
// point slot) pair owns one ROW of `row_stride` 16-bit digits; pass 1 scatters a row's entries into
// `1 << coarse_bits` BINS per window by the digit's high bits (workgroup-aggregated reservations), pass 2
// sorts one bin — 2^fine_bits buckets, a few thousand entries — by the low bits through LDS and writes
// every bucket's run (start, count).  The digits are uniform by construction (secret random weights), so a
// bin of `bin_cap` = mean + 8 sigma slots never overflows on inputs a caller can produce; if one does, the
// entries beyond it are dropped and the sub-group is flagged (kRlcOverflow): it takes the per-signature path.
struct RlcPlan {
  uint32_t n;             // items of this bucket pass ...
  uint32_t first, total;  // ... which are items first .. first + n of a sub-group of `total` (groups == 1 only: a host
                          // call may run its bucket pass in two ranges, the first while the second is still on the bus)
  uint32_t groups, sub;   // sub-groups; items of each but the last
  uint32_t items;         // items of the whole group
  int lpts, spts, fixed;  // per item: points with 252-bit scalars, points with the z themselves, fixed-base terms
  int c, half, wpk, wr, windows, nseg, nseg2;
  int fine_bits, coarse_bits;
  uint32_t kmul;        // floor(2^(wpk c) / r): the keys' scalars get a random multiple of r below it added
  uint32_t rows;        // wpk * lpts + wr * spts digit rows per sub-group
  uint32_t row_stride;  // digits per row (sub rounded up to 8: rows start 16-byte aligned)
  uint32_t bins;        // windows << coarse_bits
  uint32_t bin_cap;     // entries a bin can take
  size_t entries;       // n * rows (digit, point) pairs of this pass
  size_t buckets;       // windows << c
};
constexpr size_t kRlcMaxGroup = (size_t)1 << 22;  // point indices (at most 4 per item) are packed into 24 bits
constexpr int kRlcMaxSub = 16;                    // sub-groups per group: at most
// below this an aggregate is slower than the per-signature kernels (its tail does not shrink with the
// batch): with automatic window bits such groups skip it.  The host forms (typed objects, records in host
// memory) use kRlcMinAuto for every scheme; device-resident calls rlc_min_auto(scheme): a double or
// var-generator signature is twice the per-signature work, and their kernels are latency-bound below 2^16
// items (r06, same box, all valid, 12-bit windows: double 2^14 items 1.15 x, 2^15 1.39 x, 2^16 1.40 x;
// var-generator 1.28 / 1.25 / 1.23 x; single 2^16 items 0.96 x, 2^17 1.36 x)
constexpr size_t kRlcMinAuto = (size_t)1 << 17;
inline size_t rlc_min_auto(int scheme) { return scheme == 0 ? kRlcMinAuto : (size_t)1 << 14; }
constexpr int kRlcFsumBlocks = 256;  // (64-thread workgroups riding on k_rlc_sum<0>: 64 of them took longer than the sums themselves)
constexpr int kRlcTile = 8192;  // digits one workgroup of k_rlc_part1 partitions
// flags of one sub-group (4 words): [0] defects, [1] = 1 once the chain of kernels ran to its end.
// A sub-group is ACCEPTED iff [0] == 0 and [1] == 1: the per-signature kernels launched behind the
// aggregate take a pointer to these words as their `gate` (launch.h) and return at once when it says so.
enum : uint32_t { kRlcOffCurve = 1, kRlcTorsion = 2, kRlcSum = 4, kRlcOverflow = 8 };
// the flag block of one group: word 0 = "skip the aggregate" (the sample found a wrong item), words
// 4 + 4 g .. the flags of sub-group g
constexpr int kRlcGroupFlagWords = 4 + 4 * kRlcMaxSub;
constexpr size_t kRlcMaxGroupsPerCall = 64;  // DSV_MAX_BATCH / kRlcMaxGroup
// a group owns TWO flag blocks: [2k] its aggregate(s), [2k + 1] the sub-group aggregates of a second stage that
// runs only where the first one rejected (dsv_rlc.hip: "guarded" calls)
constexpr size_t kRlcFlagBlocks = 2 * kRlcMaxGroupsPerCall;
inline int rlc_default_bits(size_t n) {
  // (r06, all valid, same box: 2^19 items 3.005 ms at 14 bits, 3.068 at 16; 2^17 items 1.318 at 14, 1.341 at 12,
  //  1.497 at 16; 2^16 items 1.070 at 12, 1.081 at 14; 2^15 items 0.983 at 12, 1.456 at 8)
  return n >= ((size_t)1 << 20) ? 16 : n >= ((size_t)1 << 17) ? 14 : n >= ((size_t)1 << 14) ? 12 : 8;
}
// even, and the keys' windows cover 252 or 256 bits exactly (10 would need 260: a ninth scalar word)
inline bool rlc_bits_ok(int c) { return c == 4 || c == 6 || c == 8 || c == 12 || c == 14 || c == 16; }
// the plan of a range [first, first + cnt) of a group planned as `whole` (one sub-group): same bucket
// geometry, fewer pairs
inline RlcPlan rlc_range(const RlcPlan& whole, size_t first, size_t cnt) {
  RlcPlan p = whole;
  p.n = (uint32_t)cnt;
  p.first = (uint32_t)first;
  p.entries = cnt * (size_t)p.rows;
  return p;
}
// scheme: 0 single, 1 double, 2 var-generator (k_rlc.hip: k_rlc_prep says which point gets which scalar);
// `items` in `groups` sub-groups (of ceil(items / groups) items rounded up to `align`; fewer sub-groups if
// that leaves some empty)
inline RlcPlan rlc_plan(int scheme, size_t items, int c, int groups = 1, size_t align = 1) {
  RlcPlan p;
  if (groups < 1) groups = 1;
  if (groups > kRlcMaxSub) groups = kRlcMaxSub;
  size_t sub = (items + groups - 1) / groups;
  sub = (sub + align - 1) / align * align;
  if (sub > items) sub = items;
  if (sub == 0) sub = 1;
  p.groups = items ? (uint32_t)((items + sub - 1) / sub) : 1u;
  p.sub = (uint32_t)sub;
  p.items = (uint32_t)items;
  p.n = p.sub;
  p.first = 0;
  p.total = p.sub;
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
  p.fine_bits = c < 8 ? c : 8;
  p.coarse_bits = c - p.fine_bits;
  p.rows = (uint32_t)(p.wpk * p.lpts + p.wr * p.spts);
  p.row_stride = (uint32_t)((sub + 7) / 8 * 8);
  p.bins = (uint32_t)p.windows << p.coarse_bits;
  {
    // a window holds at most max(lpts, spts) * sub entries, spread uniformly over its bins
    const size_t most = (size_t)(p.lpts > p.spts ? p.lpts : p.spts) * sub;
    size_t cap = most;
    if (p.coarse_bits) {
      // (the TOP window of the keys' scalars is uniform over [0, kmul r / 2^((wpk - 1) c)) only — 0.905 of
      //  the digits when wpk c = 252, 0.962 when it is 256 —: its bins take up to 1.105 x the mean)
      const size_t mean = (most >> p.coarse_bits) * 9 / 8 + 1;
      size_t root = 1;
      while (root * root < mean) root++;
      cap = mean + 8 * root + 64;
      if (cap > most) cap = most;
    }
    p.bin_cap = (uint32_t)((cap + 63) / 64 * 64);
  }
  p.entries = sub * (size_t)p.rows;
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
// Device buffers of one group; every array holds `groups` sub-groups `*_stride` elements apart.
struct RlcBuffers {
  uint32_t* pts;       // (lpts + spts) sub x 32 words: the long points, then the negated short ones, as affine niels
  uint32_t* fsc;       // fixed x sub x 8 words: z_i u_i mod r (z'_i u_i)
  uint32_t* fpart;     // 2 x kRlcFsumBlocks x 8 words per sub-group
  uint32_t* fsum;      // 2 x 8 words per sub-group
  uint16_t* digits;    // rows x row_stride: the window digits (0: the entry enters no bucket)
  uint32_t* counters;  // bins (entries of each bin) + 256 (buckets of each run length) + 256 (cursors of k_rlc_order)
  uint32_t* binned;    // bins x bin_cap: (low digit bits << 24 | point index), bin by bin
  uint32_t* sorted;    // bins x bin_cap: point indices, bucket by bucket inside each bin
  uint32_t* start;     // buckets: first entry of the bucket's run in `sorted`
  uint32_t* cnt;       // buckets: its length
  uint32_t* order;     // buckets: bucket numbers, longest run first
  uint32_t* buckets;   // buckets x 36 words (extended niels)
  uint32_t* buckets2;  // the same for a second range (groups == 1)
  uint32_t* tmp[2];    // rlc_tmp_points(p, k) x 36 words
  uint32_t* flags;     // kRlcGroupFlagWords of this group
  size_t pts_stride, fsc_stride, digits_stride, counters_stride, bin_stride, bucket_stride, tmp_stride[2];  // in elements
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
// hash output c / valid of the group in, ok[i] = "item i is well-formed" and flags out; never synchronises;
// returns the first error of a launch
hipError_t launch_rlc(int scheme, const RlcPlan& p, const RlcBuffers& b, const RlcInputs& in, ChaChaKey key,
                      const uint32_t* tableG, const uint32_t* tableG2, uint8_t* ok, hipStream_t s);
// the same in pieces: begin (flags), the bucket pass of one range into b.buckets (second = false) or
// b.buckets2, and the rest over the whole group (`merged`: the two bucket arrays are added first)
hipError_t launch_rlc_begin(const RlcBuffers& b, hipStream_t s);
hipError_t launch_rlc_buckets(int scheme, const RlcPlan& range, const RlcBuffers& b, const RlcInputs& in,
                              ChaChaKey key, uint8_t* ok, bool second, hipStream_t s);
hipError_t launch_rlc_finish(const RlcPlan& whole, const RlcBuffers& b, const uint32_t* tableG,
                             const uint32_t* tableG2, bool merged, hipStream_t s);
// The sample check, on the device: flags[0] of the group = 1 ("skip the aggregate") when one of the `count`
// items from `first` on is well-formed (valid byte set, u < r, every key coordinate < q) and still has
// verdict 0 in sample_ok — a WRONG signature; malformed ones do not count, they stay out of the sum.
// pk1: PK' / Gen, or null.
void launch_rlc_sample_decide(const uint8_t* sample_ok, const uint8_t* valid, const uint8_t* u, const uint8_t* pk0,
                              const uint8_t* pk1, size_t first, size_t count, uint32_t* flags, hipStream_t s);
// The call's verdict from the flag blocks of its groups (kRlcGroupFlagWords apart): *accepted = every
// sub-group of every group accepted (device-accessible memory, may be null); `history` (may be null; pinned
// host memory the context owns): [0] = 8 and [2] = 128 after a call with a rejected sub-group, one less each (not
// below 0) after a call whose aggregates all accepted, unchanged if the call ran none; [1] += 1.
struct RlcVerdictArgs {
  uint32_t ngroups;
  uint32_t and_into;                     // *accepted &= ... (the second kind of a mixed batch)
  uint8_t subs[kRlcMaxGroupsPerCall];    // sub-groups of each group's DECISIVE stage; 0: the group took the per-signature path as it is
  uint8_t second[kRlcMaxGroupsPerCall];  // 1: the decisive flag block is the group's second one
};
void launch_rlc_verdict(const uint32_t* flags, RlcVerdictArgs a, uint32_t* accepted, uint32_t* history, hipStream_t s);
// Between the two stages of a guarded group: if the first stage's (single) aggregate ACCEPTED, mark the second
// stage's block "skip" and its `subs` sub-groups accepted — its kernels and the per-signature launches gated by
// it then return at once; else leave it zeroed: the second stage runs.  One thread; `second` zeroed beforehand.
void launch_rlc_chain(const uint32_t* first, uint32_t* second, uint32_t subs, hipStream_t s);

}  // namespace dsv
