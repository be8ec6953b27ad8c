// common.h — device helpers shared by the kernel translation units of libdsv (k_hash.hip,
// k_verify.hip, k_quad.hip, k_vargen.hip, k_misc.hip): element loads / stores, the fixed-base table
// lookup and accumulation, the per-lane window tables of a variable base, scalar recodings.
// Field / curve arithmetic lives in fe29.h, jubjub29.h, fr.h, halfgcd.h, decode29.h, stdrng.h.
//
// r03: the compile-time A/B arms that lost in r01 / r02 (non-temporal table traffic, global address
// space loads, stored negations, 3-bit windows, LDS-staged fixed-base tables, prefetched fixed-base
// lookups, per-lane identity entries) are gone from the product source; their measurements stay in
// profiles/r02/ab_*.txt and tools/variants/ says how to get each arm back.
#pragma once
#include <hip/hip_runtime.h>

#include "fe29.h"
#include "fr.h"
#include "jubjub29.h"
#include "launch.h"

namespace dsv {

static_assert(kLimbs == NL, "launch.h and fe29.h disagree on the limb count");

// fixed-base tables: SIGNED windows of kFixedBits bits over a scalar < 2^252.  16 bits: 16 windows
// x 32769 entries x 144 B = 75.5 MB per generator in HBM / Infinity Cache, 16 mixed additions per
// chain (r01 / most of r02: 11 bits, 23 additions, L2-resident; same-box A/B 16 bits +1.5 %,
// profiles/r02/ab_fixed_bits.txt — the lookups are independent of the accumulator, so their
// latency hides behind the additions).
constexpr int kFixedHalf = 1 << (kFixedBits - 1);

DSV_DEV void load_words8(u32 (&w)[8], const uint8_t* base, size_t idx) {
  const uint4* p = reinterpret_cast<const uint4*>(base + idx * 32);
  uint4 a = p[0], b = p[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
  w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
DSV_DEV void store_words8(uint8_t* base, size_t idx, const u32 (&w)[8]) {
  uint4* p = reinterpret_cast<uint4*>(base + idx * 32);
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// canonical LE bytes -> Montgomery fe29; returns false if the encoding is >= q
DSV_DEV bool load_fq(Fe& out, const uint8_t* base, size_t idx) {
  u32 w[8];
  load_words8(w, base, idx);
  bool ok = words_lt(w, kQ32);
  out = fe_to_mont(fe_from_words_plain(w));
  return ok;
}
// the same for -x when `negate` (the u coordinate of a point to be negated): q - x on the canonical
// words (x = 0 gives q, which is 0 again after the conversion)
DSV_DEV bool load_fq_signed(Fe& out, const uint8_t* base, size_t idx, bool negate) {
  u32 w[8];
  load_words8(w, base, idx);
  const bool ok = words_lt(w, kQ32);
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)kQ32[i] - w[i] - borrow;
    borrow = (u32)(t >> 63);
    w[i] = negate ? (u32)t : w[i];
  }
  out = fe_to_mont(fe_from_words_plain(w));
  return ok;
}
DSV_DEV void store_fq(uint8_t* base, size_t idx, const Fe& mont) {
  u32 w[8];
  fe_to_words_plain(w, fe_from_mont(mont));
  store_words8(base, idx, w);
}

// entry for signed digit d of `window`: -P swaps (v+u, v-u) and takes the stored negated 2d*uv —
// the sign costs address arithmetic only
DSV_DEV ANiels load_aniels(const u32* __restrict__ table, int window, int d) {
  const bool neg = d < 0;
  const u32 mag = (u32)(neg ? -d : d);
  const u32* p = table + ((size_t)window * kFixedEntries + mag) * kEntryWords;
  ANiels n;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    n.vpu.l[i] = p[(neg ? NL : 0) + i];
    n.vmu.l[i] = p[(neg ? 0 : NL) + i];
    n.t2d.l[i] = p[(neg ? 3 * NL : 2 * NL) + i];
  }
  return n;
}

// signed recoding of a scalar < 2^252 into kFixedWindows digits: add 2^(bits-1) to every window;
// digit = window value - 2^(bits-1).  The recoded scalar takes 288 bits (windows reach past 255).
DSV_DEV void recode_fixed(u32 (&y)[9], const u32 (&s)[8]) {
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    u32 bias = 0;
#pragma unroll
    for (int k = 0; k < kFixedWindows; k++) {
      const int pos = kFixedBits * k + kFixedBits - 1;  // bit of 2^(bits-1) in window k
      if ((pos >> 5) == i) bias |= 1u << (pos & 31);
    }
    const u64 t = (u64)(i < 8 ? s[i] : 0u) + bias + carry;
    y[i] = (u32)t;
    carry = (u32)(t >> 32);
  }
}
// next digit, LSB first (the order of the additions is irrelevant): shifting the recoded scalar
// down needs no dynamically indexed register
DSV_DEV int next_fixed_digit(u32 (&y)[9]) {
  const int d = (int)(y[0] & ((1u << kFixedBits) - 1)) - kFixedHalf;
#pragma unroll
  for (int i = 0; i < 8; i++) y[i] = __funnelshift_r(y[i], y[i + 1], kFixedBits);
  y[8] >>= kFixedBits;
  return d;
}
// acc += s * Gen from the signed-window table: kFixedWindows mixed additions, no doubling.  The
// running accumulator is passed in so that u*G + c*PK needs no separate final addition (and no
// second live point).
DSV_DEV Ext fixed_base_accumulate(Ext acc, const u32 (&s)[8], const u32* __restrict__ table) {
  u32 y[9];
  recode_fixed(y, s);
#pragma unroll 1
  for (int w = 0; w < kFixedWindows; w++) {
    const int d = next_fixed_digit(y);
    ANiels e = load_aniels(table, w, d);
    acc = ext_add_aniels(acc, e);
  }
  return acc;
}

// [ acc + s * Gen == O ]: the same, the last of the kFixedWindows additions only as far as the
// identity test needs it (ext_add_aniels_is_identity)
DSV_DEV bool fixed_base_accumulate_is_identity(Ext acc, const u32 (&s)[8], const u32* __restrict__ table) {
  u32 y[9];
  recode_fixed(y, s);
#pragma unroll 1
  for (int w = 0; w < kFixedWindows - 1; w++) {
    const int d = next_fixed_digit(y);
    ANiels e = load_aniels(table, w, d);
    acc = ext_add_aniels(acc, e);
  }
  return ext_add_aniels_is_identity(acc, load_aniels(table, kFixedWindows - 1, next_fixed_digit(y)));
}

// ---- per-lane window table of a variable base, in global memory, LANE-MAJOR ---------------
// Signed 4-bit digits d in [-8, 8): entries |d| * P for |d| = 1..8, each stored as extended niels
// (v+u, v-u, z, 2d*t), 4 x 9 words = 144 B; 1296 B per lane, contiguous (slot 0 is never written:
// digit 0 reads ONE shared identity entry).  A lookup is 4 x 36 contiguous bytes of ONE entry (-P
// swaps v+u / v-u by address and negates 2d*t in registers after the load).  The slot belongs to
// (workgroup, lane), so the verify kernels run a fixed grid with a grid-stride loop.
DSV_DEV void store_fe_words(u32* p, const Fe& a) {
#pragma unroll
  for (int i = 0; i < NL; i++) p[i] = a.l[i];
}
DSV_DEV Fe load_fe_words(const u32* p) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = p[i];
  return r;
}
DSV_DEV void store_var_entry(u32* lane_tbl, int e, const Niels& n) {
  u32* p = lane_tbl + e * kVarEntryWords;
  store_fe_words(p, n.vpu);
  store_fe_words(p + NL, n.vmu);
  store_fe_words(p + 2 * NL, n.z);
  store_fe_words(p + 3 * NL, n.t2d);
}
__device__ const u32 kIdentityEntry[4 * NL] = {
    // v+u = 1, v-u = 1, z = 1 (Montgomery form), 2d*t = 0
    DSV_ONE_LIST, DSV_ONE_LIST, DSV_ONE_LIST, 0, 0, 0, 0, 0, 0, 0, 0, 0};
DSV_DEV Niels load_var_entry(const u32* lane_tbl, int d) {
  const bool neg = d < 0;
  const int mag = neg ? -d : d;
  const u32* p = mag == 0 ? kIdentityEntry : lane_tbl + mag * kVarEntryWords;
  Niels n;
  n.vpu = load_fe_words(p + (neg ? NL : 0));
  n.vmu = load_fe_words(p + (neg ? 0 : NL));
  n.z = load_fe_words(p + 2 * NL);
  const Fe t = load_fe_words(p + 3 * NL);
  n.t2d = fe_select(neg, fe_neg2(t), t);
  return n;
}
// Software-pipelined form: the loads of the entry for the NEXT window are issued one group
// operation ahead and stay in flight while the chain works; the sign fix-up waits until the entry
// is consumed, so nothing forces an s_waitcnt right behind the loads (+0.65 % single, +1.2 %
// var-generator, profiles/r02/ab_table_prefetch.txt).
struct RawNiels {
  Fe a, b, z, t;  // vpu / vmu already swapped by address for a negative digit; t = 2d*t of +entry
  bool neg;
};
DSV_DEV RawNiels load_var_entry_raw(const u32* lane_tbl, int d) {
  RawNiels r;
  r.neg = d < 0;
  const int mag = r.neg ? -d : d;
  const u32* p = mag == 0 ? kIdentityEntry : lane_tbl + mag * kVarEntryWords;
  r.a = load_fe_words(p + (r.neg ? NL : 0));
  r.b = load_fe_words(p + (r.neg ? 0 : NL));
  r.z = load_fe_words(p + 2 * NL);
  r.t = load_fe_words(p + 3 * NL);
  return r;
}
DSV_DEV Niels finish_var_entry(const RawNiels& r) {
  Niels n;
  n.vpu = r.a;
  n.vmu = r.b;
  n.z = r.z;
  n.t2d = fe_select(r.neg, fe_neg2(r.t), r.t);
  return n;
}
// entries 1..8 of P (affine): every step is a mixed addition, and u*v of the current multiple is
// computed once for its own entry AND for the addition that produces the next one
DSV_DEV void build_var_table(u32* lane_tbl, const Fe& pu, const Fe& pv) {
  Ext p = ext_from_affine(pu, pv);
  Fe tt = fe_mul(p.t1, p.t2);
  Niels n1 = ext_to_niels_t(p, tt);
  store_var_entry(lane_tbl, 1, n1);
  const ANiels a1 = {n1.vpu, n1.vmu, n1.t2d};
  Ext cur = p;
#pragma unroll 1
  for (int i = 2; i < kVarEntries; i++) {
    cur = ext_add_aniels_t(cur, tt, a1);
    tt = fe_mul(cur.t1, cur.t2);
    store_var_entry(lane_tbl, i, ext_to_niels_t(cur, tt));
  }
}
// signed recoding: y = s + 0x8888..8; digit k of s is nibble k of y minus 8, in [-8, 7].
// Exact for s < 2^252 (nibble 63 of y is then 8 or 9, i.e. digit 63 is 0 or 1).
DSV_DEV void recode_signed4(u32 (&y)[8], const u32 (&s)[8]) {
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 t = (u64)s[i] + 0x88888888u + carry;
    y[i] = (u32)t;
    carry = (u32)(t >> 32);
  }
}
DSV_DEV int sdigit4(const u32 (&y)[8], int k) {
  return (int)((y[k >> 3] >> (4 * (k & 7))) & 0xf) - 8;
}
// index of the highest non-zero signed digit among recoded scalars OR-ed into nz (a zero digit is
// nibble 8, so the caller passes y ^ 0x8888..8)
DSV_DEV int top_digit4(const u32 (&nz)[8]) {
  int len = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (nz[i] != 0) len = 32 * i + (32 - __clz(nz[i]));
  return len > 0 ? (len - 1) >> 2 : 0;
}

// ---- JOINT window table of two variable bases (P, R): signed 2-bit digits (da, db) in [-1, 2]^2,
// one table entry da*P + db*R per window, ONE addition per two doublings.  Same additions per bit
// as two 4-bit tables (one per 2 bits against two per 4), but 11 stored entries instead of 16 and a
// table built with 61 multiplications + 12 squarings instead of 116 multiplications; the chain's
// length follows max(bitlen a, bitlen b) in steps of 2 bits instead of 4.
//   slot: 1 P | 2 R | 3 2P | 4 2R | 5 P+R | 6 P-R | 7 2P+R | 8 2P-R | 9 P+2R | 10 2R-P | 11 2P+2R
// Digit pairs map to +-slot (a negative pair reads the entry with v+u / v-u swapped and 2d*t
// negated, as in the one-base tables); (0, 0) reads the shared identity entry.  Both scalars are
// non-negative: a term that enters the equation with a minus sign has its point negated instead
// (load_fq_signed: q - u in the canonical domain, so every bound downstream is unchanged).
constexpr int kJointSlots = 12;  // slot 0 is never written
constexpr int kJointLaneWords = kJointSlots * kVarEntryWords;
static_assert(kJointLaneWords <= 2 * kVarLaneWords, "the joint table lives in the two one-base slots");
// signed slot of the pair (da, db) given as raw digits ra = da + 1, rb = db + 1 (digits in [-1, 2]:
// with this digit set a scalar of odd bit length never needs a window beyond its own length and
// one of even length only when its top field is 3 or a carry arrives — 65.7 windows per wave on
// average where the set [-2, 1] needs 66.25, tests/test_joint_windows.py)
DSV_DEV int joint_slot(u32 ra, u32 rb) {
  const u32 idx = ra * 4 + rb;
  // idx:            0    1    2    3    4   5   6   7    8   9  10  11   12  13  14  15
  // (da, db):    -1-1  -10  -11  -12  0-1  00  01  02  1-1  10  11  12  2-1  20  21  22
  // slot:           5    1    6   10    2   0   2   4    6   1   5   9    8   3   7  11
  const unsigned long long slots = 0xB73895164202A615ULL;
  const int slot = (int)((slots >> (4 * idx)) & 15u);
  const bool neg = ((0x17u >> idx) & 1u) != 0;
  return neg ? -slot : slot;
}
// y = 0x5555..5 + s: digit k of s is the 2-bit field k of y minus 1, in [-1, 2].  Exact for
// s < 2^255.  (A negative term is handled by negating its POINT: build_joint_table's caller.)
DSV_DEV void recode_signed2(u32 (&y)[8], const u32 (&s)[8]) {
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 t = (u64)s[i] + 0x55555555u + carry;
    y[i] = (u32)t;
    carry = (u32)(t >> 32);
  }
}
DSV_DEV int joint_digit(const u32 (&ya)[8], const u32 (&yb)[8], int k) {
  const int sh = 2 * (k & 15);
  return joint_slot((ya[k >> 4] >> sh) & 3u, (yb[k >> 4] >> sh) & 3u);
}
// index of the highest window with a non-zero digit in either scalar (caller: (ya | yb) ^ 0x55..5)
DSV_DEV int top_digit2(const u32 (&nz)[8]) {
  int len = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (nz[i] != 0) len = 32 * i + (32 - __clz(nz[i]));
  return len > 0 ? (len - 1) >> 1 : 0;
}
// P = (pu, pv), R = (ru, rv) affine.  Every sum / difference pair is one shared mixed addition onto
// an extended point whose t1*t2 is at hand (ext_add_sub_aniels_t: 10 M per pair).
DSV_DEV void build_joint_table(u32* tbl, const Fe& pu, const Fe& pv, const Fe& ru, const Fe& rv) {
  const Ext P = ext_from_affine(pu, pv), R = ext_from_affine(ru, rv);
  const Fe ttP = fe_mul(pu, pv), ttR = fe_mul(ru, rv);
  const Niels nP = ext_to_niels_t(P, ttP), nR = ext_to_niels_t(R, ttR);
  store_var_entry(tbl, 1, nP);
  store_var_entry(tbl, 2, nR);
  const ANiels aR = {nR.vpu, nR.vmu, nR.t2d};
  auto put = [&](int slot, const Ext& x) { store_var_entry(tbl, slot, ext_to_niels(x)); };
  Ext sum, diff;
  {
    ext_add_sub_aniels_t(sum, diff, P, ttP, aR);  // P + R, P - R
    put(5, sum);
    put(6, diff);
    put(11, ext_double(sum));
  }
  {
    const Ext P2 = ext_double_affine(pu, pv);
    const Fe tt = fe_mul(P2.t1, P2.t2);
    store_var_entry(tbl, 3, ext_to_niels_t(P2, tt));
    ext_add_sub_aniels_t(sum, diff, P2, tt, aR);  // 2P + R, 2P - R
    put(7, sum);
    put(8, diff);
  }
  {
    const Ext R2 = ext_double_affine(ru, rv);
    const Fe tt = fe_mul(R2.t1, R2.t2);
    store_var_entry(tbl, 4, ext_to_niels_t(R2, tt));
    // P's niels form comes back from its table slot: three fields fewer to keep live — or to spill —
    // across the two blocks above (57 instead of 78 spilled VGPRs in the single-equation kernel:
    // +0.9 %; reloading R's form for the second block as well and reordering the blocks leaves 25
    // but waits on the stores in flight: slower — profiles/r03/ab_joint_windows.txt)
    const u32* e1 = tbl + 1 * kVarEntryWords;
    const ANiels aP = {load_fe_words(e1), load_fe_words(e1 + NL), load_fe_words(e1 + 3 * NL)};
    ext_add_sub_aniels_t(sum, diff, R2, tt, aP);  // 2R + P, 2R - P
    put(9, sum);
    put(10, diff);
  }
}
// acc = 4 * acc
DSV_DEV Ext ext_mul4(const Ext& p) {
  Ext q = p;
  ext_double_uvz(q.u, q.v, q.z);
  return ext_double(q);
}

// acc = 16 * acc: three doublings that skip the (t1, t2) outputs nobody reads, then a full one
DSV_DEV Ext ext_mul16(const Ext& p) {
  Fe u = p.u, v = p.v, z = p.z;
#pragma unroll 1
  for (int j = 0; j < 3; j++) ext_double_uvz(u, v, z);
  Ext q;
  q.u = u;
  q.v = v;
  q.z = z;
  return ext_double(q);
}

// s * P, signed 4-bit fixed windows, MSB first: acc = 16*acc + T[digit].  TOP = index of the
// highest possibly non-zero digit (63 for a 252-bit Fr scalar); the first window is a plain
// addition onto the identity (no doublings of the identity).
template <int TOP>
DSV_DEV Ext var_base_mul(const u32 (&s)[8], const u32* lane_tbl) {
  u32 y[8];
  recode_signed4(y, s);
  Ext acc = ext_from_niels(load_var_entry(lane_tbl, sdigit4(y, TOP)));
#pragma unroll 1
  for (int k = TOP - 1; k >= 0; k--) {
    acc = ext_mul16(acc);
    acc = ext_add_niels(acc, load_var_entry(lane_tbl, sdigit4(y, k)));
  }
  return acc;
}

// marks an output element as invalid: 0xff..ff is >= q and >= r, every consumer rejects it
DSV_DEV void store_poison(uint8_t* base, size_t idx) {
  const u32 w[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
  store_words8(base, idx, w);
}

}  // namespace dsv
