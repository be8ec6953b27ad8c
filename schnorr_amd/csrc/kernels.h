// kernels.h — the device side of libdsv: every __global__ kernel of the engine and the device
// helpers they share (scalar multiplications, window tables, recodings, the mixed-batch split).
// Included once, by dsv.hip, which holds the host side (contexts, C ABI).  Field / curve / hash
// arithmetic lives in fe29.h, jubjub29.h, hades29.h, halfgcd.h, decode29.h, fr.h, stdrng.h.
#pragma once
#include <hip/hip_runtime.h>

#include "fe29.h"
#include "fr.h"
#include "hades29.h"
#include "halfgcd.h"
#include "decode29.h"
#include "stdrng.h"
#include "jubjub29.h"
#include "quad29.h"

namespace dsv {

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
#ifndef DSV_WAVES_VERIFY
#define DSV_WAVES_VERIFY 2
#endif
#ifndef DSV_WAVES_HASH
#define DSV_WAVES_HASH 2
#endif
// fixed-base tables: SIGNED windows of DSV_FIXED_BITS bits over a scalar < 2^252.  16 bits: 16
// windows x 32769 entries x 144 B = 75.5 MB per generator in HBM / Infinity Cache, 16 mixed
// additions per chain.  r01 / most of r02 ran the L2-resident 11-bit table (23 additions, 3.4 MB):
// same-box A/B 13 bits +0.8 %, 16 bits +1.5 % (profiles/r02/ab_fixed_bits.txt) — the lookups are
// independent of the accumulator, so their latency hides behind the additions.
#ifndef DSV_FIXED_BITS
#define DSV_FIXED_BITS 16
#endif
constexpr int kFixedBits = DSV_FIXED_BITS;
constexpr int kFixedWindows = (253 + kFixedBits - 1) / kFixedBits;  // +1 bit: recoding carry
constexpr int kFixedHalf = 1 << (kFixedBits - 1);
constexpr int kFixedEntries = kFixedHalf + 1;                       // |digit| = 0 .. 2^(bits-1)
constexpr int kEntryWords = 4 * NL;  // v+u, v-u, 2d*uv, -(2d*uv): 144 B, 16-byte aligned
constexpr size_t kTableBytes = (size_t)kFixedWindows * kFixedEntries * kEntryWords * 4;
static_assert(kFixedWindows * kFixedBits <= 288 && kFixedBits >= 4 && kFixedBits <= 16, "window");

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

// generic double-and-add over a 256-bit LE scalar (init-time table construction only)
DSV_DEV Ext ext_mul_words(const Ext& p, const u32 (&s)[8]) {
  Niels n = ext_to_niels(p);
  Niels id = niels_identity();
  Ext acc = ext_identity();
#pragma unroll 1
  for (int bit = 255; bit >= 0; bit--) {
    acc = ext_double(acc);
    bool b = (s[bit >> 5] >> (bit & 31)) & 1;
    Niels sel;
    sel.vpu = fe_select(b, n.vpu, id.vpu);
    sel.vmu = fe_select(b, n.vmu, id.vmu);
    sel.z = fe_select(b, n.z, id.z);
    sel.t2d = fe_select(b, n.t2d, id.t2d);
    acc = ext_add_niels(acc, sel);
  }
  return acc;
}

// ------------------------------------------------------------------------------------------
// init: fixed-base tables.  table[w][d] = affine niels of (d * 2^(kFixedBits*w)) * Gen, d = 0 ..
// 2^(kFixedBits-1), canonical limbs (+ the negated 2d*uv).
// ------------------------------------------------------------------------------------------
// bits / entry_words: the shipped table (kFixedBits, 4 fields) or the LDS-staged A/B variant
// (DSV_FIXED_LDS_BITS, 3 fields: no stored negation)
__global__ void __launch_bounds__(64) k_build_fixed_table(u32* __restrict__ table, int which, int bits,
                                                          int entry_words) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int windows = (253 + bits - 1) / bits, entries = (1 << (bits - 1)) + 1;
  if (idx >= windows * entries) return;
  const int w = idx / entries, d = idx % entries;
  const u32 gu[NL] = DSV_GEN_U, gv[NL] = DSV_GEN_V, nu[NL] = DSV_GENN_U, nv[NL] = DSV_GENN_V;
  Ext g = which == 0 ? ext_from_affine(fe_const(gu), fe_const(gv))
                     : ext_from_affine(fe_const(nu), fe_const(nv));
  // scalar = d << (bits * w), d <= 2^(bits-1) <= 2^15: at most two words
  u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // (entries whose scalar would not fit 256 bits are never looked up: a digit there is 0 or 1
  //  and 1 << pos < 2^253)
  const int pos = bits * w, wi = pos >> 5, sh = pos & 31;
  const u64 v = (u64)(u32)d << sh;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (i == wi) s[i] = (u32)v;
    if (i == wi + 1) s[i] = (u32)(v >> 32);
  }
  Ext p = ext_mul_words(g, s);
  Fe zi = fe_invert(p.z);
  Fe u = fe_mul(p.u, zi), v2 = fe_mul(p.v, zi);
  Fe vpu = fe_canon(fe_add(v2, u));
  Fe vmu = fe_canon(fe_sub2(v2, u));
  Fe t2d = fe_mul(fe_mul(u, v2), fe_const(kD2));
  Fe nt2d = fe_canon(fe_neg2(t2d));
  t2d = fe_canon(t2d);
  u32* e = table + (size_t)idx * entry_words;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    e[i] = vpu.l[i];
    e[NL + i] = vmu.l[i];
    e[2 * NL + i] = t2d.l[i];
    if (entry_words > 3 * NL) e[3 * NL + i] = nt2d.l[i];
  }
}

// ------------------------------------------------------------------------------------------
// challenge hash
// ------------------------------------------------------------------------------------------
template <bool DOUBLE>
__global__ void __launch_bounds__(256, DSV_WAVES_HASH)
k_challenge(const uint8_t* __restrict__ R_uv, const uint8_t* __restrict__ Rp_uv,
            const uint8_t* __restrict__ m, size_t n, uint8_t* __restrict__ c_out,
            uint8_t* __restrict__ valid) {
#if DSV_HADES_MFMA
  // the hashes of a wave cooperate through the matrix cores: every lane runs, spare lanes redo
  // the last item and skip the stores
  hades_mfma_load_table();
  const size_t i_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i_raw < n;
  const size_t i = live ? i_raw : n - 1;
#else
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr bool live = true;
#endif
  Fe ru, rv, mm;
  bool ok = load_fq(ru, R_uv, 2 * i);
  ok &= load_fq(rv, R_uv, 2 * i + 1);
  ok &= load_fq(mm, m, i);
  Fe h;
  if (DOUBLE) {
    Fe pu, pv;
    ok &= load_fq(pu, Rp_uv, 2 * i);
    ok &= load_fq(pv, Rp_uv, 2 * i + 1);
    h = poseidon_hash5(ru, rv, pu, pv, mm);
  } else {
    h = poseidon_hash3(ru, rv, mm);
  }
  u32 c[8];
  poseidon_truncate(c, h);
  if (!live) return;
  store_words8(c_out, i, c);
  if (valid) valid[i] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// scalar multiplications
// ------------------------------------------------------------------------------------------
// acc += u * Gen from the signed kFixedBits-bit-window table: kFixedWindows (16 for 16 bits) mixed
// additions, no doubling.  The running accumulator is passed in so that u*G + c*PK needs no
// separate final addition (and no second live point).
// -DDSV_FIXED_PREFETCH=1: the same pipelining for the fixed-base lookups (L2 hits, 23 per chain):
// measured equal (profiles/r02/ab_table_prefetch.txt), so the plain loop ships.
#ifndef DSV_FIXED_PREFETCH
#define DSV_FIXED_PREFETCH 0
#endif
DSV_DEV Ext fixed_base_accumulate(Ext acc, const u32 (&s)[8], const u32* __restrict__ table) {
  // signed recoding: add 2^(bits-1) to every window; digit = window value - 2^(bits-1).
  // s < 2^252, so the top window cannot overflow.  Windows are consumed LSB first (the order of
  // the additions is irrelevant) by shifting the recoded scalar down, which needs no
  // dynamically indexed register.
  u32 y[9];  // 288 bits: room for the windows that reach past bit 255
  {
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
#if DSV_FIXED_PREFETCH
  // Software pipelining: the entry of window w+1 is loaded before the addition of window w.  Two
  // windows per round with the two entries in their own registers (a rolled one-window loop would
  // have to copy `next` into `current`, and the copy waits for the load).
  auto next_digit = [&y]() {
    const int d = (int)(y[0] & ((1u << kFixedBits) - 1)) - kFixedHalf;
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = __funnelshift_r(y[i], y[i + 1], kFixedBits);
    y[8] >>= kFixedBits;
    return d;
  };
  ANiels e0 = load_aniels(table, 0, next_digit());
#pragma unroll 1
  for (int w = 0; w + 1 < kFixedWindows; w += 2) {
    const ANiels e1 = load_aniels(table, w + 1, next_digit());
    acc = ext_add_aniels(acc, e0);
    const bool more = w + 2 < kFixedWindows;  // even window count: the last load is a dummy
    const int d2 = next_digit();
    e0 = load_aniels(table, more ? w + 2 : w + 1, more ? d2 : 0);
    acc = ext_add_aniels(acc, e1);
  }
  if (kFixedWindows & 1) acc = ext_add_aniels(acc, e0);
#else
#pragma unroll 1
  for (int w = 0; w < kFixedWindows; w++) {
    const int d = (int)(y[0] & ((1u << kFixedBits) - 1)) - kFixedHalf;
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = __funnelshift_r(y[i], y[i + 1], kFixedBits);
    y[8] >>= kFixedBits;
    ANiels e = load_aniels(table, w, d);
    acc = ext_add_aniels(acc, e);
  }
#endif
  return acc;
}

// ---- A/B: the fixed-base table staged in LDS (north_star: "LDS-staged windowed fixed-base tables")
// -DDSV_FIXED_LDS_BITS=5: signed 5-bit windows, 51 windows x 17 entries x 108 B = 93.6 KB, copied
// into LDS once per (persistent, 512-thread) workgroup; 6 bits: 43 x 33 x 108 B = 153 KB.  The 160 KB
// of LDS cannot hold the 11-bit table (3.4 MB), so staging costs 51 (43) mixed additions instead
// of 23.  Single-equation kernel only (two generators do not fit).  Measurement: DESIGN.md §6.
#ifndef DSV_FIXED_LDS_BITS
#define DSV_FIXED_LDS_BITS 0
#endif
constexpr int kLdsBits = DSV_FIXED_LDS_BITS ? DSV_FIXED_LDS_BITS : 5;
constexpr int kLdsWindows = (253 + kLdsBits - 1) / kLdsBits;
constexpr int kLdsHalf = 1 << (kLdsBits - 1);
constexpr int kLdsEntries = kLdsHalf + 1;
constexpr int kLdsEntryWords = 3 * NL;
constexpr int kLdsTableWords = kLdsWindows * kLdsEntries * kLdsEntryWords;
constexpr int kLdsBlock = 512;  // 8 waves share one copy of the table

DSV_DEV Ext fixed_base_accumulate_lds(Ext acc, const u32 (&s)[8], const u32* lds_table) {
  u32 y[9];
  {
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      u32 bias = 0;
#pragma unroll
      for (int k = 0; k < kLdsWindows; k++) {
        const int pos = kLdsBits * k + kLdsBits - 1;
        if ((pos >> 5) == i) bias |= 1u << (pos & 31);
      }
      const u64 t = (u64)(i < 8 ? s[i] : 0u) + bias + carry;
      y[i] = (u32)t;
      carry = (u32)(t >> 32);
    }
  }
#pragma unroll 1
  for (int w = 0; w < kLdsWindows; w++) {
    const int d = (int)(y[0] & ((1u << kLdsBits) - 1)) - kLdsHalf;
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = __funnelshift_r(y[i], y[i + 1], kLdsBits);
    y[8] >>= kLdsBits;
    const bool neg = d < 0;
    const u32* p = lds_table + (w * kLdsEntries + (neg ? -d : d)) * kLdsEntryWords;
    ANiels e;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      e.vpu.l[i] = p[(neg ? NL : 0) + i];
      e.vmu.l[i] = p[(neg ? 0 : NL) + i];
      e.t2d.l[i] = p[2 * NL + i];
    }
    e.t2d = fe_select(neg, fe_neg2(e.t2d), e.t2d);
    acc = ext_add_aniels(acc, e);
  }
  return acc;
}

// ---- per-lane window table of a variable base, in global memory, LANE-MAJOR ---------------
// Signed 4-bit digits d in [-8, 8): entries |d| * P for |d| = 0..8, each stored as extended niels
// (v+u, v-u, z, 2d*t), 4 x 9 words = 144 B; 1296 B per lane, contiguous.
// A lookup is therefore 4 x 36 contiguous bytes of ONE entry (-P swaps v+u / v-u by address and
// negates 2d*t in registers), instead of 36 dwords scattered over 36
// different 256-B rows as a compiler-scratch array would give (r01 v1: 64.7 GB FETCH_SIZE per
// 2^20 batch, profiles/r01/v1_pmc_summary.json).  The slot belongs to (workgroup, lane), so the
// verify kernels run a fixed grid with a grid-stride loop.
// Entries hold four fields (144 B); a negative digit swaps v+u / v-u by address and negates 2d*t
// after the load.  -DDSV_VAR_NEG_T2D=1 is the r01 layout, which also STORED the negated 2d*t
// (180 B): r02 same-box A/B, 2^20 signatures: 65.6 M/s with the stored negation, 66.8 M/s without
// (profiles/r02/ab_table_traffic.txt) — 20 % less table-write traffic buys more than the 45
// cheap instructions per lookup cost, on a kernel that runs at its power limit.
#ifndef DSV_VAR_NEG_T2D
#define DSV_VAR_NEG_T2D 0
#endif
constexpr int kVarEntries = 9;
constexpr int kVarEntryWords = (DSV_VAR_NEG_T2D ? 5 : 4) * NL;
constexpr int kVarLaneWords = kVarEntries * kVarEntryWords;  // 324 words = 1296 B (405 / 1620 with the stored negation)
constexpr int kVerifyBlock = 64;        // ONE wave per workgroup: a finished wave's slot is refilled at
                                        // once instead of waiting for its three workgroup mates
#ifndef DSV_MAX_VERIFY_GRID
#define DSV_MAX_VERIFY_GRID 4096                              // 16 single-wave workgroups per CU
#endif
constexpr unsigned kMaxVerifyGrid = DSV_MAX_VERIFY_GRID;

// -DDSV_TABLE_NT=1 / 2: non-temporal stores (1) or stores and loads (2) for the per-lane window
// tables, so that this write-once / read-few stream does not displace the L2-resident fixed-base
// tables (A/B knob; see DESIGN.md §6 for the measurement)
#ifndef DSV_TABLE_NT
#define DSV_TABLE_NT 0
#endif
DSV_DEV void store_fe_words(u32* p, const Fe& a) {
#pragma unroll
  for (int i = 0; i < NL; i++) {
    if (DSV_TABLE_NT >= 1) __builtin_nontemporal_store(a.l[i], p + i); else p[i] = a.l[i];
  }
}
// The table pointer is the select of a per-lane workspace slot and the shared identity entry, which
// the compiler only knows as a generic pointer (flat_load).  -DDSV_TABLE_GLOBAL_AS=1 casts it to the
// global address space (global_load): measured equal to slightly slower
// (profiles/r02/ab_global_as_loads.txt), so the flat form stays.
#ifndef DSV_TABLE_GLOBAL_AS
#define DSV_TABLE_GLOBAL_AS 0
#endif
typedef const __attribute__((address_space(1))) u32* GlobalWords;
DSV_DEV Fe load_fe_words(const u32* p) {
  Fe r;
#if DSV_TABLE_GLOBAL_AS
  GlobalWords g = (GlobalWords)p;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = DSV_TABLE_NT >= 2 ? __builtin_nontemporal_load(g + i) : g[i];
#else
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = DSV_TABLE_NT >= 2 ? __builtin_nontemporal_load(p + i) : p[i];
#endif
  return r;
}
DSV_DEV void store_var_entry(u32* lane_tbl, int e, const Niels& n) {
  u32* p = lane_tbl + e * kVarEntryWords;
  store_fe_words(p, n.vpu);
  store_fe_words(p + NL, n.vmu);
  store_fe_words(p + 2 * NL, n.z);
  store_fe_words(p + 3 * NL, n.t2d);
  if (DSV_VAR_NEG_T2D) store_fe_words(p + 4 * NL, fe_neg2(n.t2d));
}
// Entry 0 (the identity) is the same for every lane: it is read from ONE shared copy instead of
// being written into every lane's table (-DDSV_SHARED_IDENTITY=0 restores the per-lane copy;
// one ninth of the table writes, A/B in DESIGN.md §3).
#ifndef DSV_SHARED_IDENTITY
#define DSV_SHARED_IDENTITY 1
#endif
__device__ const u32 kIdentityEntry[5 * NL] = {
    // v+u = 1, v-u = 1, z = 1 (Montgomery form), 2d*t = 0, -(2d*t) = 0
#define DSV_ONE_LIMBS_ DSV_ONE_LIST
    DSV_ONE_LIMBS_, DSV_ONE_LIMBS_, DSV_ONE_LIMBS_, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0
#undef DSV_ONE_LIMBS_
};
DSV_DEV Niels load_var_entry(const u32* lane_tbl, int d) {
  const bool neg = d < 0;
  const int mag = neg ? -d : d;
  const u32* p = (DSV_SHARED_IDENTITY && mag == 0) ? kIdentityEntry : lane_tbl + mag * kVarEntryWords;
  Niels n;
  n.vpu = load_fe_words(p + (neg ? NL : 0));
  n.vmu = load_fe_words(p + (neg ? 0 : NL));
  n.z = load_fe_words(p + 2 * NL);
  if (DSV_VAR_NEG_T2D) {
    n.t2d = load_fe_words(p + (neg ? 4 * NL : 3 * NL));
  } else {
    const Fe t = load_fe_words(p + 3 * NL);
    n.t2d = fe_select(neg, fe_neg2(t), t);
  }
  return n;
}
// Software-pipelined form of load_var_entry (-DDSV_VAR_PREFETCH, A/B in DESIGN.md §3): the loads
// of the entry for the NEXT window are issued one group operation ahead and stay in flight while
// the chain works; the sign fix-up waits until the entry is consumed, so nothing forces an
// s_waitcnt right behind the loads.
#ifndef DSV_VAR_PREFETCH
#define DSV_VAR_PREFETCH 1
#endif
struct RawNiels {
  Fe a, b, z, t;  // vpu / vmu already swapped by address for a negative digit; t = 2d*t of +entry
  bool neg;
};
DSV_DEV RawNiels load_var_entry_raw(const u32* lane_tbl, int d) {
  RawNiels r;
  r.neg = d < 0;
  const int mag = r.neg ? -d : d;
  const u32* p = (DSV_SHARED_IDENTITY && mag == 0) ? kIdentityEntry : lane_tbl + mag * kVarEntryWords;
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
DSV_DEV void build_var_table(u32* lane_tbl, const Fe& pu, const Fe& pv) {
  Ext p = ext_from_affine(pu, pv);
  Fe tt = fe_mul(p.t1, p.t2);  // u*v of the current multiple: used by its entry AND by the next addition
  Niels n1 = ext_to_niels_t(p, tt);
  if (!DSV_SHARED_IDENTITY) store_var_entry(lane_tbl, 0, niels_identity());
  store_var_entry(lane_tbl, 1, n1);
  const ANiels a1 = {n1.vpu, n1.vmu, n1.t2d};  // P is affine: every step is a mixed addition
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

// ---- A/B: window width of the half-scalar chain (-DDSV_HALF_WINDOW_BITS=3; shipped: 4) ----------
// W-bit signed windows: entries |d| = 0 .. 2^(W-1), digits in [-2^(W-1), 2^(W-1)).  Only
// k_verify_fixed_half uses these; every other kernel keeps the 4-bit forms above.
#ifndef DSV_HALF_WINDOW_BITS
#define DSV_HALF_WINDOW_BITS 4
#endif
constexpr int kHalfW = DSV_HALF_WINDOW_BITS;
constexpr int kHalfEntries = (1 << (kHalfW - 1)) + 1;
constexpr int kHalfLaneWords = kHalfEntries * kVarEntryWords;
static_assert(kHalfW == 3 || kHalfW == 4, "window width of the half-scalar chain");
template <int W>
DSV_DEV u32 recode_bias_word(int i) {  // bits W*k + W-1 that fall into word i
  u32 b = 0;
#pragma unroll
  for (int k = 0; k * W + W - 1 < 256; k++)
    if (((k * W + W - 1) >> 5) == i) b |= 1u << ((k * W + W - 1) & 31);
  return b;
}
template <int W>
DSV_DEV void recode_signed_w(u32 (&y)[8], const u32 (&s)[8]) {
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 t = (u64)s[i] + recode_bias_word<W>(i) + carry;
    y[i] = (u32)t;
    carry = (u32)(t >> 32);
  }
}
template <int W>
DSV_DEV int sdigit_w(const u32 (&y)[8], int k) {
  if (W == 4) return sdigit4(y, k);
  const int pos = W * k, wi = pos >> 5, off = pos & 31;
  u32 lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {  // dynamic word pick without a dynamically indexed register
    lo = i == wi ? y[i] : lo;
    hi = i == wi + 1 ? y[i] : hi;
  }
  const u64 two = ((u64)hi << 32) | lo;
  return (int)((two >> off) & ((1u << W) - 1)) - (1 << (W - 1));
}
template <int N>
DSV_DEV void build_var_table_n(u32* lane_tbl, const Fe& pu, const Fe& pv) {
  Ext p = ext_from_affine(pu, pv);
  Fe tt = fe_mul(p.t1, p.t2);  // u*v of the current multiple: used by its entry AND by the next addition
  Niels n1 = ext_to_niels_t(p, tt);
  if (!DSV_SHARED_IDENTITY) store_var_entry(lane_tbl, 0, niels_identity());
  store_var_entry(lane_tbl, 1, n1);
  const ANiels a1 = {n1.vpu, n1.vmu, n1.t2d};  // P is affine: every step is a mixed addition
  Ext cur = p;
#pragma unroll 1
  for (int i = 2; i < N; i++) {
    cur = ext_add_aniels_t(cur, tt, a1);
    tt = fe_mul(cur.t1, cur.t2);
    store_var_entry(lane_tbl, i, ext_to_niels_t(cur, tt));
  }
}
template <int W>
DSV_DEV Ext ext_mul_pow2(const Ext& p) {  // 2^W * p
  Fe u = p.u, v = p.v, z = p.z;
#pragma unroll 1
  for (int j = 0; j < W - 1; j++) ext_double_uvz(u, v, z);
  Ext q;
  q.u = u;
  q.v = v;
  q.z = z;
  return ext_double(q);
}

// s * P, signed 4-bit fixed windows, MSB first: acc = 16*acc + T[digit].  TOP = index of the
// highest possibly non-zero digit (62 for a 250-bit challenge, 63 for a 252-bit Fr scalar); the
// first window is a plain addition onto the identity (no doublings of the identity).
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
// a*P + b*Q with one shared doubling chain (Straus); a < 2^252, b < 2^252
DSV_DEV Ext var_base_mul2(const u32 (&a)[8], const u32* tp, const u32 (&b)[8], const u32* tq) {
  u32 ya[8], yb[8];
  recode_signed4(ya, a);
  recode_signed4(yb, b);
  Ext acc = ext_from_niels(load_var_entry(tp, sdigit4(ya, 63)));
  acc = ext_add_niels(acc, load_var_entry(tq, sdigit4(yb, 63)));
#if DSV_VAR_PREFETCH && !DSV_VAR_NEG_T2D
  RawNiels ea = load_var_entry_raw(tp, sdigit4(ya, 62));
  RawNiels eb = load_var_entry_raw(tq, sdigit4(yb, 62));
#pragma unroll 1
  for (int k = 62; k >= 0; k--) {
    acc = ext_mul16(acc);
    const int kn = k > 0 ? k - 1 : 0;  // last round: reloads its own entries, unused
    acc = ext_add_niels(acc, finish_var_entry(ea));
    ea = load_var_entry_raw(tp, sdigit4(ya, kn));
    acc = ext_add_niels(acc, finish_var_entry(eb));
    eb = load_var_entry_raw(tq, sdigit4(yb, kn));
  }
#else
#pragma unroll 1
  for (int k = 62; k >= 0; k--) {
    acc = ext_mul16(acc);
    acc = ext_add_niels(acc, load_var_entry(tp, sdigit4(ya, k)));
    acc = ext_add_niels(acc, load_var_entry(tq, sdigit4(yb, k)));
  }
#endif
  return acc;
}

// ------------------------------------------------------------------------------------------
// verify kernels
// ------------------------------------------------------------------------------------------
// ok[i] = ok_in & [ u*Gen + c*PK == R ]   with Gen given by its fixed-base table.
// ACCUM = false: first pass, ok_in = valid[i];  ACCUM = true: ok_in = ok[i] (double scheme).
// Order of work is chosen for register pressure: PK -> window table (global workspace) -> c*PK
// -> += u*Gen -> compare with R; each input is loaded right before its only use.
template <bool ACCUM>
__global__ void __launch_bounds__(kVerifyBlock, DSV_WAVES_VERIFY)
k_verify_fixed(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c,
               const uint8_t* __restrict__ PK_uv, const uint8_t* __restrict__ R_uv,
               const u32* __restrict__ table, const uint8_t* __restrict__ valid, size_t n,
               uint8_t* __restrict__ ok, u32* __restrict__ var_tables) {
  u32* lane_tbl = var_tables + ((size_t)blockIdx.x * kVerifyBlock + threadIdx.x) * kVarLaneWords;
#pragma unroll 1
  for (size_t base = (size_t)blockIdx.x * kVerifyBlock; base < n;
       base += (size_t)gridDim.x * kVerifyBlock) {
    const size_t i = base + threadIdx.x;
    if (i >= n) continue;
    bool good = ACCUM ? (ok[i] != 0) : (valid[i] != 0);
    {
      Fe pku, pkv;
      good &= load_fq(pku, PK_uv, 2 * i);
      good &= load_fq(pkv, PK_uv, 2 * i + 1);
      build_var_table(lane_tbl, pku, pkv);
    }
    Ext acc;
    {
      u32 cs[8];
      load_words8(cs, c, i);
      acc = var_base_mul<62>(cs, lane_tbl);
    }
    {
      u32 us[8];
      load_words8(us, u, i);
      good &= words_lt(us, kR32);
      acc = fixed_base_accumulate(acc, us, table);
    }
    Fe ru, rv;
    good &= load_fq(ru, R_uv, 2 * i);
    good &= load_fq(rv, R_uv, 2 * i + 1);
    bool eq = ext_eq_affine(acc, ru, rv);
    ok[i] = (good & eq) ? 1 : 0;
  }
}

// Same verdict, ~half the doublings (halfgcd.h): with (a, b), a = b*c (mod 8r), b odd,
//   u*G + c*PK == R   <=>   (b*u mod r)*G + a*PK - b*R == O.
// Two per-lane window tables (PK and R), one Straus chain of ~34 windows whose length is the
// lane's own max(bitlen a, bitlen b) (lanes of a wave simply leave the loop at different times).
//
// NCHAIN = 2 is PublicKeyDouble::verify (/root/reference/src/keys/public.rs:222-244) in ONE
// launch: both equations share u and c, so (a, b), both recodings and b*u mod r are computed once
// and the chain runs twice — (G, PK, R) then (G', PK', R') — through the same code (a rolled loop
// over the two operand sets: the hot loop exists once in the instruction cache) and the same two
// table slots.  r01 launched the single-equation kernel twice and repeated the shared part.
struct ChainOperands {
  const uint8_t* PK_uv;
  const uint8_t* R_uv;
  const u32* table;  // fixed-base table of the generator that goes with this (PK, R) pair
};
// BLOCK / LDS: the shipped kernel runs single-wave workgroups with the fixed-base table in L2
// (BLOCK = 64, LDS = false); the A/B variant runs 8-wave workgroups that stage a narrower table in
// LDS first (op0.table then points at the DSV_FIXED_LDS_BITS-bit table in global memory).
template <bool ACCUM, int NCHAIN, int BLOCK = kVerifyBlock, bool LDS = false>
__global__ void __launch_bounds__(BLOCK, BLOCK == kVerifyBlock ? DSV_WAVES_VERIFY : 1)
k_verify_fixed_half(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c,
                    ChainOperands op0, ChainOperands op1, const uint8_t* __restrict__ valid,
                    size_t n, uint8_t* __restrict__ ok, u32* __restrict__ var_tables) {
  extern __shared__ u32 lds_table[];
  if (LDS) {
    for (int k = threadIdx.x; k < kLdsTableWords; k += BLOCK) lds_table[k] = op0.table[k];
    __syncthreads();
  }
  constexpr int kVerifyBlock = BLOCK;  // shadows the namespace constant inside this kernel
  u32* tpk = var_tables + ((size_t)blockIdx.x * kVerifyBlock + threadIdx.x) * (2 * kHalfLaneWords);
  u32* tr = tpk + kHalfLaneWords;
#pragma unroll 1
  for (size_t base = (size_t)blockIdx.x * kVerifyBlock; base < n;
       base += (size_t)gridDim.x * kVerifyBlock) {
    const size_t i = base + threadIdx.x;
    if (i >= n) continue;
    bool good = ACCUM ? (ok[i] != 0) : (valid[i] != 0);
    u32 ya[8], yb[8], w[8];
    bool b_neg;
    int top;
    {
      u32 cs[8], a[8], b[8];
      load_words8(cs, c, i);
      half_scalars(a, b, b_neg, cs);
      recode_signed_w<kHalfW>(ya, a);
      recode_signed_w<kHalfW>(yb, b);
      // index of the highest non-zero signed digit of either scalar (a zero digit is nibble 8)
      u32 nz[8];
#pragma unroll
      for (int k = 0; k < 8; k++)
        nz[k] = (ya[k] ^ recode_bias_word<kHalfW>(k)) | (yb[k] ^ recode_bias_word<kHalfW>(k));
      const int nzbits = bitlen8(nz);
      top = nzbits > 0 ? (nzbits - 1) / kHalfW : 0;
      u32 us[8];
      load_words8(us, u, i);
      const bool u_ok = words_lt(us, kR32);
      good &= u_ok;
      if (!u_ok) us[7] &= 0x0fffffffu;  // keep fr_mul's inputs below r-ish; verdict is 0 anyway
      fr_mul(w, b, us);                 // |b| * u mod r
      if (b_neg) {                      // (b*u) mod r with b < 0
        const u32 zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u32 t[8];
        fr_sub(t, zero, w);
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = t[k];
      }
    }
    const int rsign = b_neg ? 1 : -1;
#pragma unroll 1
    for (int h = 0; h < NCHAIN; h++) {
      const ChainOperands op = h ? op1 : op0;
      {
        Fe pku, pkv;
        good &= load_fq(pku, op.PK_uv, 2 * i);
        good &= load_fq(pkv, op.PK_uv, 2 * i + 1);
        build_var_table_n<kHalfEntries>(tpk, pku, pkv);
      }
      {
        Fe ru, rv;
        good &= load_fq(ru, op.R_uv, 2 * i);
        good &= load_fq(rv, op.R_uv, 2 * i + 1);
        build_var_table_n<kHalfEntries>(tr, ru, rv);
      }
      // T = a*PK + (b_neg ? +|b| : -|b|) * R  (+ w*G below)
      Ext acc = ext_from_niels(load_var_entry(tpk, sdigit_w<kHalfW>(ya, top)));
      acc = ext_add_niels(acc, load_var_entry(tr, rsign * sdigit_w<kHalfW>(yb, top)));
#if DSV_VAR_PREFETCH && !DSV_VAR_NEG_T2D
      {
        const int k0 = top > 0 ? top - 1 : 0;
        RawNiels ea = load_var_entry_raw(tpk, sdigit_w<kHalfW>(ya, k0));
        RawNiels eb = load_var_entry_raw(tr, rsign * sdigit_w<kHalfW>(yb, k0));
#pragma unroll 1
        for (int k = top - 1; k >= 0; k--) {
          acc = ext_mul_pow2<kHalfW>(acc);
          const int kn = k > 0 ? k - 1 : 0;  // last round: reloads its own entries, unused
          acc = ext_add_niels(acc, finish_var_entry(ea));
          ea = load_var_entry_raw(tpk, sdigit_w<kHalfW>(ya, kn));
          acc = ext_add_niels(acc, finish_var_entry(eb));
          eb = load_var_entry_raw(tr, rsign * sdigit_w<kHalfW>(yb, kn));
        }
      }
#else
#pragma unroll 1
      for (int k = top - 1; k >= 0; k--) {
        acc = ext_mul_pow2<kHalfW>(acc);
        acc = ext_add_niels(acc, load_var_entry(tpk, sdigit_w<kHalfW>(ya, k)));
        acc = ext_add_niels(acc, load_var_entry(tr, rsign * sdigit_w<kHalfW>(yb, k)));
      }
#endif
      acc = LDS ? fixed_base_accumulate_lds(acc, w, lds_table) : fixed_base_accumulate(acc, w, op.table);
      // T == O  <=>  u == 0 and v == z
      good &= (bool)((int)fe_is_zero_canon(fe_canon(acc.u)) & (int)fe_equal(acc.v, acc.z));
    }
    ok[i] = good ? 1 : 0;
  }
}

// ---- small batches: four lanes per signature (quad29.h) ---------------------------------------
// Same equation, same tables, same digits as k_verify_fixed_half; lane q of a quad runs one of the
// four products of every group operation, so the Straus chain takes ~2 multiplication-times per
// operation instead of 7 / 8.  Used by the entry points for n <= kQuadMaxItems, where the chip is
// mostly idle and the call's latency is one lane's serial instruction stream.  Throughput per
// lane is ~0.7 of the one-lane kernel, so large batches never come here.
// Work split inside a quad: all four lanes run the (cheap, identical) scalar preparation; lane 0
// builds the window table of PK while lane 1 builds the one of R; all four read the entries.
constexpr int kQuadBlock = 256;              // 64 signatures per workgroup
constexpr size_t kQuadMaxItems = (size_t)1 << 14;

DSV_DEV QExt qext_mul16(QExt p, int q) {
#pragma unroll 1
  for (int j = 0; j < 3; j++) qext_double<false>(p, q);
  qext_double<true>(p, q);
  return p;
}
template <bool ACCUM, int NCHAIN>
__global__ void __launch_bounds__(kQuadBlock)
k_verify_fixed_half_quad(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c,
                         ChainOperands op0, ChainOperands op1, const uint8_t* __restrict__ valid,
                         size_t n, uint8_t* __restrict__ ok, u32* __restrict__ var_tables) {
  const size_t i = ((size_t)blockIdx.x * kQuadBlock + threadIdx.x) >> 2;
  const int q = threadIdx.x & 3;
  if (i >= n) return;  // whole quads leave together
  u32* tpk = var_tables + i * (2 * kVarLaneWords);
  u32* tr = tpk + kVarLaneWords;
  bool good = ACCUM ? (ok[i] != 0) : (valid[i] != 0);
  u32 ya[8], yb[8], w[8];
  bool b_neg;
  int top;
  {
    u32 cs[8], a[8], b[8];
    load_words8(cs, c, i);
    half_scalars(a, b, b_neg, cs);
    recode_signed4(ya, a);
    recode_signed4(yb, b);
    u32 nz[8];
#pragma unroll
    for (int k = 0; k < 8; k++) nz[k] = (ya[k] ^ 0x88888888u) | (yb[k] ^ 0x88888888u);
    const int nzbits = bitlen8(nz);
    top = nzbits > 0 ? (nzbits - 1) >> 2 : 0;
    u32 us[8];
    load_words8(us, u, i);
    const bool u_ok = words_lt(us, kR32);
    good &= u_ok;
    if (!u_ok) us[7] &= 0x0fffffffu;
    fr_mul(w, b, us);
    if (b_neg) {
      const u32 zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      u32 t[8];
      fr_sub(t, zero, w);
#pragma unroll
      for (int k = 0; k < 8; k++) w[k] = t[k];
    }
  }
  const int rsign = b_neg ? 1 : -1;
#pragma unroll 1
  for (int h = 0; h < NCHAIN; h++) {
    const ChainOperands op = h ? op1 : op0;
    {
      // lane 0: table of PK, lane 1: table of R (lanes 2, 3 wait); validity of all four coordinates
      // is checked by every lane
      Fe pu, pv, ru, rv;
      good &= load_fq(pu, op.PK_uv, 2 * i);
      good &= load_fq(pv, op.PK_uv, 2 * i + 1);
      good &= load_fq(ru, op.R_uv, 2 * i);
      good &= load_fq(rv, op.R_uv, 2 * i + 1);
      if (q < 2) build_var_table(q ? tr : tpk, q ? ru : pu, q ? rv : pv);
      // the other lanes of the wave read what lanes 0 / 1 wrote to global memory
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
    }
    QExt acc = qext_identity();
    qext_add_niels(acc, q, load_var_entry(tpk, sdigit4(ya, top)));
    qext_add_niels(acc, q, load_var_entry(tr, rsign * sdigit4(yb, top)));
#pragma unroll 1
    for (int k = top - 1; k >= 0; k--) {
      acc = qext_mul16(acc, q);
      qext_add_niels(acc, q, load_var_entry(tpk, sdigit4(ya, k)));
      qext_add_niels(acc, q, load_var_entry(tr, rsign * sdigit4(yb, k)));
    }
    {  // += w * Gen: fixed_base_accumulate with the quad addition
      u32 y[9];
      u32 carry = 0;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        u32 bias = 0;
#pragma unroll
        for (int j = 0; j < kFixedWindows; j++) {
          const int pos = kFixedBits * j + kFixedBits - 1;
          if ((pos >> 5) == k) bias |= 1u << (pos & 31);
        }
        const u64 t = (u64)(k < 8 ? w[k] : 0u) + bias + carry;
        y[k] = (u32)t;
        carry = (u32)(t >> 32);
      }
#pragma unroll 1
      for (int win = 0; win < kFixedWindows; win++) {
        const int d = (int)(y[0] & ((1u << kFixedBits) - 1)) - kFixedHalf;
#pragma unroll
        for (int k = 0; k < 8; k++) y[k] = __funnelshift_r(y[k], y[k + 1], kFixedBits);
        y[8] >>= kFixedBits;
        qext_add_aniels(acc, q, load_aniels(op.table, win, d));
      }
    }
    good &= (bool)((int)fe_is_zero_canon(fe_canon(acc.u)) & (int)fe_equal(acc.v, acc.z));
    if (NCHAIN > 1) __builtin_amdgcn_wave_barrier();  // table slots are rebuilt by lanes 0 / 1 next round
  }
  if (q == 0) ok[i] = good ? 1 : 0;
}

__global__ void __launch_bounds__(kVerifyBlock, DSV_WAVES_VERIFY)
k_verify_var(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c,
             const uint8_t* __restrict__ PK_uv, const uint8_t* __restrict__ Gen_uv,
             const uint8_t* __restrict__ R_uv, const uint8_t* __restrict__ valid, size_t n,
             uint8_t* __restrict__ ok, u32* __restrict__ var_tables) {
  u32* tp = var_tables + ((size_t)blockIdx.x * kVerifyBlock + threadIdx.x) * (2 * kVarLaneWords);
  u32* tq = tp + kVarLaneWords;
#pragma unroll 1
  for (size_t base = (size_t)blockIdx.x * kVerifyBlock; base < n;
       base += (size_t)gridDim.x * kVerifyBlock) {
    const size_t i = base + threadIdx.x;
    if (i >= n) continue;
    bool good = valid[i] != 0;
    {
      Fe gu, gv;
      good &= load_fq(gu, Gen_uv, 2 * i);
      good &= load_fq(gv, Gen_uv, 2 * i + 1);
      build_var_table(tp, gu, gv);
    }
    {
      Fe pku, pkv;
      good &= load_fq(pku, PK_uv, 2 * i);
      good &= load_fq(pkv, PK_uv, 2 * i + 1);
      build_var_table(tq, pku, pkv);
    }
    Ext acc;
    {
      u32 us[8], cs[8];
      load_words8(us, u, i);
      load_words8(cs, c, i);
      good &= words_lt(us, kR32);
      if (!words_lt(us, kR32)) us[7] &= 0x0fffffffu;  // keep the recoding in range; verdict is 0 anyway
      acc = var_base_mul2(us, tp, cs, tq);
    }
    Fe ru, rv;
    good &= load_fq(ru, R_uv, 2 * i);
    good &= load_fq(rv, R_uv, 2 * i + 1);
    bool eq = ext_eq_affine(acc, ru, rv);
    ok[i] = (good & eq) ? 1 : 0;
  }
}

// (u, v, z) -> affine (u/z, v/z), canonical bytes; flags z == 0 / non-canonical as invalid
__global__ void __launch_bounds__(256)
k_normalize_uvz(const uint8_t* __restrict__ uvz, size_t n, uint8_t* __restrict__ uv,
                uint8_t* __restrict__ valid, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe x, y, z;
  bool ok = load_fq(x, uvz, 3 * i);
  ok &= load_fq(y, uvz, 3 * i + 1);
  ok &= load_fq(z, uvz, 3 * i + 2);
  Fe zc = fe_canon(z);
  ok &= !fe_is_zero_canon(zc);
  Fe zi = fe_invert(z);
  store_fq(uv, 2 * i, fe_mul(x, zi));
  store_fq(uv, 2 * i + 1, fe_mul(y, zi));
  if (accumulate) ok &= valid[i] != 0;
  valid[i] = ok ? 1 : 0;
}
__global__ void k_and_bytes(uint8_t* __restrict__ ok, const uint8_t* __restrict__ valid, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ok[i] = ok[i] & valid[i];
}

// ------------------------------------------------------------------------------------------
// signing / key derivation ("next" row of the scope table: the step that precedes verify)
// ------------------------------------------------------------------------------------------
// marks an output element as invalid: 0xff..ff is >= q and >= r, every consumer rejects it
DSV_DEV void store_poison(uint8_t* base, size_t idx) {
  const u32 w[8] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};
  store_words8(base, idx, w);
}
DSV_DEV void store_affine(uint8_t* out_uv, size_t i, const Ext& p) {
  Fe zi = fe_invert(p.z);
  store_fq(out_uv, 2 * i, fe_mul(p.u, zi));
  store_fq(out_uv, 2 * i + 1, fe_mul(p.v, zi));
}
// out = scalar * Gen (fixed-base table), affine.  R = r*G, PK = sk*G
// (/root/reference/src/keys/secret.rs:159, public.rs:61-67)
__global__ void __launch_bounds__(256, DSV_WAVES_VERIFY)
k_fixed_base_points(const uint8_t* __restrict__ scalar, const u32* __restrict__ table, size_t n,
                    uint8_t* __restrict__ out_uv) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 s[8];
  load_words8(s, scalar, i);
  if (!words_lt(s, kR32)) {  // not a JubJubScalar: poison (no canonical point has 0xff.. coordinates)
    store_poison(out_uv, 2 * i);
    store_poison(out_uv, 2 * i + 1);
    return;
  }
  Ext acc = fixed_base_accumulate(ext_identity(), s, table);
  store_affine(out_uv, i, acc);
}
// out = scalar * P for a per-item base P (var-generator scheme: secret.rs:442, public.rs:337-344)
__global__ void __launch_bounds__(kVerifyBlock, DSV_WAVES_VERIFY)
k_var_base_points(const uint8_t* __restrict__ scalar, const uint8_t* __restrict__ P_uv, size_t n,
                  uint8_t* __restrict__ out_uv, u32* __restrict__ var_tables) {
  u32* lane_tbl = var_tables + ((size_t)blockIdx.x * kVerifyBlock + threadIdx.x) * kVarLaneWords;
#pragma unroll 1
  for (size_t base = (size_t)blockIdx.x * kVerifyBlock; base < n;
       base += (size_t)gridDim.x * kVerifyBlock) {
    const size_t i = base + threadIdx.x;
    if (i >= n) continue;
    {
      Fe pu, pv;
      load_fq(pu, P_uv, 2 * i);
      load_fq(pv, P_uv, 2 * i + 1);
      build_var_table(lane_tbl, pu, pv);
    }
    u32 s[8];
    load_words8(s, scalar, i);
    const bool canonical = words_lt(s, kR32);
    s[7] &= 0x0fffffffu;  // keeps the signed recoding in range for a non-canonical scalar
    Ext acc = var_base_mul<63>(s, lane_tbl);
    if (canonical) {
      store_affine(out_uv, i, acc);
    } else {
      store_poison(out_uv, 2 * i);
      store_poison(out_uv, 2 * i + 1);
    }
  }
}
// u = r - c * sk  in Fr  (secret.rs:165)
__global__ void __launch_bounds__(256)
k_sign_finish(const uint8_t* __restrict__ r, const uint8_t* c,  // c may alias u_out
              const uint8_t* __restrict__ sk, size_t n, uint8_t* u_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 rs[8], cs[8], ks[8], t[8], u[8];
  load_words8(rs, r, i);
  load_words8(cs, c, i);
  load_words8(ks, sk, i);
  if (!words_lt(rs, kR32) || !words_lt(ks, kR32)) {  // nonce or key not a JubJubScalar
    store_poison(u_out, i);
    return;
  }
  fr_mul(t, cs, ks);
  fr_sub(u, rs, t);
  store_words8(u_out, i, u);
}

// ------------------------------------------------------------------------------------------
// wire formats: point decompression (JubJubAffine::from_bytes) and field gathering
// ------------------------------------------------------------------------------------------
// in: one 32-byte compressed point per item at in + i*in_stride (16-byte aligned);
// out_uv: affine u || v canonical; ok[i] = (accumulate ? ok[i] : 1) & decodable
__global__ void __launch_bounds__(256, DSV_WAVES_HASH)
k_decompress(const uint8_t* __restrict__ in, size_t in_stride, size_t n,
             uint8_t* __restrict__ out_uv, uint8_t* __restrict__ ok, int accumulate, TsTables ts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[8];
  {
    const uint4* p = reinterpret_cast<const uint4*>(in + i * in_stride);
    uint4 a = p[0], b = p[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
  }
  const u32 sign = w[7] >> 31;
  w[7] &= 0x7fffffffu;
  bool good = words_lt(w, kQ32);
  const Fe v = fe_to_mont(fe_from_words_plain(w));
  const Fe v2 = fe_sqr(v);
  const Fe num = fe_sub2(v2, fe_one());                          // v^2 - 1
  const Fe den = fe_add(fe_mul(v2, fe_const(kD)), fe_one());     // 1 + d v^2  (never 0: -1/d is a non-square)
  // u = n * (n d)^(-1/2); accept iff u^2 d == n  (rejects non-squares; n == 0 gives u == 0)
  Fe u = fe_mul(num, fe_inv_sqrt(fe_mul(num, den), ts));
  good &= fe_equal(fe_mul(fe_sqr(u), den), num);
  u32 uw[8];
  fe_to_words_plain(uw, fe_from_mont(u));
  if ((uw[0] & 1u) != sign) {                                    // take the other root
    u = fe_neg2(u);
    fe_to_words_plain(uw, fe_from_mont(u));
  }
  store_words8(out_uv, 2 * i, uw);
  store_words8(out_uv, 2 * i + 1, w);
  if (accumulate) good &= ok[i] != 0;
  ok[i] = good ? 1 : 0;
}
// out[i] = 32 bytes at in + i*stride   (AoS wire records -> SoA scalar array)
__global__ void k_gather32(const uint8_t* __restrict__ in, size_t stride, size_t n,
                           uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4* p = reinterpret_cast<const uint4*>(in + i * stride);
  uint4* o = reinterpret_cast<uint4*>(out + i * 32);
  o[0] = p[0];
  o[1] = p[1];
}

// ------------------------------------------------------------------------------------------
// the reference harness's inputs: item i = (sk, message, nonce) from StdRng keystream blocks
// 3i .. 3i+2 (stdrng.h)
// ------------------------------------------------------------------------------------------
struct ChaChaKey {
  u32 w[8];
};
__global__ void __launch_bounds__(256)
k_stdrng_triples(ChaChaKey key, size_t first_item, size_t n, uint8_t* __restrict__ sk,
                 uint8_t* __restrict__ m, uint8_t* __restrict__ r) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 blk = 3 * (u64)(first_item + i);
  u32 ks[16], o[8];
  chacha12_block(ks, key.w, blk);
  fr_from_wide(o, ks);
  store_words8(sk, i, o);
  chacha12_block(ks, key.w, blk + 1);
  fq_from_wide(o, ks);
  store_words8(m, i, o);
  chacha12_block(ks, key.w, blk + 2);
  fr_from_wide(o, ks);
  store_words8(r, i, o);
}

// var-generator harness (tests/schnorr_var_generator.rs:16-22, benches/signature_var_generator.rs:
// 50-63): SecretKeyVarGen::random draws sk then the generator scalar (src/keys/secret.rs:371-373),
// then the message, then (inside sign) the nonce: item i = keystream blocks 4i .. 4i+3
__global__ void __launch_bounds__(256)
k_stdrng_quads(ChaChaKey key, size_t first_item, size_t n, uint8_t* __restrict__ sk,
               uint8_t* __restrict__ g, uint8_t* __restrict__ m, uint8_t* __restrict__ r) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 blk = 4 * (u64)(first_item + i);
  u32 ks[16], o[8];
  chacha12_block(ks, key.w, blk);
  fr_from_wide(o, ks);
  store_words8(sk, i, o);
  chacha12_block(ks, key.w, blk + 1);
  fr_from_wide(o, ks);
  store_words8(g, i, o);
  chacha12_block(ks, key.w, blk + 2);
  fq_from_wide(o, ks);
  store_words8(m, i, o);
  chacha12_block(ks, key.w, blk + 3);
  fr_from_wide(o, ks);
  store_words8(r, i, o);
}

__global__ void __launch_bounds__(256)
k_debug_fq_mul(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, size_t n,
               uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe x, y;
  load_fq(x, a, i);
  load_fq(y, b, i);
  // exercise mul, sqr, add, sub paths: out = a*b  (computed as ((a+b)^2 - a^2 - b^2) / 2 cross-checked)
  Fe p = fe_mul(x, y);
  Fe s = fe_sqr(fe_add(x, y));
  Fe t = fe_sub4(fe_sub4(s, fe_sqr(x)), fe_sqr(y));  // 2ab, < 9.2 q
  t = fe_mul(t, fe_one());                           // back to < 1.2 q before the comparison
  bool same = fe_equal(t, fe_dbl(p));
  u32 w[8];
  fe_to_words_plain(w, fe_from_mont(p));
  if (!same) w[7] |= 0x80000000u;  // poison: can never be canonical
  store_words8(out, i, w);
}

// ------------------------------------------------------------------------------------------
// mixed batches (BASELINE.json configs[4]): split a batch by kind ON THE DEVICE
// kinds[i] = 0 (single signature) / 1 (double signature); anything else is an invalid item that
// lands in neither list (its verdict stays 0).  Stable compaction in three small kernels:
// per-tile counts, one-block exclusive scan of the tile counts, per-tile write-out.  HBM-bound
// byte work (n bytes in, 4n bytes out); against the ~600 k VALU instructions per verdict it is
// noise — written for coalescing, not tuned further.
// ------------------------------------------------------------------------------------------
constexpr int kSplitThreads = 256;
constexpr int kSplitPerThread = 16;                        // one 16-byte load per thread
constexpr int kSplitTile = kSplitThreads * kSplitPerThread;  // 4096 items per workgroup

// counts of kind 0 and kind 1 among the 16 items of this thread, packed (kind1 << 16 | kind0)
DSV_DEV u32 split_thread_counts(const uint8_t* __restrict__ kinds, size_t n, size_t first,
                                uint8_t (&k)[kSplitPerThread]) {
  u32 cnt = 0;
  if (first + kSplitPerThread <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(kinds + first);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < kSplitPerThread; j++) k[j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
  } else {
#pragma unroll
    for (int j = 0; j < kSplitPerThread; j++) k[j] = first + j < n ? kinds[first + j] : (uint8_t)0xff;
  }
#pragma unroll
  for (int j = 0; j < kSplitPerThread; j++) cnt += (k[j] == 0 ? 1u : 0u) + (k[j] == 1 ? 0x10000u : 0u);
  return cnt;
}
// exclusive scan over the workgroup of one packed counter per thread; returns the block total
DSV_DEV u32 split_block_scan(u32 v, u32& exclusive) {
  __shared__ u32 wave_tot[kSplitThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u32 t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  u32 before = 0, total = 0;
#pragma unroll
  for (int w2 = 0; w2 < kSplitThreads / 64; w2++) {
    const u32 t = wave_tot[w2];
    if (w2 < wave) before += t;
    total += t;
  }
  exclusive = before + inc - v;
  __syncthreads();
  return total;
}
__global__ void __launch_bounds__(kSplitThreads)
k_kind_count(const uint8_t* __restrict__ kinds, size_t n, u32* __restrict__ tile_counts) {
  uint8_t k[kSplitPerThread];
  const size_t first = ((size_t)blockIdx.x * kSplitThreads + threadIdx.x) * kSplitPerThread;
  u32 ex;
  const u32 total = split_block_scan(split_thread_counts(kinds, n, first, k), ex);
  if (threadIdx.x == 0) {
    tile_counts[2 * blockIdx.x] = total & 0xffffu;
    tile_counts[2 * blockIdx.x + 1] = total >> 16;
  }
}
// in place: tile_counts[2t + k] -> number of kind-k items in tiles before t; totals[k] = all of them
__global__ void __launch_bounds__(1024)
k_kind_scan(u32* __restrict__ tile_counts, size_t ntiles, u32* __restrict__ totals) {
  __shared__ u32 part[2][1024];
  const size_t per = (ntiles + 1023) / 1024;
  const size_t lo = (size_t)threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
  u32 s0 = 0, s1 = 0;
  for (size_t t = lo; t < hi; t++) {
    s0 += tile_counts[2 * t];
    s1 += tile_counts[2 * t + 1];
  }
  part[0][threadIdx.x] = s0;
  part[1][threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 a = 0, b = 0;
    for (int t = 0; t < 1024; t++) {
      const u32 x = part[0][t], y = part[1][t];
      part[0][t] = a;
      part[1][t] = b;
      a += x;
      b += y;
    }
    totals[0] = a;
    totals[1] = b;
  }
  __syncthreads();
  u32 a = part[0][threadIdx.x], b = part[1][threadIdx.x];
  for (size_t t = lo; t < hi; t++) {
    const u32 x = tile_counts[2 * t], y = tile_counts[2 * t + 1];
    tile_counts[2 * t] = a;
    tile_counts[2 * t + 1] = b;
    a += x;
    b += y;
  }
}
// idx_k[j] = batch position of the j-th item of kind k (j < cap_k: a caller that understated a
// count loses the surplus instead of overrunning its buffer; totals[] tell)
__global__ void __launch_bounds__(kSplitThreads)
k_kind_write(const uint8_t* __restrict__ kinds, size_t n, const u32* __restrict__ tile_offsets,
             u32* __restrict__ idx0, size_t cap0, u32* __restrict__ idx1, size_t cap1) {
  uint8_t k[kSplitPerThread];
  const size_t first = ((size_t)blockIdx.x * kSplitThreads + threadIdx.x) * kSplitPerThread;
  u32 ex;
  split_block_scan(split_thread_counts(kinds, n, first, k), ex);
  size_t p0 = (size_t)tile_offsets[2 * blockIdx.x] + (ex & 0xffffu);
  size_t p1 = (size_t)tile_offsets[2 * blockIdx.x + 1] + (ex >> 16);
#pragma unroll
  for (int j = 0; j < kSplitPerThread; j++) {
    if (k[j] == 0) {
      if (p0 < cap0) idx0[p0] = (u32)(first + j);
      p0++;
    } else if (k[j] == 1) {
      if (p1 < cap1) idx1[p1] = (u32)(first + j);
      p1++;
    }
  }
}
// dst row j = src row idx[j]; rows of row16 * 16 bytes, one thread per 16-byte piece
__global__ void __launch_bounds__(256)
k_gather_rows(const uint4* __restrict__ src, u32 row16, const u32* __restrict__ idx, size_t count,
              uint4* __restrict__ dst) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= count * row16) return;
  const size_t j = g / row16;
  const u32 part = (u32)(g - j * row16);
  dst[g] = src[(size_t)idx[j] * row16 + part];
}
__global__ void __launch_bounds__(256)
k_scatter_bytes(const uint8_t* __restrict__ src, const u32* __restrict__ idx, size_t count,
                uint8_t* __restrict__ dst) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < count) dst[idx[j]] = src[j];
}
// a mixed call whose declared kind counts disagree with the kind vector has no usable verdicts
__global__ void __launch_bounds__(256)
k_mixed_check(const u32* __restrict__ totals, u32 want0, u32 want1, uint8_t* __restrict__ ok, size_t n) {
  if (totals[0] == want0 && totals[1] == want1) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    ok[i] = 0;
}

}  // namespace dsv
