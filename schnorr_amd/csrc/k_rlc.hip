// k_rlc.hip — SURVEY.md §8(f)-4: random-linear-combination batch verification of
// `PublicKey::verify` (/root/reference/src/keys/public.rs:121-130) as an OPTIONAL fast-accept path
// in front of the per-signature kernels (k_verify.hip), behind the same bool-vector boundary.
//
// The reference's equation is cofactorless, u*G + c*PK == R, and its types hold points with a
// small-order component (`from_bytes` checks the curve equation only, src/keys/public.rs:94-100), so
// a plain  sum z_i (u_i G + c_i PK_i - R_i) == O  is NOT the reference's verdict: torsion defects
// cancel.  What this path accepts instead is the conjunction of
//   (1) every PK_i and R_i that enters the sum lies in the prime-order subgroup, and
//   (2) (sum z_i u_i) G + sum (z_i c_i) PK_i - sum z_i R_i == O     with secret random 128-bit z_i,
// under which every per-signature verdict is `true` (error <= 2^-112: see below).  If either fails
// the caller (dsv_rlc.hip) runs the per-signature kernels and returns THEIR verdicts, so the bool vector
// is the reference's in every case; only the time differs.
//
// Both come out of ONE bucket pass.  With c-bit unsigned windows, (2) is Pippenger: bucket (w, d)
// sums the points whose scalar has digit d in window w.  Seen as a 2^(c/2) x 2^(c/2) matrix per
// window, the buckets' row and column sums give, for every BIT p of the scalars,
//   S_p = sum over { i : bit p of scalar_i is set } of P_i,
// and sum_i scalar_i P_i = sum_p 2^p S_p.  The S_p are also 252 + 128 independent random-subset sums
// of the inputs: if some PK_i0 (R_i0) has a torsion component t != 0, then "r * S_p == O for every
// p" pins every bit of z_i0 c_i0 mod r (of z_i0) to one value — probability 2^-128 over z_i0 (the
// map z -> z c mod r is injective for c != 0; for c == 0 the key does not enter the equation, in the
// reference's either).  PK and R scalars use SEPARATE windows so the two arguments stay independent.
// Cost per signature at c = 16: 16 + 8 mixed additions (7 multiplications each) against ~1900
// multiplications of the half-gcd chain, plus a sort of 24 (key, index) pairs.
//
// Kernels (launch order; every stage reads what the previous one wrote, same stream; blockIdx.y = the
// sub-group of the group, rlc.h):
//   k_rlc_prep        per item: z_i = ChaCha12(key, i), e_i = z_i c_i, f_i = z_i u_i (mod r), range and
//                     curve checks, the points as affine niels (PK_i, -R_i), one 16-bit digit per window
//   k_rlc_part1       a row of digits -> bins by the digit's high bits (workgroup-aggregated reservations)
//   k_rlc_part2       one bin -> its buckets' runs of point indices (counting sort through LDS), run starts
//                     and lengths in the same pass
//   k_rlc_lenhist / k_rlc_order   bucket numbers by run length, longest first (counting sort)
//   k_rlc_accumulate  one lane per bucket: mixed additions over its run
//   k_rlc_sum<0..3>   row / column sums, then the per-bit subset sums S_p (short chains; <1..3>: a butterfly
//                     over `count` lanes per output); extra workgroups of <0> / <1> sum the f_i mod r
//   k_rlc_scale       lanes A: r * S_p == O ?    lanes B: 2^p * S_p    one more lane: (sum f_i) * G;  the last
//                     workgroup of lanes B adds their results up: identity test -> flags
//   k_rlc_sample_decide / k_rlc_verdict   the sample's and the call's verdicts (dsv_rlc.hip: no host round trip)
// Every kernel of the chain returns at once when word 0 of the group's flag block is set (the sample
// found a wrong signature: the per-signature kernels decide).
#include "common.h"
#include "rlc.h"
#include "stdrng.h"

namespace dsv {

namespace {
constexpr int kPtWords = 32;     // affine niels (v+u, v-u, 2d*uv): 27 words, padded to one 128-byte line
constexpr int kNielsWords = 36;  // (v+u, v-u, z, 2d*t)

// signed binary expansion of r (non-adjacent form, 85 non-zero digits, top digit +2^252):
// r = kRNafPos - kRNafNeg, checked at compile time below
__device__ constexpr u32 kRNafPos[8] = {0x00004100u, 0x10a01080u, 0x11081084u, 0xa8882094u,
                                        0x01444000u, 0x08884001u, 0x85440029u, 0x1080050au};
__device__ constexpr u32 kRNafNeg[8] = {0x29091449u, 0x40090221u, 0x44400001u, 0x02200000u,
                                        0x00100500u, 0x02210500u, 0x20105080u, 0x02025020u};
constexpr u32 kRWords[8] = DSV_R32;
constexpr bool naf_is_r() {
  const u32 pos[8] = {0x00004100u, 0x10a01080u, 0x11081084u, 0xa8882094u, 0x01444000u, 0x08884001u, 0x85440029u, 0x1080050au};
  const u32 neg[8] = {0x29091449u, 0x40090221u, 0x44400001u, 0x02200000u, 0x00100500u, 0x02210500u, 0x20105080u, 0x02025020u};
  u64 borrow = 0;
  for (int i = 0; i < 8; i++) {
    const u64 d = (u64)pos[i] - neg[i] - borrow;
    if ((u32)d != kRWords[i]) return false;
    borrow = (d >> 63) & 1;
    if (pos[i] & neg[i]) return false;
  }
  return borrow == 0;
}
static_assert(naf_is_r(), "kRNafPos - kRNafNeg must be the subgroup order r");

DSV_DEV void fr_add(u32 (&out)[8], const u32 (&a)[8], const u32 (&b)[8]) {  // a + b mod r (a, b < r)
  u32 s[8], d[8];
  u32 carry = 0, borrow = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const u64 y = (u64)a[j] + b[j] + carry;
    s[j] = (u32)y;
    carry = (u32)(y >> 32);
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const u64 y = (u64)s[j] - kR32[j] - borrow;
    d[j] = (u32)y;
    borrow = (u32)(y >> 63);
  }
  const bool ge = borrow == 0;  // (r < 2^252: a + b never carries out of 256 bits)
#pragma unroll
  for (int j = 0; j < 8; j++) out[j] = ge ? d[j] : s[j];
}

DSV_DEV void store_pt(u32* p, const ANiels& n) {
  uint4* q = reinterpret_cast<uint4*>(p);
  u32 w[28];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    w[i] = n.vpu.l[i];
    w[NL + i] = n.vmu.l[i];
    w[2 * NL + i] = n.t2d.l[i];
  }
  w[27] = 0;
#pragma unroll
  for (int k = 0; k < 7; k++) q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
DSV_DEV ANiels load_pt(const u32* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  u32 w[28];
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const uint4 x = q[k];
    w[4 * k] = x.x, w[4 * k + 1] = x.y, w[4 * k + 2] = x.z, w[4 * k + 3] = x.w;
  }
  ANiels n;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    n.vpu.l[i] = w[i];
    n.vmu.l[i] = w[NL + i];
    n.t2d.l[i] = w[2 * NL + i];
  }
  return n;
}
DSV_DEV void store_niels(u32* p, const Niels& n) {
  uint4* q = reinterpret_cast<uint4*>(p);
  u32 w[kNielsWords];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    w[i] = n.vpu.l[i];
    w[NL + i] = n.vmu.l[i];
    w[2 * NL + i] = n.z.l[i];
    w[3 * NL + i] = n.t2d.l[i];
  }
#pragma unroll
  for (int k = 0; k < 9; k++) q[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
DSV_DEV Niels load_niels(const u32* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  u32 w[kNielsWords];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const uint4 x = q[k];
    w[4 * k] = x.x, w[4 * k + 1] = x.y, w[4 * k + 2] = x.z, w[4 * k + 3] = x.w;
  }
  Niels n;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    n.vpu.l[i] = w[i];
    n.vmu.l[i] = w[NL + i];
    n.z.l[i] = w[2 * NL + i];
    n.t2d.l[i] = w[3 * NL + i];
  }
  return n;
}
DSV_DEV Niels niels_neg(const Niels& n) {
  Niels r;
  r.vpu = n.vmu;
  r.vmu = n.vpu;
  r.z = n.z;
  r.t2d = fe_neg2(n.t2d);
  return r;
}

// -u^2 + v^2 == 1 + d u^2 v^2, as 2 v^2 == 2 u^2 + 2 + (2d) u^2 v^2 (u, v: fe_mul outputs)
DSV_DEV bool on_curve(const Fe& u, const Fe& v) {
  const Fe uu = fe_sqr(u), vv = fe_sqr(v);
  const Fe rhs = fe_mul(fe_mul(uu, vv), fe_const(kD2));
  const Fe a = fe_carry(fe_dbl(vv));                                           // < 3q
  const Fe b0 = fe_carry(fe_add(fe_dbl(uu), fe_dbl(fe_one())));                // < 5q
  const Fe b = fe_carry(fe_add(b0, rhs));                                      // < 6.5q, limbs < 2^29 + 8
  return fe_equal(a, b);
}
// (v+u, v-u, 2d*uv) of (u, v), or of (-u, v) — forms as ext_to_niels stores them
DSV_DEV ANiels affine_niels(const Fe& u, const Fe& v, bool negate) {
  ANiels n;
  const Fe s = fe_carry(fe_add(v, u)), d = fe_sub2(v, u);
  const Fe t = fe_mul(fe_mul(u, v), fe_const(kD2));
  n.vpu = negate ? d : s;
  n.vmu = negate ? s : d;
  n.t2d = negate ? fe_neg2(t) : t;
  return n;
}
}  // namespace

// ---- per item ---------------------------------------------------------------------------------
// What an item contributes (SCHEME 0 single, 1 double, 2 var-generator; p.lpts "long" points with
// 252-bit scalars, p.spts "short" ones with the z themselves, p.fixed fixed-base terms):
//   single  u G  + c PK      - R        : long { PK: z c },            short { -R: z },          fixed { G: z u }
//   double  ... and u G' + c PK' - R'   : long { PK: z c, PK': z' c }, short { -R: z, -R': z' }, fixed { G: z u, G': z' u }
//   vargen  u Gen + c PK - R            : long { PK: z c, Gen: z u },  short { -R: z },          fixed { }
// (/root/reference/src/keys/public.rs:121-130, :222-244, :401-415), z and z' independent.
namespace {
// which items a workgroup's sub-group covers (rlc.h: RlcPlan)
struct SubView {
  u32 g;      // the sub-group (blockIdx.y)
  u32 base;   // its first item in the group's arrays
  u32 first;  // first item of this pass inside the sub-group
  u32 n;      // items of this pass
  u32 total;  // items of the sub-group
};
DSV_DEV SubView sub_view(const RlcPlan& p) {
  SubView v;
  v.g = blockIdx.y;
  if (p.groups == 1) {
    v.base = 0, v.first = p.first, v.n = p.n, v.total = p.total;
  } else {
    v.base = v.g * p.sub;
    const u32 left = p.items - v.base;
    v.total = left < p.sub ? left : p.sub;
    v.first = 0, v.n = v.total;
  }
  return v;
}
struct PrepOut {
  size_t gi;  // the item's place in the group's arrays (inputs, ok)
  u32 i;      // ... in its sub-group (points, weights)
  u32 il;     // ... and in the range this pass covers (the digit rows)
  u32 total;  // items of the sub-group
  const RlcPlan& p;
  u32* pts;
  uint16_t* digits;
};
// loads one point, folds its range check into `good`, returns "is on the curve", stores it as affine niels
DSV_DEV bool prep_point(const PrepOut& o, const uint8_t* __restrict__ uv, int slot, bool negate, bool& good) {
  Fe pu, pv;
  good &= load_fq(pu, uv, 2 * o.gi);
  good &= load_fq(pv, uv, 2 * o.gi + 1);
  store_pt(o.pts + ((size_t)slot * o.total + o.i) * kPtWords, affine_niels(pu, pv, negate));
  return on_curve(pu, pv);
}
// e' = e + k r, k uniform below floor(2^(wpk c) / r): the same multiple of a point of the prime-order
// subgroup (any other point fails the subgroup test anyway), but uniform over ALL wpk * c bits —
// without it the top window of a 252-bit scalar has a few thousand (c = 16: 2^12) digits only, and
// its buckets get runs 16 times as long as the others: a lane per bucket would wait for those.
// Then one digit per window into the point's row of that window (digit 0: the entry enters no bucket).
DSV_DEV void emit_long(const PrepOut& o, u32 (&e)[8], u32 kr, int slot) {
  const RlcPlan& p = o.p;
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const u64 t = (u64)kR32[k] * kr + e[k] + carry;
    e[k] = (u32)t;
    carry = t >> 32;
  }
  const u32 mask = (1u << p.c) - 1u;
#pragma unroll 1
  for (int w = 0; w < p.wpk; w++) {
    const u32 d = e[0] & mask;
#pragma unroll
    for (int k = 0; k < 7; k++) e[k] = __funnelshift_r(e[k], e[k + 1], p.c);
    e[7] >>= p.c;
    o.digits[(size_t)(w * p.lpts + slot) * p.row_stride + o.il] = (uint16_t)d;
  }
}
DSV_DEV void emit_short(const PrepOut& o, const u32 (&zz)[8], int slot) {
  const RlcPlan& p = o.p;
  u32 z[5] = {zz[0], zz[1], zz[2], zz[3], zz[4]};
  const u32 mask = (1u << p.c) - 1u;
  const size_t first = (size_t)p.wpk * p.lpts;
#pragma unroll 1
  for (int w = 0; w < p.wr; w++) {
    const u32 d = z[0] & mask;
#pragma unroll
    for (int k = 0; k < 4; k++) z[k] = __funnelshift_r(z[k], z[k + 1], p.c);
    z[4] >>= p.c;
    o.digits[(first + (size_t)(w * p.spts + slot)) * p.row_stride + o.il] = (uint16_t)d;
  }
}
// wr * c >= 128 random bits from five keystream words: every window of z is uniform
DSV_DEV void draw_z(u32 (&z)[8], const u32* blk, int zbits, bool good) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int left = zbits - 32 * k;
    const u32 m = left >= 32 ? ~0u : (left > 0 ? (1u << left) - 1u : 0u);
    z[k] = (good && k < 5) ? (blk[k] & m) : 0u;
  }
}
}  // namespace

template <int SCHEME>
__global__ void __launch_bounds__(256)
k_rlc_prep(RlcInputs in, ChaChaKey key, RlcPlan p, RlcBuffers b, uint8_t* __restrict__ ok) {
  if (b.flags[0]) return;
  const SubView v = sub_view(p);
  const u32 il = blockIdx.x * 256 + threadIdx.x;
  if (il >= v.n) return;
  const u32 i = v.first + il;
  const size_t gi = (size_t)v.base + i;
  const PrepOut o{gi, i, il, v.total, p, b.pts + (size_t)v.g * b.pts_stride, b.digits + (size_t)v.g * b.digits_stride};
  u32* fsc = b.fsc + (size_t)v.g * b.fsc_stride;
  bool good = in.valid[gi] != 0;
  u32 us[8], cs[8];
  load_words8(us, in.u, gi);
  load_words8(cs, in.c, gi);
  good &= words_lt(us, kR32);
  // slots: long points first (PK, then PK' / Gen), then the short ones (-R, -R')
  bool curve = prep_point(o, in.pk[0], 0, false, good);
  if (SCHEME == 1) curve &= prep_point(o, in.pk[1], 1, false, good);
  if (SCHEME == 2) curve &= prep_point(o, in.gen, 1, false, good);
  curve &= prep_point(o, in.r[0], p.lpts, true, good);
  if (SCHEME == 1) curve &= prep_point(o, in.r[1], p.lpts + 1, true, good);
  // a point off the curve has no place in a group sum: the per-signature kernels decide the sub-group
  if (good && !curve) atomicOr(&b.flags[4 + 4 * v.g], kRlcOffCurve);
  ok[gi] = good ? 1 : 0;
  u32 blk[16];
  chacha12_block(blk, key.w, (u64)gi);
  if (!good) {
    us[7] &= 0x0fffffffu;  // keep fr_mul's inputs in range; the products are 0 anyway
    cs[7] &= 0x0fffffffu;
  }
  // an item with verdict `false` stays out of every sum (z = 0)
#pragma unroll
  for (int eq = 0; eq < (SCHEME == 1 ? 2 : 1); eq++) {
    u32 z[8], e[8];
    draw_z(z, blk + 8 * eq, p.wr * p.c, good);
    fr_mul(e, z, cs);
    emit_long(o, e, good ? blk[8 * eq + 5] % p.kmul : 0u, eq);
    fr_mul(e, z, us);
    if (SCHEME == 2) emit_long(o, e, good ? blk[6] % p.kmul : 0u, 1);
    else store_words8(reinterpret_cast<uint8_t*>(fsc), (size_t)eq * v.total + i, e);
    emit_short(o, z, eq);
  }
}

// sum of the sub-group's scalars z_i u_i mod r (fixed-base term k): stage 0 -> kRlcFsumBlocks partial sums
// (workgroup blk of them), stage 1 (one workgroup) -> fsum.  One 64-thread workgroup; runs as extra
// workgroups of k_rlc_sum<0> / k_rlc_sum<1> — 0.05 ms of launches of their own otherwise.
DSV_DEV void fsum_block(const RlcPlan& p, const RlcBuffers& b, int stage, u32 g, u32 k, u32 blk) {
  __shared__ u32 sh[64][8];
  u32 total = p.total;
  if (p.groups > 1) {
    const u32 left = p.items - g * p.sub;
    total = left < p.sub ? left : p.sub;
  }
  u32* fpart = b.fpart + ((size_t)g * 2 + k) * kRlcFsumBlocks * 8;
  const u32* in = stage == 0 ? b.fsc + (size_t)g * b.fsc_stride + (size_t)k * total * 8 : fpart;
  const size_t n = stage == 0 ? total : kRlcFsumBlocks;
  u32* out = stage == 0 ? fpart + (size_t)blk * 8 : b.fsum + ((size_t)g * 2 + k) * 8;
  const size_t nblk = stage == 0 ? kRlcFsumBlocks : 1;
  u32 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = (size_t)blk * 64 + threadIdx.x; i < n; i += nblk * 64) {
    u32 x[8], t[8];
    load_words8(x, reinterpret_cast<const uint8_t*>(in), i);
    fr_add(t, acc, x);
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = t[j];
  }
#pragma unroll
  for (int j = 0; j < 8; j++) sh[threadIdx.x][j] = acc[j];
  __syncthreads();
  for (int step = 32; step > 0; step >>= 1) {
    if ((int)threadIdx.x < step) {
      u32 x[8], y[8], t[8];
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = sh[threadIdx.x][j], y[j] = sh[threadIdx.x + step][j];
      fr_add(t, x, y);
#pragma unroll
      for (int j = 0; j < 8; j++) sh[threadIdx.x][j] = t[j];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    u32 t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = sh[0][j];
    store_words8(reinterpret_cast<uint8_t*>(out), 0, t);
  }
}

// ---- buckets ----------------------------------------------------------------------------------
// Pass 1.  One workgroup takes kRlcTile digits of ONE row (window w, point slot): a histogram of their
// bins in LDS, ONE reservation per bin and workgroup in the bin's fill counter, then every entry goes to
// its place as (low digit bits << 24 | point index).  The digits are all a row holds: the window is the
// row's, the point index follows from the position.
__global__ void __launch_bounds__(256)
k_rlc_part1(RlcPlan p, RlcBuffers b) {
  if (b.flags[0]) return;
  __shared__ u32 hist[256], base[256];
  const SubView v = sub_view(p);
  const u32 tiles = (p.row_stride + kRlcTile - 1) / kRlcTile;
  const u32 row = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  const u32 long_rows = (u32)(p.wpk * p.lpts);
  u32 w, slot;
  if (row < long_rows) w = row / p.lpts, slot = row % p.lpts;
  else w = p.wpk + (row - long_rows) / p.spts, slot = p.lpts + (row - long_rows) % p.spts;
  const uint16_t* digits = b.digits + (size_t)v.g * b.digits_stride + (size_t)row * p.row_stride;
  u32* fill = b.counters + (size_t)v.g * b.counters_stride + ((size_t)w << p.coarse_bits);
  u32* bins = b.binned + (size_t)v.g * b.bin_stride + ((size_t)w << p.coarse_bits) * p.bin_cap;
  const u32 val0 = slot * v.total + v.first;
  const u32 fine_mask = (1u << p.fine_bits) - 1u;
  hist[threadIdx.x] = 0;
  __syncthreads();
  constexpr int kPer = kRlcTile / (256 * 8);  // 16-byte loads per thread
  u32 d[kPer][8];
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const u32 il = tile * kRlcTile + (u32)k * 2048u + threadIdx.x * 8u;
    uint4 x = make_uint4(0, 0, 0, 0);
    if (il < p.row_stride) x = *reinterpret_cast<const uint4*>(digits + il);
    const u32 words[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u32 dj = (words[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      d[k][j] = il + j < v.n ? dj : 0u;  // (beyond the pass's items a row holds nothing)
      if (d[k][j]) atomicAdd(&hist[d[k][j] >> p.fine_bits], 1u);
    }
  }
  __syncthreads();
  {
    const u32 h = hist[threadIdx.x];
    base[threadIdx.x] = h ? atomicAdd(&fill[threadIdx.x], h) : 0u;  // (bins beyond 1 << coarse_bits: h == 0)
    hist[threadIdx.x] = 0;
  }
  __syncthreads();
  bool overflow = false;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const u32 il = tile * kRlcTile + (u32)k * 2048u + threadIdx.x * 8u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u32 dj = d[k][j];
      if (!dj) continue;
      const u32 bin = dj >> p.fine_bits;
      const u32 pos = base[bin] + atomicAdd(&hist[bin], 1u);
      if (pos < p.bin_cap) bins[(size_t)bin * p.bin_cap + pos] = ((dj & fine_mask) << 24) | (val0 + il + j);
      else overflow = true;
    }
  }
  if (overflow) atomicOr(&b.flags[4 + 4 * v.g], kRlcOverflow);
}
// Pass 2.  One workgroup per bin: a counting sort of its entries by the low digit bits through LDS; the
// buckets' runs (start in `sorted`, length) come out of the same pass.
__global__ void __launch_bounds__(256)
k_rlc_part2(RlcPlan p, RlcBuffers b) {
  if (b.flags[0]) return;
  __shared__ u32 hist[256], offs[256];
  const u32 g = blockIdx.y, bin = blockIdx.x;
  const u32 w = bin >> p.coarse_bits, coarse = bin & ((1u << p.coarse_bits) - 1u);
  const u32 nfine = 1u << p.fine_bits;
  u32 have = b.counters[(size_t)g * b.counters_stride + bin];
  if (have > p.bin_cap) have = p.bin_cap;
  const u32* in = b.binned + (size_t)g * b.bin_stride + (size_t)bin * p.bin_cap;
  u32* out = b.sorted + (size_t)g * b.bin_stride + (size_t)bin * p.bin_cap;
  hist[threadIdx.x] = 0;
  __syncthreads();
  for (u32 i = threadIdx.x; i < have; i += 256) atomicAdd(&hist[in[i] >> 24], 1u);
  __syncthreads();
  // exclusive prefix sums of the 256 counts (Hillis - Steele through LDS)
  const u32 mine = hist[threadIdx.x];
  offs[threadIdx.x] = mine;
  __syncthreads();
  for (u32 step = 1; step < 256; step <<= 1) {
    const u32 add = threadIdx.x >= step ? offs[threadIdx.x - step] : 0u;
    __syncthreads();
    offs[threadIdx.x] += add;
    __syncthreads();
  }
  const u32 first = offs[threadIdx.x] - mine;
  __syncthreads();
  offs[threadIdx.x] = first;  // from here on: the bucket's cursor
  if (threadIdx.x < nfine) {
    const size_t bucket = (size_t)g * b.bucket_stride + (((size_t)w << p.c) | ((size_t)coarse << p.fine_bits) | threadIdx.x);
    b.start[bucket] = bin * p.bin_cap + first;
    b.cnt[bucket] = mine;
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < have; i += 256) {
    const u32 e = in[i];
    out[atomicAdd(&offs[e >> 24], 1u)] = e & 0x00ffffffu;
  }
}
// Bucket numbers sorted by run length, longest first: every wave of the accumulation gets 64 runs of
// (nearly) the same length — with buckets in their natural order a wave waits for its longest run
// (Poisson, mean 16: the maximum of 64 is ~26).  A counting sort over the lengths (clipped to 255):
// histogram (workgroup-aggregated), then placement.
constexpr int kLenPerThread = 32;
__global__ void __launch_bounds__(256)
k_rlc_lenhist(RlcPlan p, RlcBuffers b) {
  if (b.flags[0]) return;
  __shared__ u32 hist[256];
  const u32 g = blockIdx.y;
  const u32* cnt = b.cnt + (size_t)g * b.bucket_stride;
  hist[threadIdx.x] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * 256 * kLenPerThread;
  for (int k = 0; k < kLenPerThread; k++) {
    const size_t bk = first + (size_t)k * 256 + threadIdx.x;
    if (bk < p.buckets) {
      const u32 len = cnt[bk];
      atomicAdd(&hist[len < 255u ? len : 255u], 1u);
    }
  }
  __syncthreads();
  if (hist[threadIdx.x]) atomicAdd(&b.counters[(size_t)g * b.counters_stride + p.bins + threadIdx.x], hist[threadIdx.x]);
}
__global__ void __launch_bounds__(256)
k_rlc_order(RlcPlan p, RlcBuffers b) {
  if (b.flags[0]) return;
  __shared__ u32 hist[256], base[256], scan[256];
  const u32 g = blockIdx.y;
  const u32* cnt = b.cnt + (size_t)g * b.bucket_stride;
  u32* order = b.order + (size_t)g * b.bucket_stride;
  u32* counters = b.counters + (size_t)g * b.counters_stride + p.bins;
  hist[threadIdx.x] = 0;
  // buckets with a LONGER run than t: suffix sums of the global histogram
  const u32 mine = counters[threadIdx.x];
  scan[threadIdx.x] = mine;
  __syncthreads();
  for (u32 step = 1; step < 256; step <<= 1) {
    const u32 add = threadIdx.x + step < 256 ? scan[threadIdx.x + step] : 0u;
    __syncthreads();
    scan[threadIdx.x] += add;
    __syncthreads();
  }
  const u32 longer = scan[threadIdx.x] - mine;
  const size_t first = (size_t)blockIdx.x * 256 * kLenPerThread;
  u32 len[kLenPerThread];
  for (int k = 0; k < kLenPerThread; k++) {
    const size_t bk = first + (size_t)k * 256 + threadIdx.x;
    len[k] = 256;
    if (bk < p.buckets) {
      const u32 l = cnt[bk];
      len[k] = l < 255u ? l : 255u;
      atomicAdd(&hist[len[k]], 1u);
    }
  }
  __syncthreads();
  {
    const u32 h = hist[threadIdx.x];
    base[threadIdx.x] = longer + (h ? atomicAdd(&counters[256 + threadIdx.x], h) : 0u);
    hist[threadIdx.x] = 0;
  }
  __syncthreads();
  for (int k = 0; k < kLenPerThread; k++) {
    if (len[k] > 255u) continue;
    const size_t bk = first + (size_t)k * 256 + threadIdx.x;
    const u32 pos = base[len[k]] + atomicAdd(&hist[len[k]], 1u);
    if (pos < p.buckets) order[pos] = (u32)bk;
  }
}
// one lane per bucket (in the order of `order`): its run of the sorted point indices, one mixed addition
// per entry; the next entry's point is loaded while the current one is added
__global__ void __launch_bounds__(64)
k_rlc_accumulate(RlcPlan p, RlcBuffers b, bool second) {
  if (b.flags[0]) return;
  const SubView v = sub_view(p);
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= p.buckets) return;
  const u32 bk = b.order[(size_t)v.g * b.bucket_stride + t];
  if (bk >= p.buckets) return;  // (cannot happen: k_rlc_order writes a permutation)
  const u32* vals = b.sorted + (size_t)v.g * b.bin_stride;
  const u32* pts = b.pts + (size_t)v.g * b.pts_stride;
  u32 lo = b.start[(size_t)v.g * b.bucket_stride + bk], len = b.cnt[(size_t)v.g * b.bucket_stride + bk];
  // (never read outside the arrays, whatever the counters hold)
  const u32 slots = p.bins * p.bin_cap;
  if (lo > slots) lo = slots;
  if (len > slots - lo) len = slots - lo;
  const u32 hi = lo + len;
  const u32 last_pt = (u32)(p.lpts + p.spts) * v.total - 1u;
  Ext acc = ext_identity();
  if (lo < hi) {
    // the index of entry j + 2 and the point of entry j + 1 are on their way while entry j is added
    u32 id = vals[lo];
    ANiels cur = load_pt(pts + (size_t)(id < last_pt ? id : last_pt) * kPtWords);
    id = vals[lo + 1 < hi ? lo + 1 : lo];
#pragma unroll 1
    for (u32 j = lo; j < hi; j++) {
      const ANiels nxt = load_pt(pts + (size_t)(id < last_pt ? id : last_pt) * kPtWords);
      id = vals[j + 2 < hi ? j + 2 : hi - 1];
      acc = ext_add_aniels(acc, cur);
      cur = nxt;
    }
  }
  u32* out = (second ? b.buckets2 : b.buckets) + (size_t)v.g * b.bucket_stride * kNielsWords;
  store_niels(out + (size_t)bk * kNielsWords, ext_to_niels(acc));
}

// buckets[b] += buckets2[b]: the two ranges of a staged bucket pass
__global__ void __launch_bounds__(64)
k_rlc_merge(RlcPlan p, RlcBuffers bf) {
  if (bf.flags[0]) return;
  const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= p.buckets) return;
  u32* buckets = bf.buckets + (size_t)blockIdx.y * bf.bucket_stride * kNielsWords;
  const u32* buckets2 = bf.buckets2 + (size_t)blockIdx.y * bf.bucket_stride * kNielsWords;
  Ext acc = ext_add_niels(ext_identity(), load_niels(buckets + b * kNielsWords));
  acc = ext_add_niels(acc, load_niels(buckets2 + b * kNielsWords));
  store_niels(buckets + b * kNielsWords, ext_to_niels(acc));
}

// ---- short sums of stored points: out[o] = sum_k in[addr(o, k)] ------------------------------------
//   MODE 0  buckets -> segments of the row (kind 1) / column (kind 0) sums of each window's bucket matrix
//   MODE 1  segments -> row / column sums ("lines")
//   MODE 2  lines -> segments of: sum of the lines whose index has bit j set
//   MODE 3  segments -> S[w * c + kind * half + j], the subset sum of bit (kind * half + j) of window w
// MODE 0 (196 k outputs of 16 inputs at c = 16: throughput) runs one lane per output.  MODES 1 - 3 (a few
// thousand outputs: pure latency, 16 dependent additions of ~4 us each) run up to FOUR lanes per output: every
// lane adds count / 4 inputs, then a butterfly of two exchanges (36 words through ds_bpermute) and additions
// leaves the sum in all four — 3 + 8 (count / 4 - 1) + 18 multiplications deep instead of 8 count.  (count
// lanes per output and a butterfly of log2(count) levels is as deep, but every lane adds at every level: 5 x
// the work, and with the 12 288 outputs of MODE 1 at c = 16 that is no longer latency: 73 against 64 us.)
// `extra` workgroups behind the outputs' own run fsum_block (MODE 0: stage 0, MODE 1: stage 1).
template <int MODE>
__global__ void __launch_bounds__(64)
k_rlc_sum(const u32* __restrict__ in, size_t in_stride, RlcPlan p, u32* __restrict__ out, size_t out_stride,
          RlcBuffers bf, u32 main_blocks) {
  if (bf.flags[0]) return;
  if (MODE <= 1 && blockIdx.x >= main_blocks) {  // (uniform per workgroup)
    const u32 e = blockIdx.x - main_blocks;
    if (MODE == 0) fsum_block(p, bf, 0, blockIdx.y, e / kRlcFsumBlocks, e % kRlcFsumBlocks);
    else fsum_block(p, bf, 1, blockIdx.y, e, 0);
    return;
  }
  in += (size_t)blockIdx.y * in_stride * kNielsWords;    // (strides in points)
  out += (size_t)blockIdx.y * out_stride * kNielsWords;
  const u32 side = 1u << p.half;
  u32 total, count;
  if (MODE == 0) total = (u32)p.windows * 2 * side * p.nseg, count = side / p.nseg;
  else if (MODE == 1) total = (u32)p.windows * 2 * side, count = p.nseg;
  else if (MODE == 2) total = (u32)p.windows * 2 * p.half * p.nseg2, count = side / 2 / p.nseg2;
  else total = (u32)p.windows * 2 * p.half, count = p.nseg2;
  const u32 t = blockIdx.x * 64 + threadIdx.x;
  // count: a power of two <= 16 (tests/test_rlc_plan.py)
  const u32 lanes_per = MODE == 0 ? 1u : (count < 4u ? count : 4u), per_lane = count / lanes_per;
  const u32 o_raw = t / lanes_per, lane = t % lanes_per;
  const bool live = o_raw < total;
  if (MODE == 0 && !live) return;
  const u32 o = live ? o_raw : total - 1;  // (MODES 1 - 3: idle lanes keep step with the exchanges)
  u32 base = 0, step = 1, j = 0, first = 0;
  if (MODE == 0) {
    const u32 seg = o % p.nseg, line = o / p.nseg, idx = line % side, wk = line / side, kind = wk & 1, w = wk >> 1;
    const u32 e0 = seg * count;
    if (kind) base = (w << p.c) | (idx << p.half) | e0, step = 1;           // row idx: the low half runs
    else base = (w << p.c) | (e0 << p.half) | idx, step = side;              // column idx: the high half runs
  } else if (MODE == 1 || MODE == 3) {
    base = o * count;
  } else {
    const u32 s = o % p.nseg2, q = o / p.nseg2;
    j = q % p.half;
    base = (q / p.half) * side;  // (w * 2 + kind) * side
    first = s * count;
  }
  auto addr = [&](u32 k) -> size_t {
    if (MODE == 2) {
      const u32 m = first + k;  // the m-th index with bit j set
      return (size_t)(base + ((((m >> j) << 1) | 1u) << j | (m & ((1u << j) - 1u)))) * kNielsWords;
    }
    return (size_t)(base + k * step) * kNielsWords;
  };
  if (MODE == 0) {
    Ext acc = ext_identity();
    Niels cur = load_niels(in + addr(0));
#pragma unroll 1
    for (u32 k = 0; k < count; k++) {
      const Niels nxt = load_niels(in + addr(k + 1 < count ? k + 1 : k));
      acc = ext_add_niels(acc, cur);
      cur = nxt;
    }
    store_niels(out + (size_t)o * kNielsWords, ext_to_niels(acc));
    return;
  }
  Ext acc = ext_from_niels(load_niels(in + addr(lane)));
#pragma unroll 1
  for (u32 k = 1; k < per_lane; k++) acc = ext_add_niels(acc, load_niels(in + addr(k * lanes_per + lane)));
#pragma unroll 1
  for (u32 m = 1; m < lanes_per; m <<= 1) {
    const Niels mine = ext_to_niels(acc);
    Niels other;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      other.vpu.l[i] = (u32)__shfl_xor((int)mine.vpu.l[i], (int)m);
      other.vmu.l[i] = (u32)__shfl_xor((int)mine.vmu.l[i], (int)m);
      other.z.l[i] = (u32)__shfl_xor((int)mine.z.l[i], (int)m);
      other.t2d.l[i] = (u32)__shfl_xor((int)mine.t2d.l[i], (int)m);
    }
    acc = ext_add_niels(acc, other);
  }
  // (MODE 3: o = (w * 2 + kind) * half + j is w * c + kind * half + j, the bit's place in window w)
  if (live && lane == 0) store_niels(out + (size_t)o * kNielsWords, ext_to_niels(acc));
}

// ---- the subset sums: subgroup test and weights -----------------------------------------------------
// Both are ~250 dependent point operations per lane on a handful of lanes: pure latency (a wave
// issues one multiplication's 190 instructions in ~860 cycles whatever its neighbours do).  So FOUR
// waves share every point operation: a doubling and an addition are two rounds of at most four
// independent field multiplications each — wave k does the k-th of them for all 64 points of the
// workgroup and the results change hands through LDS (one barrier per round, two buffers).  The
// running point is (u, v, z, tt) with tt = t1 t2 (the product the next addition needs) computed by
// the wave that is idle in the second round; operand forms are exactly those of ext_double /
// ext_add_niels / ext_to_niels (jubjub29.h), so their proven bounds carry over.
namespace {
struct Xp {
  Fe u, v, z, tt;
};
struct Quad {  // the four waves' round results
  Fe r[4];
};
DSV_DEV Quad exchange(u32* sh, int& buf, int wave, int lane, const Fe& mine) {
  u32* b = sh + buf * (4 * NL * 64);
#pragma unroll
  for (int i = 0; i < NL; i++) b[(wave * NL + i) * 64 + lane] = mine.l[i];
  __syncthreads();
  Quad q;
#pragma unroll
  for (int w = 0; w < 4; w++)
#pragma unroll
    for (int i = 0; i < NL; i++) q.r[w].l[i] = b[(w * NL + i) * 64 + lane];
  buf ^= 1;  // the next round writes the other buffer: nobody is still reading it (one barrier back)
  return q;
}
// Which wave multiplies what is decided by UNIFORM BRANCHES on the wave number (a scalar: readfirstlane), each
// arm preparing only its own two operands; the multiplication itself exists once, behind the arms.  (r05
// computed every operand in every wave and selected limb by limb: 81 v_cndmask per doubling, 108 per
// addition, of 672 / 700 instructions — a tenth of a kernel that is one dependent chain.  The empty asm
// statements keep the compiler from folding the arms back into selects.)
#define DSV_ARM() asm volatile("" ::: "memory")
DSV_DEV Xp xp_finish(u32* sh, int& buf, int wave, int lane, const Fe& x, const Fe& y) {
  const Quad q = exchange(sh, buf, wave, lane, fe_mul(x, y));
  Xp r;
  r.u = q.r[0], r.v = q.r[1], r.z = q.r[2], r.tt = q.r[3];
  return r;
}
DSV_DEV Xp xp_double(u32* sh, int& buf, int wave, int lane, const Xp& p) {
  Fe x;
  if (wave == 0) { DSV_ARM(); x = p.u; }
  else if (wave == 1) { DSV_ARM(); x = p.v; }
  else if (wave == 2) { DSV_ARM(); x = p.z; }
  else { DSV_ARM(); x = fe_add(p.u, p.v); }
  const Quad q = exchange(sh, buf, wave, lane, fe_sqr(x));
  // ext_double: u = cu ct, v = vpu vmu, z = vmu ct; t1 t2 = cu vpu, with
  //   vpu = vv + uu, cu = (u + v)^2 - vpu (2uv, ext_two_uv), vmu = vv - uu, ct = 2 zz - vmu
  Fe a, b;
  if (wave == 0) {
    DSV_ARM();
    const Fe vpu = fe_add(q.r[1], q.r[0]);
    a = fe_sub4w(q.r[3], vpu);
    b = fe_sub4w(fe_dbl(q.r[2]), fe_sub2_raw(q.r[1], q.r[0]));
  } else if (wave == 1) {
    DSV_ARM();
    a = fe_add(q.r[1], q.r[0]);
    b = fe_sub2_raw(q.r[1], q.r[0]);
  } else if (wave == 2) {
    DSV_ARM();
    a = fe_sub2_raw(q.r[1], q.r[0]);
    b = fe_sub4w(fe_dbl(q.r[2]), a);
  } else {
    DSV_ARM();
    b = fe_add(q.r[1], q.r[0]);
    a = fe_sub4w(q.r[3], b);
  }
  return xp_finish(sh, buf, wave, lane, a, b);
}
DSV_DEV Xp xp_add(u32* sh, int& buf, int wave, int lane, const Xp& p, const Niels& n) {
  Fe x, y;
  if (wave == 0) { DSV_ARM(); x = fe_sub2_raw(p.v, p.u); y = n.vmu; }
  else if (wave == 1) { DSV_ARM(); x = fe_add(p.v, p.u); y = n.vpu; }
  else if (wave == 2) { DSV_ARM(); x = p.tt; y = n.t2d; }
  else { DSV_ARM(); x = p.z; y = n.z; }
  const Quad q = exchange(sh, buf, wave, lane, fe_mul(x, y));  // a, b, c, z nz
  // ext_add_tail: u = cu ct, v = cv cz, z = cz ct; t1 t2 = cu cv, with d = 2 z nz,
  //   cu = b - a, cv = b + a, cz = d + c, ct = d - c
  Fe a, b;
  if (wave == 0) {
    DSV_ARM();
    a = fe_sub2_raw(q.r[1], q.r[0]);
    b = fe_sub2(fe_dbl(q.r[3]), q.r[2]);
  } else if (wave == 1) {
    DSV_ARM();
    a = fe_add(q.r[1], q.r[0]);
    b = fe_add(fe_dbl(q.r[3]), q.r[2]);
  } else if (wave == 2) {
    DSV_ARM();
    const Fe d = fe_dbl(q.r[3]);
    a = fe_add(d, q.r[2]);
    b = fe_sub2(d, q.r[2]);
  } else {
    DSV_ARM();
    a = fe_sub2_raw(q.r[1], q.r[0]);
    b = fe_add(q.r[1], q.r[0]);
  }
  return xp_finish(sh, buf, wave, lane, a, b);
}
#undef DSV_ARM
DSV_DEV Xp xp_identity() {
  Xp r;
  r.u = fe_zero(), r.v = fe_one(), r.z = fe_one(), r.tt = fe_zero();
  return r;
}
}  // namespace

// 256 threads = 4 waves x 64 points.  Workgroups [0, g): flags |= kRlcTorsion unless r * S_l == O.
// Workgroups [g, 2g): W_l = 2^pos(l) * S_l, pos = bit position inside the scalar the window belongs to
// (every lane runs the workgroup's longest chain and keeps its own result once it is there).
// Workgroup 2g: W_lanes = (sum f_i) * G (+ (sum f'_i) * G') from the fixed-base tables, one lane.
// The LAST of the g + 1 workgroups that produce a W (a counter in the sub-group's flag words) also adds
// them all up and tests the sum for the identity — r05 had a kernel of its own for that, 0.07 ms behind
// this one; here it runs in the shadow of the subgroup test's longer chains (337 point operations against
// at most 252).  The last of ALL workgroups to finish marks the chain complete.
namespace {
// sum of `count` stored points == O ?  All four waves of the workgroup, the cooperative operations above:
// 64 strided partial sums, then a tree over the 64 lanes through LDS (tree: 64 x kNielsWords words).
DSV_DEV bool xp_sum_is_identity(const u32* __restrict__ W, u32 count, u32* sh, u32* tree, int wave, int lane) {
  int buf = 0;
  Xp acc = xp_identity();
#pragma unroll 1
  for (u32 k = 0; k < count; k += 64) {
    const Niels n = k + lane < count ? load_niels(W + (size_t)(k + lane) * kNielsWords) : niels_identity();
    acc = xp_add(sh, buf, wave, lane, acc, n);
  }
#pragma unroll 1
  for (u32 step = 32; step > 0; step >>= 1) {
    // every wave holds the same acc: each computes the niels form, wave 0 hands it to the other lanes
    Niels mine;
    mine.vpu = fe_carry(fe_add(acc.v, acc.u));
    mine.vmu = fe_sub2(acc.v, acc.u);
    mine.z = acc.z;
    mine.t2d = fe_mul(acc.tt, fe_const(kD2));
    __syncthreads();  // (the previous level's reads of `tree` are done)
    if (wave == 0 && (u32)lane >= step && (u32)lane < 2 * step) store_niels(tree + (size_t)lane * kNielsWords, mine);
    __syncthreads();
    const Niels other = (u32)lane < step ? load_niels(tree + (size_t)(lane + step) * kNielsWords) : niels_identity();
    acc = xp_add(sh, buf, wave, lane, acc, other);
  }
  // lane 0 holds the total (in every wave)
  const bool identity = (bool)((int)fe_equal(acc.u, fe_zero()) & (int)fe_equal(acc.v, acc.z));
  return __shfl((int)identity, 0) != 0;
}
}  // namespace
__global__ void __launch_bounds__(256)
k_rlc_scale(const u32* __restrict__ S, size_t S_stride, const u32* __restrict__ fsum, const u32* __restrict__ tableG,
            const u32* __restrict__ tableG2, RlcPlan p, u32* __restrict__ W, size_t W_stride, u32* __restrict__ gflags) {
  if (gflags[0]) return;  // (uniform: no barrier is left waiting)
  __shared__ u32 sh[2 * 4 * NL * 64];
  __shared__ __attribute__((aligned(16))) u32 tree[64 * kNielsWords];
  __shared__ u32 sh_max, sh_last;
  S += (size_t)blockIdx.y * S_stride * kNielsWords;
  W += (size_t)blockIdx.y * W_stride * kNielsWords;
  fsum += (size_t)blockIdx.y * 16;
  u32* flags = gflags + 4 + 4 * blockIdx.y;
  const u32 lanes = (u32)p.windows * p.c, g = (lanes + 63) / 64;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const bool weigh = blockIdx.x >= g && blockIdx.x < 2 * g, fixed_base = blockIdx.x == 2 * g;
  if (fixed_base) {
    if (threadIdx.x == 0) {
      Ext fg = ext_identity();
#pragma unroll 1
      for (int k = 0; k < p.fixed; k++) {
        u32 f[8];
        load_words8(f, reinterpret_cast<const uint8_t*>(fsum), k);
        fg = fixed_base_accumulate(fg, f, k ? tableG2 : tableG);
      }
      store_niels(W + (size_t)lanes * kNielsWords, ext_to_niels(fg));
    }
  } else {
    const u32 l = (blockIdx.x - (weigh ? g : 0)) * 64 + lane;
    const bool live = l < lanes;  // the others keep step with the barriers on the identity
    const Niels s = live ? load_niels(S + (size_t)l * kNielsWords) : niels_identity();
    int buf = 0;
    Xp acc = xp_add(sh, buf, wave, lane, xp_identity(), s);
    if (weigh) {
      const u32 w = l / p.c, bit = l % p.c;
      const u32 pos = live ? (w < (u32)p.wpk ? w : w - p.wpk) * p.c + bit : 0u;
      if (threadIdx.x == 0) sh_max = 0;
      __syncthreads();
      atomicMax(&sh_max, pos);
      __syncthreads();
      const u32 longest = sh_max;
#pragma unroll 1
      for (u32 k = 0; k < longest; k++) {
        const Xp d = xp_double(sh, buf, wave, lane, acc);
        const bool take = k < pos;
        acc.u = fe_select(take, d.u, acc.u), acc.v = fe_select(take, d.v, acc.v);
        acc.z = fe_select(take, d.z, acc.z), acc.tt = fe_select(take, d.tt, acc.tt);
      }
      if (wave == 0 && live) {
        Niels n;
        n.vpu = fe_carry(fe_add(acc.v, acc.u));
        n.vmu = fe_sub2(acc.v, acc.u);
        n.z = acc.z;
        n.t2d = fe_mul(acc.tt, fe_const(kD2));
        store_niels(W + (size_t)l * kNielsWords, n);
      }
    } else {
      const Niels sn = niels_neg(s);
#pragma unroll 1
      for (int k = 251; k >= 0; k--) {  // digit 252 is the +1 acc starts from
        acc = xp_double(sh, buf, wave, lane, acc);
        const u32 pb = (kRNafPos[k >> 5] >> (k & 31)) & 1u, nb = (kRNafNeg[k >> 5] >> (k & 31)) & 1u;
        if (pb) acc = xp_add(sh, buf, wave, lane, acc, s);   // (uniform: the digits are constants)
        if (nb) acc = xp_add(sh, buf, wave, lane, acc, sn);
      }
      if (wave == 0 && live) {
        const bool identity = (bool)((int)fe_equal(acc.u, fe_zero()) & (int)fe_equal(acc.v, acc.z));
        if (!identity) atomicOr(&flags[0], kRlcTorsion);
      }
    }
  }
  if (weigh || fixed_base) {
    // this workgroup's W are written: the last producer adds them all up
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) sh_last = atomicAdd(&flags[2], 1u) == g ? 1u : 0u;
    __syncthreads();
    if (sh_last) {  // (uniform)
      __threadfence();
      const bool identity = xp_sum_is_identity(W, lanes + 1u, sh, tree, wave, lane);
      if (threadIdx.x == 0 && !identity) atomicOr(&flags[0], kRlcSum);
    }
  }
  // the last workgroup to get here marks the chain of kernels complete (flags[0] is final by then: every
  // workgroup's atomics on it come before its own count)
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&flags[3], 1u) == 2 * g) {
    __threadfence();
    flags[1] = 1;
  }
}

// ---- the sample's and the call's verdicts ----------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rlc_sample_decide(const uint8_t* __restrict__ sample_ok, const uint8_t* __restrict__ valid, const uint8_t* __restrict__ u,
                    const uint8_t* __restrict__ pk0, const uint8_t* __restrict__ pk1, size_t first, size_t count,
                    u32* __restrict__ flags) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= count) return;
  if (sample_ok[k] != 0) return;
  const size_t i = first + k;
  bool wellformed = valid[i] != 0;
  u32 w[8];
  load_words8(w, u, i);
  wellformed &= words_lt(w, kR32);
  for (int h = 0; h < 2; h++) {
    const uint8_t* pk = h ? pk1 : pk0;
    if (!pk) continue;
    load_words8(w, pk, 2 * i);
    wellformed &= words_lt(w, kQ32);
    load_words8(w, pk, 2 * i + 1);
    wellformed &= words_lt(w, kQ32);
  }
  if (wellformed) flags[0] = 1;  // well-formed and still verdict 0: a wrong signature
}
__global__ void k_rlc_verdict(const u32* __restrict__ flags, RlcVerdictArgs a, u32* __restrict__ accepted,
                              u32* __restrict__ history) {
  if (threadIdx.x || blockIdx.x) return;
  bool all = true, rejected = false, any = false;
  for (u32 k = 0; k < a.ngroups; k++) {
    const u32* f = flags + ((size_t)2 * k + a.second[k]) * kRlcGroupFlagWords;
    for (u32 g = 0; g < a.subs[k]; g++) {
      const bool acc = f[4 + 4 * g] == 0u && f[4 + 4 * g + 1] == 1u;
      rejected |= !acc;
      any = true;
    }
    if (a.subs[k] == 0) all = false;  // a group that took the per-signature path as it is
  }
  all &= !rejected;
  if (accepted) *accepted = (all && (!a.and_into || *accepted != 0u)) ? 1u : 0u;
  if (history) {
    const u32 h = history[0];
    const u32 h2 = history[2];
    if (rejected) history[0] = 8u, history[2] = 128u;
    else if (any) history[0] = h > 0 ? h - 1 : 0u, history[2] = h2 > 0 ? h2 - 1 : 0u;
    history[1] += 1u;
  }
}

__global__ void k_rlc_chain(const u32* __restrict__ first, u32* __restrict__ second, u32 subs) {
  if (threadIdx.x || blockIdx.x) return;
  if (first[0] == 0u && first[4] == 0u && first[5] == 1u) {  // (no sample in a guarded call; one sub-group in the first stage)
    second[0] = 1u;
    for (u32 g = 0; g < subs; g++) second[4 + 4 * g + 1] = 1u;
  }
}

// ---- host side ----------------------------------------------------------------------------------------
void launch_rlc_chain(const uint32_t* first, uint32_t* second, uint32_t subs, hipStream_t s) {
  hipLaunchKernelGGL(k_rlc_chain, dim3(1), dim3(64), 0, s, first, second, subs);
}
void launch_rlc_sample_decide(const uint8_t* sample_ok, const uint8_t* valid, const uint8_t* u, const uint8_t* pk0,
                              const uint8_t* pk1, size_t first, size_t count, uint32_t* flags, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(k_rlc_sample_decide, dim3(grid_for(count)), dim3(256), 0, s, sample_ok, valid, u, pk0, pk1, first,
                     count, flags);
}
void launch_rlc_verdict(const uint32_t* flags, RlcVerdictArgs a, uint32_t* accepted, uint32_t* history, hipStream_t s) {
  hipLaunchKernelGGL(k_rlc_verdict, dim3(1), dim3(64), 0, s, flags, a, accepted, history);
}

hipError_t launch_rlc_begin(const RlcBuffers& b, hipStream_t s) {
  return hipMemsetAsync(b.flags, 0, kRlcGroupFlagWords * sizeof(uint32_t), s);
}

hipError_t launch_rlc_buckets(int scheme, const RlcPlan& p, const RlcBuffers& b, const RlcInputs& in, ChaChaKey key,
                              uint8_t* ok, bool second, hipStream_t s) {
  const unsigned G = p.groups;
  // the sort's counters: bin fills, run-length histogram, placement cursors
  hipError_t err = hipMemsetAsync(b.counters, 0, (size_t)G * b.counters_stride * sizeof(uint32_t), s);
  if (err != hipSuccess) return err;
  const dim3 grid(grid_for(p.n), G), block(256);
  if (scheme == 0) hipLaunchKernelGGL(k_rlc_prep<0>, grid, block, 0, s, in, key, p, b, ok);
  else if (scheme == 1) hipLaunchKernelGGL(k_rlc_prep<1>, grid, block, 0, s, in, key, p, b, ok);
  else hipLaunchKernelGGL(k_rlc_prep<2>, grid, block, 0, s, in, key, p, b, ok);
  const unsigned tiles = (p.row_stride + kRlcTile - 1) / kRlcTile;
  hipLaunchKernelGGL(k_rlc_part1, dim3(p.rows * tiles, G), dim3(256), 0, s, p, b);
  hipLaunchKernelGGL(k_rlc_part2, dim3(p.bins, G), dim3(256), 0, s, p, b);
  const unsigned lgrid = grid_for(p.buckets, 256 * kLenPerThread);
  hipLaunchKernelGGL(k_rlc_lenhist, dim3(lgrid, G), dim3(256), 0, s, p, b);
  hipLaunchKernelGGL(k_rlc_order, dim3(lgrid, G), dim3(256), 0, s, p, b);
  hipLaunchKernelGGL(k_rlc_accumulate, dim3(grid_for(p.buckets, 64), G), dim3(64), 0, s, p, b, second);
  return hipGetLastError();
}

hipError_t launch_rlc_finish(const RlcPlan& p, const RlcBuffers& b, const uint32_t* tableG, const uint32_t* tableG2,
                             bool merged, hipStream_t s) {
  const unsigned G = p.groups;
  if (merged) hipLaunchKernelGGL(k_rlc_merge, dim3(grid_for(p.buckets, 64), G), dim3(64), 0, s, p, b);
  const unsigned side = 1u << p.half;
  // outputs x lanes per output (k_rlc_sum), + the workgroups that sum the fixed-base scalars beside them
  const unsigned g0 = grid_for((size_t)p.windows * 2 * side * p.nseg, 64);
  auto per_out = [](unsigned count) { return count < 4u ? count : 4u; };  // lanes per output of k_rlc_sum<1..3>
  const unsigned g1 = grid_for((size_t)p.windows * 2 * side * per_out(p.nseg), 64);
  const unsigned g2 = grid_for((size_t)p.windows * 2 * p.half * p.nseg2 * per_out(side / 2 / p.nseg2), 64);
  const unsigned g3 = grid_for((size_t)p.windows * 2 * p.half * per_out(p.nseg2), 64);
  hipLaunchKernelGGL(k_rlc_sum<0>, dim3(g0 + kRlcFsumBlocks * p.fixed, G), dim3(64), 0, s, b.buckets, b.bucket_stride, p,
                     b.tmp[0], b.tmp_stride[0], b, g0);
  hipLaunchKernelGGL(k_rlc_sum<1>, dim3(g1 + p.fixed, G), dim3(64), 0, s, b.tmp[0], b.tmp_stride[0], p, b.tmp[1],
                     b.tmp_stride[1], b, g1);
  hipLaunchKernelGGL(k_rlc_sum<2>, dim3(g2, G), dim3(64), 0, s, b.tmp[1], b.tmp_stride[1], p, b.tmp[0], b.tmp_stride[0], b, g2);
  hipLaunchKernelGGL(k_rlc_sum<3>, dim3(g3, G), dim3(64), 0, s, b.tmp[0], b.tmp_stride[0], p, b.tmp[1], b.tmp_stride[1], b, g3);
  const unsigned lanes = (unsigned)p.windows * p.c, g = (lanes + 63) / 64;
  hipLaunchKernelGGL(k_rlc_scale, dim3(2 * g + 1, G), dim3(256), 0, s, b.tmp[1], b.tmp_stride[1], b.fsum, tableG, tableG2, p,
                     b.tmp[0], b.tmp_stride[0], b.flags);
  return hipGetLastError();
}

hipError_t launch_rlc(int scheme, const RlcPlan& p, const RlcBuffers& b, const RlcInputs& in, ChaChaKey key,
                      const uint32_t* tableG, const uint32_t* tableG2, uint8_t* ok, hipStream_t s) {
  hipError_t err = launch_rlc_begin(b, s);
  if (err != hipSuccess) return err;
  err = launch_rlc_buckets(scheme, p, b, in, key, ok, false, s);
  if (err != hipSuccess) return err;
  return launch_rlc_finish(p, b, tableG, tableG2, false, s);
}

}  // namespace dsv
