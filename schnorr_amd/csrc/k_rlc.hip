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
// Kernels (launch order; every stage reads what the previous one wrote, same stream):
//   k_rlc_prep        per item: z_i = ChaCha12(key, i), e_i = z_i c_i, f_i = z_i u_i (mod r), range and
//                     curve checks, the points as affine niels (PK_i, -R_i), (bucket, index) pairs
//   k_rlc_fsum[2]     sum f_i mod r
//   (hipcub radix sort of the pairs by bucket)
//   k_rlc_starts      where each bucket's run of the sorted pairs begins
//   k_rlc_counts      run lengths; (hipcub sort of the bucket numbers by run length)
//   k_rlc_accumulate  one lane per bucket: mixed additions over its run
//   k_rlc_sum<0..3>   row / column sums, then the per-bit subset sums S_p        (short chains)
//   k_rlc_scale       lanes A: r * S_p == O ?    lanes B: 2^p * S_p    one more lane: (sum f_i) * G
//   k_rlc_final       sum of all of lanes B's and that lane's results, identity test -> flags
#include <hipcub/hipcub.hpp>

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
DSV_DEV bool ext_is_identity(const Ext& p) {  // u == 0 and v == z (z != 0 on the curve: complete formulas)
  return (bool)((int)fe_equal(p.u, fe_zero()) & (int)fe_equal(p.v, p.z));
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
struct PrepOut {
  size_t i;   // the item's place in its group (inputs, points, weights)
  size_t il;  // ... and in the range this pass covers (the pair arrays)
  const RlcPlan& p;
  u32* pts;
  u32* keys;
  u32* vals;
};
// loads one point, folds its range check into `good`, returns "is on the curve", stores it as affine niels
DSV_DEV bool prep_point(const PrepOut& o, const uint8_t* __restrict__ uv, int slot, bool negate, bool& good) {
  Fe pu, pv;
  good &= load_fq(pu, uv, 2 * o.i);
  good &= load_fq(pv, uv, 2 * o.i + 1);
  store_pt(o.pts + ((size_t)slot * o.p.total + o.i) * kPtWords, affine_niels(pu, pv, negate));
  return on_curve(pu, pv);
}
// e' = e + k r, k uniform below floor(2^(wpk c) / r): the same multiple of a point of the prime-order
// subgroup (any other point fails the subgroup test anyway), but uniform over ALL wpk * c bits —
// without it the top window of a 252-bit scalar has a few thousand (c = 16: 2^12) digits only, and
// its buckets get runs 16 times as long as the others: a lane per bucket would wait for those.
// Then one (bucket, point) pair per window.
DSV_DEV void emit_long(const PrepOut& o, u32 (&e)[8], u32 kr, int slot) {
  const RlcPlan& p = o.p;
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const u64 t = (u64)kR32[k] * kr + e[k] + carry;
    e[k] = (u32)t;
    carry = t >> 32;
  }
  const u32 mask = (1u << p.c) - 1u, none = (u32)p.windows << p.c;
#pragma unroll 1
  for (int w = 0; w < p.wpk; w++) {
    const u32 d = e[0] & mask;
#pragma unroll
    for (int k = 0; k < 7; k++) e[k] = __funnelshift_r(e[k], e[k + 1], p.c);
    e[7] >>= p.c;
    const size_t at = ((size_t)w * p.lpts + slot) * p.n + o.il;
    o.keys[at] = d ? (((u32)w << p.c) | d) : none;
    o.vals[at] = (u32)((size_t)slot * p.total + o.i);
  }
}
DSV_DEV void emit_short(const PrepOut& o, const u32 (&zz)[8], int slot) {
  const RlcPlan& p = o.p;
  u32 z[5] = {zz[0], zz[1], zz[2], zz[3], zz[4]};
  const u32 mask = (1u << p.c) - 1u, none = (u32)p.windows << p.c;
  const size_t first = (size_t)p.wpk * p.lpts * p.n;
#pragma unroll 1
  for (int w = 0; w < p.wr; w++) {
    const u32 d = z[0] & mask;
#pragma unroll
    for (int k = 0; k < 4; k++) z[k] = __funnelshift_r(z[k], z[k + 1], p.c);
    z[4] >>= p.c;
    const size_t at = first + ((size_t)w * p.spts + slot) * p.n + o.il;
    o.keys[at] = d ? (((u32)(p.wpk + w) << p.c) | d) : none;
    o.vals[at] = (u32)((size_t)(p.lpts + slot) * p.total + o.i);
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
k_rlc_prep(RlcInputs in, ChaChaKey key, RlcPlan p, uint8_t* __restrict__ ok, u32* __restrict__ pts,
           u32* __restrict__ fsc, u32* __restrict__ keys, u32* __restrict__ vals, u32* __restrict__ flags) {
  const size_t il = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (il >= p.n) return;
  const size_t i = p.first + il;
  const PrepOut o{i, il, p, pts, keys, vals};
  bool good = in.valid[i] != 0;
  u32 us[8], cs[8];
  load_words8(us, in.u, i);
  load_words8(cs, in.c, i);
  good &= words_lt(us, kR32);
  // slots: long points first (PK, then PK' / Gen), then the short ones (-R, -R')
  bool curve = prep_point(o, in.pk[0], 0, false, good);
  if (SCHEME == 1) curve &= prep_point(o, in.pk[1], 1, false, good);
  if (SCHEME == 2) curve &= prep_point(o, in.gen, 1, false, good);
  curve &= prep_point(o, in.r[0], p.lpts, true, good);
  if (SCHEME == 1) curve &= prep_point(o, in.r[1], p.lpts + 1, true, good);
  // a point off the curve has no place in a group sum: the per-signature kernels decide the batch
  if (good && !curve) atomicOr(&flags[0], kRlcOffCurve);
  ok[i] = good ? 1 : 0;
  u32 blk[16];
  chacha12_block(blk, key.w, (u64)i);
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
    else store_words8(reinterpret_cast<uint8_t*>(fsc), (size_t)eq * p.total + i, e);
    emit_short(o, z, eq);
  }
}

// sum of n scalars mod r: stage 0 -> kRlcFsumBlocks partial sums, stage 1 (one workgroup) -> out
__global__ void __launch_bounds__(256) k_rlc_fsum(const u32* __restrict__ in, size_t n, u32* __restrict__ out) {
  __shared__ u32 sh[256][8];
  u32 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    u32 x[8], t[8];
    load_words8(x, reinterpret_cast<const uint8_t*>(in), i);
    fr_add(t, acc, x);
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = t[k];
  }
#pragma unroll
  for (int k = 0; k < 8; k++) sh[threadIdx.x][k] = acc[k];
  __syncthreads();
  for (int step = 128; step > 0; step >>= 1) {
    if ((int)threadIdx.x < step) {
      u32 a[8], b[8], t[8];
#pragma unroll
      for (int k = 0; k < 8; k++) a[k] = sh[threadIdx.x][k], b[k] = sh[threadIdx.x + step][k];
      fr_add(t, a, b);
#pragma unroll
      for (int k = 0; k < 8; k++) sh[threadIdx.x][k] = t[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    u32 t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = sh[0][k];
    store_words8(reinterpret_cast<uint8_t*>(out), blockIdx.x, t);
  }
}

// ---- buckets ----------------------------------------------------------------------------------
// start[b] = first sorted entry with key >= b, for b = 0 .. buckets (entries with digit 0 carry the key
// `buckets` and sort behind everything): every entry fills in the keys between its predecessor's and
// its own, so empty buckets get their (empty) range too
__global__ void __launch_bounds__(256)
k_rlc_starts(const u32* __restrict__ keys, RlcPlan p, u32* __restrict__ start) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j > p.entries) return;
  const u32 last = (u32)p.buckets;
  u32 lo = j ? keys[j - 1] + 1u : 0u, hi = j < p.entries ? keys[j] : last;
  if (hi > last) hi = last;  // (cannot happen)
  for (u32 k = lo; k <= hi && j < p.entries; k++) start[k] = (u32)j;
  if (j == p.entries)
    for (u32 k = lo; k <= last; k++) start[k] = (u32)j;
}
// run lengths (clipped to 8 bits) and the identity permutation: sorted by length, longest first, they
// give every wave of the accumulation 64 runs of (nearly) the same length — with buckets in their
// natural order a wave waits for its longest run (Poisson, mean 16: the maximum of 64 is ~26)
__global__ void __launch_bounds__(256)
k_rlc_counts(const u32* __restrict__ start, RlcPlan p, u32* __restrict__ cnt, u32* __restrict__ ids) {
  const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= p.buckets) return;
  u32 len = (b & ((1u << p.c) - 1u)) ? start[b + 1] - start[b] : 0u;  // digit 0 enters no sum
  cnt[b] = len < 255u ? len : 255u;
  ids[b] = (u32)b;
}
// one lane per bucket (in the order of `order`): its run of the sorted pairs, one mixed addition per
// entry; the next entry's point is loaded while the current one is added
__global__ void __launch_bounds__(64)
k_rlc_accumulate(const u32* __restrict__ order, const u32* __restrict__ start, const u32* __restrict__ vals,
                 const u32* __restrict__ pts, RlcPlan p, u32* __restrict__ buckets) {
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= p.buckets) return;
  const u32 b = order[t];
  if (b >= p.buckets) return;  // (cannot happen: the sort moves what k_rlc_counts wrote)
  u32 lo = 0, hi = 0;
  if (b & ((1u << p.c) - 1u)) lo = start[b], hi = start[b + 1];
  // (a sort that failed leaves anything in `start`: never read outside the pair arrays; the aggregate
  //  then simply does not come out as the identity)
  if (hi > p.entries) hi = (u32)p.entries;
  if (lo > hi) lo = hi;
  const u32 last_pt = (u32)(p.lpts + p.spts) * p.total - 1u;
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
  store_niels(buckets + (size_t)b * kNielsWords, ext_to_niels(acc));
}

// buckets[b] += buckets2[b]: the two ranges of a staged bucket pass
__global__ void __launch_bounds__(64)
k_rlc_merge(u32* __restrict__ buckets, const u32* __restrict__ buckets2, RlcPlan p) {
  const size_t b = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= p.buckets) return;
  Ext acc = ext_add_niels(ext_identity(), load_niels(buckets + b * kNielsWords));
  acc = ext_add_niels(acc, load_niels(buckets2 + b * kNielsWords));
  store_niels(buckets + b * kNielsWords, ext_to_niels(acc));
}

// ---- short sums of stored points: out[o] = sum_k in[addr(o, k)] ------------------------------------
//   MODE 0  buckets -> segments of the row (kind 1) / column (kind 0) sums of each window's bucket matrix
//   MODE 1  segments -> row / column sums ("lines")
//   MODE 2  lines -> segments of: sum of the lines whose index has bit j set
//   MODE 3  segments -> S[w * c + kind * half + j], the subset sum of bit (kind * half + j) of window w
template <int MODE>
__global__ void __launch_bounds__(64)
k_rlc_sum(const u32* __restrict__ in, RlcPlan p, u32* __restrict__ out) {
  const u32 o = blockIdx.x * 64 + threadIdx.x;
  const u32 side = 1u << p.half;
  u32 total, count;
  if (MODE == 0) total = (u32)p.windows * 2 * side * p.nseg, count = side / p.nseg;
  else if (MODE == 1) total = (u32)p.windows * 2 * side, count = p.nseg;
  else if (MODE == 2) total = (u32)p.windows * 2 * p.half * p.nseg2, count = side / 2 / p.nseg2;
  else total = (u32)p.windows * 2 * p.half, count = p.nseg2;
  if (o >= total) return;
  u32 base = 0, step = 1, j = 0, first = 0;
  if (MODE == 0) {
    const u32 seg = o % p.nseg, line = o / p.nseg, idx = line % side, wk = line / side, kind = wk & 1, w = wk >> 1;
    const u32 e0 = seg * count;
    if (kind) base = (w << p.c) | (idx << p.half) | e0, step = 1;           // row idx: the low half runs
    else base = (w << p.c) | (e0 << p.half) | idx, step = side;              // column idx: the high half runs
  } else if (MODE == 1 || MODE == 3) {
    base = o * count;
  } else {
    const u32 s = o % p.nseg2, t = o / p.nseg2;
    j = t % p.half;
    base = (t / p.half) * side;  // (w * 2 + kind) * side
    first = s * count;
  }
  auto addr = [&](u32 k) -> size_t {
    if (MODE == 2) {
      const u32 m = first + k;  // the m-th index with bit j set
      return (size_t)(base + ((((m >> j) << 1) | 1u) << j | (m & ((1u << j) - 1u)))) * kNielsWords;
    }
    return (size_t)(base + k * step) * kNielsWords;
  };
  Ext acc = ext_identity();
  Niels cur = load_niels(in + addr(0));
#pragma unroll 1
  for (u32 k = 0; k < count; k++) {
    const Niels nxt = load_niels(in + addr(k + 1 < count ? k + 1 : k));
    acc = ext_add_niels(acc, cur);
    cur = nxt;
  }
  // (MODE 3: o = (w * 2 + kind) * half + j is w * c + kind * half + j, the bit's place in window w)
  store_niels(out + (size_t)o * kNielsWords, ext_to_niels(acc));
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
// (limb by limb: a select between whole structs becomes an array in scratch memory indexed by `wave`)
DSV_DEV Fe pick(int wave, const Fe& a, const Fe& b, const Fe& c, const Fe& d) {
  return fe_select(wave < 2, fe_select(wave == 0, a, b), fe_select(wave == 2, c, d));
}
DSV_DEV Xp xp_finish(u32* sh, int& buf, int wave, int lane, const Fe& cu, const Fe& ct, const Fe& cv, const Fe& cz,
                     const Fe& zl, const Fe& t2) {
  // u = cu ct, v = cv cz, z = zl ct, tt = cu t2
  const Fe x = pick(wave, cu, cv, zl, cu), y = pick(wave, ct, cz, ct, t2);
  const Quad q = exchange(sh, buf, wave, lane, fe_mul(x, y));
  Xp r;
  r.u = q.r[0], r.v = q.r[1], r.z = q.r[2], r.tt = q.r[3];
  return r;
}
DSV_DEV Xp xp_double(u32* sh, int& buf, int wave, int lane, const Xp& p) {
  const Quad q = exchange(sh, buf, wave, lane, fe_sqr(pick(wave, p.u, p.v, p.z, fe_add(p.u, p.v))));
  const Fe zz2 = fe_dbl(q.r[2]);
  const Fe vpu = fe_add(q.r[1], q.r[0]);
  const Fe cu = fe_sub4w(q.r[3], vpu);       // 2uv (ext_two_uv)
  const Fe vmu = fe_sub2_raw(q.r[1], q.r[0]);
  const Fe ct = fe_sub4w(zz2, vmu);
  return xp_finish(sh, buf, wave, lane, cu, ct, vpu, vmu, vmu, vpu);  // (ext_double: u = cu ct, v = vpu vmu, z = vmu ct; t1 t2 = cu vpu)
}
DSV_DEV Xp xp_add(u32* sh, int& buf, int wave, int lane, const Xp& p, const Niels& n) {
  const Fe x = pick(wave, fe_sub2_raw(p.v, p.u), fe_add(p.v, p.u), p.tt, p.z);
  const Fe y = pick(wave, n.vmu, n.vpu, n.t2d, n.z);
  const Quad q = exchange(sh, buf, wave, lane, fe_mul(x, y));  // a, b, c, z nz
  const Fe d = fe_dbl(q.r[3]);
  const Fe cu = fe_sub2_raw(q.r[1], q.r[0]), cv = fe_add(q.r[1], q.r[0]);
  const Fe cz = fe_add(d, q.r[2]), ct = fe_sub2(d, q.r[2]);
  return xp_finish(sh, buf, wave, lane, cu, ct, cv, cz, cz, cv);  // (ext_add_tail: u = cu ct, v = cv cz, z = cz ct; t1 t2 = cu cv)
}
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
__global__ void __launch_bounds__(256)
k_rlc_scale(const u32* __restrict__ S, const u32* __restrict__ fsum, const u32* __restrict__ tableG,
            const u32* __restrict__ tableG2, RlcPlan p, u32* __restrict__ W, u32* __restrict__ flags) {
  __shared__ u32 sh[2 * 4 * NL * 64];
  __shared__ u32 sh_max;
  const u32 lanes = (u32)p.windows * p.c, g = (lanes + 63) / 64;
  if (blockIdx.x == 2 * g) {  // (no barrier on this path)
    if (threadIdx.x) return;
    Ext fg = ext_identity();
#pragma unroll 1
    for (int k = 0; k < p.fixed; k++) {
      u32 f[8];
      load_words8(f, reinterpret_cast<const uint8_t*>(fsum), k);
      fg = fixed_base_accumulate(fg, f, k ? tableG2 : tableG);
    }
    store_niels(W + (size_t)lanes * kNielsWords, ext_to_niels(fg));
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool weigh = blockIdx.x >= g;
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
    return;
  }
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

// sum of the windows * c weighted subset sums and of (sum f_i) * G == O ?  One wave: a strided pass,
// then a tree through LDS.
__global__ void __launch_bounds__(64)
k_rlc_final(const u32* __restrict__ W, RlcPlan p, u32* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) u32 sh[64 * kNielsWords];
  const u32 count = (u32)p.windows * p.c + 1u, t = threadIdx.x;
  Ext acc = ext_identity();
#pragma unroll 1
  for (u32 k = t; k < count; k += 64) acc = ext_add_niels(acc, load_niels(W + (size_t)k * kNielsWords));
#pragma unroll 1
  for (u32 step = 32; step > 0; step >>= 1) {
    if (t >= step && t < 2 * step) store_niels(sh + t * kNielsWords, ext_to_niels(acc));
    __syncthreads();
    if (t < step) acc = ext_add_niels(acc, load_niels(sh + (t + step) * kNielsWords));
    __syncthreads();
  }
  if (t == 0) {
    if (!ext_is_identity(acc)) atomicOr(&flags[0], kRlcSum);
    flags[1] = 1;  // the chain of kernels ran to its end
  }
}

// ---- host side ----------------------------------------------------------------------------------------
size_t rlc_sort_temp_bytes(const RlcPlan& p) {
  size_t bytes = 0;
  const u32* k = nullptr;
  u32* ko = nullptr;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, ko, k, ko, p.entries, 0, p.key_bits, (hipStream_t) nullptr);
  size_t bytes2 = 0;
  (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes2, k, ko, k, ko, p.buckets, 0, 8, (hipStream_t) nullptr);
  return bytes > bytes2 ? bytes : bytes2;
}

hipError_t launch_rlc_begin(const RlcBuffers& b, hipStream_t s) { return hipMemsetAsync(b.flags, 0, 16, s); }

hipError_t launch_rlc_buckets(int scheme, const RlcPlan& p, const RlcBuffers& b, const RlcInputs& in, ChaChaKey key,
                              uint8_t* ok, bool second, hipStream_t s) {
  const dim3 grid(grid_for(p.n)), block(256);
  if (scheme == 0)
    hipLaunchKernelGGL(k_rlc_prep<0>, grid, block, 0, s, in, key, p, ok, b.pts, b.fsc, b.keys[0], b.vals[0], b.flags);
  else if (scheme == 1)
    hipLaunchKernelGGL(k_rlc_prep<1>, grid, block, 0, s, in, key, p, ok, b.pts, b.fsc, b.keys[0], b.vals[0], b.flags);
  else
    hipLaunchKernelGGL(k_rlc_prep<2>, grid, block, 0, s, in, key, p, ok, b.pts, b.fsc, b.keys[0], b.vals[0], b.flags);
  size_t temp = b.sort_temp_bytes;
  hipError_t err = hipcub::DeviceRadixSort::SortPairs(b.sort_temp, temp, b.keys[0], b.keys[1], b.vals[0], b.vals[1],
                                                      p.entries, 0, p.key_bits, s);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(k_rlc_starts, dim3(grid_for(p.entries + 1)), dim3(256), 0, s, b.keys[1], p, b.start);
  hipLaunchKernelGGL(k_rlc_counts, dim3(grid_for(p.buckets)), dim3(256), 0, s, b.start, p, b.cnt[0], b.order[0]);
  temp = b.sort_temp_bytes;
  err = hipcub::DeviceRadixSort::SortPairsDescending(b.sort_temp, temp, b.cnt[0], b.cnt[1], b.order[0], b.order[1],
                                                     p.buckets, 0, 8, s);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(k_rlc_accumulate, dim3(grid_for(p.buckets, 64)), dim3(64), 0, s, b.order[1], b.start, b.vals[1],
                     b.pts, p, second ? b.buckets2 : b.buckets);
  return hipGetLastError();
}

hipError_t launch_rlc_finish(const RlcPlan& p, const RlcBuffers& b, const uint32_t* tableG, const uint32_t* tableG2,
                             bool merged, hipStream_t s) {
  if (merged) hipLaunchKernelGGL(k_rlc_merge, dim3(grid_for(p.buckets, 64)), dim3(64), 0, s, b.buckets, b.buckets2, p);
  for (int k = 0; k < p.fixed; k++) {
    hipLaunchKernelGGL(k_rlc_fsum, dim3(kRlcFsumBlocks), dim3(256), 0, s, b.fsc + (size_t)k * p.total * 8, (size_t)p.total, b.fpart);
    hipLaunchKernelGGL(k_rlc_fsum, dim3(1), dim3(256), 0, s, b.fpart, (size_t)kRlcFsumBlocks, b.fsum + 8 * k);
  }
  const unsigned side = 1u << p.half;
  hipLaunchKernelGGL(k_rlc_sum<0>, dim3(grid_for((size_t)p.windows * 2 * side * p.nseg, 64)), dim3(64), 0, s, b.buckets, p, b.tmp[0]);
  hipLaunchKernelGGL(k_rlc_sum<1>, dim3(grid_for((size_t)p.windows * 2 * side, 64)), dim3(64), 0, s, b.tmp[0], p, b.tmp[1]);
  hipLaunchKernelGGL(k_rlc_sum<2>, dim3(grid_for((size_t)p.windows * 2 * p.half * p.nseg2, 64)), dim3(64), 0, s, b.tmp[1], p, b.tmp[0]);
  hipLaunchKernelGGL(k_rlc_sum<3>, dim3(grid_for((size_t)p.windows * 2 * p.half, 64)), dim3(64), 0, s, b.tmp[0], p, b.tmp[1]);
  const unsigned lanes = (unsigned)p.windows * p.c, g = (lanes + 63) / 64;
  hipLaunchKernelGGL(k_rlc_scale, dim3(2 * g + 1), dim3(256), 0, s, b.tmp[1], b.fsum, tableG, tableG2, p, b.tmp[0], b.flags);
  hipLaunchKernelGGL(k_rlc_final, dim3(1), dim3(64), 0, s, b.tmp[0], p, b.flags);
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
