// hades_mfma.h — the partial-round recurrence of hades29.h on the matrix cores.
//
// hades_partial_rounds_arma() spends 810 of its ~1270 v_mad_u64_u32 per round on ten products of a
// per-hash value with a CONSTANT field element.  Over the 64 hashes of a wave that is a constant
// matrix times a matrix of per-hash columns — the one place in this engine where the work has the
// shape the matrix cores want (every other product has two per-lane operands).  In bytes:
//   x_j = sum_k d_jk 2^(8k)          (32 signed digits per window value, d = byte - 128)
//   k_j = sum_c e_jc 2^(8c)          (32 balanced digits per multiplier, generator)
//   sum_j k_j x_j = sum_m 2^(8m) C_m,   C_m = sum_j sum_k e_j(m-k) d_jk     (|C_m| < 2^22.4)
// i.e. C (64 byte columns x 64 hashes) = A (64 x 320, Toeplitz blocks of the e_j) * B (320 x 64):
// 2 row tiles x 2 hash tiles x 10 terms = 40 v_mfma_i32_32x32x32_i8 per wave and round, exact in
// int32.  The VALU keeps what only it can do: the S-box (x^5), the Montgomery reduction of the
// recombined column sums, and ~250 cheap instructions per round of digit packing / recombination.
// Measured co-issue (tools/microbench/mfma_mix.hip): one such MFMA costs a 2-wave SIMD ~14 cycles
// of VALU issue, so the 810 MADs (~3 300 cycles) become ~560 + ~650 cycles.
//
// Layout (tools/microbench/mfma_layout.hip): A and B hold k = 16*(lane/32) + byte of 4 VGPRs, row /
// column = lane%32; D register i of lane l is row 8*(i/4) + 4*(l/32) + i%4, column l%32.  Hash tile
// t = hashes 32t .. 32t+31 of the wave, so the digits of a hash have to sit in ITS lane (half of
// the k range) and in the partner lane l ^ 32 (other half): one v_permlane32_swap per register.
// The same swap brings the two halves of a result column back into the hash's own lane.
//
// Every intermediate (digit ranges, biases, the constant folded into the start limbs, the bound of
// the value handed to fe_reduce_cols) is re-derived in plain integers by gen_constants.py:
// mfma_step_model, which also emits the operand table and the start limbs used here.
#pragma once
#include "fe29.h"

namespace dsv {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ const u32 g_hades_mfma_a[DSV_HADES_MFMA_A_WORDS] = {DSV_HADES_MFMA_A_LIST};
__device__ const u32 g_hades_mfma_start[DSV_HADES_MFMA_ROUNDS][NL] = {DSV_HADES_MFMA_START_LIST};
constexpr int kMfmaTerms = 10;
constexpr int kMfmaAVecs = DSV_HADES_MFMA_A_WORDS / 4;  // 16-byte operands: [term][row tile][lane]
static_assert(kMfmaAVecs == kMfmaTerms * 2 * 64, "operand table shape");
// the dense 5 x 5 layer of the full rounds, one output row = one 5-term product of the same kind
// (DSV_HADES_MFMA_MDS, shipped 1; 0 keeps fe_dot5 for A/B)
#ifndef DSV_HADES_MFMA_MDS
#define DSV_HADES_MFMA_MDS 1
#endif
__device__ const u32 g_hades_mfma_mds[DSV_HADES_MFMA_MDS_WORDS] = {DSV_HADES_MFMA_MDS_LIST};
__device__ const u32 g_hades_mfma_mds_start[DSV_HADES_WIDTH][NL] = {DSV_HADES_MFMA_MDS_START_LIST};
constexpr int kMfmaMdsRowVecs = DSV_HADES_WIDTH * 2 * 64;  // one output row: [term][row tile][lane]
constexpr int kMfmaMdsVecs = DSV_HADES_MFMA_MDS ? DSV_HADES_MFMA_MDS_WORDS / 4 : 0;
static_assert(DSV_HADES_MFMA_MDS_WORDS / 4 == DSV_HADES_WIDTH * kMfmaMdsRowVecs, "operand table shape");

// B operands of one window value, kept as the 4-register tuples the MFMA reads (separate words
// would be copied into a tuple in front of every MFMA): t0 for hash tile 0, t1 for hash tile 1
struct Dig {
  v4i t0, t1;
};

// swap lanes 32..63 of a with lanes 0..31 of b
DSV_DEV void half_swap(u32& a, u32& b) {
  const v2u r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// x: limbs < 2^29, value < 2^256.  Signed digits (byte - 128) and the cross-half exchange.
DSV_DEV Dig mfma_digits(const Fe& x) {
  u32 w[8];
  fe_to_words_plain(w, x);
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] ^= 0x80808080u;
  // before: w[0..3] = digits 0..15 (L), w[4..7] = digits 16..31 (H) of the lane's own hash.
  // after:  w[0..3] = L own (lanes < 32) | H of lane-32 (lanes >= 32)   = B of hash tile 0
  //         w[4..7] = L of lane+32 (lanes < 32) | H own (lanes >= 32)   = B of hash tile 1
#pragma unroll
  for (int i = 0; i < 4; i++) half_swap(w[i], w[4 + i]);
  Dig d;
  d.t0 = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
  d.t1 = v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
  return d;
}
// inverse of mfma_digits
DSV_DEV Fe mfma_undigits(const Dig& d) {
  u32 w[8];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    w[i] = (u32)d.t0[i];
    w[4 + i] = (u32)d.t1[i];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) half_swap(w[i], w[4 + i]);
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] ^= 0x80808080u;
  Fe r;  // 9 x 29 bits out of 8 x 32 (fe_from_words_plain drops bit 255 and up: not here)
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
    u32 lo = w[wi] >> sh;
    if (sh > 3 && wi + 1 < 8) lo |= w[wi + 1] << (32 - sh);
    r.l[i] = (i < NL - 1) ? (lo & M29) : lo;
  }
  return r;
}

// start-up rows and state rebuild of the recurrence (DSV_HADES_MFMA_EDGE, shipped 1; 0 keeps the
// limb products for A/B): used once per permutation, so their 180 KB of operands stay in global
// memory (L2) instead of LDS.  Row offsets in 16-byte operands: rows of 7, 9, 11, 13 terms, then
// five rows of 10.
#ifndef DSV_HADES_MFMA_EDGE
#define DSV_HADES_MFMA_EDGE 1
#endif
#if DSV_HADES_MFMA_EDGE
__device__ const u32 g_hades_mfma_edge[DSV_HADES_MFMA_EDGE_WORDS] = {DSV_HADES_MFMA_EDGE_LIST};
__device__ const u32 g_hades_mfma_edge_start[9][NL] = {DSV_HADES_MFMA_EDGE_START_LIST};
static_assert(DSV_HADES_MFMA_EDGE_WORDS / 4 == (7 + 9 + 11 + 13 + 5 * 10) * 2 * 64, "operand table shape");
constexpr int kMfmaEdgeInitOff[4] = {0, 7 * 128, (7 + 9) * 128, (7 + 9 + 11) * 128};
constexpr int kMfmaEdgeFinalOff = (7 + 9 + 11 + 13) * 128;
#endif

// The operand tables live in LDS (20 KB recurrence + 50 KB dense layer per workgroup of four waves,
// two workgroups per CU): every lane re-reads its 16 bytes of each A operand every time.
struct MfmaTable {
  const v4i* a;    // recurrence: [term][row tile][lane]
  const v4i* mds;  // dense layer: [row][term][row tile][lane]
  int lane;        // lane within the wave
};
DSV_DEV v4i* hades_mfma_lds() {
  __shared__ v4i lds_a[kMfmaAVecs + kMfmaMdsVecs];
  return lds_a;
}
// once per workgroup, by every thread of it, before the first hash
DSV_DEV void hades_mfma_load_table() {
  v4i* lds_a = hades_mfma_lds();
  const v4i* src = reinterpret_cast<const v4i*>(g_hades_mfma_a);
  for (int k = threadIdx.x; k < kMfmaAVecs; k += blockDim.x) lds_a[k] = src[k];
  const v4i* src2 = reinterpret_cast<const v4i*>(g_hades_mfma_mds);
  for (int k = threadIdx.x; k < kMfmaMdsVecs; k += blockDim.x) lds_a[kMfmaAVecs + k] = src2[k];
  __syncthreads();
}
DSV_DEV MfmaTable hades_mfma_table() {
  MfmaTable t;
  t.a = hades_mfma_lds();
  t.mds = t.a + kMfmaAVecs;
  t.lane = threadIdx.x & 63;
  return t;
}

// The two halves of a 4-row group: t0 = C0 + 2^8 C1 + 2^31, t1 = C2 + 2^8 C3 + 2^31 (the group is
// t0 + 2^16 t1).  |C| < 2^22.4, so each half is a signed 32-bit value and flipping its sign bit adds
// the 2^31 that makes it unsigned.
DSV_DEV u32 mfma_half(const v16i& acc, int i) {
  return (((u32)acc[i + 1] << 8) + (u32)acc[i]) ^ 0x80000000u;
}

// ---- the step, split so that the matrix cores run under the VALU work of the SAME wave ---------
// acc[hash tile][row tile] += (term j's Toeplitz block) x (digits of one window value)
struct MfmaAcc {
  v16i t[2][2];
};
DSV_DEV void mfma_clear(MfmaAcc& acc) {
  const v16i zero = {0};  // an inline constant of the first MFMA: costs nothing
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int mt = 0; mt < 2; mt++) acc.t[h][mt] = zero;
}
// base: [term][row tile][lane] operands of one output row, already offset by the lane
DSV_DEV void mfma_term(MfmaAcc& acc, int j, const Dig& d, const v4i* base) {
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
    const v4i a = base[(j * 2 + mt) * 64];
    acc.t[0][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, d.t0, acc.t[0][mt], 0, 0, 0);
    acc.t[1][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, d.t1, acc.t[1][mt], 0, 0, 0);
  }
}
// The sixteen 4-row groups of the lane's own hash, as halves: group idx sits at bit 32 idx, its
// halves p (bit 32 idx) and q (bit 32 idx + 16).  Lane l (half h = l/32) holds rows
// 32mt + 8g + 4h + (0..3) of BOTH hash tiles; after the swap every lane has, for its own hash, the
// h = 0 groups (idx = 8mt + 2g) and the h = 1 groups (idx + 1).
struct MfmaGroups {
  u32 p[16], q[16];
};
DSV_DEV MfmaGroups mfma_collect(const MfmaAcc& acc) {
  MfmaGroups w;
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
      u32 xp = mfma_half(acc.t[0][mt], 4 * g), xq = mfma_half(acc.t[0][mt], 4 * g + 2);
      u32 yp = mfma_half(acc.t[1][mt], 4 * g), yq = mfma_half(acc.t[1][mt], 4 * g + 2);
      half_swap(xp, yp);
      half_swap(xq, yq);
      const int idx = 8 * mt + 2 * g;
      w.p[idx] = xp;
      w.q[idx] = xq;
      w.p[idx + 1] = yp;
      w.q[idx + 1] = yq;
    }
  }
  return w;
}
// halves -> 17 words (one carry chain: z = P + 2^16 Q) -> 29-bit columns + start limbs ->
// Montgomery reduction.  Result limbs < 2^29, value < 2^256 (generator: mfma_step_model).
DSV_DEV Fe mfma_finish(const MfmaGroups& w, const u32* start) {
  u32 z[17];
  {
    u32 c = 0;
    z[0] = __builtin_addc(w.p[0], w.q[0] << 16, 0u, &c);
#pragma unroll
    for (int i = 1; i < 16; i++)
      z[i] = __builtin_addc(w.p[i], __funnelshift_r(w.q[i - 1], w.q[i], 16), c, &c);
    z[16] = (w.q[15] >> 16) + c;
    // take the bias of the top group off again, down to 2^515 (generator: MFMA_KTOP)
    u64 top = ((u64)z[16] << 32) | z[15];
    top -= ((u64)(0x8000u - 8u) << 32) | 0x80000000u;
    z[15] = (u32)top;
    z[16] = (u32)(top >> 32);
  }
  // 29-bit columns (column 16 takes everything from bit 464 up) + the start limbs
  u64 c[18];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int bit = 29 * k, wi = bit >> 5, sh = bit & 31;
    const u32 v = sh ? __funnelshift_r(z[wi], z[wi + 1], sh) : z[wi];
    c[k] = (u64)((v & M29) + (k < NL ? start[k] : 0u));
  }
  c[16] = (u64)__funnelshift_r(z[14], z[15], 16) | ((u64)__funnelshift_r(z[15], z[16], 16) << 32);
  c[17] = 0;
  return fe_reduce_cols(c);
}

// a_new = sum_j k_j x_j + gamma (Montgomery form) in one go (the unpipelined form, kept for A/B)
DSV_DEV Fe hades_mfma_step(const Dig (&win)[kMfmaTerms], const u32* start, const MfmaTable& tab) {
  MfmaAcc acc;
  mfma_clear(acc);
#pragma unroll
  for (int j = 0; j < kMfmaTerms; j++) mfma_term(acc, j, win[j], tab.a + tab.lane);
  return mfma_finish(mfma_collect(acc), start);
}
// one row of NT terms: operands d[], A operands at base[(term * 2 + row tile) * 64] (lane offset
// already applied; LDS or global), start limbs of the row
template <int NT>
DSV_DEV Fe mfma_dot(const Dig (&d)[NT], const v4i* base, const u32* start) {
  MfmaAcc acc;
  mfma_clear(acc);
#pragma unroll
  for (int j = 0; j < NT; j++) mfma_term(acc, j, d[j], base);
  return mfma_finish(mfma_collect(acc), start);
}
// one output row of the dense layer: sum_j M[row][j] * s_j, operands = the five S-box outputs
DSV_DEV Fe hades_mfma_mds_row(const Dig (&d)[DSV_HADES_WIDTH], int row, const MfmaTable& tab) {
  const v4i* base = tab.mds + row * kMfmaMdsRowVecs + tab.lane;
  MfmaAcc acc;
  mfma_clear(acc);
#pragma unroll
  for (int j = 0; j < DSV_HADES_WIDTH; j++) mfma_term(acc, j, d[j], base);
  return mfma_finish(mfma_collect(acc), g_hades_mfma_mds_start[row]);
}

}  // namespace dsv
