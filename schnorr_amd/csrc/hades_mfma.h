// hades_mfma.h — the linear layers of Hades (hades29.h) on the matrix cores.
//
// After the scalar recurrence of the partial rounds, 64 % of k_challenge's MADs were products of a
// per-hash value with a CONSTANT field element: ten per recurrence round, five per row of the dense
// layer of a full round.  Over the 64 hashes of a wave such a row is a constant matrix times a
// matrix of per-hash columns — the one place in this engine where the work has the shape the matrix
// cores want (every other product has two per-lane operands).  In bytes, with the POSITION of a
// byte folded into the constant modulo q:
//   x_j = sum_k b_jk 2^(8k),   K_jk = c_j * 2^(8k) mod q = sum_m e_jkm 2^(8m)   (generator)
//   sum_j c_j x_j == sum_m 2^(8m) C_m (mod q),   C_m = sum_j sum_k e_jkm (b_jk - 128) + constant
// i.e. C (32 byte rows x 64 hashes) = A (32 x 32*terms, dense) * B (32*terms x 64): one
// v_mfma_i32_32x32x32_i8 per term and hash tile (2 per term and wave), exact in int32
// (|C_m| < 2^23).  The result is a 271-bit integer congruent to the row; one Barrett step (a
// 32 x 18-bit product for the quotient, 8 MADs for quotient * q) brings it under 2^256 — no
// Montgomery reduction, because the operands carry their Montgomery factor through a linear map
// unchanged.  (r02's first version kept c_j as ONE constant: a Toeplitz band, 64 rows, two half-empty
// tiles per term and a 17-column Montgomery reduction — twice the MFMAs; profiles/r02/ab_hades_mfma.txt.)
// The VALU keeps what only it can do — the S-box — plus the glue around the MFMAs:
//  * operands: 8 words -> XOR 0x80808080 (signed digits b - 128, no carry chain) -> one
//    v_permlane32_swap per register, because a hash tile is 32 hashes wide and its K range is split
//    over the two halves of the wave; kept as the 4-register tuples the MFMA reads;
//  * result: each lane holds four 4-row groups of BOTH hash tiles; two v_lshl_add_u32 and a sign-bit
//    flip turn a group into two unsigned halves, the same swap brings the partner's groups home, one
//    9-word carry chain z = P + 2^16 Q + start, the Barrett step, 8 result words.
// Layout (tools/microbench/mfma_layout.hip): A and B hold k = 16*(lane/32) + byte of 4 VGPRs, row /
// column = lane%32; D register i of lane l is row 8*(i/4) + 4*(l/32) + i%4, column l%32.
//
// Every constant of that arithmetic (the -128 offsets, the 2^31 biases of the halves, the +1 of an
// operand stored one below its value, the round constant) is ONE field element per row, folded by
// the generator into the row's start words; gen_constants.py: mfma_linear proves the ranges for all
// operand values from the actual digits and re-derives every intermediate in plain integers
// (mfma_step_model) against the field arithmetic.
#pragma once
#include "fe29.h"

namespace dsv {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ const u32 g_hades_mfma_a[DSV_HADES_MFMA_A_WORDS] = {DSV_HADES_MFMA_A_LIST};
__device__ const u32 g_hades_mfma_start[DSV_HADES_MFMA_ROUNDS][8] = {DSV_HADES_MFMA_START_LIST};
constexpr int kMfmaTerms = 10;
constexpr int kMfmaAVecs = DSV_HADES_MFMA_A_WORDS / 4;  // 16-byte operands: [term][lane]
static_assert(kMfmaAVecs == kMfmaTerms * 64, "operand table shape");
// the dense 5 x 5 layer of the full rounds, one output row = one 5-term product of the same kind
__device__ const u32 g_hades_mfma_mds[DSV_HADES_MFMA_MDS_WORDS] = {DSV_HADES_MFMA_MDS_LIST};
__device__ const u32 g_hades_mfma_mds_start[DSV_HADES_WIDTH][8] = {DSV_HADES_MFMA_MDS_START_LIST};
constexpr int kMfmaMdsRowVecs = DSV_HADES_WIDTH * 64;  // one output row: [term][lane]
constexpr int kMfmaMdsVecs = DSV_HADES_MFMA_MDS_WORDS / 4;
static_assert(DSV_HADES_MFMA_MDS_WORDS / 4 == DSV_HADES_WIDTH * kMfmaMdsRowVecs, "operand table shape");
// start-up rows and state rebuild of the recurrence: used once per permutation, so their 90 KB of
// operands stay in global memory (L2) instead of LDS.  Row offsets in 16-byte operands: rows of 7,
// 9, 11, 13 terms, then five rows of 10.
__device__ const u32 g_hades_mfma_edge[DSV_HADES_MFMA_EDGE_WORDS] = {DSV_HADES_MFMA_EDGE_LIST};
__device__ const u32 g_hades_mfma_edge_start[9][8] = {DSV_HADES_MFMA_EDGE_START_LIST};
static_assert(DSV_HADES_MFMA_EDGE_WORDS / 4 == (7 + 9 + 11 + 13 + 5 * 10) * 64, "operand table shape");
constexpr int kMfmaEdgeInitOff[4] = {0, 7 * 64, (7 + 9) * 64, (7 + 9 + 11) * 64};
constexpr int kMfmaEdgeFinalOff = (7 + 9 + 11 + 13) * 64;
__device__ constexpr u32 kMfmaQ32[8] = DSV_Q32;

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

// 8 words -> signed digits (byte - 128) and the cross-half exchange
DSV_DEV Dig mfma_digits_words(const u32 (&x)[8]) {
  u32 w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = x[i] ^ 0x80808080u;
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
// x: limbs < 2^29, value < 2^256
DSV_DEV Dig mfma_digits(const Fe& x) {
  u32 w[8];
  fe_to_words_plain(w, x);
  return mfma_digits_words(w);
}
// The operand tables live in LDS (10 KB recurrence + 25 KB dense layer per workgroup of four waves,
// two workgroups per CU): every lane re-reads its 16 bytes of each A operand every time.
struct MfmaTable {
  const v4i* a;    // recurrence: [term][lane]
  const v4i* mds;  // dense layer: [row][term][lane]
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

// ---- one row, split so that the matrix cores can run under the VALU work of the SAME wave -------
// acc[hash tile] += (term j's 32 x 32 block) x (digits of one operand)
struct MfmaAcc {
  v16i t[2];
};
DSV_DEV void mfma_clear(MfmaAcc& acc) {
  const v16i zero = {0};  // an inline constant of the first MFMA: costs nothing
  acc.t[0] = zero;
  acc.t[1] = zero;
}
// base: [term][lane] operands of one output row, already offset by the lane (LDS or global)
DSV_DEV void mfma_term(MfmaAcc& acc, int j, const Dig& d, const v4i* base) {
  const v4i a = base[j * 64];
  acc.t[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, d.t0, acc.t[0], 0, 0, 0);
  acc.t[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, d.t1, acc.t[1], 0, 0, 0);
}
// The eight 4-row groups of the lane's own hash, as halves: group idx sits at bit 32 idx, its
// halves p (bit 32 idx) and q (bit 32 idx + 16).  Lane l (half h = l/32) holds rows
// 8g + 4h + (0..3) of BOTH hash tiles; after the swap every lane has, for its own hash, the h = 0
// groups (idx = 2g) and the h = 1 groups (idx + 1).
struct MfmaGroups {
  u32 p[8], q[8];
};
DSV_DEV MfmaGroups mfma_collect(const MfmaAcc& acc) {
  MfmaGroups w;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    u32 xp = mfma_half(acc.t[0], 4 * g), xq = mfma_half(acc.t[0], 4 * g + 2);
    u32 yp = mfma_half(acc.t[1], 4 * g), yq = mfma_half(acc.t[1], 4 * g + 2);
    half_swap(xp, yp);
    half_swap(xq, yq);
    w.p[2 * g] = xp;
    w.q[2 * g] = xq;
    w.p[2 * g + 1] = yp;
    w.q[2 * g + 1] = yq;
  }
  return w;
}
// halves + start words -> z (9 words, < 2^272) -> Barrett step -> out (8 words, < 2^256), congruent
// to the row (generator: mfma_step_model)
DSV_DEV void mfma_finish(u32 (&out)[8], const MfmaGroups& w, const u32* start) {
  u32 z[9];
  {
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32 f = i ? __funnelshift_r(w.q[i - 1], w.q[i], 16) : (w.q[0] << 16);
      s += (u64)w.p[i] + f + start[i];
      z[i] = (u32)s;
      s >>= 32;
    }
    z[8] = (u32)s + (w.q[7] >> 16);
  }
  const u32 t = __funnelshift_r(z[7], z[8], 16);  // z >> 240, < 2^32
  const u32 qhat = __umulhi(t, (u32)DSV_HADES_MFMA_MU);
  u64 m = 0;
  u32 borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    m += (u64)qhat * kMfmaQ32[i];
    out[i] = __builtin_subc(z[i], (u32)m, borrow, &borrow);
    m >>= 32;
  }
  // (word 8 of z - qhat * q is zero: the difference is below 2^256)
}

// one row of NT terms: operands d[], A operands at base[term * 64] (lane offset already applied),
// start words of the row; result as 8 words
template <int NT>
DSV_DEV void mfma_row(u32 (&out)[8], const Dig (&d)[NT], const v4i* base, const u32* start) {
  MfmaAcc acc;
  mfma_clear(acc);
#pragma unroll
  for (int j = 0; j < NT; j++) mfma_term(acc, j, d[j], base);
  mfma_finish(out, mfma_collect(acc), start);
}
template <int NT>
DSV_DEV Fe mfma_dot(const Dig (&d)[NT], const v4i* base, const u32* start) {
  u32 w[8];
  mfma_row<NT>(w, d, base, start);
  return fe_from_words_plain(w);
}
// one output row of the dense layer: sum_j M[row][j] * s_j, operands = the five S-box outputs
DSV_DEV Fe hades_mfma_mds_row(const Dig (&d)[DSV_HADES_WIDTH], int row, const MfmaTable& tab) {
  return mfma_dot<DSV_HADES_WIDTH>(d, tab.mds + row * kMfmaMdsRowVecs + tab.lane, g_hades_mfma_mds_start[row]);
}

}  // namespace dsv
