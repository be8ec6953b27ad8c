// k_verify.hip — the dominant kernel: ok &= [ u*Gen + c*PK == R ] for Gen = G / G' given by its
// fixed-base table (`PublicKey::verify`, `PublicKeyDouble::verify`,
// /root/reference/src/keys/public.rs:121-130, :222-244), evaluated with half-size scalars
// (halfgcd.h): with (a, b), a = b*c (mod 8r), b odd,
//   u*G + c*PK == R   <=>   (b*u mod r)*G + a*PK - b*R == O.
// ONE per-lane window table over both variable bases — the 11 combinations da*PK + db*R of signed
// 2-bit digits in [-1, 2] (common.h: build_joint_table) — and one Straus chain of ~66 windows of two doublings
// and ONE addition each, whose length is the lane's own max(bitlen a, bitlen b) (lanes of a wave
// simply leave the loop at different times); then 16 mixed additions from the fixed-base table;
// the verdict is an identity test.  (r01 - r03a: one 4-bit table of 8 entries per base, two
// additions per four doublings: the same chain, but 116 multiplications of table building per
// chain instead of 67 + 12 squarings and 16 entries written per lane instead of 11: +2.1 % single,
// +2.5 % double, same-box A/B profiles/r03/ab_joint_windows.txt.)
//
// NCHAIN = 2 is PublicKeyDouble::verify in ONE launch: both equations share u and c, so (a, b), both
// recodings and b*u mod r are computed once and the chain runs twice — (G, PK, R) then (G', PK', R')
// — through the same code (a rolled loop over the two operand sets: the hot loop exists once in
// the instruction cache) and the same table slot.
#include "common.h"
#include "halfgcd.h"

namespace dsv {

// ---- lanes sorted by chain length (r04) --------------------------------------------------------
// A wave runs its longest lane: 65.7 two-bit windows on average where its lanes need 64.4.  For a
// large sub-batch the scalar preparation of every item — half_scalars, both recodings, b*u mod r, the
// top window — runs first in a kernel of its own (k_half_prep; 96-byte record per item), a counting
// sort by the top window yields a permutation (k_bucket_scatter, longest chains first), and the
// verify kernel walks the permutation: the lanes of a wave then run chains of one length.
//   record (24 words): ya[8] | yb[0..5] (words 6, 7 of yb are 0x55555555: |b| < 2^160) | w[8] |
//                      meta = top | b_neg << 8 | u_ok << 9 ; word 23 unused
constexpr int kSortRecWords = 24;
constexpr int kSortBins = 128;  // top = (bit length - 1) >> 1 <= 127

DSV_DEV void scalar_prep(u32 (&ya)[8], u32 (&yb)[8], u32 (&w)[8], bool& b_neg, int& top, bool& u_ok,
                         const uint8_t* __restrict__ u, const uint8_t* __restrict__ c, size_t i) {
  u32 cs[8], a[8], b[8];
  load_words8(cs, c, i);
  half_scalars(a, b, b_neg, cs);
  // signed 2-bit digits of a and |b| (the sign of the R term goes into the point: -R below)
  recode_signed2(ya, a);
  recode_signed2(yb, b);
  u32 nz[8];
#pragma unroll
  for (int k = 0; k < 8; k++) nz[k] = (ya[k] ^ 0x55555555u) | (yb[k] ^ 0x55555555u);
  top = top_digit2(nz);
  u32 us[8];
  load_words8(us, u, i);
  u_ok = words_lt(us, kR32);
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

__global__ void __launch_bounds__(256)
k_half_prep(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c, size_t n, u32* __restrict__ rec,
            u32* __restrict__ hist) {
  __shared__ u32 lh[kSortBins];
  if (threadIdx.x < kSortBins) lh[threadIdx.x] = 0;
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    u32 ya[8], yb[8], w[8];
    bool b_neg, u_ok;
    int top;
    scalar_prep(ya, yb, w, b_neg, top, u_ok, u, c, i);
    uint4* r = reinterpret_cast<uint4*>(rec + i * kSortRecWords);
    r[0] = make_uint4(ya[0], ya[1], ya[2], ya[3]);
    r[1] = make_uint4(ya[4], ya[5], ya[6], ya[7]);
    r[2] = make_uint4(yb[0], yb[1], yb[2], yb[3]);
    r[3] = make_uint4(yb[4], yb[5], w[0], w[1]);
    r[4] = make_uint4(w[2], w[3], w[4], w[5]);
    r[5] = make_uint4(w[6], w[7], (u32)top | (b_neg ? 0x100u : 0u) | (u_ok ? 0x200u : 0u), 0u);
    atomicAdd(&lh[top], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kSortBins && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}
// perm[position] = item, longest chains first; hist[0..127] = counts (k_half_prep), hist[128..255] =
// cursors (zeroed with the counts); order inside a bin is arbitrary
__global__ void __launch_bounds__(256)
k_bucket_scatter(const u32* __restrict__ rec, size_t n, u32* __restrict__ hist, u32* __restrict__ perm) {
  __shared__ u32 start[kSortBins];
  if (threadIdx.x < kSortBins) {
    u32 s = 0;
    for (int t = threadIdx.x + 1; t < kSortBins; t++) s += hist[t];
    start[threadIdx.x] = s;
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = i < n;
  const u32 t = act ? (rec[i * kSortRecWords + 22] & 0xffu) : 0xffffffffu;
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(act);
  while (todo) {  // one round per distinct chain length in the wave
    const int leader = __ffsll((long long)todo) - 1;
    const u32 tl = __shfl(t, leader, 64);
    const unsigned long long m = __ballot(act && t == tl);
    u32 base = 0;
    if (lane == leader) base = atomicAdd(&hist[kSortBins + tl], (u32)__popcll(m));
    base = __shfl(base, leader, 64);
    if (act && t == tl) perm[start[tl] + base + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)i;
    todo &= ~m;
  }
}

template <int NCHAIN, bool SORTED>
__global__ void __launch_bounds__(kVerifyBlock, kWavesVerify)
k_verify_fixed_half(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c, ChainOperands op0,
                    ChainOperands op1, const uint8_t* __restrict__ valid, bool accumulate, size_t n,
                    uint8_t* __restrict__ ok, u32* __restrict__ var_tables, const u32* __restrict__ rec,
                    const u32* __restrict__ perm) {
  u32* tbl = var_tables + ((size_t)blockIdx.x * kVerifyBlock + threadIdx.x) * kJointLaneWords;
#pragma unroll 1
  for (size_t base = (size_t)blockIdx.x * kVerifyBlock; base < n;
       base += (size_t)gridDim.x * kVerifyBlock) {
    const size_t j = base + threadIdx.x;
    if (j >= n) continue;
    const size_t i = SORTED ? (size_t)perm[j] : j;
    bool good = accumulate ? (ok[i] != 0) : (valid[i] != 0);
    u32 ya[8], yb[8], w[8];
    bool b_neg;
    int top;
    if constexpr (SORTED) {
      const uint4* r = reinterpret_cast<const uint4*>(rec + i * kSortRecWords);
      const uint4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5];
      ya[0] = r0.x; ya[1] = r0.y; ya[2] = r0.z; ya[3] = r0.w;
      ya[4] = r1.x; ya[5] = r1.y; ya[6] = r1.z; ya[7] = r1.w;
      yb[0] = r2.x; yb[1] = r2.y; yb[2] = r2.z; yb[3] = r2.w;
      yb[4] = r3.x; yb[5] = r3.y; yb[6] = 0x55555555u; yb[7] = 0x55555555u;
      w[0] = r3.z; w[1] = r3.w; w[2] = r4.x; w[3] = r4.y; w[4] = r4.z; w[5] = r4.w;
      w[6] = r5.x; w[7] = r5.y;
      top = (int)(r5.z & 0xffu);
      b_neg = (r5.z & 0x100u) != 0;
      good &= (r5.z & 0x200u) != 0;
    } else {
      bool u_ok;
      scalar_prep(ya, yb, w, b_neg, top, u_ok, u, c, i);
      good &= u_ok;
    }
#pragma unroll 1
    for (int h = 0; h < NCHAIN; h++) {
      const ChainOperands op = h ? op1 : op0;
      {
        Fe pku, pkv, ru, rv;
        good &= load_fq(pku, op.PK_uv, 2 * i);
        good &= load_fq(pkv, op.PK_uv, 2 * i + 1);
        good &= load_fq_signed(ru, op.R_uv, 2 * i, !b_neg);  // the chain adds -|b| * R unless b < 0
        good &= load_fq(rv, op.R_uv, 2 * i + 1);
        build_joint_table(tbl, pku, pkv, ru, rv);
      }
      // T = a*PK + |b|*(-+R) (+ w*G below): one joint entry per 2-bit window, loaded one window ahead
      Ext acc = ext_from_niels(load_var_entry(tbl, joint_digit(ya, yb, top)));
      {
        RawNiels e = load_var_entry_raw(tbl, joint_digit(ya, yb, top > 0 ? top - 1 : 0));
#pragma unroll 1
        for (int k = top - 1; k >= 0; k--) {
          acc = ext_mul4(acc);
          const Niels cur = finish_var_entry(e);
          e = load_var_entry_raw(tbl, joint_digit(ya, yb, k > 0 ? k - 1 : 0));  // last: unused
          acc = ext_add_niels(acc, cur);
        }
      }
      // T + w*G == O  (T == O  <=>  u == 0 and v == z, decided inside the last addition)
      good &= fixed_base_accumulate_is_identity(acc, w, op.table);
    }
    ok[i] = good ? 1 : 0;
  }
}

void launch_verify_half(int nchain, bool accumulate, const uint8_t* u, const uint8_t* c,
                        ChainOperands op0, ChainOperands op1, const uint8_t* valid, size_t n,
                        uint8_t* ok, uint32_t* var_tables, hipStream_t s, const SortWs* sw) {
  const dim3 grid(verify_grid(n)), block(kVerifyBlock);
  const u32 *rec = sw ? sw->rec : nullptr, *perm = sw ? sw->perm : nullptr;
  if (sw) {
    // scalar preparation of every item + counting sort by chain length: three small launches
    (void)hipMemsetAsync(sw->hist, 0, 2 * kSortBins * sizeof(u32), s);
    hipLaunchKernelGGL(k_half_prep, dim3(grid_for(n)), dim3(256), 0, s, u, c, n, sw->rec, sw->hist);
    hipLaunchKernelGGL(k_bucket_scatter, dim3(grid_for(n)), dim3(256), 0, s, (const u32*)sw->rec, n, sw->hist, sw->perm);
    if (nchain == 2)
      hipLaunchKernelGGL((k_verify_fixed_half<2, true>), grid, block, 0, s, u, c, op0, op1, valid, accumulate, n, ok, var_tables, rec, perm);
    else
      hipLaunchKernelGGL((k_verify_fixed_half<1, true>), grid, block, 0, s, u, c, op0, op1, valid, accumulate, n, ok, var_tables, rec, perm);
    return;
  }
  if (nchain == 2)
    hipLaunchKernelGGL((k_verify_fixed_half<2, false>), grid, block, 0, s, u, c, op0, op1, valid, accumulate, n,
                       ok, var_tables, rec, perm);
  else
    hipLaunchKernelGGL((k_verify_fixed_half<1, false>), grid, block, 0, s, u, c, op0, op1, valid, accumulate, n,
                       ok, var_tables, rec, perm);
}

}  // namespace dsv
