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

template <int NCHAIN>
__global__ void __launch_bounds__(kVerifyBlock, kWavesVerify)
k_verify_fixed_half(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c, ChainOperands op0,
                    ChainOperands op1, const uint8_t* __restrict__ valid, bool accumulate, size_t n,
                    uint8_t* __restrict__ ok, u32* __restrict__ var_tables, const u32* __restrict__ gate) {
  if (gate_says_done(gate)) return;  // the batch fast accept decided these items (launch.h)
  u32* tbl = var_tables + ((size_t)blockIdx.x * kVerifyBlock + threadIdx.x) * kJointLaneWords;
#pragma unroll 1
  for (size_t base = (size_t)blockIdx.x * kVerifyBlock; base < n;
       base += (size_t)gridDim.x * kVerifyBlock) {
    const size_t i = base + threadIdx.x;
    if (i >= n) continue;
    bool good = accumulate ? (ok[i] != 0) : (valid[i] != 0);
    u32 ya[8], yb[8], w[8];
    bool b_neg;
    int top;
    {
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
                        uint8_t* ok, uint32_t* var_tables, hipStream_t s, const uint32_t* gate) {
  const dim3 grid(verify_grid(n)), block(kVerifyBlock);
  if (nchain == 2)
    hipLaunchKernelGGL(k_verify_fixed_half<2>, grid, block, 0, s, u, c, op0, op1, valid, accumulate, n,
                       ok, var_tables, gate);
  else
    hipLaunchKernelGGL(k_verify_fixed_half<1>, grid, block, 0, s, u, c, op0, op1, valid, accumulate, n,
                       ok, var_tables, gate);
}

}  // namespace dsv
