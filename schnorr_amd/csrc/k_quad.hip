// k_quad.hip — small batches: four lanes per signature (quad29.h).
// Same equation, same tables, same digits as k_verify_fixed_half (k_verify.hip); lane q of a quad
// runs one of the four products of every group operation, so the Straus chain takes ~2
// multiplication-times per operation instead of 7 / 8.  Used by the entry points for
// n <= kQuadMaxItems, where the chip is mostly idle and the call's latency is one lane's serial
// instruction stream (BASELINE configs[0] size: /root/reference/benches/signature.rs:48-60 shape,
// 1024 signatures).  Throughput per lane is ~0.7 of the one-lane kernel, so large batches never
// come here.
// Work split inside a quad: all four lanes run the (cheap, identical) scalar preparation; lane 0
// builds the window table of PK while lane 1 builds the one of R; all four read the entries.
#include "common.h"
#include "halfgcd.h"
#include "quad29.h"

namespace dsv {

DSV_DEV QExt qext_mul16(QExt p, int q) {
#pragma unroll 1
  for (int j = 0; j < 3; j++) qext_double<false>(p, q);
  qext_double<true>(p, q);
  return p;
}

template <int NCHAIN>
__global__ void __launch_bounds__(kQuadBlock)
k_verify_fixed_half_quad(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c,
                         ChainOperands op0, ChainOperands op1, const uint8_t* __restrict__ valid,
                         bool accumulate, size_t n, uint8_t* __restrict__ ok,
                         u32* __restrict__ var_tables) {
  const size_t i = ((size_t)blockIdx.x * kQuadBlock + threadIdx.x) >> 2;
  const int q = threadIdx.x & 3;
  if (i >= n) return;  // whole quads leave together
  u32* tpk = var_tables + i * (2 * kVarLaneWords);
  u32* tr = tpk + kVarLaneWords;
  bool good = accumulate ? (ok[i] != 0) : (valid[i] != 0);
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
    top = top_digit4(nz);
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
      recode_fixed(y, w);
#pragma unroll 1
      for (int win = 0; win < kFixedWindows; win++) {
        const int d = next_fixed_digit(y);
        qext_add_aniels(acc, q, load_aniels(op.table, win, d));
      }
    }
    good &= (bool)((int)fe_is_zero_canon(fe_canon(acc.u)) & (int)fe_equal(acc.v, acc.z));
    if (NCHAIN > 1) __builtin_amdgcn_wave_barrier();  // table slots are rebuilt by lanes 0 / 1 next round
  }
  if (q == 0) ok[i] = good ? 1 : 0;
}

void launch_verify_half_quad(int nchain, bool accumulate, const uint8_t* u, const uint8_t* c,
                             ChainOperands op0, ChainOperands op1, const uint8_t* valid, size_t n,
                             uint8_t* ok, uint32_t* var_tables, hipStream_t s) {
  const dim3 grid((unsigned)((4 * n + kQuadBlock - 1) / kQuadBlock)), block(kQuadBlock);
  if (nchain == 2)
    hipLaunchKernelGGL(k_verify_fixed_half_quad<2>, grid, block, 0, s, u, c, op0, op1, valid,
                       accumulate, n, ok, var_tables);
  else
    hipLaunchKernelGGL(k_verify_fixed_half_quad<1>, grid, block, 0, s, u, c, op0, op1, valid,
                       accumulate, n, ok, var_tables);
}

}  // namespace dsv
