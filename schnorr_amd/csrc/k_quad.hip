// k_quad.hip — small batches: EIGHT lanes per signature, two quads of four (quad29.h).
// Same equation, same tables, same digits as k_verify_fixed_half (k_verify.hip).  Used by the entry
// points for n <= kQuadMaxItems, where the chip is mostly idle and the call's latency is one lane's
// serial instruction stream (BASELINE configs[0] size: /root/reference/benches/signature.rs:48-60
// shape, 1024 signatures).  Two levels of parallelism inside ONE signature:
//   * a quad of lanes runs the four independent products of every group operation side by side
//     (r02: ~2 multiplication-times per operation instead of 7 / 8);
//   * r03: the two variable-base parts are independent until the end, so quad 0 of an octet runs
//     a*PK and quad 1 runs -b*R — each its own 4-doubling / 1-addition chain, the same instruction
//     stream on different operands — and each takes half of the 16 fixed-base additions of
//     (b*u)*G; one extended addition joins them.  128 doublings + 41 additions + 1 instead of
//     128 + 82: ~19 % fewer multiplication-times on the critical path (the doublings are done
//     twice, which costs nothing where the machine idles).
// Throughput per lane is ~0.35 of the one-lane kernel, so large batches never come here.
// Work split inside an octet: all eight lanes run the (cheap, identical) scalar preparation; lane
// 0 of quad 0 builds the window table of PK while lane 0 of quad 1 builds the one of R.
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
// value held by the lane four lanes up (quad 1 of an octet -> quad 0), for every limb
DSV_DEV Fe from_upper_quad(const Fe& x) {
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++)
    r.l[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)x.l[i], 0x104 /* row_shl:4 */, 0xf, 0xf, true);
  return r;
}

template <int NCHAIN>
__global__ void __launch_bounds__(kQuadBlock)
k_verify_fixed_half_oct(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c,
                        ChainOperands op0, ChainOperands op1, const uint8_t* __restrict__ valid,
                        bool accumulate, bool tables_ready, size_t n, uint8_t* __restrict__ ok,
                        u32* __restrict__ var_tables, const u32* __restrict__ gate) {
  if (gate_says_done(gate)) return;  // (uniform: before any barrier)
  const size_t i = ((size_t)blockIdx.x * kQuadBlock + threadIdx.x) >> 3;
  const int q = threadIdx.x & 3;
  const bool upper = (threadIdx.x & 4) != 0;  // quad 1 of the octet: the -b*R half
  if (i >= n) return;                         // whole octets leave together
  u32* tpk = var_tables + i * (2 * kVarLaneWords);
  u32* tr = tpk + kVarLaneWords;
  bool good = accumulate ? (ok[i] != 0) : (valid[i] != 0);
  u32 ys[8], w[8];  // this quad's recoded scalar (a or b), and b*u mod r
  int sgn, top;
  {
    u32 cs[8], a[8], b[8], ya[8], yb[8];
    bool b_neg;
    load_words8(cs, c, i);
    half_scalars(a, b, b_neg, cs);
    recode_signed4(ya, a);
    recode_signed4(yb, b);
    u32 nz[8];
#pragma unroll
    for (int k = 0; k < 8; k++) nz[k] = (ya[k] ^ 0x88888888u) | (yb[k] ^ 0x88888888u);
    top = top_digit4(nz);  // both quads walk the same number of windows
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
#pragma unroll
    for (int k = 0; k < 8; k++) ys[k] = upper ? yb[k] : ya[k];
    sgn = upper ? (b_neg ? 1 : -1) : 1;
  }
  const u32* tbl = upper ? tr : tpk;
#pragma unroll 1
  for (int h = 0; h < NCHAIN; h++) {
    const ChainOperands op = h ? op1 : op0;
    {
      // lane 0 of quad 0: table of PK, lane 0 of quad 1: table of R; validity of all four
      // coordinates is checked by every lane
      Fe pu, pv, ru, rv;
      good &= load_fq(pu, op.PK_uv, 2 * i);
      good &= load_fq(pv, op.PK_uv, 2 * i + 1);
      good &= load_fq(ru, op.R_uv, 2 * i);
      good &= load_fq(rv, op.R_uv, 2 * i + 1);
      // (the first equation's tables may have been built beside the hash already: k_prep_var_tables)
      if (q == 0 && !(tables_ready && h == 0)) build_var_table(upper ? tr : tpk, upper ? ru : pu, upper ? rv : pv);
      // the other lanes of the wave read what these lanes wrote to global memory
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
    }
    QExt acc = qext_identity();
    qext_add_niels(acc, q, load_var_entry(tbl, sgn * sdigit4(ys, top)));
#pragma unroll 1
    for (int k = top - 1; k >= 0; k--) {
      acc = qext_mul16(acc, q);
      qext_add_niels(acc, q, load_var_entry(tbl, sgn * sdigit4(ys, k)));
    }
    {  // += half of w * Gen: windows 0..7 in quad 0, 8..15 in quad 1 (16-bit windows: half-words)
      static_assert(kFixedBits == 16 && kFixedWindows == 16, "the window split below assumes 16 x 16 bits");
      u32 y[9];
      recode_fixed(y, w);
#pragma unroll 1
      for (int j = 0; j < kFixedWindows / 2; j++) {
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {  // word j / 2 of this quad's half, without a dynamic register index
          lo = (j >> 1) == k ? y[k] : lo;
          hi = (j >> 1) == k ? y[4 + k] : hi;
        }
        const u32 word = upper ? hi : lo;
        const int d = (int)((word >> (16 * (j & 1))) & 0xffffu) - kFixedHalf;
        qext_add_aniels(acc, q, load_aniels(op.table, j + (upper ? 8 : 0), d));
      }
    }
    {  // quad 0 += quad 1: the upper half as an extended niels operand, moved four lanes down
      Niels nb;
      nb.vpu = from_upper_quad(fe_carry(fe_add(acc.v, acc.u)));
      nb.vmu = from_upper_quad(fe_sub2(acc.v, acc.u));
      nb.z = from_upper_quad(acc.z);
      nb.t2d = from_upper_quad(fe_mul(acc.t, fe_const(kD2)));
      qext_add_niels(acc, q, nb);  // (meaningful in quad 0 only)
    }
    good &= (bool)((int)fe_is_zero_canon(fe_canon(acc.u)) & (int)fe_equal(acc.v, acc.z));
    if (NCHAIN > 1) __builtin_amdgcn_wave_barrier();  // table slots are rebuilt next round
  }
  if ((threadIdx.x & 7) == 0) ok[i] = good ? 1 : 0;
}

// The c-independent prefix of the kernel above — the window tables of PK and R (65 multiplications
// on ONE lane each, ~7 % of the kernel's critical path) — as a kernel of its own, so that the entry
// points can run it on a second stream beside k_challenge.  Two lanes per item.
__global__ void __launch_bounds__(64)
k_prep_var_tables(const uint8_t* __restrict__ PK_uv, const uint8_t* __restrict__ R_uv, size_t n,
                  u32* __restrict__ var_tables) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = t >> 1;
  if (i >= n) return;
  const bool r = (t & 1) != 0;
  Fe pu, pv;
  load_fq(pu, r ? R_uv : PK_uv, 2 * i);
  load_fq(pv, r ? R_uv : PK_uv, 2 * i + 1);
  build_var_table(var_tables + i * (2 * kVarLaneWords) + (r ? kVarLaneWords : 0), pu, pv);
}
void launch_prep_var_tables(const uint8_t* PK_uv, const uint8_t* R_uv, size_t n, uint32_t* var_tables,
                            hipStream_t s) {
  hipLaunchKernelGGL(k_prep_var_tables, dim3(grid_for(2 * n, 64)), dim3(64), 0, s, PK_uv, R_uv, n, var_tables);
}

void launch_verify_half_quad(int nchain, bool accumulate, bool tables_ready, const uint8_t* u,
                             const uint8_t* c, ChainOperands op0, ChainOperands op1,
                             const uint8_t* valid, size_t n, uint8_t* ok, uint32_t* var_tables,
                             hipStream_t s, const uint32_t* gate) {
  const dim3 grid((unsigned)((8 * n + kQuadBlock - 1) / kQuadBlock)), block(kQuadBlock);
  if (nchain == 2)
    hipLaunchKernelGGL(k_verify_fixed_half_oct<2>, grid, block, 0, s, u, c, op0, op1, valid,
                       accumulate, tables_ready, n, ok, var_tables, gate);
  else
    hipLaunchKernelGGL(k_verify_fixed_half_oct<1>, grid, block, 0, s, u, c, op0, op1, valid,
                       accumulate, tables_ready, n, ok, var_tables, gate);
}

}  // namespace dsv
