// k_vargen.hip — both bases variable: `PublicKeyVarGen::verify`
// (/root/reference/src/keys/public.rs:401-415): ok = valid & [ u*Gen + c*PK == R ], and the
// variable-base multiplication of the var-generator signer / key derivation
// (/root/reference/src/keys/secret.rs:442, src/keys/public.rs:337-344).
#include "common.h"
#include "decode29.h"

namespace dsv {

// a*P + b*Q with one shared doubling chain (Straus); a < 2^252, b < 2^252; the entries of the next
// window are loaded one group operation ahead
DSV_DEV Ext var_base_mul2(const u32 (&a)[8], const u32* tp, const u32 (&b)[8], const u32* tq) {
  u32 ya[8], yb[8];
  recode_signed4(ya, a);
  recode_signed4(yb, b);
  Ext acc = ext_from_niels(load_var_entry(tp, sdigit4(ya, 63)));
  acc = ext_add_niels(acc, load_var_entry(tq, sdigit4(yb, 63)));
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
  return acc;
}

__global__ void __launch_bounds__(kVerifyBlock, kWavesVerify)
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

// out = scalar * P for a per-item base P, affine
__global__ void __launch_bounds__(kVerifyBlock, kWavesVerify)
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
      Fe zi = fe_invert(acc.z);
      store_fq(out_uv, 2 * i, fe_mul(acc.u, zi));
      store_fq(out_uv, 2 * i + 1, fe_mul(acc.v, zi));
    } else {
      store_poison(out_uv, 2 * i);
      store_poison(out_uv, 2 * i + 1);
    }
  }
}

void launch_verify_var(const uint8_t* u, const uint8_t* c, const uint8_t* PK_uv, const uint8_t* Gen_uv,
                       const uint8_t* R_uv, const uint8_t* valid, size_t n, uint8_t* ok,
                       uint32_t* var_tables, hipStream_t s) {
  hipLaunchKernelGGL(k_verify_var, dim3(verify_grid(n)), dim3(kVerifyBlock), 0, s, u, c, PK_uv, Gen_uv,
                     R_uv, valid, n, ok, var_tables);
}
void launch_var_base_points(const uint8_t* scalar, const uint8_t* P_uv, size_t n, uint8_t* out_uv,
                            uint32_t* var_tables, hipStream_t s) {
  hipLaunchKernelGGL(k_var_base_points, dim3(verify_grid(n)), dim3(kVerifyBlock), 0, s, scalar, P_uv, n,
                     out_uv, var_tables);
}

}  // namespace dsv
