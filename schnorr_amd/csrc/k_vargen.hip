// k_vargen.hip — both bases variable: `PublicKeyVarGen::verify`
// (/root/reference/src/keys/public.rs:401-415): ok = valid & [ u*Gen + c*PK == R ], and the
// variable-base multiplication of the var-generator signer / key derivation
// (/root/reference/src/keys/secret.rs:442, src/keys/public.rs:337-344).
#include "common.h"
#include "decode29.h"
#include "inv29.h"
#include "lattice3.h"
#include "quad29.h"

namespace dsv {

// ok = valid & [ x*Gen + y*PK - z*R == O ] with (x, y, z) the short lattice vector of lattice3.h
// (x = z*u, y = z*c mod 8r, z odd): a three-base Straus chain of ~43 signed 4-bit windows instead of
// the reference equation's two-base chain of 63 (r01 / r02).  Three per-lane window tables (Gen, PK,
// R); the entry of the NEXT addition is loaded while the current one runs, so two entries are live
// at any time, as in the two-base kernels.
__global__ void __launch_bounds__(kVerifyBlock, kWavesVerify)
k_verify_var(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c,
             const uint8_t* __restrict__ PK_uv, const uint8_t* __restrict__ Gen_uv,
             const uint8_t* __restrict__ R_uv, const uint8_t* __restrict__ valid, size_t n,
             uint8_t* __restrict__ ok, u32* __restrict__ var_tables, const u32* __restrict__ gate) {
  if (gate_says_done(gate)) return;  // the batch fast accept decided these items (launch.h)
  u32* tg = var_tables + ((size_t)blockIdx.x * kVerifyBlock + threadIdx.x) * (3 * kVarLaneWords);
  u32* tp = tg + kVarLaneWords;
  u32* tr = tp + kVarLaneWords;
#pragma unroll 1
  for (size_t base = (size_t)blockIdx.x * kVerifyBlock; base < n;
       base += (size_t)gridDim.x * kVerifyBlock) {
    const size_t i = base + threadIdx.x;
    if (i >= n) continue;
    bool good = valid[i] != 0;
    u32 yx[8], yy[8], yz[8];
    int sx, sy, sz, top;
    {
      u32 us[8], cs[8], mx[8], my[8], mz[8];
      bool nx, ny, nz;
      load_words8(us, u, i);
      load_words8(cs, c, i);
      good &= words_lt(us, kR32);
      if (!words_lt(us, kR32)) us[7] &= 0x0fffffffu;  // keep the scalars in range; verdict is 0 anyway
      lattice3_scalars(mx, my, mz, nx, ny, nz, us, cs);
      recode_signed4(yx, mx);
      recode_signed4(yy, my);
      recode_signed4(yz, mz);
      u32 nzd[8];
#pragma unroll
      for (int k = 0; k < 8; k++)
        nzd[k] = (yx[k] ^ 0x88888888u) | (yy[k] ^ 0x88888888u) | (yz[k] ^ 0x88888888u);
      top = top_digit4(nzd);
      sx = nx ? -1 : 1;
      sy = ny ? -1 : 1;
      sz = nz ? 1 : -1;  // the chain adds (-z) * R
    }
    {
      Fe gu, gv;
      good &= load_fq(gu, Gen_uv, 2 * i);
      good &= load_fq(gv, Gen_uv, 2 * i + 1);
      build_var_table(tg, gu, gv);
    }
    {
      Fe pku, pkv;
      good &= load_fq(pku, PK_uv, 2 * i);
      good &= load_fq(pkv, PK_uv, 2 * i + 1);
      build_var_table(tp, pku, pkv);
    }
    {
      Fe ru, rv;
      good &= load_fq(ru, R_uv, 2 * i);
      good &= load_fq(rv, R_uv, 2 * i + 1);
      build_var_table(tr, ru, rv);
    }
    Ext acc = ext_from_niels(load_var_entry(tg, sx * sdigit4(yx, top)));
    acc = ext_add_niels(acc, load_var_entry(tp, sy * sdigit4(yy, top)));
    acc = ext_add_niels(acc, load_var_entry(tr, sz * sdigit4(yz, top)));
    {
      const int k0 = top > 0 ? top - 1 : 0;
      RawNiels ea = load_var_entry_raw(tg, sx * sdigit4(yx, k0));
#pragma unroll 1
      for (int k = top - 1; k >= 0; k--) {
        acc = ext_mul16(acc);
        const RawNiels eb = load_var_entry_raw(tp, sy * sdigit4(yy, k));
        acc = ext_add_niels(acc, finish_var_entry(ea));
        const RawNiels ec = load_var_entry_raw(tr, sz * sdigit4(yz, k));
        acc = ext_add_niels(acc, finish_var_entry(eb));
        const int kn = k > 0 ? k - 1 : 0;  // last round: reloads its own entry, unused
        ea = load_var_entry_raw(tg, sx * sdigit4(yx, kn));
        acc = ext_add_niels(acc, finish_var_entry(ec));
      }
    }
    // T == O  <=>  u == 0 and v == z
    good &= (bool)((int)fe_is_zero_canon(fe_canon(acc.u)) & (int)fe_equal(acc.v, acc.z));
    ok[i] = good ? 1 : 0;
  }
}

// The same verdict with SIXTEEN lanes per signature (small batches and the first chunk of a host call
// that finds the GPU idle: below ~2^14 items the one-lane kernel above is one wave's serial instruction
// stream — 1.2 ms whatever the count — on a chip that is mostly idle).  One DPP row of 16 lanes owns a
// signature: quad 0 runs x*Gen, quad 1 y*PK, quad 2 -z*R — the same 4-doublings / 1-addition chain on
// three operands, each quad's four lanes sharing the four products of every group operation (quad29.h:
// two multiplication-times per operation instead of 7 / 8) — and quad 3 keeps step on quad 2's operands
// (its result is not read).  Then quad 0 += quad 1 += quad 2 through row shifts.  Same lattice vector,
// same tables, same digits as k_verify_var: 44 windows of (4 doublings + 1 addition) on the critical
// path instead of 44 x (4 + 3), each at 2 / 7 of the latency.  Throughput per lane is ~0.25 of the
// one-lane kernel's, so only launches of at most kVarHexMaxItems items come here (launch.h).
DSV_DEV Fe from_quad_up(const Fe& x, int quads) {  // the value held 4 * quads lanes up in the row
  Fe r;
#pragma unroll
  for (int i = 0; i < NL; i++)
    r.l[i] = quads == 1 ? (u32)__builtin_amdgcn_update_dpp(0, (int)x.l[i], 0x104 /* row_shl:4 */, 0xf, 0xf, true)
                        : (u32)__builtin_amdgcn_update_dpp(0, (int)x.l[i], 0x108 /* row_shl:8 */, 0xf, 0xf, true);
  return r;
}
DSV_DEV Niels qext_to_niels(const QExt& p) {
  Niels n;
  n.vpu = fe_carry(fe_add(p.v, p.u));
  n.vmu = fe_sub2(p.v, p.u);
  n.z = p.z;
  n.t2d = fe_mul(p.t, fe_const(kD2));
  return n;
}
__global__ void __launch_bounds__(kQuadBlock)
k_verify_var_hex(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c,
                 const uint8_t* __restrict__ PK_uv, const uint8_t* __restrict__ Gen_uv,
                 const uint8_t* __restrict__ R_uv, const uint8_t* __restrict__ valid, size_t n,
                 uint8_t* __restrict__ ok, u32* __restrict__ var_tables, const u32* __restrict__ gate) {
  if (gate_says_done(gate)) return;
  const size_t i = ((size_t)blockIdx.x * kQuadBlock + threadIdx.x) >> 4;
  const int q = threadIdx.x & 3, quad = (threadIdx.x >> 2) & 3;
  if (i >= n) return;  // whole rows leave together
  const int base = quad < 2 ? quad : 2;  // 0 Gen, 1 PK, 2 R (quad 3 shadows quad 2)
  u32* tbl = var_tables + i * (3 * kVarLaneWords) + (size_t)base * kVarLaneWords;
  bool good = valid[i] != 0;
  u32 ys[8];
  int sgn, top;
  {
    u32 us[8], cs[8], mx[8], my[8], mz[8], yx[8], yy[8], yz[8];
    bool nx, ny, nz;
    load_words8(us, u, i);
    load_words8(cs, c, i);
    good &= words_lt(us, kR32);
    if (!words_lt(us, kR32)) us[7] &= 0x0fffffffu;  // keep the scalars in range; verdict is 0 anyway
    lattice3_scalars(mx, my, mz, nx, ny, nz, us, cs);
    recode_signed4(yx, mx);
    recode_signed4(yy, my);
    recode_signed4(yz, mz);
    u32 nzd[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
      nzd[k] = (yx[k] ^ 0x88888888u) | (yy[k] ^ 0x88888888u) | (yz[k] ^ 0x88888888u);
    top = top_digit4(nzd);  // all quads walk the same number of windows
#pragma unroll
    for (int k = 0; k < 8; k++) ys[k] = base == 0 ? yx[k] : (base == 1 ? yy[k] : yz[k]);
    sgn = base == 0 ? (nx ? -1 : 1) : (base == 1 ? (ny ? -1 : 1) : (nz ? 1 : -1));  // the chain adds (-z) * R
  }
  {
    // every lane checks all six coordinates; lane 0 of quads 0 .. 2 builds its base's window table
    Fe gu, gv, pu, pv, ru, rv;
    good &= load_fq(gu, Gen_uv, 2 * i);
    good &= load_fq(gv, Gen_uv, 2 * i + 1);
    good &= load_fq(pu, PK_uv, 2 * i);
    good &= load_fq(pv, PK_uv, 2 * i + 1);
    good &= load_fq(ru, R_uv, 2 * i);
    good &= load_fq(rv, R_uv, 2 * i + 1);
    if (q == 0 && quad < 3)
      build_var_table(tbl, base == 0 ? gu : (base == 1 ? pu : ru), base == 0 ? gv : (base == 1 ? pv : rv));
    // the other lanes of the wave read what these lanes wrote to global memory
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
  }
  QExt acc = qext_identity();
  qext_add_niels(acc, q, load_var_entry(tbl, sgn * sdigit4(ys, top)));
#pragma unroll 1
  for (int k = top - 1; k >= 0; k--) {
#pragma unroll 1
    for (int j = 0; j < 3; j++) qext_double<false>(acc, q);
    qext_double<true>(acc, q);
    qext_add_niels(acc, q, load_var_entry(tbl, sgn * sdigit4(ys, k)));
  }
  {  // quad 0 += quad 1, += quad 2: their results as extended niels operands, moved down the row
    const Niels mine = qext_to_niels(acc);
#pragma unroll 1
    for (int up = 1; up <= 2; up++) {
      Niels nb;
      nb.vpu = from_quad_up(mine.vpu, up);
      nb.vmu = from_quad_up(mine.vmu, up);
      nb.z = from_quad_up(mine.z, up);
      nb.t2d = from_quad_up(mine.t2d, up);
      qext_add_niels(acc, q, nb);  // (meaningful in quad 0 only)
    }
  }
  good &= (bool)((int)fe_is_zero_canon(fe_canon(acc.u)) & (int)fe_equal(acc.v, acc.z));
  if ((threadIdx.x & 15) == 0) ok[i] = good ? 1 : 0;
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
      Fe zi = fe_invert_euclid(acc.z);
      store_fq(out_uv, 2 * i, fe_mul(acc.u, zi));
      store_fq(out_uv, 2 * i + 1, fe_mul(acc.v, zi));
    } else {
      store_poison(out_uv, 2 * i);
      store_poison(out_uv, 2 * i + 1);
    }
  }
}

// introspection for tests: the scalars the kernel above would use for (u, c): out = |x| || |y| ||
// |z| (3 x 32 bytes LE) || sign bytes of x, y, z || 29 zero bytes (128 B per item)
__global__ void __launch_bounds__(64)
k_debug_lattice3(const uint8_t* __restrict__ u, const uint8_t* __restrict__ c, size_t n,
                 uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 us[8], cs[8], mx[8], my[8], mz[8];
  bool nx, ny, nz;
  load_words8(us, u, i);
  load_words8(cs, c, i);
  us[7] &= 0x0fffffffu;
  cs[7] &= 0x03ffffffu;
  lattice3_scalars(mx, my, mz, nx, ny, nz, us, cs);
  store_words8(out, 4 * i, mx);
  store_words8(out, 4 * i + 1, my);
  store_words8(out, 4 * i + 2, mz);
  const u32 sg[8] = {(nx ? 1u : 0u) | (ny ? 0x100u : 0u) | (nz ? 0x10000u : 0u), 0, 0, 0, 0, 0, 0, 0};
  store_words8(out, 4 * i + 3, sg);
}
// the same for halfgcd.h: out = |a| || |b| (2 x 32 bytes LE) || sign byte of b || 31 zero bytes (96 B)
__global__ void __launch_bounds__(64)
k_debug_half_scalars(const uint8_t* __restrict__ c, size_t n, uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 cs[8], a[8], b[8];
  bool b_neg;
  load_words8(cs, c, i);
  cs[7] &= 0x03ffffffu;
  half_scalars(a, b, b_neg, cs);
  store_words8(out, 3 * i, a);
  store_words8(out, 3 * i + 1, b);
  const u32 sg[8] = {b_neg ? 1u : 0u, 0, 0, 0, 0, 0, 0, 0};
  store_words8(out, 3 * i + 2, sg);
}
void launch_debug_half_scalars(const uint8_t* c, size_t n, uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_half_scalars, dim3(grid_for(n, 64)), dim3(64), 0, s, c, n, out);
}
void launch_debug_lattice3(const uint8_t* u, const uint8_t* c, size_t n, uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_lattice3, dim3(grid_for(n, 64)), dim3(64), 0, s, u, c, n, out);
}

void launch_verify_var(const uint8_t* u, const uint8_t* c, const uint8_t* PK_uv, const uint8_t* Gen_uv,
                       const uint8_t* R_uv, const uint8_t* valid, size_t n, uint8_t* ok,
                       uint32_t* var_tables, hipStream_t s, const uint32_t* gate) {
  static const bool hex_on = !(getenv("DSV_QUAD") && atoi(getenv("DSV_QUAD")) == 0);  // (the switch of the fixed-base oct kernel)
  if (hex_on && n <= kVarHexMaxItems) {
    hipLaunchKernelGGL(k_verify_var_hex, dim3((unsigned)((16 * n + kQuadBlock - 1) / kQuadBlock)), dim3(kQuadBlock), 0, s, u,
                       c, PK_uv, Gen_uv, R_uv, valid, n, ok, var_tables, gate);
    return;
  }
  hipLaunchKernelGGL(k_verify_var, dim3(verify_grid(n)), dim3(kVerifyBlock), 0, s, u, c, PK_uv, Gen_uv,
                     R_uv, valid, n, ok, var_tables, gate);
}
void launch_var_base_points(const uint8_t* scalar, const uint8_t* P_uv, size_t n, uint8_t* out_uv,
                            uint32_t* var_tables, hipStream_t s) {
  hipLaunchKernelGGL(k_var_base_points, dim3(verify_grid(n)), dim3(kVerifyBlock), 0, s, scalar, P_uv, n,
                     out_uv, var_tables);
}

}  // namespace dsv
