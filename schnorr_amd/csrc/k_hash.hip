// k_hash.hip — the challenge hash c = trunc250(Poseidon(R.u, R.v[, R'.u, R'.v], m)) of
// `challenge_hash` / `challenge_hash_double` (/root/reference/src/signatures.rs:127-134, :275-290),
// about a fifth of a verification.  Arithmetic: hades29.h (S-boxes on the VALU) + hades_mfma.h (every
// product with a constant as an exact int8 product on the matrix cores).
#define DSV_HOST_TABLES 1  // this unit uploads the round constants
#include "common.h"
#include "hades29.h"

namespace dsv {

// DOUBLE: the 5-input sponge = two permutations ([0, Ru, Rv, R'u, R'v] -> perm -> word 1 += m,
// word 2 += 1 -> perm).  r02 inlined the permutation twice and kept m live across the first one:
// 190 spilled VGPRs, 764 B of scratch per lane, 4.6 % SLOWER than two single hashes.  Now the two
// permutations are two trips through ONE copy of the permutation body with only the 5-word state
// live between them; m is loaded (and checked) when it is absorbed.  Both trips take the generic
// first round (the capacity word enters as an ordinary 0: one S-box more than the constant-folded
// form, 1 of 198), the second one ends with the one-row last round.
template <bool DOUBLE>
__global__ void __launch_bounds__(256, kWavesHash)
k_challenge(const uint8_t* __restrict__ R_uv, const uint8_t* __restrict__ Rp_uv,
            const uint8_t* __restrict__ m, size_t n, uint8_t* __restrict__ c_out,
            uint8_t* __restrict__ valid, const uint8_t* __restrict__ valid_in) {
  // the hashes of a wave cooperate through the matrix cores: every lane runs, spare lanes redo
  // the last item and skip the stores
  hades_mfma_load_table();
  const size_t i_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i_raw < n;
  const size_t i = live ? i_raw : n - 1;
  Fe s[5];
  s[0] = fe_zero();
  bool ok = load_fq(s[1], R_uv, 2 * i);
  ok &= load_fq(s[2], R_uv, 2 * i + 1);
  if (DOUBLE) {
    ok &= load_fq(s[3], Rp_uv, 2 * i);
    ok &= load_fq(s[4], Rp_uv, 2 * i + 1);
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
      if (pass) {
        Fe mm;
        ok &= load_fq(mm, m, i);
        s[1] = fe_add(s[1], mm);
        s[2] = fe_add(s[2], fe_one());
      }
      hades_permute<0>(s, pass != 0);
    }
  } else {
    ok &= load_fq(s[3], m, i);
    s[4] = fe_one();
    hades_permute<2>(s, true);
  }
  u32 c[8];
  poseidon_truncate(c, s[1]);
  if (!live) return;
  store_words8(c_out, i, c);
  // valid_in (may be null): what an earlier stage found out about the item — the normalisation of
  // projective input (z = 0, a coordinate >= q), point decompression — folded in here instead of a
  // kernel of its own AND-ing the verdicts afterwards (r05: one launch less per sub-batch)
  if (valid) valid[i] = (ok && (!valid_in || valid_in[i] != 0)) ? 1 : 0;
}

hipError_t hash_upload_constants() {
  // __constant__ symbols exist once per device: these copies go to the device currently selected
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_hades_rc), DSV_HADES_RC_HOST, sizeof(DSV_HADES_RC_HOST));
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol(HIP_SYMBOL(c_hades_k0), DSV_HADES_K0_HOST, sizeof(DSV_HADES_K0_HOST));
}

void launch_challenge(bool dbl, const uint8_t* R_uv, const uint8_t* Rp_uv, const uint8_t* m, size_t n,
                      uint8_t* c, uint8_t* valid, hipStream_t s, const uint8_t* valid_in) {
  if (dbl)
    hipLaunchKernelGGL(k_challenge<true>, dim3(grid_for(n)), dim3(256), 0, s, R_uv, Rp_uv, m, n, c, valid, valid_in);
  else
    hipLaunchKernelGGL(k_challenge<false>, dim3(grid_for(n)), dim3(256), 0, s, R_uv, Rp_uv, m, n, c, valid, valid_in);
}

}  // namespace dsv
