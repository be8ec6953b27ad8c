// dsv_inputs.hip — input generation for tests and benchmarks (signing, key derivation, the
// bit-compatible StdRng stream: SURVEY.md §8(f)-1, (f)-3) and the debug probes of include/dsv.h.
#include "dsv_constants.h"
#include "dsv_host.h"

using namespace dsvh;

extern "C" {

// ---- signing / key derivation: INPUT GENERATION for tests and benchmarks ----------------------
// NOT a replacement for SecretKey::sign in production: the fixed- and variable-base multiplications
// index tables in global memory with digits of the secret scalar (addresses depend on secrets;
// dusk-jubjub's multiplication is constant-time), and secrets pass through library-owned staging.
// The host entry points scrub that staging before they return; the *_dev ones never own secrets.
namespace {
void launch_sign_single(Context& ctx, const void* sk, const void* m, const void* r, size_t n, void* u,
                        void* R_uv, hipStream_t s) {
  launch_fixed_base_points((const uint8_t*)r, (const u32*)ctx.table[0], n, (uint8_t*)R_uv, s);
  // scratch use: c is written to u (32 B per item) before k_sign_finish overwrites it in place
  launch_challenge(false, (const uint8_t*)R_uv, (const uint8_t*)nullptr, (const uint8_t*)m, n, (uint8_t*)u, (uint8_t*)nullptr, s);
  launch_sign_finish((const uint8_t*)r, (const uint8_t*)u, (const uint8_t*)sk, n, (uint8_t*)u, s);
}
void launch_sign_double(Context& ctx, const void* sk, const void* m, const void* r, size_t n, void* u,
                        void* R_uv, void* Rp_uv, hipStream_t s) {
  launch_fixed_base_points((const uint8_t*)r, (const u32*)ctx.table[0], n, (uint8_t*)R_uv, s);
  launch_fixed_base_points((const uint8_t*)r, (const u32*)ctx.table[1], n, (uint8_t*)Rp_uv, s);
  launch_challenge(true, (const uint8_t*)R_uv, (const uint8_t*)Rp_uv, (const uint8_t*)m, n, (uint8_t*)u, (uint8_t*)nullptr, s);
  launch_sign_finish((const uint8_t*)r, (const uint8_t*)u, (const uint8_t*)sk, n, (uint8_t*)u, s);
}
// host-side check of what the kernels would otherwise poison: scalars must be < r
int check_canonical_scalars(const uint8_t* s, size_t n, const char* what) {
  static const uint32_t kR[8] = DSV_R32;
  for (size_t i = 0; i < n; i++) {
    uint32_t w[8];
    memcpy(w, s + 32 * i, 32);
    bool lt = false;
    for (int k = 7; k >= 0; k--) {
      if (w[k] != kR[k]) {
        lt = w[k] < kR[k];
        break;
      }
    }
    if (!lt) return fail(DSV_ERR_INVALID_ARGUMENT, "%s[%zu] is not a canonical JubJubScalar (>= r)", what, i);
  }
  return DSV_OK;
}
}  // namespace

int dsv_public_keys_dev(const void* sk, int which, size_t n, void* PK_uv, void* stream) {
  if (n && (!sk || !PK_uv || which < 0 || which > 1)) return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  DSV_DEV_PROLOGUE(n, PK_uv);
  launch_fixed_base_points((const uint8_t*)sk, (const u32*)ctx.table[which], n, (uint8_t*)PK_uv, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_sign_single_dev(const void* sk, const void* m, const void* r, size_t n, void* u, void* R_uv,
                        void* stream) {
  if (n && (!sk || !m || !r || !u || !R_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, u);
  launch_sign_single(ctx, sk, m, r, n, u, R_uv, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_sign_double_dev(const void* sk, const void* m, const void* r, size_t n, void* u, void* R_uv,
                        void* Rp_uv, void* stream) {
  if (n && (!sk || !m || !r || !u || !R_uv || !Rp_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, u);
  launch_sign_double(ctx, sk, m, r, n, u, R_uv, Rp_uv, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// variable-base forms (var-generator scheme): PK = sk * Gen; R = r * Gen, c, u.  `workspace`:
// dsv_workspace_bytes(n) device bytes (the per-lane window tables live there)
int dsv_public_keys_vargen_dev(const void* sk, const void* Gen_uv, size_t n, void* PK_uv,
                               void* workspace, void* stream) {
  if (n && (!sk || !Gen_uv || !PK_uv || !workspace)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, PK_uv);
  launch_var_base_points((const uint8_t*)sk, (const uint8_t*)Gen_uv, n, (uint8_t*)PK_uv, reinterpret_cast<u32*>(workspace), (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_sign_vargen_dev(const void* sk, const void* Gen_uv, const void* m, const void* r, size_t n,
                        void* u, void* R_uv, void* workspace, void* stream) {
  if (n && (!sk || !Gen_uv || !m || !r || !u || !R_uv || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, u);
  hipStream_t s = (hipStream_t)stream;
  launch_var_base_points((const uint8_t*)r, (const uint8_t*)Gen_uv, n, (uint8_t*)R_uv, reinterpret_cast<u32*>(workspace), s);
  launch_challenge(false, (const uint8_t*)R_uv, (const uint8_t*)nullptr, (const uint8_t*)m, n, (uint8_t*)u, (uint8_t*)nullptr, s);
  launch_sign_finish((const uint8_t*)r, (const uint8_t*)u, (const uint8_t*)sk, n, (uint8_t*)u, s);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

int dsv_public_keys(const uint8_t* sk, int which, const uint8_t* gen_uv, size_t n, uint8_t* PK_uv) {
  if (n && (!sk || !PK_uv || which < 0 || which > 1)) return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  DSV_HOST_PROLOGUE(n);
  if (int r = check_canonical_scalars(sk, n, "sk")) return r;
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, align_up(n * 32, 256) + 2 * align_up(n * 64, 256) +
                                    var_table_bytes(n, 1) + 256))
    return r;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dg = st.take(n * 64), *dpk = st.take(n * 64),
          *dtab = st.take(var_table_bytes(n, 1));
  H2D(dsk, sk, n * 32);
  if (gen_uv) {
    H2D(dg, gen_uv, n * 64);
    launch_var_base_points((const uint8_t*)dsk, (const uint8_t*)dg, n, dpk, reinterpret_cast<u32*>(dtab), 0);
  } else {
    launch_fixed_base_points((const uint8_t*)dsk, (const u32*)ctx.table[which], n, dpk, 0);
  }
  HIP_TRY(hipGetLastError());
  D2H(PK_uv, dpk, n * 64);
  // scrub the secret keys and the window tables derived from them
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));
  if (gen_uv) HIP_TRY(hipMemsetAsync(dtab, 0, var_table_bytes(n, 1), 0));
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_sign_single(const uint8_t* sk, const uint8_t* m, const uint8_t* r, size_t n, uint8_t* u,
                    uint8_t* R_uv) {
  if (n && (!sk || !m || !r || !u || !R_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  if (int rc = check_canonical_scalars(sk, n, "sk")) return rc;
  if (int rc = check_canonical_scalars(r, n, "nonce")) return rc;
  DSV_HOST_LOCK();
  if (int rc = ensure_stage(ctx, 4 * align_up(n * 32, 256) + align_up(n * 64, 256))) return rc;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dm = st.take(n * 32), *dr = st.take(n * 32),
          *du = st.take(n * 32), *dR = st.take(n * 64);
  H2D(dsk, sk, n * 32);
  H2D(dm, m, n * 32);
  H2D(dr, r, n * 32);
  launch_sign_single(ctx, dsk, dm, dr, n, du, dR, 0);
  HIP_TRY(hipGetLastError());
  D2H(u, du, n * 32);
  D2H(R_uv, dR, n * 64);
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));  // scrub key and nonce
  HIP_TRY(hipMemsetAsync(dr, 0, n * 32, 0));
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_sign_double(const uint8_t* sk, const uint8_t* m, const uint8_t* r, size_t n, uint8_t* u,
                    uint8_t* R_uv, uint8_t* Rp_uv) {
  if (n && (!sk || !m || !r || !u || !R_uv || !Rp_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  if (int rc = check_canonical_scalars(sk, n, "sk")) return rc;
  if (int rc = check_canonical_scalars(r, n, "nonce")) return rc;
  DSV_HOST_LOCK();
  if (int rc = ensure_stage(ctx, 4 * align_up(n * 32, 256) + 2 * align_up(n * 64, 256))) return rc;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dm = st.take(n * 32), *dr = st.take(n * 32),
          *du = st.take(n * 32), *dR = st.take(n * 64), *dRp = st.take(n * 64);
  H2D(dsk, sk, n * 32);
  H2D(dm, m, n * 32);
  H2D(dr, r, n * 32);
  launch_sign_double(ctx, dsk, dm, dr, n, du, dR, dRp, 0);
  HIP_TRY(hipGetLastError());
  D2H(u, du, n * 32);
  D2H(R_uv, dR, n * 64);
  D2H(Rp_uv, dRp, n * 64);
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));
  HIP_TRY(hipMemsetAsync(dr, 0, n * 32, 0));
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_sign_vargen(const uint8_t* sk, const uint8_t* Gen_uv, const uint8_t* m, const uint8_t* r,
                    size_t n, uint8_t* u, uint8_t* R_uv) {
  if (n && (!sk || !Gen_uv || !m || !r || !u || !R_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  if (int rc = check_canonical_scalars(sk, n, "sk")) return rc;
  if (int rc = check_canonical_scalars(r, n, "nonce")) return rc;
  DSV_HOST_LOCK();
  if (int rc = ensure_stage(ctx, 4 * align_up(n * 32, 256) + 2 * align_up(n * 64, 256) +
                                     var_table_bytes(n, 1) + 256))
    return rc;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dm = st.take(n * 32), *dr = st.take(n * 32),
          *du = st.take(n * 32), *dG = st.take(n * 64), *dR = st.take(n * 64),
          *dtab = st.take(var_table_bytes(n, 1));
  H2D(dsk, sk, n * 32);
  H2D(dm, m, n * 32);
  H2D(dr, r, n * 32);
  H2D(dG, Gen_uv, n * 64);
  launch_var_base_points((const uint8_t*)dr, (const uint8_t*)dG, n, dR, reinterpret_cast<u32*>(dtab), 0);
  launch_challenge(false, (const uint8_t*)dR, (const uint8_t*)nullptr, (const uint8_t*)dm, n, du, (uint8_t*)nullptr, 0);
  launch_sign_finish((const uint8_t*)dr, (const uint8_t*)du, (const uint8_t*)dsk, n, du, 0);
  HIP_TRY(hipGetLastError());
  D2H(u, du, n * 32);
  D2H(R_uv, dR, n * 64);
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));
  HIP_TRY(hipMemsetAsync(dr, 0, n * 32, 0));
  HIP_TRY(hipMemsetAsync(dtab, 0, var_table_bytes(n, 1), 0));  // multiples of Gen, not secret — cheap anyway
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}


// ---- reference-harness inputs ---------------------------------------------------------------
namespace {
// rand_core 0.6 SeedableRng::seed_from_u64: eight PCG32 outputs = the ChaCha key
ChaChaKey stdrng_key(uint64_t state) {
  ChaChaKey k;
  for (int i = 0; i < 8; i++) {
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    const uint32_t rot = (uint32_t)(state >> 59);
    k.w[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
  }
  return k;
}
}  // namespace

int dsv_stdrng_sign_inputs_dev(uint64_t seed, size_t first_item, size_t n, void* sk, void* m,
                               void* r, void* stream) {
  if (n && (!sk || !m || !r)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, sk);
  launch_stdrng_triples(stdrng_key(seed), first_item, n, (uint8_t*)sk, (uint8_t*)m, (uint8_t*)r, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_stdrng_sign_inputs(uint64_t seed, size_t first_item, size_t n, uint8_t* sk, uint8_t* m,
                           uint8_t* r) {
  if (n && (!sk || !m || !r)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int rc = ensure_stage(ctx, 3 * align_up(n * 32, 256))) return rc;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dm = st.take(n * 32), *dr = st.take(n * 32);
  launch_stdrng_triples(stdrng_key(seed), first_item, n, dsk, dm, dr, 0);
  HIP_TRY(hipGetLastError());
  D2H(sk, dsk, n * 32);
  D2H(m, dm, n * 32);
  D2H(r, dr, n * 32);
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));
  HIP_TRY(hipMemsetAsync(dr, 0, n * 32, 0));
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_stdrng_vargen_inputs_dev(uint64_t seed, size_t first_item, size_t n, void* sk, void* g,
                                 void* m, void* r, void* stream) {
  if (n && (!sk || !g || !m || !r)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, sk);
  launch_stdrng_quads(stdrng_key(seed), first_item, n, (uint8_t*)sk, (uint8_t*)g, (uint8_t*)m, (uint8_t*)r, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

int dsv_debug_table_entry(int which, int window, int digit, uint8_t out96[96]) {
  if (which < 0 || which > 1 || window < 0 || window >= kFixedWindows || digit < 0 ||
      digit >= kFixedEntries || !out96)
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad table coordinates");
  Context* ctxp = nullptr;
  if (int r = host_context(ctxp)) return r;
  Context& ctx = *ctxp;
  DSV_HOST_LOCK();
  u32 e[kEntryWords];
  HIP_TRY(hipMemcpy(e, ctx.table[which] + ((size_t)window * kFixedEntries + digit) * kEntryWords,
                    sizeof e, hipMemcpyDeviceToHost));
  // entries are Montgomery (R = 2^261) canonical limbs; hand back the raw limbs as 3 x 9 x 29-bit
  // packed LE integers so the test can undo the Montgomery factor with Python integers.
  for (int f = 0; f < 3; f++) {
    unsigned __int128 acc = 0;
    int bits = 0, o = 0;
    uint8_t* dst = out96 + 32 * f;
    memset(dst, 0, 32);
    for (int i = 0; i < kLimbs; i++) {
      acc |= (unsigned __int128)e[f * kLimbs + i] << bits;
      bits += 29;
      while (bits >= 8 && o < 32) {
        dst[o++] = (uint8_t)acc;
        acc >>= 8;
        bits -= 8;
      }
    }
    while (o < 32) {
      dst[o++] = (uint8_t)acc;
      acc >>= 8;
    }
  }
  return DSV_OK;
}

int dsv_fixed_window_bits(void) { return kFixedBits; }

int dsv_debug_lattice3(const uint8_t* u, const uint8_t* c, size_t n, uint8_t* out128) {
  if (n && (!u || !c || !out128)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, 2 * align_up(n * 32, 256) + align_up(n * 128, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *du = st.take(n * 32), *dc = st.take(n * 32), *dout = st.take(n * 128);
  H2D(du, u, n * 32);
  H2D(dc, c, n * 32);
  launch_debug_lattice3(du, dc, n, dout, 0);
  HIP_TRY(hipGetLastError());
  D2H(out128, dout, n * 128);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

int dsv_debug_half_scalars(const uint8_t* c, size_t n, uint8_t* out96) {
  if (n && (!c || !out96)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, align_up(n * 32, 256) + align_up(n * 96, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *dc = st.take(n * 32), *dout = st.take(n * 96);
  H2D(dc, c, n * 32);
  launch_debug_half_scalars(dc, n, dout, 0);
  HIP_TRY(hipGetLastError());
  D2H(out96, dout, n * 96);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

int dsv_debug_fq_mul(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  if (n && (!a || !b || !out)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, 3 * align_up(n * 32, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *da = st.take(n * 32), *db = st.take(n * 32), *dout = st.take(n * 32);
  H2D(da, a, n * 32);
  H2D(db, b, n * 32);
  launch_debug_fq_mul((const uint8_t*)da, (const uint8_t*)db, n, dout, 0);
  HIP_TRY(hipGetLastError());
  D2H(out, dout, n * 32);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

}  // extern "C"
