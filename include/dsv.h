/* dsv.h — C ABI of the MI355X batch Schnorr-verify engine (libdsv.so).
 *
 * Drop-in boundary for the native verification path of dusk-schnorr 0.18.  The reference has
 * no FFI of its own; the entry points below are what a Rust `extern "C"` block in that crate
 * would bind to offload its three verify functions (binding shown in INTEGRATION.md):
 *
 *   dsv_verify_single   replaces  PublicKey::verify        /root/reference/src/keys/public.rs:121-130
 *   dsv_verify_double   replaces  PublicKeyDouble::verify  /root/reference/src/keys/public.rs:222-244
 *   dsv_verify_vargen   replaces  PublicKeyVarGen::verify  /root/reference/src/keys/public.rs:401-415
 *   dsv_challenge_*     exposes   challenge_hash{,_double} /root/reference/src/signatures.rs:127-134, 275-290
 *   dsv_sign_*          replaces  SecretKey::sign / sign_double / SecretKeyVarGen::sign
 *                                                          /root/reference/src/keys/secret.rs:150-168, 217-240, 433-451
 *   dsv_public_keys     replaces  PublicKey::from(&SecretKey)  /root/reference/src/keys/public.rs:61-67, 265-272
 *
 * Data layout (all entry points): structure-of-arrays, caller-owned, one batch = n items.
 *   scalar  (JubJubScalar u, sk, c; BlsScalar message m) : 32 bytes, canonical little-endian
 *                                                           (what `to_bytes()` yields)
 *   point   (R, R', PK, PK', Gen)                         : 64 bytes = affine u || v, each a
 *                                                           canonical LE BlsScalar, i.e. the pair
 *                                                           `JubJubExtended::to_hash_inputs()` returns
 *   ext point (the *_ext entry points)                     : 96 bytes = u || v || z of a
 *                                                           JubJubExtended with arbitrary z != 0
 *   verdict ok[i]                                          : one byte, 1 = verify() true, 0 = false
 * Verdicts are bit-exact with the reference's `verify` for every input its types can hold
 * (any on-curve point incl. identity / small order, any scalar).  Encodings the Rust types
 * cannot hold (scalar >= r, coordinate or message >= q) give ok[i] = 0; off-curve
 * coordinates are out of contract (result unspecified, never a fault).
 *
 * Host entry points take HOST pointers (pageable is fine), stage through library-owned pinned
 * and device buffers in chunks of 2^17 items (DSV_HOST_THREADS copy threads, default 4) and block
 * until the verdicts are in `ok`; they serialise on one internal lock.  The *_dev entry points
 * take DEVICE pointers (hipMalloc'd, 16-byte aligned) plus a hipStream_t passed as void*, enqueue
 * only, and never synchronise — they are what the bench times with inputs resident in HBM.
 *
 * Return value: DSV_OK (0) or a negative dsv_status; dsv_last_error() gives the text for
 * the calling thread.  The library keeps no pointer after a call returns.  Calls on
 * different streams may run concurrently; dsv_init is idempotent and thread-safe.
 */
#ifndef DSV_H
#define DSV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  DSV_OK = 0,
  DSV_ERR_NOT_INITIALIZED = -1,
  DSV_ERR_INVALID_ARGUMENT = -2,
  DSV_ERR_HIP = -3,
  DSV_ERR_NO_DEVICE = -4,
  DSV_ERR_TOO_LARGE = -5
} dsv_status;

#define DSV_MAX_BATCH ((size_t)1 << 28)

/* ---- lifecycle ---- */
int dsv_init(int device);             /* select GPU, build the fixed-base tables for G and G' */
int dsv_shutdown(void);
const char *dsv_version(void);
const char *dsv_last_error(void);
int dsv_device_count(void);

/* ---- verify, host buffers ---- */
int dsv_verify_single(const uint8_t *u, const uint8_t *R_uv, const uint8_t *PK_uv,
                      const uint8_t *m, size_t n, uint8_t *ok);
int dsv_verify_double(const uint8_t *u, const uint8_t *R_uv, const uint8_t *Rp_uv,
                      const uint8_t *PK_uv, const uint8_t *PKp_uv, const uint8_t *m, size_t n,
                      uint8_t *ok);
int dsv_verify_vargen(const uint8_t *u, const uint8_t *R_uv, const uint8_t *PK_uv,
                      const uint8_t *Gen_uv, const uint8_t *m, size_t n, uint8_t *ok);
/* R and PK as projective (u, v, z), 96 B each: for callers holding un-normalised
 * JubJubExtended values (the device does the z inversion of to_hash_inputs) */
int dsv_verify_single_ext(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                          const uint8_t *m, size_t n, uint8_t *ok);

/* ---- verify, device buffers (enqueue only) ----
 * workspace: device scratch of dsv_workspace_bytes(n) bytes, 256-byte aligned, owned by the call
 * until its work on `stream` has completed (two calls in flight need two workspaces).
 * Ordering: everything the call enqueues happens after the work already on `stream` and before
 * anything enqueued on `stream` afterwards.  Batches of >= 2^17 items are cut into 2^16-item
 * parts that run on two library-owned streams forked from / joined to `stream` by events
 * (DSV_SPLIT=0 in the environment at dsv_init keeps every launch on `stream` itself). */
size_t dsv_workspace_bytes(size_t n);
int dsv_verify_single_dev(const void *u, const void *R_uv, const void *PK_uv, const void *m,
                          size_t n, void *ok, void *workspace, void *stream);
int dsv_verify_double_dev(const void *u, const void *R_uv, const void *Rp_uv, const void *PK_uv,
                          const void *PKp_uv, const void *m, size_t n, void *ok, void *workspace,
                          void *stream);
int dsv_verify_vargen_dev(const void *u, const void *R_uv, const void *PK_uv, const void *Gen_uv,
                          const void *m, size_t n, void *ok, void *workspace, void *stream);

/* second stage alone: ok[i] = (accumulate ? ok[i] : valid[i]) & [u*Gen + c*PK == R], Gen = G
 * (which = 0) or G' (which = 1); c and valid as produced by dsv_challenge_*_dev */
int dsv_verify_core_dev(const void *u, const void *c, const void *valid, const void *PK_uv,
                        const void *R_uv, int which, int accumulate, size_t n, void *ok,
                        void *workspace, void *stream);

/* ---- challenge hash only (c = trunc250(Poseidon(R.., m))), 32 B LE per item ---- */
int dsv_challenge_single(const uint8_t *R_uv, const uint8_t *m, size_t n, uint8_t *c);
int dsv_challenge_double(const uint8_t *R_uv, const uint8_t *Rp_uv, const uint8_t *m, size_t n,
                         uint8_t *c);
int dsv_challenge_single_dev(const void *R_uv, const void *m, size_t n, void *c, void *valid,
                             void *stream);
int dsv_challenge_double_dev(const void *R_uv, const void *Rp_uv, const void *m, size_t n,
                             void *c, void *valid, void *stream);

/* ---- signing / key derivation (fixed-base only; input generation and the "next" row) ----
 * sk, m, r canonical 32 B.  r is the caller-drawn nonce (the reference draws it from its RNG
 * inside sign()).  Outputs: u (32 B), R_uv / Rp_uv (64 B).  gen_uv == NULL means the standard
 * generator G;  for the var-generator scheme pass the per-key generator (variable base). */
int dsv_sign_single(const uint8_t *sk, const uint8_t *m, const uint8_t *r, size_t n, uint8_t *u,
                    uint8_t *R_uv);
int dsv_sign_double(const uint8_t *sk, const uint8_t *m, const uint8_t *r, size_t n, uint8_t *u,
                    uint8_t *R_uv, uint8_t *Rp_uv);
int dsv_sign_vargen(const uint8_t *sk, const uint8_t *Gen_uv, const uint8_t *m, const uint8_t *r,
                    size_t n, uint8_t *u, uint8_t *R_uv);
/* PK = sk * G (which = 0), sk * G' (which = 1); or sk * Gen when gen_uv != NULL */
int dsv_public_keys(const uint8_t *sk, int which, const uint8_t *gen_uv, size_t n,
                    uint8_t *PK_uv);
int dsv_sign_single_dev(const void *sk, const void *m, const void *r, size_t n, void *u,
                        void *R_uv, void *stream);
int dsv_sign_double_dev(const void *sk, const void *m, const void *r, size_t n, void *u,
                        void *R_uv, void *Rp_uv, void *stream);
int dsv_public_keys_dev(const void *sk, int which, size_t n, void *PK_uv, void *stream);

/* ---- wire formats (the reference's Serializable impls) ------------------------------------
 * compressed point = JubJubAffine::to_bytes(): canonical v with bit 255 = lowest bit of u.
 * dsv_decompress_points: JubJubAffine::from_bytes for n points; ok[i] = 0 where the reference
 * would return Err (v >= q, or no square root).  The _dev form reads one 32-byte record every
 * in_stride bytes (16-byte aligned) and, with accumulate != 0, ANDs into ok[] instead of
 * overwriting it. */
int dsv_compress_points(const uint8_t *in_uv, size_t n, uint8_t *out32); /* JubJubAffine::to_bytes */
int dsv_decompress_points(const uint8_t *in32, size_t n, uint8_t *out_uv, uint8_t *ok);
int dsv_decompress_points_dev(const void *in, size_t in_stride, size_t n, void *out_uv, void *ok,
                              int accumulate, void *stream);
/* verify straight from serialized values (host buffers, array-of-records as the Rust
 * `to_bytes()` produce them):
 *   single : Signature (64 B = u || compressed R)            PublicKey (32 B)
 *            /root/reference/src/signatures.rs:106-123, src/keys/public.rs:87-101
 *   double : SignatureDouble (96 B = u || R || R')           PublicKeyDouble (64 B = pk || pk')
 *            /root/reference/src/signatures.rs:245-270, src/keys/public.rs:282-299
 *   vargen : SignatureVarGen (64 B)                          PublicKeyVarGen (64 B = pk || generator)
 *            /root/reference/src/signatures.rs:387-404, src/keys/public.rs:347-372
 * ok[i] = 1 iff every from_bytes would succeed AND verify() would return true. */
int dsv_verify_single_wire(const uint8_t *sig64, const uint8_t *pk32, const uint8_t *m, size_t n,
                           uint8_t *ok);
int dsv_verify_double_wire(const uint8_t *sig96, const uint8_t *pk64, const uint8_t *m, size_t n,
                           uint8_t *ok);
int dsv_verify_vargen_wire(const uint8_t *sig64, const uint8_t *pk64, const uint8_t *m, size_t n,
                           uint8_t *ok);

/* ---- the reference harness's input generator (rand 0.8 StdRng::seed_from_u64(seed) = ChaCha12,
 * draw order per item: sk = JubJubScalar::random, m = BlsScalar::random, nonce r =
 * JubJubScalar::random; /root/reference/tests/schnorr.rs:16-22, benches/signature.rs:48-60).
 * Items first_item .. first_item + n - 1 of that stream; 32 B canonical each. */
int dsv_stdrng_sign_inputs(uint64_t seed, size_t first_item, size_t n, uint8_t *sk, uint8_t *m,
                           uint8_t *r);
int dsv_stdrng_sign_inputs_dev(uint64_t seed, size_t first_item, size_t n, void *sk, void *m,
                               void *r, void *stream);

/* ---- introspection for tests: copy one fixed-base table entry (affine niels v+u, v-u, 2duv
 * as canonical LE, 96 B) for generator `which` (0 = G, 1 = G'), window w (8-bit), digit d ---- */
int dsv_debug_table_entry(int which, int window, int digit, uint8_t out96[96]);
int dsv_fixed_window_bits(void); /* width of the (signed) fixed-base windows; digit <= 2^(bits-1) */
/* ---- introspection: field-multiplier self test on the device: out = a*b mod q (canonical) */
int dsv_debug_fq_mul(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* DSV_H */
