/* schnorr_oracle.h — CPU restatement of dusk-schnorr 0.18 native sign/verify.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under schnorr_amd/ (the product) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED.  The reference (/root/reference, Rust) cannot be compiled in the
 * authoring container (no cargo/rustc) and its arithmetic lives in third-party crates
 * that are not under /root/reference and are pinned only by caret ranges (no
 * Cargo.lock): dusk-jubjub 0.14, dusk-bls12_381 0.13, dusk-poseidon 0.33 (+ dusk-hades),
 * dusk-bytes 0.1, ff 0.13  (/root/reference/Cargo.toml:20-29).  The reference's own
 * tests hold no golden vectors (tests/schnorr.rs etc. are sign->verify relations driven
 * by RNG seeds).  This file restates the published algorithms of those crates
 *  and follows the reference's call sites literally:
 *
 *   verify          /root/reference/src/keys/public.rs:121-130
 *   verify (double) /root/reference/src/keys/public.rs:222-244
 *   verify (vargen) /root/reference/src/keys/public.rs:401-415
 *   challenge_hash  /root/reference/src/signatures.rs:127-134
 *   ..._double      /root/reference/src/signatures.rs:275-290
 *   sign            /root/reference/src/keys/secret.rs:150-168
 *   sign_double     /root/reference/src/keys/secret.rs:217-240
 *   sign (vargen)   /root/reference/src/keys/secret.rs:433-451
 *   pk from sk      /root/reference/src/keys/public.rs:61-67, 265-272, 337-344
 *
 * Field/curve constants are verified mathematically (tests/test_oracle.py re-derives
 * them with Python integers); the Hades constants are recipe-derived and unverified
 * against the crate blobs (oracle/gen_constants.py).
 *
 * All byte arrays are little-endian canonical (non-Montgomery) field elements, 32 B
 * each, structure-of-arrays, exactly the layout of include/dsv.h.
 */
#ifndef SCHNORR_ORACLE_H
#define SCHNORR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- field elements (4 x u64 LE limbs, Montgomery form R = 2^256) ---- */
typedef struct { uint64_t l[4]; } ofq_t; /* BlsScalar  (JubJub base field)   */
typedef struct { uint64_t l[4]; } ofr_t; /* JubJubScalar (prime-order subgroup) */
typedef struct { ofq_t u, v, z, t1, t2; } oext_t;      /* JubJubExtended */
typedef struct { ofq_t vpu, vmu, z, t2d; } oniels_t;   /* ExtendedNielsPoint */

/* Fq */
int  ofq_from_bytes(ofq_t *r, const uint8_t b[32]);          /* 0 if b >= q */
void ofq_to_bytes(uint8_t b[32], const ofq_t *a);
void ofq_from_bytes_wide(ofq_t *r, const uint8_t b[64]);
void ofq_add(ofq_t *r, const ofq_t *a, const ofq_t *b);
void ofq_sub(ofq_t *r, const ofq_t *a, const ofq_t *b);
void ofq_neg(ofq_t *r, const ofq_t *a);
void ofq_mul(ofq_t *r, const ofq_t *a, const ofq_t *b);
void ofq_square(ofq_t *r, const ofq_t *a);
int  ofq_invert(ofq_t *r, const ofq_t *a);                   /* 0 if a == 0 */
int  ofq_sqrt(ofq_t *r, const ofq_t *a);                     /* 0 if non-residue */
int  ofq_eq(const ofq_t *a, const ofq_t *b);
/* Fr */
int  ofr_from_bytes(ofr_t *r, const uint8_t b[32]);          /* 0 if b >= r */
void ofr_to_bytes(uint8_t b[32], const ofr_t *a);
void ofr_from_bytes_wide(ofr_t *r, const uint8_t b[64]);
void ofr_mul(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_sub(ofr_t *r, const ofr_t *a, const ofr_t *b);
void ofr_add(ofr_t *r, const ofr_t *a, const ofr_t *b);

/* JubJub */
void oext_identity(oext_t *r);
void oext_generator(oext_t *r);        /* GENERATOR_EXTENDED      */
void oext_generator_nums(oext_t *r);   /* GENERATOR_NUMS_EXTENDED */
void oext_from_affine(oext_t *r, const ofq_t *u, const ofq_t *v);
void oext_double(oext_t *r, const oext_t *p);
void oext_to_niels(oniels_t *r, const oext_t *p);
void oext_add_niels(oext_t *r, const oext_t *p, const oniels_t *n);
void oext_add(oext_t *r, const oext_t *p, const oext_t *q);
void oext_mul(oext_t *r, const oext_t *p, const uint8_t scalar_le[32]); /* 252-step */
int  oext_eq(const oext_t *a, const oext_t *b);              /* projective eq */
int  oext_to_affine(ofq_t *u, ofq_t *v, const oext_t *p);    /* to_hash_inputs; 0 if z==0 */
int  oext_is_on_curve(const oext_t *p);
/* JubJubAffine::to_bytes / from_bytes (compressed 32 B) */
int  ojub_compress(uint8_t out[32], const oext_t *p);
int  ojub_decompress(oext_t *r, const uint8_t in[32]);       /* 0 on reject */

/* Hades / Poseidon */
void ohades_permute(ofq_t state[5]);
void oposeidon_sponge_hash(ofq_t *out, const ofq_t *msgs, size_t n);
void oposeidon_truncated_hash(uint8_t out_fr_le[32], const ofq_t *msgs, size_t n);

/* challenge hashes (canonical bytes of the JubJubScalar c) */
void ochallenge_hash(uint8_t c_le[32], const oext_t *R, const ofq_t *m);
void ochallenge_hash_double(uint8_t c_le[32], const oext_t *R, const oext_t *Rp, const ofq_t *m);

/* ---- batch byte API, same SoA layout as include/dsv.h ------------------------------
 * points are affine (u||v), 64 B each; scalars / messages 32 B; ok[i] in {0,1}.
 * Non-canonical encodings (u >= r, coordinate or message >= q) give ok[i] = 0.
 * nthreads <= 1 : single thread; > 1 : that many pthreads over contiguous chunks. */
int oracle_verify_single(const uint8_t *u, const uint8_t *R_uv, const uint8_t *PK_uv,
                         const uint8_t *m, size_t n, uint8_t *ok, int nthreads);
int oracle_verify_double(const uint8_t *u, const uint8_t *R_uv, const uint8_t *Rp_uv,
                         const uint8_t *PK_uv, const uint8_t *PKp_uv, const uint8_t *m,
                         size_t n, uint8_t *ok, int nthreads);
int oracle_verify_vargen(const uint8_t *u, const uint8_t *R_uv, const uint8_t *PK_uv,
                         const uint8_t *Gen_uv, const uint8_t *m, size_t n, uint8_t *ok,
                         int nthreads);
/* same, but points given as full JubJubExtended (u,v,z,t1,t2 canonical, 160 B each) —
 * exercises to_hash_inputs with z != 1 (tests/keys.rs:33-59 semantics). */
int oracle_verify_single_ext(const uint8_t *u, const uint8_t *R_ext, const uint8_t *PK_ext,
                             const uint8_t *m, size_t n, uint8_t *ok);
int oracle_verify_double_ext(const uint8_t *u, const uint8_t *R_ext, const uint8_t *Rp_ext,
                             const uint8_t *PK_ext, const uint8_t *PKp_ext, const uint8_t *m,
                             size_t n, uint8_t *ok);
int oracle_verify_vargen_ext(const uint8_t *u, const uint8_t *R_ext, const uint8_t *PK_ext,
                             const uint8_t *Gen_ext, const uint8_t *m, size_t n, uint8_t *ok);

/* the same on the reference's IN-MEMORY representation: every 32-byte element is the four u64
 * Montgomery limbs (R = 2^256) of a BlsScalar / JubJubScalar as the Rust types hold them
 * (/root/reference/src/signatures.rs:58-61, src/keys/public.rs:59, Cargo.toml:25-26); points as
 * the limbs of u || v || z (96 B).  Limbs >= the modulus or z = 0: ok[i] = 0. */
int oracle_verify_single_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                              const uint8_t *m, size_t n, uint8_t *ok);
int oracle_verify_double_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *Rp_uvz,
                              const uint8_t *PK_uvz, const uint8_t *PKp_uvz, const uint8_t *m,
                              size_t n, uint8_t *ok);
int oracle_verify_vargen_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                              const uint8_t *Gen_uvz, const uint8_t *m, size_t n, uint8_t *ok);
/* canonical bytes <-> in-memory limbs; which = 0: BlsScalar, 1: JubJubScalar; 0 if an input >= modulus */
int oracle_to_mont(int which, const uint8_t *canonical, size_t n, uint8_t *limbs);
int oracle_from_mont(int which, const uint8_t *limbs, size_t n, uint8_t *canonical);

/* challenge scalar only (for kernel-level parity of the hash stage) */
int oracle_challenge_single(const uint8_t *R_uv, const uint8_t *m, size_t n, uint8_t *c);
int oracle_challenge_double(const uint8_t *R_uv, const uint8_t *Rp_uv, const uint8_t *m,
                            size_t n, uint8_t *c);

/* signing / key derivation (input generation for the harness).
 * sk_wide, r_wide: 64 uniformly random bytes each -> Fr::from_bytes_wide (ff::Field::random)
 * m_wide         : 64 random bytes -> BlsScalar::from_bytes_wide (BlsScalar::random)
 * outputs canonical: sk[32], m[32], u[32], R_uv[64], PK_uv[64] (+ Rp, PKp / Gen). */
int oracle_keygen_sign_single(const uint8_t *sk_wide, const uint8_t *m_wide,
                              const uint8_t *r_wide, size_t n, uint8_t *sk, uint8_t *m,
                              uint8_t *u, uint8_t *R_uv, uint8_t *PK_uv, int nthreads);
int oracle_keygen_sign_double(const uint8_t *sk_wide, const uint8_t *m_wide,
                              const uint8_t *r_wide, size_t n, uint8_t *sk, uint8_t *m,
                              uint8_t *u, uint8_t *R_uv, uint8_t *Rp_uv, uint8_t *PK_uv,
                              uint8_t *PKp_uv, int nthreads);
int oracle_keygen_sign_vargen(const uint8_t *sk_wide, const uint8_t *g_wide,
                              const uint8_t *m_wide, const uint8_t *r_wide, size_t n,
                              uint8_t *sk, uint8_t *m, uint8_t *u, uint8_t *R_uv,
                              uint8_t *PK_uv, uint8_t *Gen_uv, int nthreads);

/* scalar-mult helper on bytes: out_uv = scalar * P (affine in/out) */
int oracle_scalar_mul(const uint8_t *scalar, const uint8_t *P_uv, size_t n, uint8_t *out_uv);
/* fixed-base window table entry check: out = (digit << (8*window)) * Gen as affine-niels
 * canonical (v+u, v-u, 2d*u*v), 96 B */
int oracle_fixed_base_entry(int which_gen, int window_bits, int window, uint32_t digit,
                            uint8_t out96[96]);

/* wire formats: from_bytes (scalar canonicity + point decompression) then verify */
int oracle_decompress(const uint8_t *in32, size_t n, uint8_t *out_uv, uint8_t *ok);
int oracle_verify_single_wire(const uint8_t *sig64, const uint8_t *pk32, const uint8_t *m, size_t n,
                              uint8_t *ok);
int oracle_verify_double_wire(const uint8_t *sig96, const uint8_t *pk64, const uint8_t *m, size_t n,
                              uint8_t *ok);
int oracle_verify_vargen_wire(const uint8_t *sig64, const uint8_t *pk64, const uint8_t *m, size_t n,
                              uint8_t *ok);

const char *oracle_banner(void);

#ifdef __cplusplus
}
#endif
#endif
