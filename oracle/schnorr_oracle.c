/* schnorr_oracle.c — see schnorr_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.
 *
 * Literal CPU restatement (4 x u64 Montgomery limbs, unsigned __int128) of the algorithms
 * the reference's verify/sign path executes through dusk-jubjub / dusk-bls12_381 /
 * dusk-poseidon.  Deliberately naive: generic 252-step double-and-add for every base,
 * dense Hades permutation, one inversion per to_hash_inputs — the same work the Rust
 * reference does, so that it also serves as the "port" CPU baseline.
 */
#include "schnorr_oracle.h"
#include "hades_constants.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ generic 4x64 Montgomery */
typedef struct {
  uint64_t p[4];   /* modulus */
  uint64_t inv;    /* -p^{-1} mod 2^64 */
  uint64_t r[4];   /* R   mod p */
  uint64_t r2[4];  /* R^2 mod p */
  uint64_t r3[4];  /* R^3 mod p */
} field_t;

/* SURVEY.md Appendix A.1 (verified with Python integers in tests/test_oracle.py) */
static const field_t FQ = {
    {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
    0xfffffffeffffffffULL,
    {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL},
    {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL},
    {0xc62c1807439b73afULL, 0x1b3e0d188cf06990ULL, 0x73d13c71c7b5f418ULL, 0x6e2a5bb9c8db33e9ULL}};
/* Appendix A.2 */
static const field_t FR = {
    {0xd0970e5ed6f72cb7ULL, 0xa6682093ccc81082ULL, 0x06673b0101343b00ULL, 0x0e7db4ea6533afa9ULL},
    0x1ba3a358ef788ef9ULL,
    {0x25f80bb3b99607d9ULL, 0xf315d62f66b6e750ULL, 0x932514eeeb8814f4ULL, 0x09a6fc6f479155c6ULL},
    {0x67719aa495e57731ULL, 0x51b0cef09ce3fc26ULL, 0x69dab7fac026e9a5ULL, 0x04f6547b8d127688ULL},
    {0xe0d6c6563d830544ULL, 0x323e3883598d0f85ULL, 0xf0fea3004c2e2ba8ULL, 0x05874f84946737ecULL}};

static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t *carry) {
  u128 t = (u128)a + b + *carry;
  *carry = (uint64_t)(t >> 64);
  return (uint64_t)t;
}
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t *borrow) {
  u128 t = (u128)a - b - (*borrow >> 63);
  *borrow = (uint64_t)(t >> 64);
  return (uint64_t)t;
}
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t *carry) {
  u128 t = (u128)a + (u128)b * c + *carry;
  *carry = (uint64_t)(t >> 64);
  return (uint64_t)t;
}

/* r = a - p if a >= p else a   (a < 2p) */
static void f_sub_mod_once(const field_t *f, uint64_t r[4], const uint64_t a[4]) {
  uint64_t b = 0, t[4];
  for (int i = 0; i < 4; i++) t[i] = sbb(a[i], f->p[i], &b);
  uint64_t mask = b; /* all-ones if borrow (a < p) */
  uint64_t c = 0;
  for (int i = 0; i < 4; i++) r[i] = adc(t[i], f->p[i] & mask, &c);
}
static void f_add(const field_t *f, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t c = 0, t[4];
  for (int i = 0; i < 4; i++) t[i] = adc(a[i], b[i], &c);
  /* both moduli are < 2^255, so no carry out */
  f_sub_mod_once(f, r, t);
}
static void f_sub(const field_t *f, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t bw = 0, t[4];
  for (int i = 0; i < 4; i++) t[i] = sbb(a[i], b[i], &bw);
  uint64_t c = 0;
  for (int i = 0; i < 4; i++) r[i] = adc(t[i], f->p[i] & bw, &c);
}
static void f_neg(const field_t *f, uint64_t r[4], const uint64_t a[4]) {
  uint64_t z[4] = {0, 0, 0, 0};
  f_sub(f, r, z, a);
}
static void f_mont_reduce(const field_t *f, uint64_t r[4], const uint64_t t[8]) {
  uint64_t x[9];
  memcpy(x, t, 64);
  x[8] = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t k = x[i] * f->inv, carry = 0;
    (void)mac(x[i], k, f->p[0], &carry);
    for (int j = 1; j < 4; j++) x[i + j] = mac(x[i + j], k, f->p[j], &carry);
    uint64_t c2 = 0;
    x[i + 4] = adc(x[i + 4], carry, &c2);
    for (int j = i + 5; j < 9 && c2; j++) x[j] = adc(x[j], 0, &c2);
  }
  /* result x[4..7] (+ x[8] cannot be set: t < p*R) */
  f_sub_mod_once(f, r, x + 4);
}
static void f_mul(const field_t *f, uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; j++) t[i + j] = mac(t[i + j], a[i], b[j], &carry);
    t[i + 4] = carry;
  }
  f_mont_reduce(f, r, t);
}
static int f_is_canonical(const field_t *f, const uint64_t a[4]) {
  uint64_t b = 0;
  for (int i = 0; i < 4; i++) (void)sbb(a[i], f->p[i], &b);
  return b != 0; /* borrow => a < p */
}
static void load_le(uint64_t l[4], const uint8_t b[32]) {
  for (int i = 0; i < 4; i++) {
    uint64_t v = 0;
    for (int k = 7; k >= 0; k--) v = (v << 8) | b[8 * i + k];
    l[i] = v;
  }
}
static void store_le(uint8_t b[32], const uint64_t l[4]) {
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(l[i] >> (8 * k));
}
static int f_from_bytes(const field_t *f, uint64_t r[4], const uint8_t b[32]) {
  uint64_t t[4];
  load_le(t, b);
  int ok = f_is_canonical(f, t);
  f_mul(f, r, t, f->r2); /* canonical -> Montgomery (from_raw) */
  return ok;
}
static void f_to_canonical(const field_t *f, uint64_t r[4], const uint64_t a[4]) {
  uint64_t t[8] = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
  f_mont_reduce(f, r, t); /* "reduce()" upstream */
}
static void f_to_bytes(const field_t *f, uint8_t b[32], const uint64_t a[4]) {
  uint64_t t[4];
  f_to_canonical(f, t, a);
  store_le(b, t);
}
/* from_bytes_wide: lo*R2 + hi*R3  (Appendix A.1) */
static void f_from_bytes_wide(const field_t *f, uint64_t r[4], const uint8_t b[64]) {
  uint64_t lo[4], hi[4], a[4], c[4];
  load_le(lo, b);
  load_le(hi, b + 32);
  f_mul(f, a, lo, f->r2);
  f_mul(f, c, hi, f->r3);
  f_add(f, r, a, c);
}
static int f_eq(const uint64_t a[4], const uint64_t b[4]) {
  return ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3])) == 0;
}
static int f_is_zero(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
/* r = a^e, e as 4 x u64 LE (square-and-multiply, MSB first) */
static void f_pow(const field_t *f, uint64_t r[4], const uint64_t a[4], const uint64_t e[4]) {
  uint64_t acc[4];
  memcpy(acc, f->r, 32);
  for (int i = 255; i >= 0; i--) {
    f_mul(f, acc, acc, acc);
    if ((e[i / 64] >> (i % 64)) & 1) f_mul(f, acc, acc, a);
  }
  memcpy(r, acc, 32);
}

/* ------------------------------------------------------------------ Fq / Fr wrappers */
int ofq_from_bytes(ofq_t *r, const uint8_t b[32]) { return f_from_bytes(&FQ, r->l, b); }
void ofq_to_bytes(uint8_t b[32], const ofq_t *a) { f_to_bytes(&FQ, b, a->l); }
void ofq_from_bytes_wide(ofq_t *r, const uint8_t b[64]) { f_from_bytes_wide(&FQ, r->l, b); }
void ofq_add(ofq_t *r, const ofq_t *a, const ofq_t *b) { f_add(&FQ, r->l, a->l, b->l); }
void ofq_sub(ofq_t *r, const ofq_t *a, const ofq_t *b) { f_sub(&FQ, r->l, a->l, b->l); }
void ofq_neg(ofq_t *r, const ofq_t *a) { f_neg(&FQ, r->l, a->l); }
void ofq_mul(ofq_t *r, const ofq_t *a, const ofq_t *b) { f_mul(&FQ, r->l, a->l, b->l); }
void ofq_square(ofq_t *r, const ofq_t *a) { f_mul(&FQ, r->l, a->l, a->l); }
int ofq_eq(const ofq_t *a, const ofq_t *b) { return f_eq(a->l, b->l); }
int ofq_invert(ofq_t *r, const ofq_t *a) {
  /* Fermat: a^(q-2); upstream uses an addition chain for the same exponent */
  uint64_t e[4] = {FQ.p[0] - 2, FQ.p[1], FQ.p[2], FQ.p[3]};
  if (f_is_zero(a->l)) {
    memset(r, 0, sizeof *r);
    return 0;
  }
  f_pow(&FQ, r->l, a->l, e);
  return 1;
}
/* Tonelli-Shanks, q - 1 = 2^32 * t.  Returns *a* root; callers fix the sign. */
int ofq_sqrt(ofq_t *r, const ofq_t *a) {
  if (f_is_zero(a->l)) {
    memset(r, 0, sizeof *r);
    return 1;
  }
  /* t = (q-1) >> 32 ; (t-1)/2 */
  uint64_t t[4], tm1h[4];
  t[0] = (FQ.p[0] >> 32) | (FQ.p[1] << 32);
  t[1] = (FQ.p[1] >> 32) | (FQ.p[2] << 32);
  t[2] = (FQ.p[2] >> 32) | (FQ.p[3] << 32);
  t[3] = (FQ.p[3] >> 32);
  uint64_t tm1[4] = {t[0] - 1, t[1], t[2], t[3]}; /* t is odd */
  tm1h[0] = (tm1[0] >> 1) | (tm1[1] << 63);
  tm1h[1] = (tm1[1] >> 1) | (tm1[2] << 63);
  tm1h[2] = (tm1[2] >> 1) | (tm1[3] << 63);
  tm1h[3] = (tm1[3] >> 1);
  /* root of unity c = 7^t  (7 generates Fq^*) */
  ofq_t seven, c, w, x, b;
  uint8_t sb[32] = {7};
  ofq_from_bytes(&seven, sb);
  f_pow(&FQ, c.l, seven.l, t);
  f_pow(&FQ, w.l, a->l, tm1h);       /* w = a^((t-1)/2) */
  ofq_mul(&x, a, &w);                /* x = a^((t+1)/2) */
  ofq_mul(&b, &x, &w);               /* b = a^t */
  int v = 32;
  ofq_t one;
  memcpy(one.l, FQ.r, 32);
  while (!ofq_eq(&b, &one)) {
    int k = 0;
    ofq_t b2 = b;
    while (!ofq_eq(&b2, &one)) {
      ofq_square(&b2, &b2);
      k++;
      if (k == v) return 0; /* non-residue */
    }
    ofq_t cc = c;
    for (int i = 0; i < v - k - 1; i++) ofq_square(&cc, &cc);
    ofq_mul(&x, &x, &cc);
    ofq_square(&c, &cc);
    ofq_mul(&b, &b, &c);
    v = k;
  }
  /* verify */
  ofq_t chk;
  ofq_square(&chk, &x);
  if (!ofq_eq(&chk, a)) return 0;
  *r = x;
  return 1;
}

int ofr_from_bytes(ofr_t *r, const uint8_t b[32]) { return f_from_bytes(&FR, r->l, b); }
void ofr_to_bytes(uint8_t b[32], const ofr_t *a) { f_to_bytes(&FR, b, a->l); }
void ofr_from_bytes_wide(ofr_t *r, const uint8_t b[64]) { f_from_bytes_wide(&FR, r->l, b); }
void ofr_mul(ofr_t *r, const ofr_t *a, const ofr_t *b) { f_mul(&FR, r->l, a->l, b->l); }
void ofr_sub(ofr_t *r, const ofr_t *a, const ofr_t *b) { f_sub(&FR, r->l, a->l, b->l); }
void ofr_add(ofr_t *r, const ofr_t *a, const ofr_t *b) { f_add(&FR, r->l, a->l, b->l); }

/* ------------------------------------------------------------------ JubJub (Appendix A.3) */
static ofq_t fq_const(const char *hex_be) { /* 64 hex digits, big-endian */
  uint8_t b[32];
  for (int i = 0; i < 32; i++) {
    unsigned v = 0;
    for (int k = 0; k < 2; k++) {
      char ch = hex_be[2 * i + k];
      v = (v << 4) | (unsigned)(ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10);
    }
    b[31 - i] = (uint8_t)v;
  }
  ofq_t r;
  ofq_from_bytes(&r, b);
  return r;
}
static ofq_t FQ_ONE, FQ_ZERO, EDWARDS_D, EDWARDS_D2, GEN_U, GEN_V, GENN_U, GENN_V;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_consts(void) {
  memset(&FQ_ZERO, 0, sizeof FQ_ZERO);
  memcpy(FQ_ONE.l, FQ.r, 32);
  EDWARDS_D = fq_const("2a9318e74bfa2b48f5fd9207e6bd7fd4292d7f6d37579d2601065fd6d6343eb1");
  EDWARDS_D2 = fq_const("552631ce97f45691ebfb240fcd7affa8525afeda6eaf3a4c020cbfadac687d62");
  GEN_U = fq_const("3fd2814c43ac65a6f1fbf02d0fd6cce62e3ebb21fd6c54ed4df7b7ffec7beaca");
  GEN_V = fq_const("0000000000000000000000000000000000000000000000000000000000000012");
  GENN_U = fq_const("5e67b8f316f414f7bd9514c773fd4456931e316a39fe4541921710179df76377");
  GENN_V = fq_const("43d80eb3b2f3eb1b7b162dbeeb3b34fd9949ba0f82a5507a6705b707162e3ef8");
}
static void ensure_init(void) { pthread_once(&g_once, init_consts); }

void oext_identity(oext_t *r) {
  ensure_init();
  r->u = FQ_ZERO; r->v = FQ_ONE; r->z = FQ_ONE; r->t1 = FQ_ZERO; r->t2 = FQ_ZERO;
}
void oext_from_affine(oext_t *r, const ofq_t *u, const ofq_t *v) {
  ensure_init();
  r->u = *u; r->v = *v; r->z = FQ_ONE; r->t1 = *u; r->t2 = *v;
}
void oext_generator(oext_t *r) { ensure_init(); oext_from_affine(r, &GEN_U, &GEN_V); }
void oext_generator_nums(oext_t *r) { ensure_init(); oext_from_affine(r, &GENN_U, &GENN_V); }

/* CompletedPoint -> extended */
static void completed_to_ext(oext_t *r, const ofq_t *u, const ofq_t *v, const ofq_t *z,
                             const ofq_t *t) {
  oext_t o;
  ofq_mul(&o.u, u, t);
  ofq_mul(&o.v, v, z);
  ofq_mul(&o.z, z, t);
  o.t1 = *u;
  o.t2 = *v;
  *r = o;
}
void oext_double(oext_t *r, const oext_t *p) {
  ofq_t uu, vv, zz2, uv2, s, vpu, vmu, cu, ct;
  ofq_square(&uu, &p->u);
  ofq_square(&vv, &p->v);
  ofq_square(&zz2, &p->z);
  ofq_add(&zz2, &zz2, &zz2);
  ofq_add(&s, &p->u, &p->v);
  ofq_square(&uv2, &s);
  ofq_add(&vpu, &vv, &uu);
  ofq_sub(&vmu, &vv, &uu);
  ofq_sub(&cu, &uv2, &vpu);
  ofq_sub(&ct, &zz2, &vmu);
  completed_to_ext(r, &cu, &vpu, &vmu, &ct);
}
void oext_to_niels(oniels_t *r, const oext_t *p) {
  ensure_init();
  ofq_t t;
  ofq_add(&r->vpu, &p->v, &p->u);
  ofq_sub(&r->vmu, &p->v, &p->u);
  r->z = p->z;
  ofq_mul(&t, &p->t1, &p->t2);
  ofq_mul(&r->t2d, &t, &EDWARDS_D2);
}
void oext_add_niels(oext_t *r, const oext_t *p, const oniels_t *n) {
  ofq_t a, b, c, d, t, cu, cv, cz, ct;
  ofq_sub(&t, &p->v, &p->u);
  ofq_mul(&a, &t, &n->vmu);
  ofq_add(&t, &p->v, &p->u);
  ofq_mul(&b, &t, &n->vpu);
  ofq_mul(&t, &p->t1, &p->t2);
  ofq_mul(&c, &t, &n->t2d);
  ofq_mul(&d, &p->z, &n->z);
  ofq_add(&d, &d, &d);
  ofq_sub(&cu, &b, &a);
  ofq_add(&cv, &b, &a);
  ofq_add(&cz, &d, &c);
  ofq_sub(&ct, &d, &c);
  completed_to_ext(r, &cu, &cv, &cz, &ct);
}
void oext_add(oext_t *r, const oext_t *p, const oext_t *q) {
  oniels_t n;
  oext_to_niels(&n, q);
  oext_add_niels(r, p, &n);
}
/* impl Mul<&Fr>: to_niels().multiply(bytes): 252 steps MSB->LSB, top 4 bits skipped */
void oext_mul(oext_t *r, const oext_t *p, const uint8_t s[32]) {
  ensure_init();
  oniels_t n, zero;
  oext_to_niels(&n, p);
  zero.vpu = FQ_ONE; zero.vmu = FQ_ONE; zero.z = FQ_ONE; zero.t2d = FQ_ZERO;
  oext_t acc;
  oext_identity(&acc);
  for (int bit = 251; bit >= 0; bit--) {
    oext_double(&acc, &acc);
    int b = (s[bit >> 3] >> (bit & 7)) & 1;
    oext_add_niels(&acc, &acc, b ? &n : &zero);
  }
  *r = acc;
}
int oext_eq(const oext_t *a, const oext_t *b) {
  ofq_t l, r_;
  ofq_mul(&l, &a->u, &b->z);
  ofq_mul(&r_, &b->u, &a->z);
  int e1 = ofq_eq(&l, &r_);
  ofq_mul(&l, &a->v, &b->z);
  ofq_mul(&r_, &b->v, &a->z);
  return e1 & ofq_eq(&l, &r_);
}
int oext_to_affine(ofq_t *u, ofq_t *v, const oext_t *p) {
  ofq_t zi;
  int ok = ofq_invert(&zi, &p->z);
  ofq_mul(u, &p->u, &zi);
  ofq_mul(v, &p->v, &zi);
  return ok;
}
int oext_is_on_curve(const oext_t *p) {
  ensure_init();
  ofq_t u, v, u2, v2, l, r_, t;
  if (!oext_to_affine(&u, &v, p)) return 0;
  ofq_square(&u2, &u);
  ofq_square(&v2, &v);
  ofq_sub(&l, &v2, &u2);
  ofq_mul(&t, &u2, &v2);
  ofq_mul(&t, &t, &EDWARDS_D);
  ofq_add(&r_, &FQ_ONE, &t);
  return ofq_eq(&l, &r_);
}
int ojub_compress(uint8_t out[32], const oext_t *p) {
  ofq_t u, v;
  uint8_t ub[32];
  if (!oext_to_affine(&u, &v, p)) return 0;
  ofq_to_bytes(out, &v);
  ofq_to_bytes(ub, &u);
  out[31] |= (uint8_t)(ub[0] << 7);
  return 1;
}
int ojub_decompress(oext_t *r, const uint8_t in[32]) {
  ensure_init();
  uint8_t b[32];
  memcpy(b, in, 32);
  int sign = b[31] >> 7;
  b[31] &= 0x7f;
  ofq_t v, v2, num, den, di, u2, u;
  if (!ofq_from_bytes(&v, b)) return 0;
  ofq_square(&v2, &v);
  ofq_sub(&num, &v2, &FQ_ONE);
  ofq_mul(&den, &v2, &EDWARDS_D);
  ofq_add(&den, &den, &FQ_ONE);
  if (!ofq_invert(&di, &den)) memset(&di, 0, sizeof di);
  ofq_mul(&u2, &num, &di);
  if (!ofq_sqrt(&u, &u2)) return 0;
  uint8_t ub[32];
  ofq_to_bytes(ub, &u);
  if ((ub[0] & 1) != sign) ofq_neg(&u, &u);
  oext_from_affine(r, &u, &v);
  return 1;
}

/* ------------------------------------------------------------------ Hades / Poseidon (A.4) */
static void sbox(ofq_t *x) {
  ofq_t x2, x4;
  ofq_square(&x2, x);
  ofq_square(&x4, &x2);
  ofq_mul(x, &x4, x);
}
static void mds_mul(ofq_t s[5]) {
  ofq_t r[5];
  for (int k = 0; k < 5; k++) {
    memset(&r[k], 0, sizeof r[k]);
    for (int j = 0; j < 5; j++) {
      ofq_t t;
      ofq_mul(&t, (const ofq_t *)HADES_MDS[k][j], &s[j]);
      ofq_add(&r[k], &r[k], &t);
    }
  }
  memcpy(s, r, sizeof r);
}
void ohades_permute(ofq_t s[5]) {
  int ci = 0;
  for (int round = 0; round < HADES_FULL_ROUNDS + HADES_PARTIAL_ROUNDS; round++) {
    int full = round < HADES_FULL_ROUNDS / 2 || round >= HADES_FULL_ROUNDS / 2 + HADES_PARTIAL_ROUNDS;
    for (int k = 0; k < 5; k++) ofq_add(&s[k], &s[k], (const ofq_t *)HADES_ROUND_CONSTANTS[ci++]);
    if (full)
      for (int k = 0; k < 5; k++) sbox(&s[k]);
    else
      sbox(&s[4]); /* partial round: last word only */
    mds_mul(s);
  }
}
void oposeidon_sponge_hash(ofq_t *out, const ofq_t *msgs, size_t n) {
  ensure_init();
  ofq_t st[5];
  memset(st, 0, sizeof st);
  const size_t rate = 4;
  size_t nchunks = (n + rate - 1) / rate;
  if (n == 0) nchunks = 0;
  for (size_t c = 0; c < nchunks; c++) {
    size_t len = (c + 1 == nchunks) ? n - c * rate : rate;
    for (size_t k = 0; k < len; k++) ofq_add(&st[1 + k], &st[1 + k], &msgs[c * rate + k]);
    if (c + 1 == nchunks) {
      if (len < rate) {
        ofq_add(&st[len + 1], &st[len + 1], &FQ_ONE);
      } else {
        ohades_permute(st);
        ofq_add(&st[1], &st[1], &FQ_ONE);
      }
    }
    ohades_permute(st);
  }
  *out = st[1];
}
void oposeidon_truncated_hash(uint8_t out[32], const ofq_t *msgs, size_t n) {
  ofq_t h;
  oposeidon_sponge_hash(&h, msgs, n);
  ofq_to_bytes(out, &h);
  out[31] &= 0x03; /* keep the low 250 bits: canonical & (2^250 - 1) */
}
void ochallenge_hash(uint8_t c[32], const oext_t *R, const ofq_t *m) {
  ofq_t in[3];
  oext_to_affine(&in[0], &in[1], R); /* to_hash_inputs */
  in[2] = *m;
  oposeidon_truncated_hash(c, in, 3);
}
void ochallenge_hash_double(uint8_t c[32], const oext_t *R, const oext_t *Rp, const ofq_t *m) {
  ofq_t in[5];
  oext_to_affine(&in[0], &in[1], R);
  oext_to_affine(&in[2], &in[3], Rp);
  in[4] = *m;
  oposeidon_truncated_hash(c, in, 5);
}

/* ------------------------------------------------------------------ single-item verify */
static int load_point(oext_t *p, const uint8_t uv[64]) {
  ofq_t u, v;
  int ok = ofq_from_bytes(&u, uv) & ofq_from_bytes(&v, uv + 32);
  oext_from_affine(p, &u, &v);
  return ok;
}
static int load_ext(oext_t *p, const uint8_t b[160]) {
  int ok = ofq_from_bytes(&p->u, b) & ofq_from_bytes(&p->v, b + 32) &
           ofq_from_bytes(&p->z, b + 64) & ofq_from_bytes(&p->t1, b + 96) &
           ofq_from_bytes(&p->t2, b + 128);
  return ok;
}
static void store_point(uint8_t uv[64], const oext_t *p) {
  ofq_t u, v;
  oext_to_affine(&u, &v, p);
  ofq_to_bytes(uv, &u);
  ofq_to_bytes(uv + 32, &v);
}
/* public.rs:121-130 */
static int verify_one(const oext_t *pk, const uint8_t u[32], const oext_t *R, const ofq_t *m) {
  uint8_t c[32];
  oext_t g, a, b, p1;
  ochallenge_hash(c, R, m);
  oext_generator(&g);
  oext_mul(&a, &g, u);
  oext_mul(&b, pk, c);
  oext_add(&p1, &a, &b);
  return oext_eq(&p1, R);
}
/* public.rs:222-244 */
static int verify_one_double(const oext_t *pk, const oext_t *pkp, const uint8_t u[32],
                             const oext_t *R, const oext_t *Rp, const ofq_t *m) {
  uint8_t c[32];
  oext_t g, gn, a, b, p1, p2;
  ochallenge_hash_double(c, R, Rp, m);
  oext_generator(&g);
  oext_generator_nums(&gn);
  oext_mul(&a, &g, u);
  oext_mul(&b, pk, c);
  oext_add(&p1, &a, &b);
  oext_mul(&a, &gn, u);
  oext_mul(&b, pkp, c);
  oext_add(&p2, &a, &b);
  return oext_eq(&p1, R) && oext_eq(&p2, Rp);
}
/* public.rs:401-415 */
static int verify_one_vargen(const oext_t *pk, const oext_t *gen, const uint8_t u[32],
                             const oext_t *R, const ofq_t *m) {
  uint8_t c[32];
  oext_t a, b, p1;
  ochallenge_hash(c, R, m);
  oext_mul(&a, gen, u);
  oext_mul(&b, pk, c);
  oext_add(&p1, &a, &b);
  return oext_eq(&p1, R);
}

/* ------------------------------------------------------------------ batch drivers */
typedef struct job_s {
  void (*fn)(struct job_s *, size_t lo, size_t hi);
  const uint8_t *in[8];
  uint8_t *out[8];
  size_t lo, hi;
} job_t;
static void *job_thread(void *arg) {
  job_t *j = (job_t *)arg;
  j->fn(j, j->lo, j->hi);
  return NULL;
}
static void run_job(job_t *tmpl, size_t n, int nthreads) {
  ensure_init();
  if (nthreads <= 1 || n < 2) {
    tmpl->fn(tmpl, 0, n);
    return;
  }
  if ((size_t)nthreads > n) nthreads = (int)n;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)nthreads);
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = *tmpl;
    jobs[t].lo = n * (size_t)t / (size_t)nthreads;
    jobs[t].hi = n * (size_t)(t + 1) / (size_t)nthreads;
    pthread_create(&th[t], NULL, job_thread, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th);
  free(jobs);
}

static void job_verify_single(job_t *j, size_t lo, size_t hi) {
  for (size_t i = lo; i < hi; i++) {
    oext_t R, pk;
    ofq_t m;
    ofr_t us;
    int ok = ofr_from_bytes(&us, j->in[0] + 32 * i);
    ok &= load_point(&R, j->in[1] + 64 * i);
    ok &= load_point(&pk, j->in[2] + 64 * i);
    ok &= ofq_from_bytes(&m, j->in[3] + 32 * i);
    j->out[0][i] = (uint8_t)(ok ? verify_one(&pk, j->in[0] + 32 * i, &R, &m) : 0);
  }
}
int oracle_verify_single(const uint8_t *u, const uint8_t *R, const uint8_t *PK, const uint8_t *m,
                         size_t n, uint8_t *ok, int nthreads) {
  job_t j = {job_verify_single, {u, R, PK, m}, {ok}, 0, 0};
  run_job(&j, n, nthreads);
  return 0;
}
static void job_verify_double(job_t *j, size_t lo, size_t hi) {
  for (size_t i = lo; i < hi; i++) {
    oext_t R, Rp, pk, pkp;
    ofq_t m;
    ofr_t us;
    int ok = ofr_from_bytes(&us, j->in[0] + 32 * i);
    ok &= load_point(&R, j->in[1] + 64 * i);
    ok &= load_point(&Rp, j->in[2] + 64 * i);
    ok &= load_point(&pk, j->in[3] + 64 * i);
    ok &= load_point(&pkp, j->in[4] + 64 * i);
    ok &= ofq_from_bytes(&m, j->in[5] + 32 * i);
    j->out[0][i] =
        (uint8_t)(ok ? verify_one_double(&pk, &pkp, j->in[0] + 32 * i, &R, &Rp, &m) : 0);
  }
}
int oracle_verify_double(const uint8_t *u, const uint8_t *R, const uint8_t *Rp, const uint8_t *PK,
                         const uint8_t *PKp, const uint8_t *m, size_t n, uint8_t *ok,
                         int nthreads) {
  job_t j = {job_verify_double, {u, R, Rp, PK, PKp, m}, {ok}, 0, 0};
  run_job(&j, n, nthreads);
  return 0;
}
static void job_verify_vargen(job_t *j, size_t lo, size_t hi) {
  for (size_t i = lo; i < hi; i++) {
    oext_t R, pk, gen;
    ofq_t m;
    ofr_t us;
    int ok = ofr_from_bytes(&us, j->in[0] + 32 * i);
    ok &= load_point(&R, j->in[1] + 64 * i);
    ok &= load_point(&pk, j->in[2] + 64 * i);
    ok &= load_point(&gen, j->in[3] + 64 * i);
    ok &= ofq_from_bytes(&m, j->in[4] + 32 * i);
    j->out[0][i] = (uint8_t)(ok ? verify_one_vargen(&pk, &gen, j->in[0] + 32 * i, &R, &m) : 0);
  }
}
int oracle_verify_vargen(const uint8_t *u, const uint8_t *R, const uint8_t *PK, const uint8_t *Gen,
                         const uint8_t *m, size_t n, uint8_t *ok, int nthreads) {
  job_t j = {job_verify_vargen, {u, R, PK, Gen, m}, {ok}, 0, 0};
  run_job(&j, n, nthreads);
  return 0;
}
int oracle_verify_single_ext(const uint8_t *u, const uint8_t *R_ext, const uint8_t *PK_ext,
                             const uint8_t *m, size_t n, uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, pk;
    ofq_t mm;
    ofr_t us;
    int good = ofr_from_bytes(&us, u + 32 * i);
    good &= load_ext(&R, R_ext + 160 * i);
    good &= load_ext(&pk, PK_ext + 160 * i);
    good &= ofq_from_bytes(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one(&pk, u + 32 * i, &R, &mm) : 0);
  }
  return 0;
}
/* PublicKeyDouble::verify / PublicKeyVarGen::verify on un-normalised JubJubExtended values
 * (/root/reference/src/keys/public.rs:222-244, :401-415): to_hash_inputs inverts each z */
int oracle_verify_double_ext(const uint8_t *u, const uint8_t *R_ext, const uint8_t *Rp_ext,
                             const uint8_t *PK_ext, const uint8_t *PKp_ext, const uint8_t *m,
                             size_t n, uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, Rp, pk, pkp;
    ofq_t mm;
    ofr_t us;
    int good = ofr_from_bytes(&us, u + 32 * i);
    good &= load_ext(&R, R_ext + 160 * i);
    good &= load_ext(&Rp, Rp_ext + 160 * i);
    good &= load_ext(&pk, PK_ext + 160 * i);
    good &= load_ext(&pkp, PKp_ext + 160 * i);
    good &= ofq_from_bytes(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one_double(&pk, &pkp, u + 32 * i, &R, &Rp, &mm) : 0);
  }
  return 0;
}
int oracle_verify_vargen_ext(const uint8_t *u, const uint8_t *R_ext, const uint8_t *PK_ext,
                             const uint8_t *Gen_ext, const uint8_t *m, size_t n, uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, pk, gen;
    ofq_t mm;
    ofr_t us;
    int good = ofr_from_bytes(&us, u + 32 * i);
    good &= load_ext(&R, R_ext + 160 * i);
    good &= load_ext(&pk, PK_ext + 160 * i);
    good &= load_ext(&gen, Gen_ext + 160 * i);
    good &= ofq_from_bytes(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one_vargen(&pk, &gen, u + 32 * i, &R, &mm) : 0);
  }
  return 0;
}
/* ---- the reference's in-memory representation -------------------------------------------------
 * BlsScalar / JubJubScalar / the coordinates of JubJubExtended hold [u64; 4] Montgomery limbs,
 * R = 2^256 (dusk-bls12_381 0.13, dusk-jubjub 0.14: /root/reference/Cargo.toml:25-26) — exactly
 * this file's ofq_t / ofr_t.  The *_mont functions take those limbs (32 B little-endian per
 * element) as the fields of Signature / PublicKey hold them (/root/reference/src/signatures.rs:
 * 58-61, src/keys/public.rs:59) and run the reference's verify on them with no conversion at all;
 * a point comes as the limbs of u || v || z (96 B) and is completed to a JubJubExtended of the
 * same projective class, (u z, v z, z^2, t1 = u, t2 = v): every step of verify (to_hash_inputs,
 * Mul, Add, PartialEq) depends on the class only.  Limbs >= the modulus (the Rust types cannot
 * hold them) or z = 0 (to_hash_inputs would panic): ok = 0. */
static int load_mont_fq(ofq_t *r, const uint8_t b[32]) {
  load_le(r->l, b);
  return f_is_canonical(&FQ, r->l);
}
static int load_mont_point(oext_t *p, const uint8_t b[96]) {
  ofq_t u, v, z;
  int ok = load_mont_fq(&u, b) & load_mont_fq(&v, b + 32) & load_mont_fq(&z, b + 64);
  static const ofq_t zero = {{0, 0, 0, 0}};
  ok &= !ofq_eq(&z, &zero);
  ofq_mul(&p->u, &u, &z);
  ofq_mul(&p->v, &v, &z);
  ofq_square(&p->z, &z);
  p->t1 = u;
  p->t2 = v;
  return ok;
}
static int load_mont_u(uint8_t u_le[32], const uint8_t b[32]) {
  ofr_t us;
  load_le(us.l, b);
  const int ok = f_is_canonical(&FR, us.l);
  ofr_to_bytes(u_le, &us); /* what Mul<&JubJubScalar> starts with: scalar.to_bytes() */
  return ok;
}
int oracle_verify_single_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                              const uint8_t *m, size_t n, uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, pk;
    ofq_t mm;
    uint8_t ub[32];
    int good = load_mont_u(ub, u + 32 * i);
    good &= load_mont_point(&R, R_uvz + 96 * i);
    good &= load_mont_point(&pk, PK_uvz + 96 * i);
    good &= load_mont_fq(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one(&pk, ub, &R, &mm) : 0);
  }
  return 0;
}
int oracle_verify_double_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *Rp_uvz,
                              const uint8_t *PK_uvz, const uint8_t *PKp_uvz, const uint8_t *m,
                              size_t n, uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, Rp, pk, pkp;
    ofq_t mm;
    uint8_t ub[32];
    int good = load_mont_u(ub, u + 32 * i);
    good &= load_mont_point(&R, R_uvz + 96 * i);
    good &= load_mont_point(&Rp, Rp_uvz + 96 * i);
    good &= load_mont_point(&pk, PK_uvz + 96 * i);
    good &= load_mont_point(&pkp, PKp_uvz + 96 * i);
    good &= load_mont_fq(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one_double(&pk, &pkp, ub, &R, &Rp, &mm) : 0);
  }
  return 0;
}
int oracle_verify_vargen_mont(const uint8_t *u, const uint8_t *R_uvz, const uint8_t *PK_uvz,
                              const uint8_t *Gen_uvz, const uint8_t *m, size_t n, uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, pk, gen;
    ofq_t mm;
    uint8_t ub[32];
    int good = load_mont_u(ub, u + 32 * i);
    good &= load_mont_point(&R, R_uvz + 96 * i);
    good &= load_mont_point(&pk, PK_uvz + 96 * i);
    good &= load_mont_point(&gen, Gen_uvz + 96 * i);
    good &= load_mont_fq(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one_vargen(&pk, &gen, ub, &R, &mm) : 0);
  }
  return 0;
}
/* canonical bytes <-> the in-memory limbs (`from_bytes` / `to_bytes` of the two scalar types);
 * which = 0: BlsScalar (mod q), 1: JubJubScalar (mod r).  Returns 0 if an input is not below the
 * modulus (its output is then unspecified). */
int oracle_to_mont(int which, const uint8_t *canonical, size_t n, uint8_t *limbs) {
  const field_t *f = which ? &FR : &FQ;
  int all = 1;
  for (size_t i = 0; i < n; i++) {
    uint64_t t[4];
    all &= f_from_bytes(f, t, canonical + 32 * i);
    store_le(limbs + 32 * i, t);
  }
  return all;
}
int oracle_from_mont(int which, const uint8_t *limbs, size_t n, uint8_t *canonical) {
  const field_t *f = which ? &FR : &FQ;
  int all = 1;
  for (size_t i = 0; i < n; i++) {
    uint64_t t[4], c[4];
    load_le(t, limbs + 32 * i);
    all &= f_is_canonical(f, t);
    f_to_canonical(f, c, t);
    store_le(canonical + 32 * i, c);
  }
  return all;
}

int oracle_challenge_single(const uint8_t *R_uv, const uint8_t *m, size_t n, uint8_t *c) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R;
    ofq_t mm;
    load_point(&R, R_uv + 64 * i);
    ofq_from_bytes(&mm, m + 32 * i);
    ochallenge_hash(c + 32 * i, &R, &mm);
  }
  return 0;
}
int oracle_challenge_double(const uint8_t *R_uv, const uint8_t *Rp_uv, const uint8_t *m, size_t n,
                            uint8_t *c) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, Rp;
    ofq_t mm;
    load_point(&R, R_uv + 64 * i);
    load_point(&Rp, Rp_uv + 64 * i);
    ofq_from_bytes(&mm, m + 32 * i);
    ochallenge_hash_double(c + 32 * i, &R, &Rp, &mm);
  }
  return 0;
}

/* ------------------------------------------------------------------ keygen + sign */
/* secret.rs:150-168 : r random; R = G*r; c = H(R,m); u = r - c*sk */
static void job_sign_single(job_t *j, size_t lo, size_t hi) {
  oext_t g;
  oext_generator(&g);
  for (size_t i = lo; i < hi; i++) {
    ofr_t sk, r, c, u, t;
    ofq_t m;
    uint8_t skb[32], rb[32], cb[32];
    ofr_from_bytes_wide(&sk, j->in[0] + 64 * i);
    ofq_from_bytes_wide(&m, j->in[1] + 64 * i);
    ofr_from_bytes_wide(&r, j->in[2] + 64 * i);
    ofr_to_bytes(skb, &sk);
    ofr_to_bytes(rb, &r);
    oext_t R, pk;
    oext_mul(&R, &g, rb);
    ochallenge_hash(cb, &R, &m);
    ofr_from_bytes(&c, cb);
    ofr_mul(&t, &c, &sk);
    ofr_sub(&u, &r, &t);
    oext_mul(&pk, &g, skb); /* public.rs:61-67 */
    memcpy(j->out[0] + 32 * i, skb, 32);
    ofq_to_bytes(j->out[1] + 32 * i, &m);
    ofr_to_bytes(j->out[2] + 32 * i, &u);
    store_point(j->out[3] + 64 * i, &R);
    store_point(j->out[4] + 64 * i, &pk);
  }
}
int oracle_keygen_sign_single(const uint8_t *sk_wide, const uint8_t *m_wide, const uint8_t *r_wide,
                              size_t n, uint8_t *sk, uint8_t *m, uint8_t *u, uint8_t *R_uv,
                              uint8_t *PK_uv, int nthreads) {
  job_t j = {job_sign_single, {sk_wide, m_wide, r_wide}, {sk, m, u, R_uv, PK_uv}, 0, 0};
  run_job(&j, n, nthreads);
  return 0;
}
/* secret.rs:217-240 */
static void job_sign_double(job_t *j, size_t lo, size_t hi) {
  oext_t g, gn;
  oext_generator(&g);
  oext_generator_nums(&gn);
  for (size_t i = lo; i < hi; i++) {
    ofr_t sk, r, c, u, t;
    ofq_t m;
    uint8_t skb[32], rb[32], cb[32];
    ofr_from_bytes_wide(&sk, j->in[0] + 64 * i);
    ofq_from_bytes_wide(&m, j->in[1] + 64 * i);
    ofr_from_bytes_wide(&r, j->in[2] + 64 * i);
    ofr_to_bytes(skb, &sk);
    ofr_to_bytes(rb, &r);
    oext_t R, Rp, pk, pkp;
    oext_mul(&R, &g, rb);
    oext_mul(&Rp, &gn, rb);
    ochallenge_hash_double(cb, &R, &Rp, &m);
    ofr_from_bytes(&c, cb);
    ofr_mul(&t, &c, &sk);
    ofr_sub(&u, &r, &t);
    oext_mul(&pk, &g, skb);   /* public.rs:265-272 */
    oext_mul(&pkp, &gn, skb);
    memcpy(j->out[0] + 32 * i, skb, 32);
    ofq_to_bytes(j->out[1] + 32 * i, &m);
    ofr_to_bytes(j->out[2] + 32 * i, &u);
    store_point(j->out[3] + 64 * i, &R);
    store_point(j->out[4] + 64 * i, &Rp);
    store_point(j->out[5] + 64 * i, &pk);
    store_point(j->out[6] + 64 * i, &pkp);
  }
}
int oracle_keygen_sign_double(const uint8_t *sk_wide, const uint8_t *m_wide, const uint8_t *r_wide,
                              size_t n, uint8_t *sk, uint8_t *m, uint8_t *u, uint8_t *R_uv,
                              uint8_t *Rp_uv, uint8_t *PK_uv, uint8_t *PKp_uv, int nthreads) {
  job_t j = {job_sign_double, {sk_wide, m_wide, r_wide}, {sk, m, u, R_uv, Rp_uv, PK_uv, PKp_uv},
             0, 0};
  run_job(&j, n, nthreads);
  return 0;
}
/* secret.rs:367-376 (random: sk then generator scalar), :433-451 (sign) */
static void job_sign_vargen(job_t *j, size_t lo, size_t hi) {
  oext_t g;
  oext_generator(&g);
  for (size_t i = lo; i < hi; i++) {
    ofr_t sk, gs, r, c, u, t;
    ofq_t m;
    uint8_t skb[32], gb[32], rb[32], cb[32];
    ofr_from_bytes_wide(&sk, j->in[0] + 64 * i);
    ofr_from_bytes_wide(&gs, j->in[1] + 64 * i);
    ofq_from_bytes_wide(&m, j->in[2] + 64 * i);
    ofr_from_bytes_wide(&r, j->in[3] + 64 * i);
    ofr_to_bytes(skb, &sk);
    ofr_to_bytes(gb, &gs);
    ofr_to_bytes(rb, &r);
    oext_t gen, R, pk;
    oext_mul(&gen, &g, gb);
    oext_mul(&R, &gen, rb);
    ochallenge_hash(cb, &R, &m);
    ofr_from_bytes(&c, cb);
    ofr_mul(&t, &c, &sk);
    ofr_sub(&u, &r, &t);
    oext_mul(&pk, &gen, skb); /* public.rs:337-344 */
    memcpy(j->out[0] + 32 * i, skb, 32);
    ofq_to_bytes(j->out[1] + 32 * i, &m);
    ofr_to_bytes(j->out[2] + 32 * i, &u);
    store_point(j->out[3] + 64 * i, &R);
    store_point(j->out[4] + 64 * i, &pk);
    store_point(j->out[5] + 64 * i, &gen);
  }
}
int oracle_keygen_sign_vargen(const uint8_t *sk_wide, const uint8_t *g_wide, const uint8_t *m_wide,
                              const uint8_t *r_wide, size_t n, uint8_t *sk, uint8_t *m, uint8_t *u,
                              uint8_t *R_uv, uint8_t *PK_uv, uint8_t *Gen_uv, int nthreads) {
  job_t j = {job_sign_vargen, {sk_wide, g_wide, m_wide, r_wide}, {sk, m, u, R_uv, PK_uv, Gen_uv},
             0, 0};
  run_job(&j, n, nthreads);
  return 0;
}

int oracle_scalar_mul(const uint8_t *scalar, const uint8_t *P_uv, size_t n, uint8_t *out_uv) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t p, r;
    load_point(&p, P_uv + 64 * i);
    oext_mul(&r, &p, scalar + 32 * i);
    store_point(out_uv + 64 * i, &r);
  }
  return 0;
}
int oracle_fixed_base_entry(int which_gen, int window_bits, int window, uint32_t digit,
                            uint8_t out96[96]) {
  ensure_init();
  oext_t g, acc;
  if (which_gen == 0) oext_generator(&g); else oext_generator_nums(&g);
  /* scalar = digit << (window_bits*window) as a 256-bit LE integer; use double-and-add
     on the integer directly (may exceed 252 bits only if caller asks for it) */
  uint8_t s[40] = {0};
  int sh = window_bits * window;
  uint64_t d = digit;
  for (int k = 0; k < 8; k++) {
    int bitpos = sh + 8 * k;
    int byte = bitpos >> 3, off = bitpos & 7;
    uint32_t v = (uint32_t)((d >> (8 * k)) & 0xff) << off;
    if (byte < 39) { s[byte] |= (uint8_t)v; s[byte + 1] |= (uint8_t)(v >> 8); }
  }
  oext_identity(&acc);
  oniels_t n;
  oext_to_niels(&n, &g);
  for (int bit = 255; bit >= 0; bit--) {
    oext_double(&acc, &acc);
    if ((s[bit >> 3] >> (bit & 7)) & 1) oext_add_niels(&acc, &acc, &n);
  }
  ofq_t u, v, t, a;
  oext_to_affine(&u, &v, &acc);
  ofq_add(&a, &v, &u);
  ofq_to_bytes(out96, &a);
  ofq_sub(&a, &v, &u);
  ofq_to_bytes(out96 + 32, &a);
  ofq_mul(&t, &u, &v);
  ofq_mul(&t, &t, &EDWARDS_D2);
  ofq_to_bytes(out96 + 64, &t);
  return 0;
}

/* ---- wire formats: Serializable::from_bytes then verify ---------------------------------
 * signatures.rs:117-122 (u = JubJubScalar::from_slice, R = JubJubAffine::from_slice),
 * keys/public.rs:94-100 */
int oracle_decompress(const uint8_t *in32, size_t n, uint8_t *out_uv, uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t p;
    ok[i] = (uint8_t)ojub_decompress(&p, in32 + 32 * i);
    if (ok[i]) store_point(out_uv + 64 * i, &p); else memset(out_uv + 64 * i, 0, 64);
  }
  return 0;
}
int oracle_verify_single_wire(const uint8_t *sig64, const uint8_t *pk32, const uint8_t *m, size_t n,
                              uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, pk;
    ofq_t mm;
    ofr_t us;
    int good = ofr_from_bytes(&us, sig64 + 64 * i);
    good &= ojub_decompress(&R, sig64 + 64 * i + 32);
    good &= ojub_decompress(&pk, pk32 + 32 * i);
    good &= ofq_from_bytes(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one(&pk, sig64 + 64 * i, &R, &mm) : 0);
  }
  return 0;
}
int oracle_verify_double_wire(const uint8_t *sig96, const uint8_t *pk64, const uint8_t *m, size_t n,
                              uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, Rp, pk, pkp;
    ofq_t mm;
    ofr_t us;
    int good = ofr_from_bytes(&us, sig96 + 96 * i);
    good &= ojub_decompress(&R, sig96 + 96 * i + 32);
    good &= ojub_decompress(&Rp, sig96 + 96 * i + 64);
    good &= ojub_decompress(&pk, pk64 + 64 * i);
    good &= ojub_decompress(&pkp, pk64 + 64 * i + 32);
    good &= ofq_from_bytes(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one_double(&pk, &pkp, sig96 + 96 * i, &R, &Rp, &mm) : 0);
  }
  return 0;
}
int oracle_verify_vargen_wire(const uint8_t *sig64, const uint8_t *pk64, const uint8_t *m, size_t n,
                              uint8_t *ok) {
  ensure_init();
  for (size_t i = 0; i < n; i++) {
    oext_t R, pk, gen;
    ofq_t mm;
    ofr_t us;
    int good = ofr_from_bytes(&us, sig64 + 64 * i);
    good &= ojub_decompress(&R, sig64 + 64 * i + 32);
    good &= ojub_decompress(&pk, pk64 + 64 * i);
    good &= ojub_decompress(&gen, pk64 + 64 * i + 32);
    good &= ofq_from_bytes(&mm, m + 32 * i);
    ok[i] = (uint8_t)(good ? verify_one_vargen(&pk, &gen, sig64 + 64 * i, &R, &mm) : 0);
  }
  return 0;
}

const char *oracle_banner(void) {
  return "schnorr_oracle: CPU restatement of dusk-schnorr 0.18 verify/sign — TEST "
         "INFRASTRUCTURE, PARITY UNPINNED (no reference golden vectors exist; Hades "
         "constants recipe-derived)";
}
