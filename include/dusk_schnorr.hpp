// dusk_schnorr.hpp — host-side mirror (C++17, header-only) of the dusk-schnorr 0.18 key /
// signature surface for the native verification path, on top of the C ABI in dsv.h.
//
// The reference is a Rust crate; this image has no Rust toolchain, so the compiled-language host
// layer is C++ with the reference's names, argument meaning, error behaviour AND in-memory
// representation:
//
//   BlsScalar, JubJubScalar                         four u64 limbs in Montgomery form, R = 2^256, as
//                                                   dusk-bls12_381 0.13 / dusk-jubjub 0.14 hold them
//                                                   (/root/reference/Cargo.toml:25-26); `to_bytes()`
//                                                   is a Montgomery reduction, `from_bytes()` a
//                                                   multiplication by R^2 — what they cost upstream
//   JubJubExtended{u, v, z, t1, t2}                 160 B, five BlsScalar
//   SecretKey::random / sign / sign_double          /root/reference/src/keys/secret.rs:79-86, 150-168, 217-240
//   SecretKeyVarGen::{new, random, sign}            /root/reference/src/keys/secret.rs:352-376, 433-451
//   PublicKey::from(&sk) / verify                   /root/reference/src/keys/public.rs:61-67, 121-130
//   PublicKeyDouble::from(&sk) / verify             /root/reference/src/keys/public.rs:222-244, 265-272
//   PublicKeyVarGen::from(&sk) / verify             /root/reference/src/keys/public.rs:337-344, 401-415
//   Signature{u, R}, SignatureDouble{u, R, R_prime}, SignatureVarGen{u, R}
//                                                   /root/reference/src/signatures.rs:58-73, 180-203, 337-353
//   Serializable: to_bytes / from_bytes of every key and signature type
//                                                   /root/reference/src/signatures.rs:106-123, 245-270, 387-404
//                                                   /root/reference/src/keys/public.rs:87-101, 282-299, 347-372
//                                                   /root/reference/src/keys/secret.rs:89-103, 313-336
//     (`Result<Self, BytesError>` becomes std::optional: empty = the reference's Err)
//   verify_batch / verify_batch_double / verify_batch_var_gen : the new batch entry points
//
// `verify` is infallible and returns bool exactly like the reference; a failure of the engine
// itself (no GPU, HIP error) is not a verdict and is thrown as std::runtime_error.
//
// verify() and verify_batch*() do NO field arithmetic on the host: they hand the engine the limbs
// where they lie (dsv_verify_*_mont_cols: one strided column per field of the typed objects; the
// engine's copy threads gather them into pinned staging while the GPU works on the previous
// chunk) — no `to_bytes()`, i.e. none of the 8 (single) / 14 (double) / 11 (var-generator)
// Montgomery reductions per signature a byte-oriented binding would run on one host thread, and
// no `to_hash_inputs` inversion.  The host arithmetic below (Montgomery multiplication, inversion,
// projective equality) serves the parts of the surface that are host work in the reference too:
// (de)serialisation, `random`, `PartialEq`.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "dsv.h"

namespace dusk_schnorr {

namespace detail {
inline void check(int rc, const char* what) {
  if (rc != DSV_OK)
    throw std::runtime_error(std::string(what) + ": dsv error " + std::to_string(rc) + ": " +
                             dsv_last_error());
}
inline void ensure_init() {
  // the GPUs named in $DSV_DEVICES, else every GPU of the node: verify_batch* shard over all of
  // them.  First call: two 75.5 MB window tables per device, ~50 ms each.
  static const bool once = [] {
    const int rc = dsv_init_visible();
    if (rc < 0) check(rc, "dsv_init_visible");  // e.g. DSV_ERR_NO_DEVICE
    return true;
  }();
  (void)once;
}

// One prime field in Montgomery form, R = 2^256 (every constant is
// re-derived from the modulus by tests/test_cpp_mirror.py)
struct FieldParams {
  uint64_t p[4];   // modulus
  uint64_t inv;    // -p^-1 mod 2^64
  uint64_t r[4];   // R mod p   (= one)
  uint64_t r2[4];  // R^2 mod p
  uint64_t r3[4];  // R^3 mod p
};
inline constexpr FieldParams kFq = {
    {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL},
    0xfffffffeffffffffULL,
    {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL},
    {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL},
    {0xc62c1807439b73afULL, 0x1b3e0d188cf06990ULL, 0x73d13c71c7b5f418ULL, 0x6e2a5bb9c8db33e9ULL}};
inline constexpr FieldParams kFr = {
    {0xd0970e5ed6f72cb7ULL, 0xa6682093ccc81082ULL, 0x06673b0101343b00ULL, 0x0e7db4ea6533afa9ULL},
    0x1ba3a358ef788ef9ULL,
    {0x25f80bb3b99607d9ULL, 0xf315d62f66b6e750ULL, 0x932514eeeb8814f4ULL, 0x09a6fc6f479155c6ULL},
    {0x67719aa495e57731ULL, 0x51b0cef09ce3fc26ULL, 0x69dab7fac026e9a5ULL, 0x04f6547b8d127688ULL},
    {0xe0d6c6563d830544ULL, 0x323e3883598d0f85ULL, 0xf0fea3004c2e2ba8ULL, 0x05874f84946737ecULL}};

using u128 = unsigned __int128;
inline bool geq(const uint64_t a[4], const uint64_t b[4]) {
  for (int i = 3; i >= 0; i--)
    if (a[i] != b[i]) return a[i] > b[i];
  return true;
}
inline void sub_in_place(uint64_t a[4], const uint64_t b[4]) {
  u128 borrow = 0;
  for (int i = 0; i < 4; i++) {
    const u128 t = (u128)a[i] - b[i] - (uint64_t)borrow;
    a[i] = (uint64_t)t;
    borrow = (t >> 64) & 1;
  }
}
// out = a * b * 2^-256 mod p   (operand scanning; a, b < p)
inline void mont_mul(const FieldParams& f, uint64_t out[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a[j] * b[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * f.inv;
    c = ((u128)m * f.p[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * f.p[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || geq(t, f.p)) sub_in_place(t, f.p);
  std::memcpy(out, t, 32);
}
}  // namespace detail

// A field element as the reference holds it: `.0` = four u64 Montgomery limbs.
template <const detail::FieldParams& F>
struct MontScalar {
  uint64_t l[4] = {0, 0, 0, 0};  // x * 2^256 mod p, little-endian limbs

  static MontScalar zero() { return MontScalar{}; }
  static MontScalar one() { return from_limbs(F.r); }
  static MontScalar from_limbs(const uint64_t limbs[4]) {  // the reference's tuple field, as is
    MontScalar s;
    std::memcpy(s.l, limbs, 32);
    return s;
  }
  // `from_raw([u64; 4])`: canonical integer limbs -> Montgomery form
  static MontScalar from_raw(const uint64_t raw[4]) {
    MontScalar s;
    detail::mont_mul(F, s.l, raw, F.r2);
    return s;
  }
  static MontScalar from(uint64_t x) {
    const uint64_t raw[4] = {x, 0, 0, 0};
    return from_raw(raw);
  }
  // `from_bytes_wide`: 512-bit little-endian integer mod p = lo * R^2 + hi * R^3 (Montgomery products)
  static MontScalar from_bytes_wide(const uint8_t wide[64]) {
    uint64_t lo[4], hi[4];
    std::memcpy(lo, wide, 32);
    std::memcpy(hi, wide + 32, 32);
    // the halves may exceed p; mont_mul only needs a * b < p * 2^256, true for any 256-bit a and b < p
    MontScalar a, b;
    detail::mont_mul(F, a.l, lo, F.r2);
    detail::mont_mul(F, b.l, hi, F.r3);
    return a + b;
  }
  // `Field::random`: 64 bytes from the rng, reduced.  Rng: void operator()(uint8_t*, size_t)
  template <class Rng>
  static MontScalar random(Rng& rng) {
    uint8_t wide[64];
    rng(wide, 64);
    return from_bytes_wide(wide);
  }
  // `to_bytes()`: one Montgomery reduction, canonical little-endian
  std::array<uint8_t, 32> to_bytes() const {
    const uint64_t one_raw[4] = {1, 0, 0, 0};
    uint64_t c[4];
    detail::mont_mul(F, c, l, one_raw);
    std::array<uint8_t, 32> b;
    std::memcpy(b.data(), c, 32);
    return b;
  }
  // `Serializable::from_bytes`: rejects encodings >= the modulus
  static std::optional<MontScalar> from_bytes(const uint8_t b[32]) {
    uint64_t raw[4];
    std::memcpy(raw, b, 32);
    if (detail::geq(raw, F.p)) return std::nullopt;
    return from_raw(raw);
  }
  static std::array<uint8_t, 32> modulus_bytes() {
    std::array<uint8_t, 32> b;
    std::memcpy(b.data(), F.p, 32);
    return b;
  }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const MontScalar& o) const { return std::memcmp(l, o.l, 32) == 0; }
  bool operator!=(const MontScalar& o) const { return !(*this == o); }
  MontScalar operator+(const MontScalar& o) const {
    MontScalar s;
    detail::u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (detail::u128)l[i] + o.l[i];
      s.l[i] = (uint64_t)c;
      c >>= 64;
    }
    if (detail::geq(s.l, F.p)) detail::sub_in_place(s.l, F.p);  // both moduli < 2^255: no carry out
    return s;
  }
  MontScalar operator-() const {
    if (is_zero()) return *this;
    MontScalar s = from_limbs(F.p);
    detail::sub_in_place(s.l, l);
    return s;
  }
  MontScalar operator-(const MontScalar& o) const { return *this + (-o); }
  MontScalar operator*(const MontScalar& o) const {
    MontScalar s;
    detail::mont_mul(F, s.l, l, o.l);
    return s;
  }
  // x^(p-2): `invert()`; empty for zero
  std::optional<MontScalar> invert() const {
    if (is_zero()) return std::nullopt;
    uint64_t e[4];
    std::memcpy(e, F.p, 32);
    e[0] -= 2;  // neither modulus ends in 0 or 1
    MontScalar acc = one();
    for (int bit = 255; bit >= 0; bit--) {
      acc = acc * acc;
      if ((e[bit >> 6] >> (bit & 63)) & 1) acc = acc * *this;
    }
    return acc;
  }
};
using BlsScalar = MontScalar<detail::kFq>;
using JubJubScalar = MontScalar<detail::kFr>;
static_assert(sizeof(BlsScalar) == 32 && sizeof(JubJubScalar) == 32, "four u64 limbs");

struct JubJubAffine {  // (u, v) — the pair to_hash_inputs() returns
  BlsScalar u, v = BlsScalar::one();
  const BlsScalar& get_u() const { return u; }
  const BlsScalar& get_v() const { return v; }
  bool operator==(const JubJubAffine& o) const { return u == o.u && v == o.v; }
  // compressed form: canonical v, bit 255 = lowest bit of canonical u
  std::array<uint8_t, 32> to_bytes() const {
    std::array<uint8_t, 32> out = v.to_bytes();
    out[31] |= (uint8_t)((u.to_bytes()[0] & 1) << 7);
    return out;
  }
  // JubJubAffine::from_bytes: the square root runs on the device (k_decompress)
  static std::optional<JubJubAffine> from_bytes(const uint8_t b[32]) {
    detail::ensure_init();
    uint8_t uv[64], ok = 0;
    detail::check(dsv_decompress_points(b, 1, uv, &ok), "dsv_decompress_points");
    if (!ok) return std::nullopt;
    return JubJubAffine{*BlsScalar::from_bytes(uv), *BlsScalar::from_bytes(uv + 32)};
  }
  static JubJubAffine from_canonical(const uint8_t uv[64]) {  // what the engine's kernels emit
    auto u = BlsScalar::from_bytes(uv), v = BlsScalar::from_bytes(uv + 32);
    if (!u || !v) throw std::runtime_error("engine returned a non-canonical coordinate");
    return JubJubAffine{*u, *v};
  }
  std::array<uint8_t, 64> to_canonical() const {
    std::array<uint8_t, 64> out;
    std::memcpy(out.data(), u.to_bytes().data(), 32);
    std::memcpy(out.data() + 32, v.to_bytes().data(), 32);
    return out;
  }
};

// (u, v, z, t1, t2) with u/z, v/z the affine coordinates and t1 * t2 = u v / z: the reference's
// JubJubExtended, 160 B.  Points that come out of this library's kernels (sign, key derivation,
// from_bytes) have z = 1; a caller may hand in any projective representation
// (from_raw_unchecked): verify() treats equal points alike whatever their z
// (/root/reference/tests/keys.rs:33-59).
struct JubJubExtended {
  BlsScalar u, v = BlsScalar::one(), z = BlsScalar::one(), t1, t2;  // identity, like Default upstream
  static JubJubExtended from(const JubJubAffine& a) { return JubJubExtended{a.u, a.v, BlsScalar::one(), a.u, a.v}; }
  static JubJubExtended from_raw_unchecked(const BlsScalar& u, const BlsScalar& v, const BlsScalar& z,
                                           const BlsScalar& t1, const BlsScalar& t2) {
    return JubJubExtended{u, v, z, t1, t2};
  }
  const BlsScalar& get_u() const { return u; }
  const BlsScalar& get_v() const { return v; }
  const BlsScalar& get_z() const { return z; }
  // JubJubExtended::to_hash_inputs / JubJubAffine::from(ext): (u/z, v/z); z = 0 panics upstream
  JubJubAffine to_affine() const {
    const auto zi = z.invert();
    if (!zi) throw std::domain_error("JubJubExtended with z = 0 (the reference panics here)");
    return JubJubAffine{u * *zi, v * *zi};
  }
  std::array<BlsScalar, 2> to_hash_inputs() const {
    const JubJubAffine a = to_affine();
    return {a.u, a.v};
  }
  std::array<uint8_t, 32> to_bytes() const { return to_affine().to_bytes(); }
  static std::optional<JubJubExtended> from_bytes(const uint8_t b[32]) {
    auto a = JubJubAffine::from_bytes(b);
    if (!a) return std::nullopt;
    return from(*a);
  }
  // PartialEq of the reference: u1 z2 == u2 z1 and v1 z2 == v2 z1 — four host multiplications, never
  // a panic (z = 0 simply compares by the products)
  bool operator==(const JubJubExtended& o) const { return u * o.z == o.u * z && v * o.z == o.v * z; }
  bool operator!=(const JubJubExtended& o) const { return !(*this == o); }
};
static_assert(sizeof(JubJubExtended) == 160 && offsetof(JubJubExtended, u) == 0 &&
                  offsetof(JubJubExtended, v) == 32 && offsetof(JubJubExtended, z) == 64,
              "u || v || z are the first 96 bytes: what dsv_verify_*_mont reads");

struct Signature {
  static constexpr size_t SIZE = 64;  // u || compressed R
  JubJubScalar u_;
  JubJubExtended R_;
  const JubJubScalar& u() const { return u_; }
  const JubJubExtended& R() const { return R_; }
  bool operator==(const Signature& o) const { return u_ == o.u_ && R_ == o.R_; }
  std::array<uint8_t, SIZE> to_bytes() const {
    std::array<uint8_t, SIZE> b;
    std::memcpy(b.data(), u_.to_bytes().data(), 32);
    std::memcpy(b.data() + 32, R_.to_bytes().data(), 32);
    return b;
  }
  static std::optional<Signature> from_bytes(const std::array<uint8_t, SIZE>& b) {
    auto u = JubJubScalar::from_bytes(b.data());
    auto R = JubJubExtended::from_bytes(b.data() + 32);
    if (!u || !R) return std::nullopt;
    return Signature{*u, *R};
  }
};
struct SignatureDouble {
  static constexpr size_t SIZE = 96;  // u || compressed R || compressed R'
  JubJubScalar u_;
  JubJubExtended R_, R_prime_;
  const JubJubScalar& u() const { return u_; }
  const JubJubExtended& R() const { return R_; }
  const JubJubExtended& R_prime() const { return R_prime_; }
  bool operator==(const SignatureDouble& o) const {
    return u_ == o.u_ && R_ == o.R_ && R_prime_ == o.R_prime_;
  }
  std::array<uint8_t, SIZE> to_bytes() const {
    std::array<uint8_t, SIZE> b;
    std::memcpy(b.data(), u_.to_bytes().data(), 32);
    std::memcpy(b.data() + 32, R_.to_bytes().data(), 32);
    std::memcpy(b.data() + 64, R_prime_.to_bytes().data(), 32);
    return b;
  }
  static std::optional<SignatureDouble> from_bytes(const std::array<uint8_t, SIZE>& b) {
    auto u = JubJubScalar::from_bytes(b.data());
    auto R = JubJubExtended::from_bytes(b.data() + 32);
    auto Rp = JubJubExtended::from_bytes(b.data() + 64);
    if (!u || !R || !Rp) return std::nullopt;
    return SignatureDouble{*u, *R, *Rp};
  }
};
struct SignatureVarGen {
  static constexpr size_t SIZE = 64;
  JubJubScalar u_;
  JubJubExtended R_;
  const JubJubScalar& u() const { return u_; }
  const JubJubExtended& R() const { return R_; }
  bool operator==(const SignatureVarGen& o) const { return u_ == o.u_ && R_ == o.R_; }
  std::array<uint8_t, SIZE> to_bytes() const {
    std::array<uint8_t, SIZE> b;
    std::memcpy(b.data(), u_.to_bytes().data(), 32);
    std::memcpy(b.data() + 32, R_.to_bytes().data(), 32);
    return b;
  }
  static std::optional<SignatureVarGen> from_bytes(const std::array<uint8_t, SIZE>& b) {
    auto u = JubJubScalar::from_bytes(b.data());
    auto R = JubJubExtended::from_bytes(b.data() + 32);
    if (!u || !R) return std::nullopt;
    return SignatureVarGen{*u, *R};
  }
};
static_assert(sizeof(Signature) == 192 && sizeof(SignatureDouble) == 352 && sizeof(SignatureVarGen) == 192,
              "the Rust structs' sizes");

namespace detail {
inline JubJubScalar scalar_from_engine(const uint8_t b[32]) {
  auto s = JubJubScalar::from_bytes(b);
  if (!s) throw std::runtime_error("engine returned a non-canonical scalar");
  return *s;
}
// sk * G (which = 0), sk * G' (1), or sk * gen: the engine's key-derivation kernels take canonical bytes
inline JubJubExtended derive(const JubJubScalar& sk, int which, const JubJubExtended* gen, const char* what) {
  ensure_init();
  uint8_t out[64];
  if (gen) {
    const auto g = gen->to_affine().to_canonical();
    check(dsv_public_keys(sk.to_bytes().data(), which, g.data(), 1, out), what);
  } else {
    check(dsv_public_keys(sk.to_bytes().data(), which, nullptr, 1, out), what);
  }
  return JubJubExtended::from(JubJubAffine::from_canonical(out));
}
}  // namespace detail

struct SecretKey {
  JubJubScalar sk;
  template <class Rng>
  static SecretKey random(Rng& rng) { return SecretKey{JubJubScalar::random(rng)}; }
  bool operator==(const SecretKey& o) const { return sk == o.sk; }
  std::array<uint8_t, 32> to_bytes() const { return sk.to_bytes(); }
  static std::optional<SecretKey> from_bytes(const std::array<uint8_t, 32>& b) {
    auto s = JubJubScalar::from_bytes(b.data());
    if (!s) return std::nullopt;
    return SecretKey{*s};
  }
  // sign: r <- rng; R = r*G; c = H(R, m); u = r - c*sk
  template <class Rng>
  Signature sign(Rng& rng, const BlsScalar& message) const {
    detail::ensure_init();
    const JubJubScalar r = JubJubScalar::random(rng);
    uint8_t u[32], R[64];
    detail::check(dsv_sign_single(sk.to_bytes().data(), message.to_bytes().data(), r.to_bytes().data(), 1,
                                  u, R), "dsv_sign_single");
    return Signature{detail::scalar_from_engine(u), JubJubExtended::from(JubJubAffine::from_canonical(R))};
  }
  template <class Rng>
  SignatureDouble sign_double(Rng& rng, const BlsScalar& message) const {
    detail::ensure_init();
    const JubJubScalar r = JubJubScalar::random(rng);
    uint8_t u[32], R[64], Rp[64];
    detail::check(dsv_sign_double(sk.to_bytes().data(), message.to_bytes().data(), r.to_bytes().data(), 1,
                                  u, R, Rp), "dsv_sign_double");
    return SignatureDouble{detail::scalar_from_engine(u), JubJubExtended::from(JubJubAffine::from_canonical(R)),
                           JubJubExtended::from(JubJubAffine::from_canonical(Rp))};
  }
};

struct PublicKey {
  JubJubExtended pk;
  static PublicKey from(const SecretKey& sk) { return PublicKey{detail::derive(sk.sk, 0, nullptr, "dsv_public_keys")}; }
  const JubJubExtended& as_ref() const { return pk; }
  // public.rs:142 from_raw_unchecked: any coordinates, no validation
  static PublicKey from_raw_unchecked(const JubJubExtended& p) { return PublicKey{p}; }
  std::array<uint8_t, 32> to_bytes() const { return pk.to_bytes(); }
  static std::optional<PublicKey> from_bytes(const std::array<uint8_t, 32>& b) {
    auto p = JubJubExtended::from_bytes(b.data());
    if (!p) return std::nullopt;
    return PublicKey{*p};
  }
  // u*G + c*PK == R  with c = H(R || m): the limbs of (u, R, pk, m) go to the engine as they are
  bool verify(const Signature& sig, const BlsScalar& message) const {
    detail::ensure_init();
    uint8_t ok = 0;
    const dsv_column cols[4] = {{&sig.u_, sizeof sig}, {&sig.R_, sizeof sig}, {&pk, sizeof *this}, {&message, 32}};
    detail::check(dsv_verify_single_mont_cols(cols, 1, &ok), "dsv_verify_single_mont_cols");
    return ok == 1;
  }
  bool operator==(const PublicKey& o) const { return pk == o.pk; }
};

struct PublicKeyDouble {
  JubJubExtended pk_, pk_prime_;
  static PublicKeyDouble from(const SecretKey& sk) {
    return PublicKeyDouble{detail::derive(sk.sk, 0, nullptr, "pk"), detail::derive(sk.sk, 1, nullptr, "pk'")};
  }
  static PublicKeyDouble from_raw_unchecked(const JubJubExtended& p, const JubJubExtended& pp) {
    return PublicKeyDouble{p, pp};
  }
  const JubJubExtended& pk() const { return pk_; }
  const JubJubExtended& pk_prime() const { return pk_prime_; }
  bool operator==(const PublicKeyDouble& o) const { return pk_ == o.pk_ && pk_prime_ == o.pk_prime_; }
  std::array<uint8_t, 64> to_bytes() const {  // pk || pk'
    std::array<uint8_t, 64> b;
    std::memcpy(b.data(), pk_.to_bytes().data(), 32);
    std::memcpy(b.data() + 32, pk_prime_.to_bytes().data(), 32);
    return b;
  }
  static std::optional<PublicKeyDouble> from_bytes(const std::array<uint8_t, 64>& b) {
    auto p = JubJubExtended::from_bytes(b.data());
    auto pp = JubJubExtended::from_bytes(b.data() + 32);
    if (!p || !pp) return std::nullopt;
    return PublicKeyDouble{*p, *pp};
  }
  bool verify(const SignatureDouble& sig, const BlsScalar& message) const {
    detail::ensure_init();
    uint8_t ok = 0;
    const dsv_column cols[6] = {{&sig.u_, sizeof sig},        {&sig.R_, sizeof sig}, {&sig.R_prime_, sizeof sig},
                                {&pk_, sizeof *this},         {&pk_prime_, sizeof *this}, {&message, 32}};
    detail::check(dsv_verify_double_mont_cols(cols, 1, &ok), "dsv_verify_double_mont_cols");
    return ok == 1;
  }
};

struct SecretKeyVarGen {
  JubJubScalar sk;
  JubJubExtended generator_;
  static SecretKeyVarGen make(const JubJubScalar& sk, const JubJubExtended& generator) {
    return SecretKeyVarGen{sk, generator};
  }
  // random: sk, then a generator scalar g; generator = g * G   (secret.rs:367-376)
  template <class Rng>
  static SecretKeyVarGen random(Rng& rng) {
    SecretKeyVarGen k;
    k.sk = JubJubScalar::random(rng);
    const JubJubScalar g = JubJubScalar::random(rng);
    k.generator_ = detail::derive(g, 0, nullptr, "gen");
    return k;
  }
  const JubJubExtended& generator() const { return generator_; }
  bool operator==(const SecretKeyVarGen& o) const { return sk == o.sk && generator_ == o.generator_; }
  std::array<uint8_t, 64> to_bytes() const {  // sk || compressed generator
    std::array<uint8_t, 64> b;
    std::memcpy(b.data(), sk.to_bytes().data(), 32);
    std::memcpy(b.data() + 32, generator_.to_bytes().data(), 32);
    return b;
  }
  static std::optional<SecretKeyVarGen> from_bytes(const std::array<uint8_t, 64>& b) {
    auto s = JubJubScalar::from_bytes(b.data());
    auto g = JubJubExtended::from_bytes(b.data() + 32);
    if (!s || !g) return std::nullopt;
    return SecretKeyVarGen{*s, *g};
  }
  template <class Rng>
  SignatureVarGen sign(Rng& rng, const BlsScalar& message) const {
    detail::ensure_init();
    const JubJubScalar r = JubJubScalar::random(rng);
    const auto gen = generator_.to_affine().to_canonical();  // the signing kernels take affine bases
    uint8_t u[32], R[64];
    detail::check(dsv_sign_vargen(sk.to_bytes().data(), gen.data(), message.to_bytes().data(),
                                  r.to_bytes().data(), 1, u, R), "dsv_sign_vargen");
    return SignatureVarGen{detail::scalar_from_engine(u), JubJubExtended::from(JubJubAffine::from_canonical(R))};
  }
};

struct PublicKeyVarGen {
  JubJubExtended pk_, generator_;
  static PublicKeyVarGen from(const SecretKeyVarGen& sk) {
    return PublicKeyVarGen{detail::derive(sk.sk, 0, &sk.generator_, "dsv_public_keys(gen)"), sk.generator_};
  }
  static PublicKeyVarGen from_raw_unchecked(const JubJubExtended& pk, const JubJubExtended& gen) {
    return PublicKeyVarGen{pk, gen};
  }
  const JubJubExtended& public_key() const { return pk_; }
  const JubJubExtended& generator() const { return generator_; }
  bool operator==(const PublicKeyVarGen& o) const { return pk_ == o.pk_ && generator_ == o.generator_; }
  std::array<uint8_t, 64> to_bytes() const {  // compressed pk || compressed generator
    std::array<uint8_t, 64> b;
    std::memcpy(b.data(), pk_.to_bytes().data(), 32);
    std::memcpy(b.data() + 32, generator_.to_bytes().data(), 32);
    return b;
  }
  static std::optional<PublicKeyVarGen> from_bytes(const std::array<uint8_t, 64>& b) {
    auto p = JubJubExtended::from_bytes(b.data());
    auto g = JubJubExtended::from_bytes(b.data() + 32);
    if (!p || !g) return std::nullopt;
    return PublicKeyVarGen{*p, *g};
  }
  bool verify(const SignatureVarGen& sig, const BlsScalar& message) const {
    detail::ensure_init();
    uint8_t ok = 0;
    const dsv_column cols[5] = {{&sig.u_, sizeof sig}, {&sig.R_, sizeof sig}, {&pk_, sizeof *this},
                                {&generator_, sizeof *this}, {&message, 32}};
    detail::check(dsv_verify_vargen_mont_cols(cols, 1, &ok), "dsv_verify_vargen_mont_cols");
    return ok == 1;
  }
};
static_assert(sizeof(PublicKey) == 160 && sizeof(PublicKeyDouble) == 320 && sizeof(PublicKeyVarGen) == 320,
              "the Rust structs' sizes");

// ---- the new batch entry points (north_star): out[i] == pks[i].verify(sigs[i], msgs[i]) --------
// No copy and no arithmetic here: the engine reads u / R / pk / m out of the typed objects through
// one strided column per field (dsv_verify_*_mont_cols) and shards over every initialised GPU.
// `verify_batch_bytes*` return the engine's verdict bytes (1 = true) for callers that do not want
// the std::vector<bool> packing pass.
inline std::vector<uint8_t> verify_batch_bytes(const Signature* sigs, const PublicKey* pks,
                                               const BlsScalar* msgs, size_t n) {
  detail::ensure_init();
  std::vector<uint8_t> ok(n);
  if (!n) return ok;
  const dsv_column cols[4] = {{&sigs->u_, sizeof *sigs}, {&sigs->R_, sizeof *sigs}, {&pks->pk, sizeof *pks}, {msgs, 32}};
  detail::check(dsv_verify_single_mont_cols(cols, n, ok.data()), "dsv_verify_single_mont_cols");
  return ok;
}
inline std::vector<uint8_t> verify_batch_double_bytes(const SignatureDouble* sigs, const PublicKeyDouble* pks,
                                                      const BlsScalar* msgs, size_t n) {
  detail::ensure_init();
  std::vector<uint8_t> ok(n);
  if (!n) return ok;
  const dsv_column cols[6] = {{&sigs->u_, sizeof *sigs},  {&sigs->R_, sizeof *sigs},       {&sigs->R_prime_, sizeof *sigs},
                              {&pks->pk_, sizeof *pks},   {&pks->pk_prime_, sizeof *pks},  {msgs, 32}};
  detail::check(dsv_verify_double_mont_cols(cols, n, ok.data()), "dsv_verify_double_mont_cols");
  return ok;
}
inline std::vector<uint8_t> verify_batch_var_gen_bytes(const SignatureVarGen* sigs, const PublicKeyVarGen* pks,
                                                       const BlsScalar* msgs, size_t n) {
  detail::ensure_init();
  std::vector<uint8_t> ok(n);
  if (!n) return ok;
  const dsv_column cols[5] = {{&sigs->u_, sizeof *sigs}, {&sigs->R_, sizeof *sigs}, {&pks->pk_, sizeof *pks},
                              {&pks->generator_, sizeof *pks}, {msgs, 32}};
  detail::check(dsv_verify_vargen_mont_cols(cols, n, ok.data()), "dsv_verify_vargen_mont_cols");
  return ok;
}
namespace detail {
// verdict bytes -> std::vector<bool> (true iff the byte is 1).  The element-wise loop costs ~2 ms per
// 2^20 verdicts — 10 % of a whole verify_batch on one MI355X — and is an artefact of C++'s bit-packed
// vector<bool> (a Rust Vec<bool> IS the byte vector): with libstdc++ 64 verdicts are packed per
// store through the word pointer its iterator exposes, ~0.1 ms per 2^20.
inline std::vector<bool> to_bools(const std::vector<uint8_t>& ok) {
  std::vector<bool> out(ok.size());
  size_t i = 0;
#if defined(__GLIBCXX__) && !defined(_GLIBCXX_DEBUG)
  if constexpr (sizeof(std::_Bit_type) == 8) {
    std::_Bit_type* w = out.begin()._M_p;
    for (; i + 64 <= ok.size(); i += 64) {
      uint64_t word = 0;
      for (int b = 0; b < 8; b++) {
        uint64_t x;
        std::memcpy(&x, ok.data() + i + 8 * b, 8);
        x ^= 0x0101010101010101ULL;  // a byte of x is zero iff the verdict byte is exactly 1
        const uint64_t nz = ((x & 0x7f7f7f7f7f7f7f7fULL) + 0x7f7f7f7f7f7f7f7fULL) | x;  // bit 7 set iff non-zero
        const uint64_t ones = (~nz >> 7) & 0x0101010101010101ULL;
        word |= ((ones * 0x0102040810204080ULL) >> 56) << (8 * b);  // the eight low bits side by side
      }
      *w++ = word;
    }
  }
#endif
  for (; i < ok.size(); i++) out[i] = ok[i] == 1;
  return out;
}
inline void same_len(size_t a, size_t b, size_t c, const char* what) {
  if (a != b || a != c) throw std::invalid_argument(std::string(what) + ": slice lengths differ");
}
}  // namespace detail
inline std::vector<bool> verify_batch(const std::vector<Signature>& sigs, const std::vector<PublicKey>& pks,
                                      const std::vector<BlsScalar>& msgs) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch");
  return detail::to_bools(verify_batch_bytes(sigs.data(), pks.data(), msgs.data(), sigs.size()));
}
inline std::vector<bool> verify_batch_double(const std::vector<SignatureDouble>& sigs,
                                             const std::vector<PublicKeyDouble>& pks,
                                             const std::vector<BlsScalar>& msgs) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch_double");
  return detail::to_bools(verify_batch_double_bytes(sigs.data(), pks.data(), msgs.data(), sigs.size()));
}
inline std::vector<bool> verify_batch_var_gen(const std::vector<SignatureVarGen>& sigs,
                                              const std::vector<PublicKeyVarGen>& pks,
                                              const std::vector<BlsScalar>& msgs) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch_var_gen");
  return detail::to_bools(verify_batch_var_gen_bytes(sigs.data(), pks.data(), msgs.data(), sigs.size()));
}

// ---- batches in flight -----------------------------------------------------------------------
// ---- batches expected to be entirely valid: the fast accept (engine: dsv_verify_*_mont_cols_rlc) ----
// The SAME verdict vector as verify_batch*; what changes is the time.  One random-linear-combination
// aggregate over the whole batch (<= 2^22 items; larger ones take the ordinary path) decides "every
// item is valid" ~2x faster than checking them one by one; a batch that holds anything else — a wrong
// signature, a point with a small-order component — is then checked item by item as usual (and has
// paid for both).  `*accepted` (optional) reports which of the two happened.  Error probability of a
// wrong accept <= 2^-112 (weights from getrandom, fresh per call); verify_batch* has none.
inline std::vector<uint8_t> verify_batch_fast_bytes(const Signature* sigs, const PublicKey* pks,
                                                    const BlsScalar* msgs, size_t n, bool* accepted = nullptr) {
  detail::ensure_init();
  std::vector<uint8_t> ok(n);
  int acc = 0;
  if (n) {
    const dsv_column cols[4] = {{&sigs->u_, sizeof *sigs}, {&sigs->R_, sizeof *sigs}, {&pks->pk, sizeof *pks}, {msgs, 32}};
    detail::check(dsv_verify_single_mont_cols_rlc(cols, n, ok.data(), &acc), "dsv_verify_single_mont_cols_rlc");
  }
  if (accepted) *accepted = acc == 1;
  return ok;
}
inline std::vector<uint8_t> verify_batch_double_fast_bytes(const SignatureDouble* sigs, const PublicKeyDouble* pks,
                                                           const BlsScalar* msgs, size_t n, bool* accepted = nullptr) {
  detail::ensure_init();
  std::vector<uint8_t> ok(n);
  int acc = 0;
  if (n) {
    const dsv_column cols[6] = {{&sigs->u_, sizeof *sigs},  {&sigs->R_, sizeof *sigs},       {&sigs->R_prime_, sizeof *sigs},
                                {&pks->pk_, sizeof *pks},   {&pks->pk_prime_, sizeof *pks},  {msgs, 32}};
    detail::check(dsv_verify_double_mont_cols_rlc(cols, n, ok.data(), &acc), "dsv_verify_double_mont_cols_rlc");
  }
  if (accepted) *accepted = acc == 1;
  return ok;
}
inline std::vector<uint8_t> verify_batch_var_gen_fast_bytes(const SignatureVarGen* sigs, const PublicKeyVarGen* pks,
                                                            const BlsScalar* msgs, size_t n, bool* accepted = nullptr) {
  detail::ensure_init();
  std::vector<uint8_t> ok(n);
  int acc = 0;
  if (n) {
    const dsv_column cols[5] = {{&sigs->u_, sizeof *sigs}, {&sigs->R_, sizeof *sigs}, {&pks->pk_, sizeof *pks},
                                {&pks->generator_, sizeof *pks}, {msgs, 32}};
    detail::check(dsv_verify_vargen_mont_cols_rlc(cols, n, ok.data(), &acc), "dsv_verify_vargen_mont_cols_rlc");
  }
  if (accepted) *accepted = acc == 1;
  return ok;
}
inline std::vector<bool> verify_batch_fast(const std::vector<Signature>& sigs, const std::vector<PublicKey>& pks,
                                           const std::vector<BlsScalar>& msgs, bool* accepted = nullptr) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch_fast");
  return detail::to_bools(verify_batch_fast_bytes(sigs.data(), pks.data(), msgs.data(), sigs.size(), accepted));
}
inline std::vector<bool> verify_batch_double_fast(const std::vector<SignatureDouble>& sigs,
                                                  const std::vector<PublicKeyDouble>& pks,
                                                  const std::vector<BlsScalar>& msgs, bool* accepted = nullptr) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch_double_fast");
  return detail::to_bools(verify_batch_double_fast_bytes(sigs.data(), pks.data(), msgs.data(), sigs.size(), accepted));
}
inline std::vector<bool> verify_batch_var_gen_fast(const std::vector<SignatureVarGen>& sigs,
                                                   const std::vector<PublicKeyVarGen>& pks,
                                                   const std::vector<BlsScalar>& msgs, bool* accepted = nullptr) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch_var_gen_fast");
  return detail::to_bools(verify_batch_var_gen_fast_bytes(sigs.data(), pks.data(), msgs.data(), sigs.size(), accepted));
}

// `verify_batch*_submit` start a batch and return at once; `wait()` blocks and returns what
// verify_batch* would have returned.  Two batches in flight per GPU overlap: the second one's ramp
// (gathering and transferring its first chunk, small first sub-batches) runs while the first one's
// last chunks are on the GPU (dsv.h: dsv_verify_*_mont_cols_submit).  The typed objects must outlive
// the wait; a BatchJob that is dropped waits in its destructor.
class BatchJob {
 public:
  BatchJob() = default;
  BatchJob(BatchJob&& o) noexcept : job_(o.job_), ok_(std::move(o.ok_)) { o.job_ = nullptr; }
  BatchJob& operator=(BatchJob&& o) noexcept {
    if (this != &o) {
      drop();
      job_ = o.job_;
      ok_ = std::move(o.ok_);
      o.job_ = nullptr;
    }
    return *this;
  }
  BatchJob(const BatchJob&) = delete;
  BatchJob& operator=(const BatchJob&) = delete;
  ~BatchJob() { drop(); }
  bool done() const { return !job_ || dsv_job_done(job_) == 1; }
  std::vector<uint8_t> wait_bytes() {  // the engine's verdict bytes (1 = true)
    if (job_) {
      dsv_job* j = job_;
      job_ = nullptr;
      detail::check(dsv_job_wait(j), "dsv_job_wait");
    }
    return std::move(ok_);
  }
  std::vector<bool> wait() { return detail::to_bools(wait_bytes()); }

 private:
  friend BatchJob verify_batch_submit(const Signature*, const PublicKey*, const BlsScalar*, size_t);
  friend BatchJob verify_batch_double_submit(const SignatureDouble*, const PublicKeyDouble*, const BlsScalar*, size_t);
  friend BatchJob verify_batch_var_gen_submit(const SignatureVarGen*, const PublicKeyVarGen*, const BlsScalar*, size_t);
  void drop() noexcept {
    if (job_) (void)dsv_job_wait(job_);
    job_ = nullptr;
  }
  dsv_job* job_ = nullptr;
  std::vector<uint8_t> ok_;
};
inline BatchJob verify_batch_submit(const Signature* sigs, const PublicKey* pks, const BlsScalar* msgs, size_t n) {
  detail::ensure_init();
  BatchJob j;
  j.ok_.assign(n, 0);
  if (!n) return j;
  const dsv_column cols[4] = {{&sigs->u_, sizeof *sigs}, {&sigs->R_, sizeof *sigs}, {&pks->pk, sizeof *pks}, {msgs, 32}};
  detail::check(dsv_verify_single_mont_cols_submit(cols, n, j.ok_.data(), &j.job_), "dsv_verify_single_mont_cols_submit");
  return j;
}
inline BatchJob verify_batch_double_submit(const SignatureDouble* sigs, const PublicKeyDouble* pks,
                                           const BlsScalar* msgs, size_t n) {
  detail::ensure_init();
  BatchJob j;
  j.ok_.assign(n, 0);
  if (!n) return j;
  const dsv_column cols[6] = {{&sigs->u_, sizeof *sigs},  {&sigs->R_, sizeof *sigs},       {&sigs->R_prime_, sizeof *sigs},
                              {&pks->pk_, sizeof *pks},   {&pks->pk_prime_, sizeof *pks},  {msgs, 32}};
  detail::check(dsv_verify_double_mont_cols_submit(cols, n, j.ok_.data(), &j.job_), "dsv_verify_double_mont_cols_submit");
  return j;
}
inline BatchJob verify_batch_var_gen_submit(const SignatureVarGen* sigs, const PublicKeyVarGen* pks,
                                            const BlsScalar* msgs, size_t n) {
  detail::ensure_init();
  BatchJob j;
  j.ok_.assign(n, 0);
  if (!n) return j;
  const dsv_column cols[5] = {{&sigs->u_, sizeof *sigs}, {&sigs->R_, sizeof *sigs}, {&pks->pk_, sizeof *pks},
                              {&pks->generator_, sizeof *pks}, {msgs, 32}};
  detail::check(dsv_verify_vargen_mont_cols_submit(cols, n, j.ok_.data(), &j.job_), "dsv_verify_vargen_mont_cols_submit");
  return j;
}
inline BatchJob verify_batch_submit(const std::vector<Signature>& sigs, const std::vector<PublicKey>& pks,
                                    const std::vector<BlsScalar>& msgs) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch_submit");
  return verify_batch_submit(sigs.data(), pks.data(), msgs.data(), sigs.size());
}
inline BatchJob verify_batch_double_submit(const std::vector<SignatureDouble>& sigs,
                                           const std::vector<PublicKeyDouble>& pks,
                                           const std::vector<BlsScalar>& msgs) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch_double_submit");
  return verify_batch_double_submit(sigs.data(), pks.data(), msgs.data(), sigs.size());
}
inline BatchJob verify_batch_var_gen_submit(const std::vector<SignatureVarGen>& sigs,
                                            const std::vector<PublicKeyVarGen>& pks,
                                            const std::vector<BlsScalar>& msgs) {
  detail::same_len(sigs.size(), pks.size(), msgs.size(), "verify_batch_var_gen_submit");
  return verify_batch_var_gen_submit(sigs.data(), pks.data(), msgs.data(), sigs.size());
}

}  // namespace dusk_schnorr
