// dusk_schnorr.hpp — host-side mirror (C++17, header-only) of the dusk-schnorr 0.18 key /
// signature surface for the native verification path, on top of the C ABI in dsv.h.
//
// The reference is a Rust crate; this image has no Rust toolchain, so the compiled-language host
// layer is C++ with the reference's names, argument meaning and error behaviour:
//
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
// All arithmetic happens on the GPU (libdsv.so); the only host arithmetic here is the 512-bit
// reduction of `random()` (ff::Field::random = from_bytes_wide of 64 random bytes).
// Keys and signatures hold JubJubExtended points (u, v, z) like the reference's types do; the
// normalisation `to_hash_inputs` that the reference's verify starts with
// (/root/reference/src/signatures.rs:131, :280-281) runs on the device inside the dsv_verify_*_ext
// entry points — verify() and verify_batch*() do no field arithmetic on the host.
#pragma once
#include <array>
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
  // them (dsv_verify_*_ext_multi).  First call: two 75.5 MB window tables per device, ~50 ms each.
  static const bool once = [] {
    const int rc = dsv_init_visible();
    if (rc < 0) check(rc, "dsv_init_visible");  // e.g. DSV_ERR_NO_DEVICE
    return true;
  }();
  (void)once;
}
// x mod m for a 512-bit little-endian x and a 256-bit little-endian modulus (shift-subtract)
inline std::array<uint8_t, 32> mod_wide(const uint8_t x[64], const uint8_t m[32]) {
  uint64_t r[5] = {0, 0, 0, 0, 0}, mod[4];
  std::memcpy(mod, m, 32);
  for (int bit = 511; bit >= 0; bit--) {
    for (int i = 4; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
    r[0] = (r[0] << 1) | ((x[bit >> 3] >> (bit & 7)) & 1);
    // if r >= mod: r -= mod
    uint64_t d[5];
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 5; i++) {
      unsigned __int128 t = (unsigned __int128)r[i] - (i < 4 ? mod[i] : 0) - (uint64_t)borrow;
      d[i] = (uint64_t)t;
      borrow = (t >> 64) & 1;
    }
    if (!borrow) std::memcpy(r, d, sizeof d);
  }
  std::array<uint8_t, 32> out;
  std::memcpy(out.data(), r, 32);
  return out;
}
inline constexpr uint8_t kFrModulus[32] = {
    0xb7, 0x2c, 0xf7, 0xd6, 0x5e, 0x0e, 0x97, 0xd0, 0x82, 0x10, 0xc8, 0xcc, 0x93, 0x20, 0x68, 0xa6,
    0x00, 0x3b, 0x34, 0x01, 0x01, 0x3b, 0x67, 0x06, 0xa9, 0xaf, 0x33, 0x65, 0xea, 0xb4, 0x7d, 0x0e};
inline constexpr uint8_t kFqModulus[32] = {
    0x01, 0x00, 0x00, 0x00, 0xff, 0xff, 0xff, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0x02, 0xa4, 0xbd, 0x53,
    0x05, 0xd8, 0xa1, 0x09, 0x08, 0xd8, 0x39, 0x33, 0x48, 0x7d, 0x9d, 0x29, 0x53, 0xa7, 0xed, 0x73};
}  // namespace detail

template <const uint8_t* MOD>
struct Scalar32 {
  std::array<uint8_t, 32> bytes{};  // canonical little-endian, as `to_bytes()`
  static Scalar32 from_u64(uint64_t x) {
    Scalar32 s;
    std::memcpy(s.bytes.data(), &x, 8);
    return s;
  }
  static Scalar32 from_bytes_wide(const uint8_t wide[64]) {
    Scalar32 s;
    s.bytes = detail::mod_wide(wide, MOD);
    return s;
  }
  // `Field::random`: 64 bytes from the rng, reduced.  Rng: void operator()(uint8_t*, size_t)
  template <class Rng>
  static Scalar32 random(Rng& rng) {
    uint8_t wide[64];
    rng(wide, 64);
    return from_bytes_wide(wide);
  }
  const std::array<uint8_t, 32>& to_bytes() const { return bytes; }
  // Serializable::from_bytes: rejects encodings >= the modulus
  static std::optional<Scalar32> from_bytes(const uint8_t b[32]) {
    for (int i = 31; i >= 0; i--) {
      if (b[i] < MOD[i]) {
        Scalar32 s;
        std::memcpy(s.bytes.data(), b, 32);
        return s;
      }
      if (b[i] > MOD[i]) return std::nullopt;
    }
    return std::nullopt;  // equal to the modulus
  }
  bool operator==(const Scalar32& o) const { return bytes == o.bytes; }
};
using BlsScalar = Scalar32<detail::kFqModulus>;
using JubJubScalar = Scalar32<detail::kFrModulus>;

struct JubJubAffine {  // (u, v), canonical LE — the pair to_hash_inputs() returns
  std::array<uint8_t, 64> uv{};
  bool operator==(const JubJubAffine& o) const { return uv == o.uv; }
  // compressed form: canonical v, bit 255 = lowest bit of u
  std::array<uint8_t, 32> to_bytes() const {
    std::array<uint8_t, 32> out;
    detail::check(dsv_compress_points(uv.data(), 1, out.data()), "dsv_compress_points");
    return out;
  }
  static std::optional<JubJubAffine> from_bytes(const uint8_t b[32]) {
    detail::ensure_init();
    JubJubAffine p;
    uint8_t ok = 0;
    detail::check(dsv_decompress_points(b, 1, p.uv.data(), &ok), "dsv_decompress_points");
    if (!ok) return std::nullopt;
    return p;
  }
};

// (u, v, z) with z != 0, canonical LE each: the coordinates a JubJubExtended holds.  Points that
// come out of this library's own kernels (sign, key derivation, from_bytes) have z = 1; a caller
// may hand in any projective representation (from_raw_unchecked): verify() treats equal points
// alike whatever their z (/root/reference/tests/keys.rs:33-59).
struct JubJubExtended {
  std::array<uint8_t, 96> uvz{};
  JubJubExtended() { uvz[32] = 1; uvz[64] = 1; }  // identity (0, 1, 1), like Default in the reference
  static JubJubExtended from(const JubJubAffine& a) {
    JubJubExtended p;
    std::memcpy(p.uvz.data(), a.uv.data(), 64);
    std::memset(p.uvz.data() + 64, 0, 32);
    p.uvz[64] = 1;
    return p;
  }
  // JubJubExtended::to_hash_inputs / JubJubAffine::from(ext): (u/z, v/z), computed on the GPU
  JubJubAffine to_affine() const {
    ensure_init_();
    JubJubAffine a;
    uint8_t ok = 0;
    detail::check(dsv_to_hash_inputs(uvz.data(), 1, a.uv.data(), &ok), "dsv_to_hash_inputs");
    if (!ok) throw std::domain_error("JubJubExtended with z = 0 (the reference panics here)");
    return a;
  }
  std::array<uint8_t, 32> to_bytes() const { return to_affine().to_bytes(); }
  static std::optional<JubJubExtended> from_bytes(const uint8_t b[32]) {
    auto a = JubJubAffine::from_bytes(b);
    if (!a) return std::nullopt;
    return from(*a);
  }
  // PartialEq of the reference: projective equality (u1 z2 == u2 z1 and v1 z2 == v2 z1)
  bool operator==(const JubJubExtended& o) const { return uvz == o.uvz || to_affine() == o.to_affine(); }

 private:
  static void ensure_init_() { detail::ensure_init(); }
};

struct Signature {
  static constexpr size_t SIZE = 64;  // u || compressed R
  JubJubScalar u_;
  JubJubExtended R_;
  const JubJubScalar& u() const { return u_; }
  const JubJubExtended& R() const { return R_; }
  bool operator==(const Signature& o) const { return u_ == o.u_ && R_ == o.R_; }
  std::array<uint8_t, SIZE> to_bytes() const {
    std::array<uint8_t, SIZE> b;
    std::memcpy(b.data(), u_.bytes.data(), 32);
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
    std::memcpy(b.data(), u_.bytes.data(), 32);
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
    std::memcpy(b.data(), u_.bytes.data(), 32);
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

struct SecretKey {
  JubJubScalar sk;
  template <class Rng>
  static SecretKey random(Rng& rng) { return SecretKey{JubJubScalar::random(rng)}; }
  bool operator==(const SecretKey& o) const { return sk == o.sk; }
  std::array<uint8_t, 32> to_bytes() const { return sk.bytes; }
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
    Signature s;
    JubJubAffine R;
    detail::check(dsv_sign_single(sk.bytes.data(), message.bytes.data(), r.bytes.data(), 1,
                                  s.u_.bytes.data(), R.uv.data()), "dsv_sign_single");
    s.R_ = JubJubExtended::from(R);
    return s;
  }
  template <class Rng>
  SignatureDouble sign_double(Rng& rng, const BlsScalar& message) const {
    detail::ensure_init();
    const JubJubScalar r = JubJubScalar::random(rng);
    SignatureDouble s;
    JubJubAffine R, Rp;
    detail::check(dsv_sign_double(sk.bytes.data(), message.bytes.data(), r.bytes.data(), 1,
                                  s.u_.bytes.data(), R.uv.data(), Rp.uv.data()),
                  "dsv_sign_double");
    s.R_ = JubJubExtended::from(R);
    s.R_prime_ = JubJubExtended::from(Rp);
    return s;
  }
};

namespace detail {
inline JubJubExtended derive(const JubJubScalar& sk, int which, const uint8_t* gen_uv, const char* what) {
  ensure_init();
  JubJubAffine a;
  check(dsv_public_keys(sk.bytes.data(), which, gen_uv, 1, a.uv.data()), what);
  return JubJubExtended::from(a);
}
}  // namespace detail

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
  // u*G + c*PK == R  with c = H(R || m)
  bool verify(const Signature& sig, const BlsScalar& message) const {
    detail::ensure_init();
    uint8_t ok = 0;
    detail::check(dsv_verify_single_ext(sig.u_.bytes.data(), sig.R_.uvz.data(), pk.uvz.data(),
                                        message.bytes.data(), 1, &ok), "dsv_verify_single_ext");
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
    detail::check(dsv_verify_double_ext(sig.u_.bytes.data(), sig.R_.uvz.data(), sig.R_prime_.uvz.data(),
                                        pk_.uvz.data(), pk_prime_.uvz.data(), message.bytes.data(), 1,
                                        &ok), "dsv_verify_double_ext");
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
    detail::ensure_init();
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
    std::memcpy(b.data(), sk.bytes.data(), 32);
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
    SignatureVarGen s;
    const JubJubAffine gen = generator_.to_affine();  // the signing kernels take affine bases
    JubJubAffine R;
    detail::check(dsv_sign_vargen(sk.bytes.data(), gen.uv.data(), message.bytes.data(),
                                  r.bytes.data(), 1, s.u_.bytes.data(), R.uv.data()),
                  "dsv_sign_vargen");
    s.R_ = JubJubExtended::from(R);
    return s;
  }
};

struct PublicKeyVarGen {
  JubJubExtended pk_, generator_;
  static PublicKeyVarGen from(const SecretKeyVarGen& sk) {
    const JubJubAffine gen = sk.generator_.to_affine();
    return PublicKeyVarGen{detail::derive(sk.sk, 0, gen.uv.data(), "dsv_public_keys(gen)"), sk.generator_};
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
    detail::check(dsv_verify_vargen_ext(sig.u_.bytes.data(), sig.R_.uvz.data(), pk_.uvz.data(),
                                        generator_.uvz.data(), message.bytes.data(), 1, &ok),
                  "dsv_verify_vargen_ext");
    return ok == 1;
  }
};

// ---- the new batch entry points (north_star): out[i] == pks[i].verify(sigs[i], msgs[i]) --------
inline std::vector<bool> verify_batch(const std::vector<Signature>& sigs,
                                      const std::vector<PublicKey>& pks,
                                      const std::vector<BlsScalar>& msgs) {
  if (sigs.size() != pks.size() || sigs.size() != msgs.size())
    throw std::invalid_argument("verify_batch: slice lengths differ");
  detail::ensure_init();
  const size_t n = sigs.size();
  std::vector<uint8_t> u(32 * n), r(96 * n), pk(96 * n), m(32 * n), ok(n);
  for (size_t i = 0; i < n; i++) {  // byte copies only: no to_hash_inputs on the host
    std::memcpy(&u[32 * i], sigs[i].u_.bytes.data(), 32);
    std::memcpy(&r[96 * i], sigs[i].R_.uvz.data(), 96);
    std::memcpy(&pk[96 * i], pks[i].pk.uvz.data(), 96);
    std::memcpy(&m[32 * i], msgs[i].bytes.data(), 32);
  }
  detail::check(dsv_verify_single_ext_multi(u.data(), r.data(), pk.data(), m.data(), n, ok.data()),
                "dsv_verify_single_ext_multi");
  std::vector<bool> out(n);
  for (size_t i = 0; i < n; i++) out[i] = ok[i] == 1;
  return out;
}
inline std::vector<bool> verify_batch_double(const std::vector<SignatureDouble>& sigs,
                                             const std::vector<PublicKeyDouble>& pks,
                                             const std::vector<BlsScalar>& msgs) {
  if (sigs.size() != pks.size() || sigs.size() != msgs.size())
    throw std::invalid_argument("verify_batch_double: slice lengths differ");
  detail::ensure_init();
  const size_t n = sigs.size();
  std::vector<uint8_t> u(32 * n), r(96 * n), rp(96 * n), pk(96 * n), pkp(96 * n), m(32 * n), ok(n);
  for (size_t i = 0; i < n; i++) {
    std::memcpy(&u[32 * i], sigs[i].u_.bytes.data(), 32);
    std::memcpy(&r[96 * i], sigs[i].R_.uvz.data(), 96);
    std::memcpy(&rp[96 * i], sigs[i].R_prime_.uvz.data(), 96);
    std::memcpy(&pk[96 * i], pks[i].pk_.uvz.data(), 96);
    std::memcpy(&pkp[96 * i], pks[i].pk_prime_.uvz.data(), 96);
    std::memcpy(&m[32 * i], msgs[i].bytes.data(), 32);
  }
  detail::check(dsv_verify_double_ext_multi(u.data(), r.data(), rp.data(), pk.data(), pkp.data(),
                                            m.data(), n, ok.data()), "dsv_verify_double_ext_multi");
  std::vector<bool> out(n);
  for (size_t i = 0; i < n; i++) out[i] = ok[i] == 1;
  return out;
}
inline std::vector<bool> verify_batch_var_gen(const std::vector<SignatureVarGen>& sigs,
                                              const std::vector<PublicKeyVarGen>& pks,
                                              const std::vector<BlsScalar>& msgs) {
  if (sigs.size() != pks.size() || sigs.size() != msgs.size())
    throw std::invalid_argument("verify_batch_var_gen: slice lengths differ");
  detail::ensure_init();
  const size_t n = sigs.size();
  std::vector<uint8_t> u(32 * n), r(96 * n), pk(96 * n), g(96 * n), m(32 * n), ok(n);
  for (size_t i = 0; i < n; i++) {
    std::memcpy(&u[32 * i], sigs[i].u_.bytes.data(), 32);
    std::memcpy(&r[96 * i], sigs[i].R_.uvz.data(), 96);
    std::memcpy(&pk[96 * i], pks[i].pk_.uvz.data(), 96);
    std::memcpy(&g[96 * i], pks[i].generator_.uvz.data(), 96);
    std::memcpy(&m[32 * i], msgs[i].bytes.data(), 32);
  }
  detail::check(dsv_verify_vargen_ext_multi(u.data(), r.data(), pk.data(), g.data(), m.data(), n,
                                            ok.data()), "dsv_verify_vargen_ext_multi");
  std::vector<bool> out(n);
  for (size_t i = 0; i < n; i++) out[i] = ok[i] == 1;
  return out;
}

}  // namespace dusk_schnorr
