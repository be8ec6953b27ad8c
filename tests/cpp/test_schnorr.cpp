// C++ counterpart of the reference's native tests, through include/dusk_schnorr.hpp -> libdsv.so:
//   /root/reference/tests/schnorr.rs:14-40              sign_verify, test_wrong_keys
//   /root/reference/tests/schnorr_double.rs:14-41       same for SignatureDouble
//   /root/reference/tests/schnorr_var_generator.rs:14-40 same for SignatureVarGen
//   to_from_bytes of all three (tests/schnorr.rs:42-51, schnorr_double.rs:43-52,
//   schnorr_var_generator.rs:43-52) and of the key types, with the Err cases
// plus the new verify_batch entry points.  Exit code 0 = all passed.
#include <cstdio>
#include <cstdlib>

#include "dusk_schnorr.hpp"

using namespace dusk_schnorr;

// deterministic test RNG (splitmix64); the reference uses StdRng::seed_from_u64(2321)
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
  }
  void operator()(uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(next() >> 32);
  }
};

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                        \
    }                                                                      \
  } while (0)

static void sign_verify() {
  Rng rng(2321);
  SecretKey sk = SecretKey::random(rng);
  BlsScalar message = BlsScalar::random(rng);
  PublicKey pk = PublicKey::from(sk);
  Signature sig = sk.sign(rng, message);
  CHECK(pk.verify(sig, message));
}
static void test_wrong_keys() {
  Rng rng(2321);
  SecretKey sk = SecretKey::random(rng);
  BlsScalar message = BlsScalar::random(rng);
  Signature sig = sk.sign(rng, message);
  SecretKey wrong_sk = SecretKey::random(rng);
  PublicKey pk = PublicKey::from(wrong_sk);
  CHECK(!pk.verify(sig, message));
}
static void sign_verify_double() {
  Rng rng(2321);
  SecretKey sk = SecretKey::random(rng);
  BlsScalar message = BlsScalar::random(rng);
  PublicKeyDouble pk = PublicKeyDouble::from(sk);
  SignatureDouble sig = sk.sign_double(rng, message);
  CHECK(pk.verify(sig, message));
  SecretKey wrong_sk = SecretKey::random(rng);
  CHECK(!PublicKeyDouble::from(wrong_sk).verify(sig, message));
}
static void sign_verify_var_gen() {
  Rng rng(2321);
  SecretKeyVarGen sk = SecretKeyVarGen::random(rng);
  BlsScalar message = BlsScalar::random(rng);
  PublicKeyVarGen pk = PublicKeyVarGen::from(sk);
  SignatureVarGen sig = sk.sign(rng, message);
  CHECK(pk.verify(sig, message));
  SecretKeyVarGen wrong_sk = SecretKeyVarGen::random(rng);
  CHECK(!PublicKeyVarGen::from(wrong_sk).verify(sig, message));
}
static void batch() {
  Rng rng(77);
  const size_t n = 100;
  std::vector<Signature> sigs;
  std::vector<PublicKey> pks;
  std::vector<BlsScalar> msgs;
  for (size_t i = 0; i < n; i++) {
    SecretKey sk = SecretKey::random(rng);
    BlsScalar m = BlsScalar::random(rng);
    sigs.push_back(sk.sign(rng, m));
    pks.push_back(PublicKey::from(sk));
    msgs.push_back(m);
  }
  pks[3] = pks[4];                                   // wrong key
  msgs[10] = msgs[10] + BlsScalar::one();            // wrong message
  sigs[20].u_ = sigs[20].u_ + JubJubScalar::from(64);  // corrupted u
  std::vector<bool> ok = verify_batch(sigs, pks, msgs);
  CHECK(ok.size() == n);
  for (size_t i = 0; i < n; i++) {
    const bool want = !(i == 3 || i == 10 || i == 20);
    CHECK(ok[i] == want);
    CHECK(pks[i].verify(sigs[i], msgs[i]) == want);
  }
  // batches in flight: three jobs over the same objects (one more than the engine keeps in flight),
  // a job that is moved, one that is dropped unwaited; 200-verdict words through the 64-at-a-time packing
  {
    BatchJob a = verify_batch_submit(sigs, pks, msgs), b = verify_batch_submit(sigs, pks, msgs);
    BatchJob c = verify_batch_submit(sigs, pks, msgs);
    BatchJob moved = std::move(b);
    CHECK(a.wait() == ok && moved.wait() == ok);
    CHECK(c.done() || true);  // (dropped: its destructor waits)
    std::vector<Signature> s2;
    std::vector<PublicKey> p2;
    std::vector<BlsScalar> m2;
    for (size_t i = 0; i < 200; i++) {
      s2.push_back(sigs[i % n]);
      p2.push_back(pks[i % n]);
      m2.push_back(msgs[i % n]);
    }
    const std::vector<bool> ok2 = verify_batch_submit(s2, p2, m2).wait();
    CHECK(ok2.size() == 200);
    for (size_t i = 0; i < 200; i++) CHECK(ok2[i] == ok[i % n]);
    CHECK(verify_batch_submit(std::vector<Signature>{}, {}, {}).wait().empty());
  }
  bool threw = false;
  try {
    msgs.pop_back();
    verify_batch(sigs, pks, msgs);
  } catch (const std::invalid_argument&) {
    threw = true;
  }
  CHECK(threw);
  CHECK(verify_batch({}, {}, {}).empty());
}
// verify_batch_double / verify_batch_var_gen over typed objects (six and five strided columns), and a
// batch long enough for several pipeline chunks (the objects of a small signed set repeated)
static void batch_double_vargen_and_chunks() {
  Rng rng(99);
  const size_t n = 60;
  std::vector<SignatureDouble> dsigs;
  std::vector<PublicKeyDouble> dpks;
  std::vector<SignatureVarGen> vsigs;
  std::vector<PublicKeyVarGen> vpks;
  std::vector<Signature> sigs;
  std::vector<PublicKey> pks;
  std::vector<BlsScalar> msgs;
  for (size_t i = 0; i < n; i++) {
    SecretKey sk = SecretKey::random(rng);
    BlsScalar m = BlsScalar::random(rng);
    msgs.push_back(m);
    dsigs.push_back(sk.sign_double(rng, m));
    dpks.push_back(PublicKeyDouble::from(sk));
    sigs.push_back(sk.sign(rng, m));
    pks.push_back(PublicKey::from(sk));
    SecretKeyVarGen skv = SecretKeyVarGen::random(rng);
    vsigs.push_back(skv.sign(rng, m));
    vpks.push_back(PublicKeyVarGen::from(skv));
  }
  std::swap(dsigs[5].R_, dsigs[5].R_prime_);            // R and R' exchanged
  dpks[11] = dpks[12];                                   // wrong key pair
  vpks[7] = PublicKeyVarGen::from_raw_unchecked(vpks[7].public_key(), vpks[8].generator());  // wrong generator
  vsigs[20].u_ = vsigs[20].u_ + JubJubScalar::one();
  const std::vector<bool> okd = verify_batch_double(dsigs, dpks, msgs), okv = verify_batch_var_gen(vsigs, vpks, msgs);
  for (size_t i = 0; i < n; i++) {
    CHECK(okd[i] == !(i == 5 || i == 11) && okd[i] == dpks[i].verify(dsigs[i], msgs[i]));
    CHECK(okv[i] == !(i == 7 || i == 20) && okv[i] == vpks[i].verify(vsigs[i], msgs[i]));
  }
  // 70 020 single signatures: two chunks and a ragged sub-batch through the strided gather
  sigs[3].R_ = sigs[4].R_;
  const size_t reps = 1167;
  std::vector<Signature> big_s;
  std::vector<PublicKey> big_p;
  std::vector<BlsScalar> big_m;
  for (size_t r = 0; r < reps; r++) {
    big_s.insert(big_s.end(), sigs.begin(), sigs.end());
    big_p.insert(big_p.end(), pks.begin(), pks.end());
    big_m.insert(big_m.end(), msgs.begin(), msgs.end());
  }
  const std::vector<uint8_t> okb = verify_batch_bytes(big_s.data(), big_p.data(), big_m.data(), big_s.size());
  CHECK(okb.size() == n * reps);
  for (size_t i = 0; i < okb.size(); i++) CHECK(okb[i] == (i % n == 3 ? 0 : 1));
  // the fast accept over the same objects (verify_batch_fast*): the same verdicts.  70 020 items are too
  // few for an aggregate (< 2^17: the ordinary path); doubled, the wrong items send it to the
  // per-signature kernels, and with those repaired the aggregate decides
  bool accepted = true;
  CHECK(verify_batch_fast_bytes(big_s.data(), big_p.data(), big_m.data(), big_s.size(), &accepted) == okb && !accepted);
  auto twice = [](auto& v) { v.insert(v.end(), v.begin(), v.begin() + (long)v.size()); };
  twice(big_s), twice(big_p), twice(big_m);
  CHECK(big_s.size() >= ((size_t)1 << 17));
  std::vector<uint8_t> okb2 = okb;
  twice(okb2);
  CHECK(verify_batch_fast_bytes(big_s.data(), big_p.data(), big_m.data(), big_s.size(), &accepted) == okb2 && !accepted);
  for (size_t r = 0; r < 2 * reps; r++) big_s[r * n + 3] = big_s[r * n + 4], big_p[r * n + 3] = big_p[r * n + 4], big_m[r * n + 3] = big_m[r * n + 4];
  const std::vector<bool> fast = verify_batch_fast(big_s, big_p, big_m, &accepted);
  CHECK(accepted && fast.size() == 2 * n * reps);
  for (size_t i = 0; i < fast.size(); i++) CHECK(fast[i]);
  // double and var-generator: tampered -> item by item; the valid ones alone, repeated past 2^17 -> aggregate
  CHECK(verify_batch_double_fast(dsigs, dpks, msgs, &accepted) == okd && !accepted);
  CHECK(verify_batch_var_gen_fast(vsigs, vpks, msgs, &accepted) == okv && !accepted);
  std::vector<SignatureDouble> gd;
  std::vector<PublicKeyDouble> gdp;
  std::vector<SignatureVarGen> gv;
  std::vector<PublicKeyVarGen> gvp;
  std::vector<BlsScalar> gdm, gvm;
  for (size_t r = 0; r < 2300; r++)
    for (size_t i = 0; i < n; i++) {
      if (okd[i]) gd.push_back(dsigs[i]), gdp.push_back(dpks[i]), gdm.push_back(msgs[i]);
      if (okv[i]) gv.push_back(vsigs[i]), gvp.push_back(vpks[i]), gvm.push_back(msgs[i]);
    }
  const std::vector<bool> fd = verify_batch_double_fast(gd, gdp, gdm, &accepted);
  CHECK(accepted && fd.size() == 2300 * (n - 2) && fd.size() >= ((size_t)1 << 17));
  for (bool b : fd) CHECK(b);
  const std::vector<bool> fv = verify_batch_var_gen_fast(gv, gvp, gvm, &accepted);
  CHECK(accepted && fv.size() == 2300 * (n - 2));
  for (bool b : fv) CHECK(b);
  CHECK(verify_batch_fast({}, {}, {}, &accepted).empty() && !accepted);
}
static void to_from_bytes() {
  Rng rng(2321);
  SecretKey sk = SecretKey::random(rng);
  BlsScalar message = BlsScalar::random(rng);
  Signature sig = sk.sign(rng, message);
  CHECK(sig == *Signature::from_bytes(sig.to_bytes()));
  SignatureDouble sigd = sk.sign_double(rng, message);
  CHECK(sigd == *SignatureDouble::from_bytes(sigd.to_bytes()));
  SecretKeyVarGen skv = SecretKeyVarGen::random(rng);
  SignatureVarGen sigv = skv.sign(rng, message);
  CHECK(sigv == *SignatureVarGen::from_bytes(sigv.to_bytes()));
  // keys
  CHECK(sk == *SecretKey::from_bytes(sk.to_bytes()));
  PublicKey pk = PublicKey::from(sk);
  CHECK(pk == *PublicKey::from_bytes(pk.to_bytes()));
  PublicKeyDouble pkd = PublicKeyDouble::from(sk);
  CHECK(pkd == *PublicKeyDouble::from_bytes(pkd.to_bytes()));
  CHECK(skv == *SecretKeyVarGen::from_bytes(skv.to_bytes()));
  PublicKeyVarGen pkv = PublicKeyVarGen::from(skv);
  CHECK(pkv == *PublicKeyVarGen::from_bytes(pkv.to_bytes()));
  // a deserialised signature and key still verify
  CHECK(PublicKey::from_bytes(pk.to_bytes())->verify(*Signature::from_bytes(sig.to_bytes()), message));
  // Err cases: scalar >= r, v >= q, v with no square root
  auto bad = sig.to_bytes();
  std::memset(bad.data(), 0xff, 32);
  CHECK(!Signature::from_bytes(bad));
  bad = sig.to_bytes();
  std::memset(bad.data() + 32, 0xff, 32);
  bad[63] = 0x7f;
  CHECK(!Signature::from_bytes(bad));
  int rejected = 0;
  for (uint8_t v = 2; v < 40; v++) {  // about half of all v have no u on the curve
    std::array<uint8_t, 32> b{};
    b[0] = v;
    rejected += !PublicKey::from_bytes(b);
  }
  CHECK(rejected > 5 && rejected < 33);
  std::array<uint8_t, 32> r_bytes = JubJubScalar::modulus_bytes();
  CHECK(!SecretKey::from_bytes(r_bytes));  // exactly r
  r_bytes[0] -= 1;
  CHECK(SecretKey::from_bytes(r_bytes).has_value());  // r - 1
}
// /root/reference/tests/keys.rs:17-60, :62-75, :77-127: the same point in another projective
// representation (all coordinates different) is the same key, verifies the same signatures, and
// serialises to the same bytes — here (k u, k v, k z, t1, k t2) against (u, v, z, t1, t2)
static JubJubExtended rescaled(const JubJubExtended& p, uint64_t k = 2) {
  const BlsScalar f = BlsScalar::from(k);
  return JubJubExtended::from_raw_unchecked(p.u * f, p.v * f, p.z * f, p.t1, p.t2 * f);
}
static void partial_eq_and_projective_inputs() {
  Rng rng(77);
  SecretKey sk = SecretKey::random(rng);
  BlsScalar message = BlsScalar::random(rng);
  PublicKey pk = PublicKey::from(sk);
  PublicKey pk2 = PublicKey::from_raw_unchecked(rescaled(rescaled(pk.as_ref())));
  CHECK(pk.as_ref().u != pk2.as_ref().u && pk.as_ref().v != pk2.as_ref().v && pk.as_ref().z != pk2.as_ref().z);
  CHECK(pk == pk2);
  CHECK(pk.to_bytes() == pk2.to_bytes());
  PublicKey other = PublicKey::from(SecretKey::random(rng));
  CHECK(!(pk2 == other));
  Signature sig = sk.sign(rng, message);
  Signature sig2 = sig;
  sig2.R_ = rescaled(sig.R_);
  CHECK(pk.verify(sig2, message) && pk2.verify(sig, message) && pk2.verify(sig2, message));
  CHECK(!other.verify(sig2, message));
  // batch entry point on re-represented points
  std::vector<bool> ok = verify_batch({sig, sig2, sig2}, {pk2, pk, other}, {message, message, message});
  CHECK(ok[0] && ok[1] && !ok[2]);
  // double and var-generator keys
  PublicKeyDouble pkd = PublicKeyDouble::from(sk);
  PublicKeyDouble pkd2 = PublicKeyDouble::from_raw_unchecked(rescaled(pkd.pk()), rescaled(rescaled(pkd.pk_prime())));
  CHECK(pkd == pkd2);
  SignatureDouble sigd = sk.sign_double(rng, message);
  sigd.R_prime_ = rescaled(sigd.R_prime_);
  CHECK(pkd2.verify(sigd, message) && verify_batch_double({sigd}, {pkd2}, {message})[0]);
  SecretKeyVarGen skv = SecretKeyVarGen::random(rng);
  PublicKeyVarGen pkv = PublicKeyVarGen::from(skv);
  PublicKeyVarGen pkv2 = PublicKeyVarGen::from_raw_unchecked(rescaled(pkv.public_key()), rescaled(pkv.generator()));
  CHECK(pkv == pkv2);
  SignatureVarGen sigv = skv.sign(rng, message);
  CHECK(pkv2.verify(sigv, message) && verify_batch_var_gen({sigv}, {pkv2}, {message})[0]);
  // a signing key whose generator is held un-normalised signs the same way
  SecretKeyVarGen skv2 = SecretKeyVarGen::make(skv.sk, rescaled(skv.generator()));
  CHECK(pkv.verify(skv2.sign(rng, message), message));
}
static void random_is_reduced() {
  uint8_t wide[64];
  for (int i = 0; i < 64; i++) wide[i] = 0xff;
  JubJubScalar s = JubJubScalar::from_bytes_wide(wide);
  // (2^512 - 1) mod r, computed with Python integers
  static const uint8_t want[32] = {0x30, 0x77, 0xe5, 0x95, 0xa4, 0x9a, 0x71, 0x67, 0x26, 0xfc, 0xe3,
                                   0x9c, 0xf0, 0xce, 0xb0, 0x51, 0xa5, 0xe9, 0x26, 0xc0, 0xfa, 0xb7,
                                   0xda, 0x69, 0x88, 0x76, 0x12, 0x8d, 0x7b, 0x54, 0xf6, 0x04};
  CHECK(std::memcmp(s.to_bytes().data(), want, 32) == 0);
}
// The types hold what the reference's hold: Montgomery limbs, R = 2^256 (SURVEY.md Appendix A.1 /
// A.2), and the host arithmetic around them is a field.
static void in_memory_representation() {
  // BlsScalar::one().0 == R mod q; JubJubScalar::one().0 == R mod r
  const uint64_t rq[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
  const uint64_t rr[4] = {0x25f80bb3b99607d9ULL, 0xf315d62f66b6e750ULL, 0x932514eeeb8814f4ULL, 0x09a6fc6f479155c6ULL};
  CHECK(std::memcmp(BlsScalar::one().l, rq, 32) == 0 && std::memcmp(JubJubScalar::one().l, rr, 32) == 0);
  CHECK(std::memcmp(BlsScalar::from(1).l, rq, 32) == 0);
  std::array<uint8_t, 32> one_bytes{};
  one_bytes[0] = 1;
  CHECK(BlsScalar::one().to_bytes() == one_bytes && *JubJubScalar::from_bytes(one_bytes.data()) == JubJubScalar::one());
  Rng rng(5);
  for (int k = 0; k < 50; k++) {
    const BlsScalar a = BlsScalar::random(rng), b = BlsScalar::random(rng), c = BlsScalar::random(rng);
    CHECK((a + b) * c == a * c + b * c);
    CHECK(a - a == BlsScalar::zero() && a + (-a) == BlsScalar::zero());
    CHECK(*BlsScalar::from_bytes(a.to_bytes().data()) == a);
    if (!a.is_zero()) CHECK(a * *a.invert() == BlsScalar::one());
    const JubJubScalar x = JubJubScalar::random(rng), y = JubJubScalar::random(rng);
    CHECK(x * y == y * x && *JubJubScalar::from_bytes(x.to_bytes().data()) == x);
    CHECK(x * *x.invert() == JubJubScalar::one());
  }
  CHECK(!BlsScalar::zero().invert());
  // -1 is the largest canonical value: to_bytes() = q - 1
  auto m1 = (-BlsScalar::one()).to_bytes(), q = BlsScalar::modulus_bytes();
  q[0] -= 1;
  CHECK(m1 == q);
  // PartialEq never panics: a z = 0 "point" compares by cross products (the reference's derive)
  JubJubExtended z0 = JubJubExtended::from_raw_unchecked(BlsScalar::one(), BlsScalar::one(), BlsScalar::zero(),
                                                         BlsScalar::one(), BlsScalar::one());
  CHECK(!(z0 == JubJubExtended{}));
  bool threw = false;
  try {
    z0.to_affine();
  } catch (const std::domain_error&) {
    threw = true;
  }
  CHECK(threw);
  // such a key never verifies anything (the device rejects z = 0), and verify does not throw
  SecretKey sk = SecretKey::random(rng);
  BlsScalar m = BlsScalar::random(rng);
  Signature sig = sk.sign(rng, m);
  CHECK(!PublicKey::from_raw_unchecked(z0).verify(sig, m));
  CHECK(PublicKey::from(sk).verify(sig, m));
}

int main() {
  in_memory_representation();
  to_from_bytes();
  random_is_reduced();
  partial_eq_and_projective_inputs();
  sign_verify();
  test_wrong_keys();
  sign_verify_double();
  sign_verify_var_gen();
  batch();
  batch_double_vargen_and_chunks();
  std::printf("ok: %s\n", dsv_version());
  return 0;
}
