// verify_batch_e2e — times the entry point north_star names, `verify_batch(&[Signature], &[PublicKey],
// &[BlsScalar]) -> Vec<bool>`, END TO END from the typed objects: the C++ mirror
// include/dusk_schnorr.hpp holds the reference's in-memory representation (Montgomery limbs,
// 160-byte JubJubExtended with z != 1: /root/reference/src/keys/public.rs:59, 61-67,
// src/signatures.rs:58-61), so what is timed is what a Rust caller would pay: gathering the
// fields out of 2^20 objects, PCIe, the engine, and the Vec<bool> packing.
//
// Built as a small shared library (tools/libvb_e2e.so, g++; __graft_entry__.build()) and loaded
// by bench.py into ITS process, which hands over the same GPU-signed batch it timed (canonical
// affine bytes).  Not product code: a measuring harness.
//
//   vb_e2e_prepare  canonical bytes -> typed objects (host multiplications by R^2 and by a random z
//                   per point, on several threads; NOT timed): the objects then look like the
//                   reference's — every point projective, every element in Montgomery form
//   vb_e2e_run      one timed verify_batch over all objects; verdicts out
//   vb_e2e_run_fast one timed verify_batch_fast (the batch fast accept) over the objects a mask selects
//   vb_e2e_to_bytes_path  the r03 shim's way for comparison: 8 `to_bytes()` per signature in a
//                   serial loop (a Montgomery reduction each), then dsv_verify_single_ext_multi
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

#include "dusk_schnorr.hpp"

using namespace dusk_schnorr;

namespace {
std::vector<Signature> g_sigs;
std::vector<PublicKey> g_pks;
std::vector<BlsScalar> g_msgs;
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// (u, v) canonical -> JubJubExtended (u z, v z, z, t1 = u, t2 = v z) with a per-item z
JubJubExtended projective(const uint8_t uv[64], uint64_t salt) {
  const BlsScalar u = *BlsScalar::from_bytes(uv), v = *BlsScalar::from_bytes(uv + 32);
  const uint64_t zr[4] = {salt * 0x9e3779b97f4a7c15ULL | 1, salt ^ 0xd1b54a32d192ed03ULL, salt * 3 + 7, salt >> 3};
  const BlsScalar z = BlsScalar::from_raw(zr);
  return JubJubExtended::from_raw_unchecked(u * z, v * z, z, u, v * z);
}
}  // namespace

extern "C" {

// returns 0, or -1 when an input is not canonical (then nothing is kept)
int vb_e2e_prepare(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv, const uint8_t* m, size_t n,
                   int threads) {
  g_sigs.assign(n, Signature{});
  g_pks.assign(n, PublicKey{});
  g_msgs.assign(n, BlsScalar{});
  std::vector<int> bad((size_t)threads, 0);
  auto work = [&](int t) {
    for (size_t i = n * t / threads; i < n * (size_t)(t + 1) / threads; i++) {
      auto us = JubJubScalar::from_bytes(u + 32 * i);
      auto ms = BlsScalar::from_bytes(m + 32 * i);
      bool ok = us && ms;
      for (int k = 0; k < 4 && ok; k++)
        ok = BlsScalar::from_bytes((k < 2 ? R_uv : PK_uv) + 64 * i + 32 * (k & 1)).has_value();
      if (!ok) {  // the harness's tamper classes include encodings the types cannot hold: such an item
        bad[t]++;  // becomes u = 1, R = pk = identity — 1*G + c*O = G != O, false under every scheme
        g_sigs[i].u_ = JubJubScalar::one();  // (a DEFAULT object, u = 0, would verify: 0*G + c*O == O)
        continue;
      }
      g_sigs[i].u_ = *us;
      g_msgs[i] = *ms;
      g_sigs[i].R_ = projective(R_uv + 64 * i, 2 * i + 1);
      g_pks[i].pk = projective(PK_uv + 64 * i, 2 * i + 2);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < threads; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  int total = 0;
  for (int b : bad) total += b;
  return total;
}

// one verify_batch over the prepared objects; ms = wall time of the call incl. the Vec<bool>
// result; ok[i] = verdict.  Returns 0 or -1 (engine error: message on stderr).
int vb_e2e_run(uint8_t* ok, double* ms) {
  try {
    const double t0 = now_ms();
    const std::vector<bool> out = verify_batch(g_sigs, g_pks, g_msgs);
    *ms = now_ms() - t0;
    for (size_t i = 0; i < out.size(); i++) ok[i] = out[i] ? 1 : 0;
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vb_e2e_run: %s\n", e.what());
    return -1;
  }
}

// verify_batch_fast over the objects whose mask byte is non-zero (all of them if mask is null): the bench
// keeps the VALID items of the prepared batch, so that the aggregate decides.  The selection (a copy of
// the chosen objects) is outside the timed region; `accepted` = the aggregate decided.
int vb_e2e_run_fast(const uint8_t* mask, uint8_t* ok, size_t* count, int* accepted, double* ms) {
  try {
    std::vector<Signature> sigs;
    std::vector<PublicKey> pks;
    std::vector<BlsScalar> msgs;
    for (size_t i = 0; i < g_sigs.size(); i++)
      if (!mask || mask[i]) {
        sigs.push_back(g_sigs[i]);
        pks.push_back(g_pks[i]);
        msgs.push_back(g_msgs[i]);
      }
    bool acc = false;
    const double t0 = now_ms();
    const std::vector<bool> out = verify_batch_fast(sigs, pks, msgs, &acc);
    *ms = now_ms() - t0;
    for (size_t i = 0; i < out.size(); i++) ok[i] = out[i] ? 1 : 0;
    *count = out.size();
    *accepted = acc ? 1 : 0;
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vb_e2e_run_fast: %s\n", e.what());
    return -1;
  }
}

// `calls` verify_batch_fast calls over the selected objects from `in_flight` threads (the blocking entry
// point may be called from several threads: the engine keeps two fast-accept calls in flight per device,
// one filling its arena while the other's aggregate runs).  ms = wall time / calls.
int vb_e2e_run_fast_streamed(const uint8_t* mask, int calls, int in_flight, size_t* count, int* all_accepted,
                             double* ms) {
  try {
    std::vector<Signature> sigs;
    std::vector<PublicKey> pks;
    std::vector<BlsScalar> msgs;
    for (size_t i = 0; i < g_sigs.size(); i++)
      if (!mask || mask[i]) {
        sigs.push_back(g_sigs[i]);
        pks.push_back(g_pks[i]);
        msgs.push_back(g_msgs[i]);
      }
    std::atomic<int> next{0}, good{0};
    std::atomic<bool> failed{false};
    auto worker = [&] {
      try {
        while (next.fetch_add(1) < calls) {
          bool acc = false;
          const std::vector<uint8_t> out = verify_batch_fast_bytes(sigs.data(), pks.data(), msgs.data(), sigs.size(), &acc);
          bool all = acc;
          for (uint8_t b : out) all = all && b == 1;
          if (all) good.fetch_add(1);
        }
      } catch (const std::exception& e) {
        std::fprintf(stderr, "vb_e2e_run_fast_streamed: %s\n", e.what());
        failed.store(true);
      }
    };
    const double t0 = now_ms();
    std::vector<std::thread> th;
    for (int k = 0; k < in_flight; k++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    *ms = (now_ms() - t0) / calls;
    *count = sigs.size();
    *all_accepted = good.load() == calls ? 1 : 0;
    return failed.load() ? -1 : 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vb_e2e_run_fast_streamed: %s\n", e.what());
    return -1;
  }
}

// the byte-oriented binding of r03 on the first `count` objects: u.to_bytes(), three to_bytes() per
// point, m.to_bytes() in ONE serial loop, then the projective entry point.  convert_ms = the loop
// alone, total_ms = loop + engine call.
int vb_e2e_to_bytes_path(size_t count, uint8_t* ok, double* convert_ms, double* total_ms) {
  if (count > g_sigs.size()) count = g_sigs.size();
  std::vector<uint8_t> u(32 * count), r(96 * count), pk(96 * count), m(32 * count);
  const double t0 = now_ms();
  for (size_t i = 0; i < count; i++) {
    std::memcpy(&u[32 * i], g_sigs[i].u_.to_bytes().data(), 32);
    const JubJubExtended* pts[2] = {&g_sigs[i].R_, &g_pks[i].pk};
    uint8_t* dst[2] = {&r[96 * i], &pk[96 * i]};
    for (int k = 0; k < 2; k++) {
      std::memcpy(dst[k], pts[k]->get_u().to_bytes().data(), 32);
      std::memcpy(dst[k] + 32, pts[k]->get_v().to_bytes().data(), 32);
      std::memcpy(dst[k] + 64, pts[k]->get_z().to_bytes().data(), 32);
    }
    std::memcpy(&m[32 * i], g_msgs[i].to_bytes().data(), 32);
  }
  *convert_ms = now_ms() - t0;
  const int rc = dsv_verify_single_ext_multi(u.data(), r.data(), pk.data(), m.data(), count, ok);
  *total_ms = now_ms() - t0;
  return rc;
}

// ---- the double scheme: verify_batch_double over SignatureDouble (352 B) / PublicKeyDouble (320 B) ----
static std::vector<SignatureDouble> g_dsigs;
static std::vector<PublicKeyDouble> g_dpks;
static std::vector<BlsScalar> g_dmsgs;

int vb_e2e_prepare_double(const uint8_t* u, const uint8_t* R_uv, const uint8_t* Rp_uv, const uint8_t* PK_uv,
                          const uint8_t* PKp_uv, const uint8_t* m, size_t n, int threads) {
  g_dsigs.assign(n, SignatureDouble{});
  g_dpks.assign(n, PublicKeyDouble{});
  g_dmsgs.assign(n, BlsScalar{});
  std::vector<int> bad((size_t)threads, 0);
  auto work = [&](int t) {
    for (size_t i = n * t / threads; i < n * (size_t)(t + 1) / threads; i++) {
      auto us = JubJubScalar::from_bytes(u + 32 * i);
      auto ms = BlsScalar::from_bytes(m + 32 * i);
      const uint8_t* pts[4] = {R_uv + 64 * i, Rp_uv + 64 * i, PK_uv + 64 * i, PKp_uv + 64 * i};
      bool ok = us && ms;
      for (int k = 0; k < 8 && ok; k++) ok = BlsScalar::from_bytes(pts[k >> 1] + 32 * (k & 1)).has_value();
      if (!ok) {
        bad[t]++;
        g_dsigs[i].u_ = JubJubScalar::one();  // guaranteed invalid, see vb_e2e_prepare
        continue;
      }
      g_dsigs[i].u_ = *us;
      g_dmsgs[i] = *ms;
      g_dsigs[i].R_ = projective(pts[0], 4 * i + 1);
      g_dsigs[i].R_prime_ = projective(pts[1], 4 * i + 2);
      g_dpks[i].pk_ = projective(pts[2], 4 * i + 3);
      g_dpks[i].pk_prime_ = projective(pts[3], 4 * i + 4);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < threads; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  int total = 0;
  for (int b : bad) total += b;
  return total;
}
int vb_e2e_run_double(uint8_t* ok, double* ms) {
  try {
    const double t0 = now_ms();
    const std::vector<bool> out = verify_batch_double(g_dsigs, g_dpks, g_dmsgs);
    *ms = now_ms() - t0;
    for (size_t i = 0; i < out.size(); i++) ok[i] = out[i] ? 1 : 0;
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vb_e2e_run_double: %s\n", e.what());
    return -1;
  }
}

// verify_batch_double_fast / verify_batch_var_gen_fast over the objects a mask selects (scheme 1 / 2): as
// vb_e2e_run_fast
extern "C++" {
template <class S, class P, class F>
static int run_fast_masked(const std::vector<S>& all_s, const std::vector<P>& all_p, const std::vector<BlsScalar>& all_m,
                           const uint8_t* mask, uint8_t* ok, size_t* count, int* accepted, double* ms, F verify) {
  try {
    std::vector<S> sigs;
    std::vector<P> pks;
    std::vector<BlsScalar> msgs;
    for (size_t i = 0; i < all_s.size(); i++)
      if (!mask || mask[i]) {
        sigs.push_back(all_s[i]);
        pks.push_back(all_p[i]);
        msgs.push_back(all_m[i]);
      }
    bool acc = false;
    const double t0 = now_ms();
    const std::vector<bool> out = verify(sigs, pks, msgs, &acc);
    *ms = now_ms() - t0;
    for (size_t i = 0; i < out.size(); i++) ok[i] = out[i] ? 1 : 0;
    *count = out.size();
    *accepted = acc ? 1 : 0;
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vb_e2e_run_fast_scheme: %s\n", e.what());
    return -1;
  }
}
}  // extern "C++"

// ---- the var-generator scheme: verify_batch_var_gen over SignatureVarGen (192 B) / PublicKeyVarGen (320 B) ----
static std::vector<SignatureVarGen> g_vsigs;
static std::vector<PublicKeyVarGen> g_vpks;
static std::vector<BlsScalar> g_vmsgs;

int vb_e2e_prepare_vargen(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv, const uint8_t* Gen_uv,
                          const uint8_t* m, size_t n, int threads) {
  g_vsigs.assign(n, SignatureVarGen{});
  g_vpks.assign(n, PublicKeyVarGen{});
  g_vmsgs.assign(n, BlsScalar{});
  std::vector<int> bad((size_t)threads, 0);
  auto work = [&](int t) {
    for (size_t i = n * t / threads; i < n * (size_t)(t + 1) / threads; i++) {
      auto us = JubJubScalar::from_bytes(u + 32 * i);
      auto ms = BlsScalar::from_bytes(m + 32 * i);
      const uint8_t* pts[3] = {R_uv + 64 * i, PK_uv + 64 * i, Gen_uv + 64 * i};
      bool ok = us && ms;
      for (int k = 0; k < 6 && ok; k++) ok = BlsScalar::from_bytes(pts[k >> 1] + 32 * (k & 1)).has_value();
      if (!ok) {
        bad[t]++;
        // guaranteed invalid: generator = pk = identity, so u*Gen + c*PK = O whatever u is; R = (0, -1),
        // the point of order two, is not O
        g_vsigs[i].R_ = JubJubExtended::from(JubJubAffine{BlsScalar(), -BlsScalar::one()});
        continue;
      }
      g_vsigs[i].u_ = *us;
      g_vmsgs[i] = *ms;
      g_vsigs[i].R_ = projective(pts[0], 3 * i + 1);
      g_vpks[i].pk_ = projective(pts[1], 3 * i + 2);
      g_vpks[i].generator_ = projective(pts[2], 3 * i + 3);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < threads; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  int total = 0;
  for (int b : bad) total += b;
  return total;
}
int vb_e2e_run_vargen(uint8_t* ok, double* ms) {
  try {
    const double t0 = now_ms();
    const std::vector<bool> out = verify_batch_var_gen(g_vsigs, g_vpks, g_vmsgs);
    *ms = now_ms() - t0;
    for (size_t i = 0; i < out.size(); i++) ok[i] = out[i] ? 1 : 0;
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vb_e2e_run_vargen: %s\n", e.what());
    return -1;
  }
}

// ---- streamed: `calls` batches over the same objects, `in_flight` of them in flight at any time
// (verify_batch*_submit / BatchJob::wait; 1 = back-to-back blocking calls).  ms = wall time of the whole
// sequence incl. every Vec<bool>; ok = the LAST batch's verdicts; every batch's result is compared
// with the first one's (returns 1 on a mismatch, -1 on an engine error).  scheme: 0 single, 1 double,
// 2 var-generator.
int vb_e2e_run_fast_scheme(int scheme, const uint8_t* mask, uint8_t* ok, size_t* count, int* accepted, double* ms) {
  if (scheme == 1)
    return run_fast_masked(g_dsigs, g_dpks, g_dmsgs, mask, ok, count, accepted, ms,
                           [](auto& s, auto& p, auto& m, bool* a) { return verify_batch_double_fast(s, p, m, a); });
  if (scheme == 2)
    return run_fast_masked(g_vsigs, g_vpks, g_vmsgs, mask, ok, count, accepted, ms,
                           [](auto& s, auto& p, auto& m, bool* a) { return verify_batch_var_gen_fast(s, p, m, a); });
  return -1;
}

int vb_e2e_run_streamed(int scheme, int calls, int in_flight, uint8_t* ok, double* ms) {
  try {
    if (calls < 1 || in_flight < 1) return -1;
    auto submit = [&]() -> BatchJob {
      if (scheme == 0) return verify_batch_submit(g_sigs, g_pks, g_msgs);
      if (scheme == 1) return verify_batch_double_submit(g_dsigs, g_dpks, g_dmsgs);
      return verify_batch_var_gen_submit(g_vsigs, g_vpks, g_vmsgs);
    };
    std::vector<BatchJob> jobs;
    std::vector<bool> first, out;
    int mismatches = 0;
    const double t0 = now_ms();
    int submitted = 0, waited = 0;
    std::vector<BatchJob> ring((size_t)in_flight);
    for (; submitted < calls && submitted < in_flight; submitted++) ring[(size_t)submitted] = submit();
    for (; waited < calls; waited++) {
      out = ring[(size_t)(waited % in_flight)].wait();
      if (submitted < calls) ring[(size_t)(waited % in_flight)] = submit(), submitted++;
      if (waited == 0) first = out;
      else if (out != first) mismatches++;
    }
    *ms = now_ms() - t0;
    for (size_t i = 0; i < out.size(); i++) ok[i] = out[i] ? 1 : 0;
    return mismatches ? 1 : 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "vb_e2e_run_streamed: %s\n", e.what());
    return -1;
  }
}

void vb_e2e_release(void) {
  std::vector<SignatureVarGen>().swap(g_vsigs);
  std::vector<PublicKeyVarGen>().swap(g_vpks);
  std::vector<BlsScalar>().swap(g_vmsgs);
  std::vector<SignatureDouble>().swap(g_dsigs);
  std::vector<PublicKeyDouble>().swap(g_dpks);
  std::vector<BlsScalar>().swap(g_dmsgs);
  std::vector<Signature>().swap(g_sigs);
  std::vector<PublicKey>().swap(g_pks);
  std::vector<BlsScalar>().swap(g_msgs);
}

}  // extern "C"
