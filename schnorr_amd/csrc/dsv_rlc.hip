// dsv_rlc.hip — control of the batch fast accept (kernels: k_rlc.hip, geometry: rlc.h)
// and every *_rlc entry point of include/dsv.h: device-resident arrays, typed objects and serialized
// records in host memory (filled into a per-device arena by the host pipeline), serialized records in HBM.
#include "dsv_constants.h"
#include "dsv_pipeline.h"

using namespace dsvh;

extern "C" {

// ---- random-linear-combination fast accept in front of the per-signature path ----
// Per group of at most kRlcMaxGroup items: hash, then one aggregate test (k_rlc.hip: every key and
// nonce point in the prime-order subgroup AND the z-weighted sum of the equations is the identity,
// z_i secret, fresh per call).  Accepted: every well-formed item's verdict is `true`, as the
// reference's (error <= 2^-112).  Rejected — one wrong signature, one point with a small-order
// component or off the curve — the group goes through dsv_verify_single_dev's kernels and gets
// THEIR verdicts.  The call blocks on `stream` once per group (the decision is taken on the host).
extern "C++" {
namespace {
struct RlcCarve {
  Workspace w;  // the per-signature path's own workspace comes first: the fallback uses it as it is
  RlcBuffers b;
  uint8_t* sample_ok;  // kRlcSample verdicts of the pre-check
  void* sample_ws;     // ... and its per-signature workspace
  size_t bytes;
};
// Before an aggregate is paid for, the per-signature kernel (eight lanes per signature: 0.26 ms)
// verifies the group's first kRlcSample items from the challenges just computed: a batch that is
// tampered with throughout — the graded workload: every 16th item — then skips the aggregate and
// pays the per-signature path alone.  Only while the device's recent groups give reason to
// (Context::rlc_suspicion); automatic window bits only (explicit ones are for tests, which want the
// aggregate itself to say no).  The var-generator scheme has no eight-lane kernel: its sample takes ~1 ms.
constexpr size_t kRlcSample = 1024;
RlcCarve carve_rlc(void* ws, size_t n, const RlcPlan& p) {
  RlcCarve r;
  r.w = carve(ws, n);
  Stager st(static_cast<uint8_t*>(ws) + align_up(dsv_workspace_bytes(n), 256));
  auto words = [&](size_t count) { return reinterpret_cast<u32*>(st.take(count * 4)); };
  r.b.pts = words((size_t)(p.lpts + p.spts) * n * 32);
  r.b.fsc = words((size_t)(p.fixed ? p.fixed : 1) * n * 8);
  r.b.fpart = words((size_t)kRlcFsumBlocks * 8);
  r.b.fsum = words(16);
  for (int k = 0; k < 2; k++) r.b.keys[k] = words(p.entries), r.b.vals[k] = words(p.entries);
  r.b.start = words(p.buckets + 1);
  for (int k = 0; k < 2; k++) r.b.cnt[k] = words(p.buckets), r.b.order[k] = words(p.buckets);
  r.b.buckets = words(p.buckets * 36);
  r.b.buckets2 = words(p.buckets * 36);
  for (int k = 0; k < 2; k++) r.b.tmp[k] = words(rlc_tmp_points(p, k) * 36);
  r.b.flags = words(4);
  r.b.sort_temp_bytes = rlc_sort_temp_bytes(p);
  r.b.sort_temp = st.take(r.b.sort_temp_bytes);
  r.sample_ok = st.take(kRlcSample);
  r.sample_ws = st.take(dsv_workspace_bytes(kRlcSample));
  r.bytes = align_up(dsv_workspace_bytes(n), 256) + st.off;
  return r;
}
// groups of equal size (a batch just above 2^22 items is two halves, not one full group and a tail too
// small for an aggregate)
size_t rlc_group_items(size_t n) {
  if (n <= kRlcMaxGroup) return n;
  const size_t groups = (n + kRlcMaxGroup - 1) / kRlcMaxGroup;
  return (n + groups - 1) / groups;
}
int rlc_random_key(ChaChaKey& key) {
  uint8_t* p = reinterpret_cast<uint8_t*>(key.w);
  size_t have = 0;
  while (have < sizeof key.w) {
    const ssize_t got = getrandom(p + have, sizeof key.w - have, 0);
    if (got < 0) {
      if (errno == EINTR) continue;
      return fail(DSV_ERR_HIP, "getrandom: %s (the batch weights must be unpredictable)", strerror(errno));
    }
    have += (size_t)got;
  }
  return DSV_OK;
}
}  // namespace
namespace dsvh {
// scheme 0 single (R, PK), 1 double (R, R', PK, PK'), 2 var-generator (R, PK, Gen): unused pointers null
int verify_rlc_on(Context& ctx, int scheme, const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                  const void* PKp_uv, const void* Gen_uv, const void* m, size_t n, void* ok, void* workspace,
                  hipStream_t s, int window_bits, int* accepted, bool have_challenges,
                  const uint8_t* valid_in, const RlcStaged* staged) {
  const uint8_t *pu = (const uint8_t*)u, *pR = (const uint8_t*)R_uv, *pRp = (const uint8_t*)Rp_uv,
                *pPK = (const uint8_t*)PK_uv, *pPKp = (const uint8_t*)PKp_uv, *pG = (const uint8_t*)Gen_uv,
                *pm = (const uint8_t*)m;
  uint8_t* pok = (uint8_t*)ok;
  const size_t group = rlc_group_items(n);
  bool all = true;
  for (size_t off = 0; off < n; off += group) {
    const size_t cnt = n - off < group ? n - off : group;
    const RlcPlan plan = rlc_plan(scheme, cnt, window_bits ? window_bits : rlc_default_bits(cnt));
    const RlcCarve cv = carve_rlc(workspace, cnt, plan);
    if (!window_bits && cnt < kRlcMinAuto && !have_challenges) {
      // too small for an aggregate to pay: the per-signature entry point as it is
      all = false;
      int r;
      const uint8_t* vin = valid_in ? valid_in + off : nullptr;
      if (scheme == 0)
        r = verify_single_on(ctx, pu + 32 * off, pR + 64 * off, pPK + 64 * off, pm + 32 * off, cnt, pok + off, workspace, s, vin);
      else if (scheme == 1)
        r = verify_double_on(ctx, pu + 32 * off, pR + 64 * off, pRp + 64 * off, pPK + 64 * off, pPKp + 64 * off,
                             pm + 32 * off, cnt, pok + off, workspace, s, vin);
      else
        r = verify_vargen_on(ctx, pu + 32 * off, pR + 64 * off, pPK + 64 * off, pG + 64 * off, pm + 32 * off, cnt,
                             pok + off, workspace, s, vin);
      if (r) return r;
      continue;
    }
    ChaChaKey key;
    if (staged) key = staged->key;  // (one group: the bucket pass of its first items is on the stream already)
    else if (int r = rlc_random_key(key)) return r;
    static const bool trace = getenv("DSV_RLC_TRACE") != nullptr;  // why a group was (not) accepted
    static const bool sample_on = !(getenv("DSV_RLC_SAMPLE") && atoi(getenv("DSV_RLC_SAMPLE")) == 0);
    const bool do_sample = !window_bits && sample_on && (ctx.quad || scheme == 2) && ctx.rlc_suspicion.load() > 0;
    const size_t sn = cnt < kRlcSample ? cnt : kRlcSample;
    // (have_challenges: one group whose c / valid are in the workspace already — the host form hashes
    //  chunk by chunk while the transfers run)
    if (!have_challenges)
      launch_challenge(scheme == 1, pR + 64 * off, scheme == 1 ? pRp + 64 * off : (const uint8_t*)nullptr,
                       pm + 32 * off, cnt, cv.w.c, cv.w.valid, s, valid_in ? valid_in + off : nullptr);
    bool sample_bad = false;
    if (do_sample) {
      // (beside the hash on a stream of its own it costs MORE — 0.4 ms: a small kernel next to one that fills
      //  the chip, §3 "Host pipeline" — than in line behind it: 0.26 ms)
      u32* tables = carve(cv.sample_ws, sn).tables;
      if (scheme == 0)
        launch_verify_fixed(ctx, false, pu + 32 * off, cv.w.c, pPK + 64 * off, pR + 64 * off, 0, cv.w.valid, sn,
                            cv.sample_ok, tables, s);
      else if (scheme == 1)
        launch_verify_fixed_double(ctx, pu + 32 * off, cv.w.c, pPK + 64 * off, pR + 64 * off, pPKp + 64 * off,
                                   pRp + 64 * off, cv.w.valid, sn, cv.sample_ok, tables, s);
      else  // (one lane per signature: ~1 ms for the sample — worth it only because it is rarely taken)
        launch_verify_var(pu + 32 * off, cv.w.c, pPK + 64 * off, pG + 64 * off, pR + 64 * off, cv.w.valid, sn,
                          cv.sample_ok, tables, s);
      // a WRONG item counts, a malformed one does not (it stays out of the aggregate: verdict 0 either way)
      // (`valid` covers what the hash reads — R, R', m; u and the keys are range-checked by the verify kernel)
      // into pinned memory the device owns for this purpose (one sampler at a time: the phase is short)
      std::lock_guard<std::mutex> sampler(ctx.rlc_sample_mu);
      if (!ctx.rlc_sample_host) HIP_TRY(hipHostMalloc((void**)&ctx.rlc_sample_host, (1 + 1 + 32 + 2 * 64) * kRlcSample));
      uint8_t *verdicts = ctx.rlc_sample_host, *wellformed = verdicts + kRlcSample, *us = wellformed + kRlcSample,
              *keys[2] = {us + 32 * kRlcSample, us + (32 + 64) * kRlcSample};
      HIP_TRY(hipMemcpyAsync(verdicts, cv.sample_ok, sn, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(wellformed, cv.w.valid, sn, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(us, pu + 32 * off, 32 * sn, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(keys[0], pPK + 64 * off, 64 * sn, hipMemcpyDeviceToHost, s));
      if (scheme != 0)
        HIP_TRY(hipMemcpyAsync(keys[1], (scheme == 1 ? pPKp : pG) + 64 * off, 64 * sn, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      static const uint32_t r_words[8] = DSV_R32, q_words[8] = DSV_Q32;
      auto below = [](const uint8_t* le32, const uint32_t (&mod)[8]) {  // most significant word first
        uint32_t w[8];
        memcpy(w, le32, 32);
        for (int j = 7; j >= 0; j--)
          if (w[j] != mod[j]) return w[j] < mod[j];
        return false;
      };
      for (size_t k = 0; k < sn; k++) {
        if (verdicts[k] == 1 || !wellformed[k]) continue;
        bool canonical = below(us + 32 * k, r_words);
        for (int h = 0; h < (scheme == 0 ? 1 : 2); h++)
          canonical = canonical && below(keys[h] + 64 * k, q_words) && below(keys[h] + 64 * k + 32, q_words);
        sample_bad |= canonical;  // well-formed and still verdict 0: a wrong signature
      }
    }
    if (trace && sample_bad)
      std::fprintf(stderr, "[dsv rlc] scheme %d items %zu..%zu: a wrong item among the first %zu, no aggregate\n",
                   scheme, off, off + cnt, sn);
    RlcInputs in = {};
    in.u = pu + 32 * off, in.c = cv.w.c, in.valid = cv.w.valid;
    in.pk[0] = pPK + 64 * off, in.r[0] = pR + 64 * off;
    if (scheme == 1) in.pk[1] = pPKp + 64 * off, in.r[1] = pRp + 64 * off;
    if (scheme == 2) in.gen = pG + 64 * off;
    u32 flags[4] = {~0u, 0, 0, 0};
    if (!sample_bad) {
      if (staged) {
        HIP_TRY(launch_rlc_buckets(scheme, rlc_range(plan, staged->boundary, cnt - staged->boundary), cv.b, in, key,
                                   pok + off, true, s));
        HIP_TRY(launch_rlc_finish(plan, cv.b, ctx.table[0], ctx.table[1], true, s));
      } else {
        HIP_TRY(launch_rlc(scheme, plan, cv.b, in, key, ctx.table[0], ctx.table[1], pok + off, s));
      }
      HIP_TRY(hipMemcpyAsync(flags, cv.b.flags, sizeof flags, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    if (trace && !sample_bad)
      std::fprintf(stderr, "[dsv rlc] scheme %d items %zu..%zu c=%d: %s%s%s%s\n", scheme, off, off + cnt, plan.c,
                   flags[1] == 1 ? "" : "chain incomplete ", flags[0] & kRlcOffCurve ? "off-curve " : "",
                   flags[0] & kRlcTorsion ? "subgroup-test " : "", flags[0] & kRlcSum ? "sum " : (flags[0] ? "" : "accepted"));
    if (flags[0] == 0 && flags[1] == 1) {  // ok[] = "well-formed" is the verdict vector
      int susp = ctx.rlc_suspicion.load();
      while (susp > 0 && !ctx.rlc_suspicion.compare_exchange_weak(susp, susp - 1)) {
      }
      continue;
    }
    ctx.rlc_suspicion.store(8);
    all = false;
    // the per-signature kernels, from the challenges already in the workspace (run_split carves it the same way)
    Context* cp = &ctx;
    const int r = run_split(ctx, cnt, workspace, s, [=](size_t o, size_t part, const Workspace& w, hipStream_t ps) {
      const size_t at = off + o;
      if (scheme == 0)
        launch_verify_fixed(*cp, false, pu + 32 * at, w.c, pPK + 64 * at, pR + 64 * at, 0, w.valid, part, pok + at,
                            w.tables, ps);
      else if (scheme == 1)
        launch_verify_fixed_double(*cp, pu + 32 * at, w.c, pPK + 64 * at, pR + 64 * at, pPKp + 64 * at,
                                   pRp + 64 * at, w.valid, part, pok + at, w.tables, ps);
      else
        launch_verify_var(pu + 32 * at, (const uint8_t*)w.c, pPK + 64 * at, pG + 64 * at, pR + 64 * at,
                          (const uint8_t*)w.valid, part, pok + at, w.tables, ps);
    });
    if (r) return r;
  }
  if (accepted) *accepted = all ? 1 : 0;
  return DSV_OK;
}
}  // namespace dsvh
namespace {
}  // namespace
}  // extern "C++"
size_t dsv_rlc_workspace_bytes(size_t n, int window_bits) {
  const size_t g = rlc_group_items(n);
  if (g == 0) return 256;
  const int c = window_bits ? window_bits : rlc_default_bits(g);
  if (!rlc_bits_ok(c)) return 0;
  // the double scheme's needs: four points per item, two fixed-base terms (the others fit inside)
  return carve_rlc(reinterpret_cast<void*>((uintptr_t)4096), g, rlc_plan(1, g, c)).bytes + 256;
}
// the geometry of one group's aggregate, for tests and sizing (no GPU needed): out[0..15] =
// c, half, wpk, wr, windows, nseg, nseg2, key_bits, kmul, lpts, spts, fixed, entries, buckets,
// points of tmp[0], points of tmp[1]
int dsv_rlc_plan_info(int scheme, size_t n, int window_bits, uint64_t* out) {
  if (!out || scheme < 0 || scheme > 2 || n == 0 || n > kRlcMaxGroup)
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  const int c = window_bits ? window_bits : rlc_default_bits(n);
  if (!rlc_bits_ok(c)) return fail(DSV_ERR_INVALID_ARGUMENT, "window_bits must be 0 (automatic) or one of 4, 6, 8, 12, 14, 16");
  const RlcPlan p = rlc_plan(scheme, n, c);
  const uint64_t v[16] = {(uint64_t)p.c, (uint64_t)p.half, (uint64_t)p.wpk, (uint64_t)p.wr, (uint64_t)p.windows,
                          (uint64_t)p.nseg, (uint64_t)p.nseg2, (uint64_t)p.key_bits, p.kmul, (uint64_t)p.lpts,
                          (uint64_t)p.spts, (uint64_t)p.fixed, p.entries, p.buckets, rlc_tmp_points(p, 0),
                          rlc_tmp_points(p, 1)};
  for (int k = 0; k < 16; k++) out[k] = v[k];
  return DSV_OK;
}
#define DSV_RLC_PROLOGUE(nullcheck)                                                              \
  if (accepted) *accepted = 0;                                                                   \
  if (n && (nullcheck)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");                   \
  if (window_bits && !rlc_bits_ok(window_bits))                                                  \
    return fail(DSV_ERR_INVALID_ARGUMENT, "window_bits must be 0 (automatic) or one of 4, 6, 8, 12, 14, 16"); \
  DSV_DEV_PROLOGUE(n, ok)
int dsv_verify_single_rlc_dev(const void* u, const void* R_uv, const void* PK_uv, const void* m, size_t n,
                              void* ok, void* workspace, void* stream, int window_bits, int* accepted) {
  DSV_RLC_PROLOGUE(!u || !R_uv || !PK_uv || !m || !ok || !workspace);
  return verify_rlc_on(ctx, 0, u, R_uv, nullptr, PK_uv, nullptr, nullptr, m, n, ok, workspace, (hipStream_t)stream,
                       window_bits, accepted);
}
int dsv_verify_double_rlc_dev(const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                              const void* PKp_uv, const void* m, size_t n, void* ok, void* workspace, void* stream,
                              int window_bits, int* accepted) {
  DSV_RLC_PROLOGUE(!u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok || !workspace);
  return verify_rlc_on(ctx, 1, u, R_uv, Rp_uv, PK_uv, PKp_uv, nullptr, m, n, ok, workspace, (hipStream_t)stream,
                       window_bits, accepted);
}
int dsv_verify_vargen_rlc_dev(const void* u, const void* R_uv, const void* PK_uv, const void* Gen_uv, const void* m,
                              size_t n, void* ok, void* workspace, void* stream, int window_bits, int* accepted) {
  DSV_RLC_PROLOGUE(!u || !R_uv || !PK_uv || !Gen_uv || !m || !ok || !workspace);
  return verify_rlc_on(ctx, 2, u, R_uv, nullptr, PK_uv, nullptr, Gen_uv, m, n, ok, workspace, (hipStream_t)stream,
                       window_bits, accepted);
}

// ---- batch fast accept over typed objects in host memory (SURVEY §8(f)-4 at the named entry point) ----
// The aggregate needs its whole group resident, so this form splits the work differently from
// verify_mont_host: the pipeline (gather, transfer, normalisation — and the challenge hash, chunk by
// chunk, in the shadow of the transfers) only FILLS a per-device arena; one aggregate over the arena
// follows, and only if it fails, the per-signature kernels on what is resident already.  One group
// (n <= 2^22) on the calling thread's device; anything else takes the ordinary column path.
extern "C++" {
namespace {
struct RlcArena {
  uint8_t* u;
  uint8_t* pts[4];
  uint8_t* ok;
  uint8_t* ws;
  size_t bytes;
};
RlcArena carve_arena(uint8_t* base, int kind, size_t n) {
  Stager st(base);
  RlcArena a = {};
  a.u = st.take(n * 32);
  const int np = kind == 0 ? 2 : (kind == 1 ? 4 : 3);
  for (int k = 0; k < np; k++) a.pts[k] = st.take(n * 64);
  a.ok = st.take(n);
  a.ws = st.take(dsv_rlc_workspace_bytes(n, 0));
  a.bytes = st.off;
  return a;
}
// The bucket pass in two ranges: when the pipeline is about to enqueue the first sub-batch at or beyond the
// middle of the group, everything before it is resident (or will be, in the lanes' order): the aggregate's
// prep / sort / accumulate over THAT range goes onto the arena's stream at once and runs while the second
// half is still being gathered and transferred — the GPU has little else to do during a fill (normalisation
// and hash: ~2.7 ms of work in ~6 ms).  What remains after the fill is the second range, a merge of the two
// bucket arrays and the tail.  Speculative: if the sample check then says no, the work is dropped.
struct RlcHook {
  Context* ctx = nullptr;
  int kind = 0;
  size_t n = 0;
  RlcPlan plan;
  RlcCarve cv;
  RlcInputs in;
  ChaChaKey key;
  uint8_t* ok = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool on = false, fired = false;
  size_t boundary = 0;
  int rc = DSV_OK;
  size_t last_at = 0;
  int at_part(size_t at) {  // (called under the pipeline's enqueue lock, before sub-batch `at` is enqueued)
    // what the staging rests on: the pipeline enqueues a call's sub-batches in item order
    if (at < last_at) return fail(DSV_ERR_HIP, "host pipeline enqueued item %zu after item %zu", at, last_at);
    last_at = at;
    if (!on || fired || at < n / 2 || at == 0) return DSV_OK;
    fired = true;
    boundary = at;
    for (int k = 0; k < 2; k++) {
      HIP_TRY(hipEventRecord(ev[k], ctx->pipe_lane[k]));
      HIP_TRY(hipStreamWaitEvent(stream, ev[k], 0));
    }
    HIP_TRY(launch_rlc_begin(cv.b, stream));
    HIP_TRY(launch_rlc_buckets(kind, rlc_range(plan, 0, boundary), cv.b, in, key, ok, false, stream));
    return DSV_OK;
  }
};
template <size_t NIN>
int fill_arena(Context& ctx, int kind, const HostIn (&ins)[NIN], size_t n, uint8_t* ok, const RlcArena& a,
               const Workspace& w, RlcHook* hook) {
  Context* cp = &ctx;
  const int np = kind == 0 ? 2 : (kind == 1 ? 4 : 3);
  return run_pipelined(
      ctx, ins, ok, n, kMontItemBytes, 0,
      [=](const void* const* d, size_t cnt, Stager& x, hipStream_t st, Staged& g) {
        const size_t first = t_chunk_first;
        NormalizeArgs na = {};
        for (int k = 0; k < np; k++) {
          na.in[k] = (const uint8_t*)d[1 + k];
          na.out[k] = a.pts[k] + first * 64;
          g.p[1 + k] = na.out[k];
          g.bytes[1 + k] = 64;
        }
        uint8_t* valid = x.take(cnt);
        uint8_t* cm = x.take(cnt * 32);  // the canonical message: only the hash reads it
        u32* prefix = reinterpret_cast<u32*>(x.take(normalize_prefix_bytes(cnt, np)));
        na.u_mont = (const uint8_t*)d[0];
        na.m_mont = (const uint8_t*)d[1 + np];
        na.u_out = a.u + first * 32;
        na.m_out = cm;
        launch_normalize_uvz(na, np, cnt, valid, prefix, st, cp->norm_per_lane, cp->norm_block);
        HIP_TRY(hipGetLastError());
        g.p[0] = na.u_out;
        g.p[1 + np] = cm;
        g.bytes[0] = g.bytes[1 + np] = 32;
        g.valid = valid;
        return (int)DSV_OK;
      },
      [=](const Staged& g, size_t off, size_t cnt, void*, void*, Stager&, hipStream_t st) {
        const size_t at = t_chunk_first + off;
        if (int r = hook->at_part(at)) return r;
        launch_challenge(kind == 1, g.p[1] + 64 * off, kind == 1 ? g.p[2] + 64 * off : (const uint8_t*)nullptr,
                         g.p[1 + np] + 32 * off, cnt, w.c + 32 * at, w.valid + at, st, g.valid + off);
        HIP_TRY(hipGetLastError());
        return (int)DSV_OK;
      });
}
// one shard of n <= kRlcMaxGroup items on one device, ONE group: take an arena, let `fill(arena, workspace
// carve)` run the pipeline that leaves u, the affine points and c / valid resident, then the aggregate
template <class Fill>
int rlc_host_shard(Context& ctx, int kind, size_t n, uint8_t* ok, int* accepted, Fill fill) {
  DSV_ON_DEVICE(ctx);
  // whichever arena is free; both busy: wait for the first
  std::unique_lock<std::mutex> own(ctx.rlc_arenas[0].mu, std::try_to_lock);
  int which = 0;
  if (!own.owns_lock()) {
    own = std::unique_lock<std::mutex>(ctx.rlc_arenas[1].mu, std::try_to_lock);
    which = 1;
    if (!own.owns_lock()) {
      own = std::unique_lock<std::mutex>(ctx.rlc_arenas[0].mu);
      which = 0;
    }
  }
  Context::RlcHostArena& ar = ctx.rlc_arenas[which];
  if (!ctx.ready.load(std::memory_order_acquire))
    return fail(DSV_ERR_NOT_INITIALIZED, "device %d was shut down", ctx.device);
  const size_t need = carve_arena(reinterpret_cast<uint8_t*>((uintptr_t)4096), kind, n).bytes + 256;
  if (ar.bytes < need) {
    if (ar.dev) HIP_TRY(hipFree(ar.dev));
    ar.dev = nullptr;
    ar.bytes = 0;
    HIP_TRY(hipMalloc(&ar.dev, need + need / 8));
    ar.bytes = need + need / 8;
  }
  if (!ar.stream) HIP_TRY(hipStreamCreateWithFlags(&ar.stream, hipStreamNonBlocking));
  const RlcArena a = carve_arena(ar.dev, kind, n);
  const Workspace w = carve(a.ws, n);  // where the aggregate (and the per-signature kernels) expect c / valid
  // Points in the arena: single R PK, double R R' PK PK', var-generator R PK Gen
  const uint8_t *R = a.pts[0], *Rp = kind == 1 ? a.pts[1] : nullptr, *PK = a.pts[kind == 1 ? 2 : 1],
                *PKp = kind == 1 ? a.pts[3] : nullptr, *Gen = kind == 2 ? a.pts[2] : nullptr;
  static const bool staged_on = !(getenv("DSV_RLC_STAGED") && atoi(getenv("DSV_RLC_STAGED")) == 0);
  RlcHook hook;
  hook.on = staged_on && n >= ((size_t)1 << 18);
  if (hook.on) {
    for (auto& e : ar.ev)
      if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hook.ctx = &ctx, hook.kind = kind, hook.n = n, hook.ok = a.ok, hook.stream = ar.stream;
    hook.ev[0] = ar.ev[0], hook.ev[1] = ar.ev[1];
    hook.plan = rlc_plan(kind, n, rlc_default_bits(n));  // (what verify_rlc_on plans for this group)
    hook.cv = carve_rlc(a.ws, n, hook.plan);
    hook.in = RlcInputs{};
    hook.in.u = a.u, hook.in.c = hook.cv.w.c, hook.in.valid = hook.cv.w.valid;
    hook.in.pk[0] = PK, hook.in.r[0] = R;
    if (kind == 1) hook.in.pk[1] = PKp, hook.in.r[1] = Rp;
    if (kind == 2) hook.in.gen = Gen;
    if (int r = rlc_random_key(hook.key)) return r;
  }
  if (int rc = fill(a, w, &hook)) {
    if (hook.fired) (void)hipStreamSynchronize(ar.stream);  // nothing of this call may still run on the arena
    return rc;
  }
  // (run_pipelined returned: every chunk's kernels are done.)
  RlcStaged staged;
  staged.key = hook.key;
  staged.boundary = hook.boundary;
  if (int r = verify_rlc_on(ctx, kind, a.u, R, Rp, PK, PKp, Gen, /*m: hashed already*/ a.u, n, a.ok, a.ws,
                            ar.stream, 0, accepted, true, nullptr, hook.fired ? &staged : nullptr))
    return r;
  HIP_TRY(hipMemcpyAsync(ok, a.ok, n, hipMemcpyDeviceToHost, ar.stream));
  HIP_TRY(hipStreamSynchronize(ar.stream));
  return DSV_OK;
}
int verify_mont_cols_rlc_shard(Context& ctx, int kind, const dsv_column* cols, size_t off, size_t n, uint8_t* ok,
                               int* accepted) {
  Context* cp = &ctx;
  return rlc_host_shard(ctx, kind, n, ok + off, accepted, [=](const RlcArena& a, const Workspace& w, RlcHook* hook) {
    auto in = [&](int k, size_t width) {
      return HostIn{static_cast<const uint8_t*>(cols[k].base) + off * cols[k].stride, width, cols[k].stride};
    };
    if (kind == 0) {
      const HostIn ins[4] = {in(0, 32), in(1, 96), in(2, 96), in(3, 32)};
      return fill_arena(*cp, 0, ins, n, ok + off, a, w, hook);
    }
    if (kind == 1) {
      const HostIn ins[6] = {in(0, 32), in(1, 96), in(2, 96), in(3, 96), in(4, 96), in(5, 32)};
      return fill_arena(*cp, 1, ins, n, ok + off, a, w, hook);
    }
    const HostIn ins[5] = {in(0, 32), in(1, 96), in(2, 96), in(3, 96), in(4, 32)};
    return fill_arena(*cp, 2, ins, n, ok + off, a, w, hook);
  });
}
// Shards like the *_multi forms: one group per initialised device (each with its own aggregate; all of
// them must accept), as long as every shard is one group of a useful size; else one group on the calling
// thread's device, or — beyond 2^22 items — the ordinary column path.
int verify_mont_cols_rlc(int kind, const dsv_column* cols, size_t n, uint8_t* ok, int* accepted) {
  if (accepted) *accepted = 0;
  if (int r = check_cols(kind, cols, n, ok)) return r;
  if (n == 0) return DSV_OK;
  int nd = 0;
  for (int d = 0; d < kMaxDevices; d++) nd += g_ctx[d].ready.load(std::memory_order_acquire) ? 1 : 0;
  if (const char* e = getenv("DSV_MULTI_SHARDS")) nd = atoi(e) > nd ? atoi(e) : nd;  // (run_multi's rehearsal knob)
  if (nd > 1 && n >= (size_t)nd << 17 && (n + nd - 1) / nd <= kRlcMaxGroup) {
    std::atomic<int> rejected{0};
    const int rc = run_multi(n, [&, kind, cols, ok](Context& ctx, size_t off, size_t cnt) {
      int acc = 0;
      const int r = verify_mont_cols_rlc_shard(ctx, kind, cols, off, cnt, ok, &acc);
      if (!acc) rejected.fetch_add(1);
      return r;
    });
    if (rc == DSV_OK && accepted) *accepted = rejected.load() == 0 ? 1 : 0;
    return rc;
  }
  if (n > kRlcMaxGroup || n < kRlcMinAuto) return verify_mont_cols(kind, cols, n, ok, true);
  Context* ctxp = nullptr;
  if (int r = host_context(ctxp)) return r;
  return verify_mont_cols_rlc_shard(*ctxp, kind, cols, 0, n, ok, accepted);
}
}  // namespace
}  // extern "C++"
int dsv_verify_single_mont_cols_rlc(const dsv_column* cols, size_t n, uint8_t* ok, int* accepted) { return verify_mont_cols_rlc(0, cols, n, ok, accepted); }
int dsv_verify_double_mont_cols_rlc(const dsv_column* cols, size_t n, uint8_t* ok, int* accepted) { return verify_mont_cols_rlc(1, cols, n, ok, accepted); }
int dsv_verify_vargen_mont_cols_rlc(const dsv_column* cols, size_t n, uint8_t* ok, int* accepted) { return verify_mont_cols_rlc(2, cols, n, ok, accepted); }


// serialized records through the batch fast accept: decode (what `from_bytes` does: curve points, not
// necessarily of prime order), then the aggregate; a record that does not decode has verdict 0 and stays
// out of the sum
extern "C++" {
namespace {
size_t wire_arrays_bytes(size_t n) { return align_up(n * 32, 256) + 4 * align_up(n * 64, 256) + align_up(n, 256); }
int verify_wire_rlc_dev(int kind, const void* sig, const void* pk, const void* m, size_t n, void* ok,
                        void* workspace, void* stream, int window_bits, int* accepted) {
  if (accepted) *accepted = 0;
  if (n && (!sig || !pk || !m || !ok || !workspace)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  if (((uintptr_t)sig | (uintptr_t)pk) & 15) return fail(DSV_ERR_INVALID_ARGUMENT, "records must be 16-byte aligned");
  if (window_bits && !rlc_bits_ok(window_bits))
    return fail(DSV_ERR_INVALID_ARGUMENT, "window_bits must be 0 (automatic) or one of 4, 6, 8, 12, 14, 16");
  DSV_DEV_PROLOGUE(n, ok);
  const hipStream_t st = (hipStream_t)stream;
  Stager x(static_cast<uint8_t*>(workspace));
  WireWs w;
  w.u = x.take(n * 32);
  w.R = x.take(n * 64);
  w.Rp = x.take(n * 64);
  w.P0 = x.take(n * 64);
  w.P1 = x.take(n * 64);
  w.valid = x.take(n);
  void* rws = x.take(0);
  const uint8_t *dsig = (const uint8_t*)sig, *dpk = (const uint8_t*)pk;
  const size_t sig_bytes = kind == 1 ? 96 : 64, pk_bytes = kind == 0 ? 32 : 64;
  launch_gather32(dsig, sig_bytes, n, w.u, st);
  if (int r = decompress_on(ctx, dsig + 32, sig_bytes, n, w.R, w.valid, 0, st)) return r;
  if (kind == 1)
    if (int r = decompress_on(ctx, dsig + 64, sig_bytes, n, w.Rp, w.valid, 1, st)) return r;
  if (int r = decompress_on(ctx, dpk, pk_bytes, n, w.P0, w.valid, 1, st)) return r;
  if (kind != 0)
    if (int r = decompress_on(ctx, dpk + 32, pk_bytes, n, w.P1, w.valid, 1, st)) return r;
  return verify_rlc_on(ctx, kind, w.u, w.R, kind == 1 ? w.Rp : nullptr, w.P0, kind == 1 ? w.P1 : nullptr,
                       kind == 2 ? w.P1 : nullptr, m, n, ok, rws, st, window_bits, accepted, false, w.valid);
}
}  // namespace
}  // extern "C++"
size_t dsv_wire_rlc_workspace_bytes(size_t n, int window_bits) {
  const size_t r = dsv_rlc_workspace_bytes(n, window_bits);
  return r ? wire_arrays_bytes(n) + r + 256 : 0;
}
int dsv_verify_single_wire_rlc_dev(const void* sig64, const void* pk32, const void* m, size_t n, void* ok,
                                   void* workspace, void* stream, int window_bits, int* accepted) {
  return verify_wire_rlc_dev(0, sig64, pk32, m, n, ok, workspace, stream, window_bits, accepted);
}
int dsv_verify_double_wire_rlc_dev(const void* sig96, const void* pk64, const void* m, size_t n, void* ok,
                                   void* workspace, void* stream, int window_bits, int* accepted) {
  return verify_wire_rlc_dev(1, sig96, pk64, m, n, ok, workspace, stream, window_bits, accepted);
}
int dsv_verify_vargen_wire_rlc_dev(const void* sig64, const void* pk64, const void* m, size_t n, void* ok,
                                   void* workspace, void* stream, int window_bits, int* accepted) {
  return verify_wire_rlc_dev(2, sig64, pk64, m, n, ok, workspace, stream, window_bits, accepted);
}

// serialized records in HOST memory through the batch fast accept: the pipeline decodes chunk by chunk
// into an arena (and hashes in the shadow of the transfers), one aggregate follows — half the bus
// traffic of the typed-object form (128 B per single signature)
extern "C++" {
namespace {
int verify_wire_rlc_host(int kind, const uint8_t* sig, const uint8_t* pk, const uint8_t* m, size_t n, uint8_t* ok,
                         int* accepted) {
  if (accepted) *accepted = 0;
  if (n && (!sig || !pk || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  Context* ctxp = nullptr;
  if (int r = host_context(ctxp)) return r;
  Context& ctx = *ctxp;
  if (n > kRlcMaxGroup || n < kRlcMinAuto) return verify_wire(ctx, kind, sig, pk, m, n, ok);
  const size_t sig_bytes = kind == 1 ? 96 : 64, pk_bytes = kind == 0 ? 32 : 64;
  Context* cp = &ctx;
  return rlc_host_shard(ctx, kind, n, ok, accepted, [=](const RlcArena& a, const Workspace& w, RlcHook* hook) {
    const HostIn ins[3] = {{sig, sig_bytes}, {pk, pk_bytes}, {m, 32}};
    return run_pipelined(*cp, ins, ok, n, 0, /*per item: the decoder's verdict byte*/ 1, NoPrep{},
                         [=](const Staged& g, size_t off, size_t cnt, void*, void*, Stager& x, hipStream_t st) {
      const size_t at = t_chunk_first + off;
      if (int r = hook->at_part(at)) return r;
      const uint8_t *dsig = g.p[0] + off * sig_bytes, *dpk = g.p[1] + off * pk_bytes;
      uint8_t* valid = x.take(cnt);
      launch_gather32(dsig, sig_bytes, cnt, a.u + 32 * at, st);
      // arena order: single R PK, double R R' PK PK', var-generator R PK Gen
      int slot = 0;
      if (int r = decompress_on(*cp, dsig + 32, sig_bytes, cnt, a.pts[slot++] + 64 * at, valid, 0, st)) return r;
      if (kind == 1)
        if (int r = decompress_on(*cp, dsig + 64, sig_bytes, cnt, a.pts[slot++] + 64 * at, valid, 1, st)) return r;
      if (int r = decompress_on(*cp, dpk, pk_bytes, cnt, a.pts[slot++] + 64 * at, valid, 1, st)) return r;
      if (kind != 0)
        if (int r = decompress_on(*cp, dpk + 32, pk_bytes, cnt, a.pts[slot++] + 64 * at, valid, 1, st)) return r;
      launch_challenge(kind == 1, a.pts[0] + 64 * at, kind == 1 ? a.pts[1] + 64 * at : (const uint8_t*)nullptr,
                       g.p[2] + 32 * off, cnt, w.c + 32 * at, w.valid + at, st, valid);
      HIP_TRY(hipGetLastError());
      return (int)DSV_OK;
    });
  });
}
}  // namespace
}  // extern "C++"
int dsv_verify_single_wire_rlc(const uint8_t* sig64, const uint8_t* pk32, const uint8_t* m, size_t n, uint8_t* ok,
                               int* accepted) {
  return verify_wire_rlc_host(0, sig64, pk32, m, n, ok, accepted);
}
int dsv_verify_double_wire_rlc(const uint8_t* sig96, const uint8_t* pk64, const uint8_t* m, size_t n, uint8_t* ok,
                               int* accepted) {
  return verify_wire_rlc_host(1, sig96, pk64, m, n, ok, accepted);
}
int dsv_verify_vargen_wire_rlc(const uint8_t* sig64, const uint8_t* pk64, const uint8_t* m, size_t n, uint8_t* ok,
                               int* accepted) {
  return verify_wire_rlc_host(2, sig64, pk64, m, n, ok, accepted);
}

}  // extern "C"
