// dsv_rlc.hip — control of the batch fast accept (kernels: k_rlc.hip, geometry: rlc.h)
// and every *_rlc entry point of include/dsv.h: device-resident arrays, typed objects and serialized
// records in host memory (filled into a per-device arena by the host pipeline), serialized records in HBM.
#include "dsv_constants.h"
#include "dsv_pipeline.h"

using namespace dsvh;

extern "C" {

// ---- random-linear-combination fast accept in front of the per-signature path ----
// Per group of at most kRlcMaxGroup items: hash, then one aggregate test per SUB-GROUP (k_rlc.hip: every
// key and nonce point in the prime-order subgroup AND the z-weighted sum of the equations is the identity,
// z_i secret, fresh per call).  Accepted: every well-formed item's verdict is `true`, as the reference's
// (error <= 2^-112).  Rejected — one wrong signature, one point with a small-order component or off the
// curve — the sub-group goes through dsv_verify_single_dev's kernels and gets THEIR verdicts.
//
// ENQUEUE-ONLY (r06; r05 took the decision on the host, one stream synchronisation per group): the
// per-signature kernels are launched unconditionally behind the aggregate with a pointer to their
// sub-group's flag words as `gate` (launch.h) and return in their first instructions when the aggregate
// accepted; the call's verdict is written by a one-thread kernel into memory the device can reach.
//
// Failure localisation (SURVEY's "bisect on failure", done the way this chip allows): not sequential
// halving — every level would pay the aggregate's latency-bound tail again — but sub-groups UP FRONT while
// the device's recent calls give reason to: a group is cut into up to 16 sub-groups of >= 2^16 items that share
// the hash pass and ONE set of launches (blockIdx.y = sub-group), so the tail is paid once; only a
// failing sub-group's per-signature kernels do any work.  "Reason to" is a counter in pinned host memory
// that the verdict kernel itself maintains (8 after a call with a rejected sub-group, one less after an
// accepted call; 1 after dsv_init): the host reads it, possibly one call late, and never waits for it.
// While it is > 0 a SAMPLE is verified first as well (below).
//
// The FIRST rejected batch after a run of valid ones would still pay both paths in full (one sub-group:
// 15.4 ms against 12.3 per 2^20).  A second, slower counter (128 after a rejected aggregate, one less per
// accepted call) marks callers whose batches do fail now and then: while it is > 0 and the short one is 0, a
// group of at least four sub-groups runs GUARDED — its single aggregate as in the steady state, and behind it,
// enqueued unconditionally, a second stage of sub-group aggregates that a one-thread kernel (k_rlc_chain)
// switches off when the first one accepted (its kernels and the per-signature launches behind it then return
// at once: ~25 early exits, +0.03 ms on an accepted 2^20-item call); when the first stage rejects, the second
// localises the failure and only one sub-group reaches the per-signature kernels: 9.4 instead of 15.8 ms
// (1.31 x the per-signature path instead of 0.78 x; double 16.9 instead of 30.3 ms).  It pays from one failing
// batch in ~200 on, hence the counter's 128.  A caller whose batches never fail never enqueues it;
// DSV_RLC_GUARD=0 switches it off.
extern "C++" {
namespace {
// Before an aggregate is paid for, the per-signature kernel (eight lanes per signature: 0.26 ms)
// verifies kRlcSample consecutive items — at a position drawn from the call's secret key — from the
// challenges just computed, and a one-wave kernel sets the group's "skip" word if a WRONG item is among
// them (malformed ones do not count: they stay out of the sum): every kernel of the aggregate then
// returns at once and the per-signature kernels decide.  A batch that is tampered with throughout pays
// hash + sample + per-signature path instead of all three stages.  A heuristic, and only that: it runs
// only while the history counter is > 0, never with explicit window bits (tests want the aggregate itself
// to say no), DSV_RLC_SAMPLE=0 switches it off.  The var-generator scheme has no eight-lane kernel: its
// sample takes ~1 ms.
constexpr size_t kRlcSample = 1024;
struct RlcCarve {
  Workspace w;  // the per-signature path's own workspace comes first: the fallback uses it as it is
  RlcBuffers b;
  u32* flags_area;     // kRlcFlagBlocks flag blocks, two per group (b.flags = the one in use)
  uint8_t* sample_ok;  // kRlcSample verdicts of the pre-check
  void* sample_ws;     // ... and its per-signature workspace
  size_t bytes;
};
// nmax: items of the largest group of the call (the per-signature workspace in front is sized for it);
// cnt: items of this group (its c / valid lie where run_split's carve of the fallback expects them)
RlcCarve carve_rlc(void* ws, size_t nmax, size_t cnt, const RlcPlan& p) {
  RlcCarve r;
  r.w = carve(ws, cnt);
  Stager st(static_cast<uint8_t*>(ws) + align_up(dsv_workspace_bytes(nmax), 256));
  auto words = [&](size_t count) { return reinterpret_cast<u32*>(st.take(count * 4)); };
  const size_t G = p.groups, sub = p.sub;
  // fixed places first: the same whatever the plan
  r.flags_area = words(kRlcFlagBlocks * kRlcGroupFlagWords);
  r.b.flags = r.flags_area;
  r.sample_ok = st.take(kRlcSample);
  r.sample_ws = st.take(dsv_workspace_bytes(kRlcSample));
  RlcBuffers& b = r.b;
  b.pts_stride = (size_t)(p.lpts + p.spts) * sub * 32;
  b.pts = words(G * b.pts_stride);
  b.fsc_stride = (size_t)(p.fixed ? p.fixed : 1) * sub * 8;
  b.fsc = words(G * b.fsc_stride);
  b.fpart = words(G * 2 * kRlcFsumBlocks * 8);
  b.fsum = words(G * 16);
  b.digits_stride = align_up((size_t)p.rows * p.row_stride, 128);
  b.digits = reinterpret_cast<uint16_t*>(st.take(G * b.digits_stride * 2));
  b.counters_stride = (size_t)p.bins + 512;
  b.counters = words(G * b.counters_stride);
  b.bin_stride = (size_t)p.bins * p.bin_cap;
  b.binned = words(G * b.bin_stride);
  b.sorted = words(G * b.bin_stride);
  b.bucket_stride = p.buckets;
  b.start = words(G * p.buckets);
  b.cnt = words(G * p.buckets);
  b.order = words(G * p.buckets);
  b.buckets = words(G * p.buckets * 36);
  b.buckets2 = G == 1 ? words(p.buckets * 36) : nullptr;
  for (int k = 0; k < 2; k++) {
    b.tmp_stride[k] = rlc_tmp_points(p, k);
    b.tmp[k] = words(G * b.tmp_stride[k] * 36);
  }
  r.bytes = align_up(dsv_workspace_bytes(nmax), 256) + st.off;
  return r;
}
// groups of equal size (a batch just above 2^22 items is two halves, not one full group and a tail too
// small for an aggregate)
size_t rlc_group_items(size_t n) {
  if (n <= kRlcMaxGroup) return n;
  const size_t groups = (n + kRlcMaxGroup - 1) / kRlcMaxGroup;
  return (n + groups - 1) / groups;
}
// sub-groups a group of cnt items is cut into while the history says "batches fail": sub-groups of
// 2^16 items (DSV_RLC_SUB_LOG2; measured, 2^20 items with one wrong signature: 6.5 ms in sixteen sub-groups, 7.1 in eight, 8.1 in four, 10.8 in two; 2^18 items: 2.96 ms in four, 3.48 in two), at least two from 2^18 items on, at most kRlcMaxSub
int rlc_split_groups(size_t cnt, int window_bits) {
  static const int sub_log2 = [] {
    const char* e = getenv("DSV_RLC_SUB_LOG2");
    const int v = e ? atoi(e) : 16;
    return v < 10 ? 10 : (v > 22 ? 22 : v);
  }();
  size_t g = cnt >> sub_log2;
  if (g < 2 && (window_bits ? cnt >= 2 : cnt >= 2 * kRlcMinAuto)) g = 2;
  if (g > (size_t)kRlcMaxSub) g = kRlcMaxSub;
  return g < 1 ? 1 : (int)g;
}
// the plan of one group: window bits from the SUB-group's size; sub-groups are whole sub-batches of the
// per-signature path (kSplitItems), so that no launch of the fallback straddles two of them
RlcPlan rlc_group_plan(int scheme, size_t cnt, int window_bits, int groups) {
  if (groups <= 1) return rlc_plan(scheme, cnt, window_bits ? window_bits : rlc_default_bits(cnt));
  const size_t align = cnt >= 2 * kSplitItems ? kSplitItems : 64;
  RlcPlan probe = rlc_plan(scheme, cnt, window_bits ? window_bits : 8, groups, align);
  return rlc_plan(scheme, cnt, window_bits ? window_bits : rlc_default_bits(probe.sub), groups, align);
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
// sub-groups per group, forced (DSV_RLC_SUBGROUPS, dsv_debug_rlc_subgroups; tests and tools); 0: by the history
std::atomic<int> g_rlc_force_groups{getenv("DSV_RLC_SUBGROUPS") ? atoi(getenv("DSV_RLC_SUBGROUPS")) : 0};
// pinned host words the device writes: [0] the history counter, [1] calls completed, [2] the long history
// counter, [4 ..] a ring of verdict slots for callers whose `accepted` is pageable memory
constexpr u32 kRlcSlots = 60, kRlcSlot0 = 4;
int ensure_rlc_pinned(Context& ctx) {
  std::lock_guard<std::mutex> lk(ctx.rlc_pinned_mu);
  if (ctx.rlc_pinned) return DSV_OK;
  u32* p = nullptr;
  HIP_TRY(hipHostMalloc((void**)&p, (kRlcSlot0 + kRlcSlots) * sizeof(u32), hipHostMallocDefault));
  for (u32 k = 0; k < kRlcSlot0 + kRlcSlots; k++) p[k] = 0;
  p[0] = 1;  // the first call of a device checks a sample and runs in sub-groups
  ctx.rlc_pinned = p;
  return DSV_OK;
}
}  // namespace
namespace dsvh {
int rlc_history(Context& ctx) {
  if (ensure_rlc_pinned(ctx) != DSV_OK) return -1;
  return (int)*reinterpret_cast<volatile u32*>(ctx.rlc_pinned);
}
// where the verdict kernel writes `accepted`: the caller's own word if the device can reach it (device or
// pinned / registered host memory), else a pinned slot of the context — the call then blocks at its end
int rlc_verdict_target(Context& ctx, int* accepted, RlcVerdictTarget& t) {
  t = RlcVerdictTarget{};
  if (!accepted) return DSV_OK;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, accepted) == hipSuccess &&
      (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeHost) &&
      attr.devicePointer) {
    t.dev = static_cast<u32*>(attr.devicePointer);
    return DSV_OK;
  }
  (void)hipGetLastError();  // a failed query leaves a sticky "invalid value"
  if (int r = ensure_rlc_pinned(ctx)) return r;
  const u32 k = ctx.rlc_slot.fetch_add(1) % kRlcSlots;
  t.slot = ctx.rlc_pinned + kRlcSlot0 + k;
  t.dev = t.slot;
  t.out = accepted;
  return DSV_OK;
}
// *accepted = 0 wherever it lives (calls that enqueue nothing)
int rlc_clear_accepted(int* accepted) {
  if (!accepted) return DSV_OK;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, accepted) == hipSuccess &&
      (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged)) {
    HIP_TRY(hipMemset(accepted, 0, sizeof(int)));
    return DSV_OK;
  }
  (void)hipGetLastError();
  *accepted = 0;
  return DSV_OK;
}
int rlc_verdict_wait(const RlcVerdictTarget& t, hipStream_t s) {
  if (!t.out) return DSV_OK;
  HIP_TRY(hipStreamSynchronize(s));
  *t.out = (int)*reinterpret_cast<volatile u32*>(t.slot);
  return DSV_OK;
}
}  // namespace dsvh
namespace {
// one call's arguments, as the pieces below need them
struct RlcCall {
  Context& ctx;
  int scheme;  // 0 single (R, PK), 1 double (R, R', PK, PK'), 2 var-generator (R, PK, Gen): unused pointers null
  const uint8_t *u, *R, *Rp, *PK, *PKp, *G, *m;
  uint8_t* ok;
  void* workspace;
  hipStream_t s;
};
struct RlcTraced {  // what DSV_RLC_TRACE reports about one group
  size_t off, cnt;
  RlcPlan plan, plan2;  // plan2.groups != 0: the second stage of a guarded group
  bool sampled;
};
// the sample check of the group at `off`: kRlcSample consecutive items at a position drawn from the key through
// the per-signature kernel, then the one-wave decision into the group's flag block
void rlc_enqueue_sample(const RlcCall& c, const RlcCarve& cv, const ChaChaKey& key, size_t off, size_t cnt) {
  // (beside the hash on a stream of its own it costs MORE — 0.4 ms: a small kernel next to one that fills
  //  the chip — than in line behind it: 0.26 ms)
  const size_t sn = cnt < kRlcSample ? cnt : kRlcSample;
  const size_t at = off + (size_t)(key.w[7] % (u32)((cnt - sn) / 64 + 1)) * 64;  // (the key's last word: no weight uses it)
  const size_t rel = at - off;
  u32* tables = carve(cv.sample_ws, sn).tables;
  if (c.scheme == 0)
    launch_verify_fixed(c.ctx, false, c.u + 32 * at, cv.w.c + 32 * rel, c.PK + 64 * at, c.R + 64 * at, 0, cv.w.valid + rel, sn,
                        cv.sample_ok, tables, c.s);
  else if (c.scheme == 1)
    launch_verify_fixed_double(c.ctx, c.u + 32 * at, cv.w.c + 32 * rel, c.PK + 64 * at, c.R + 64 * at, c.PKp + 64 * at,
                               c.Rp + 64 * at, cv.w.valid + rel, sn, cv.sample_ok, tables, c.s);
  else  // (one lane per signature beyond 2^13 items, sixteen below: ~0.4 ms for the sample)
    launch_verify_var(c.u + 32 * at, cv.w.c + 32 * rel, c.PK + 64 * at, c.G + 64 * at, c.R + 64 * at, cv.w.valid + rel, sn,
                      cv.sample_ok, tables, c.s);
  // a WRONG item counts, a malformed one does not (`valid` covers what the hash reads — R, R', m; u and
  // the keys are range-checked here as the verify kernel does)
  launch_rlc_sample_decide(cv.sample_ok, cv.w.valid + rel, c.u + 32 * at, c.PK + 64 * at,
                           c.scheme == 0 ? (const uint8_t*)nullptr : (c.scheme == 1 ? c.PKp : c.G) + 64 * at, 0, sn, cv.b.flags,
                           c.s);
}
// The per-signature kernels of the group at `off`, from the challenges already in the workspace (run_split
// carves it the same way), each launch gated by its sub-group's flag words in `gflags`: no work where the
// aggregate accepted.  split: sub-batches alternating between the two internal streams (what a rejected
// sub-group wants); else one launch per sub-group on the caller's stream (what launches that are expected to
// return at once want: sixteen gated sub-batch launches cost 0.05 ms of an accepted call's 5.3).
int rlc_enqueue_fallback(const RlcCall& c, const Workspace& w0, const u32* gflags, size_t sub, bool split, size_t off,
                         size_t cnt) {
  Context* cp = &c.ctx;
  const RlcCall call = c;
  auto fallback = [=](size_t o, size_t part, const Workspace& w, hipStream_t ps) {
    // (sub-groups are whole sub-batches; an unsplit launch covers one sub-group or is cut here)
    for (size_t done = 0; done < part;) {
      const size_t at = off + o + done, g = (o + done) / sub;
      size_t take = (g + 1) * sub - (o + done);
      if (take > part - done) take = part - done;
      const u32* gate = gflags + 4 + 4 * g;
      if (call.scheme == 0)
        launch_verify_fixed(*cp, false, call.u + 32 * at, w.c + 32 * done, call.PK + 64 * at, call.R + 64 * at, 0, w.valid + done,
                            take, call.ok + at, w.tables, ps, false, gate);
      else if (call.scheme == 1)
        launch_verify_fixed_double(*cp, call.u + 32 * at, w.c + 32 * done, call.PK + 64 * at, call.R + 64 * at,
                                   call.PKp + 64 * at, call.Rp + 64 * at, w.valid + done, take, call.ok + at, w.tables, ps, false,
                                   gate);
      else
        launch_verify_var(call.u + 32 * at, (const uint8_t*)w.c + 32 * done, call.PK + 64 * at, call.G + 64 * at,
                          call.R + 64 * at, (const uint8_t*)w.valid + done, take, call.ok + at, w.tables, ps, gate);
      done += take;
    }
  };
  if (split) return run_split(c.ctx, cnt, c.workspace, c.s, fallback);
  fallback((size_t)0, cnt, w0, c.s);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// DSV_RLC_TRACE: waits for the stream and prints, per group, stage and sub-group, why it was (not) accepted
int rlc_trace(const RlcCall& c, const u32* flags_area, const RlcVerdictArgs& va, const RlcTraced* traced, u32 history,
              u32 history_long) {
  HIP_TRY(hipStreamSynchronize(c.s));
  std::vector<u32> f(kRlcFlagBlocks * kRlcGroupFlagWords);
  HIP_TRY(hipMemcpy(f.data(), flags_area, f.size() * sizeof(u32), hipMemcpyDeviceToHost));
  for (u32 k = 0; k < va.ngroups; k++) {
    const RlcTraced& t = traced[k];
    if (!va.subs[k]) {
      std::fprintf(stderr, "[dsv rlc] scheme %d items %zu..%zu: too small for an aggregate, per-signature path\n", c.scheme,
                   t.off, t.off + t.cnt);
      continue;
    }
    for (int stage = 0; stage < (t.plan2.groups ? 2 : 1); stage++) {
      const RlcPlan& pl = stage ? t.plan2 : t.plan;
      const u32* gf = f.data() + ((size_t)2 * k + stage) * kRlcGroupFlagWords;
      if (gf[0] && !stage)
        std::fprintf(stderr, "[dsv rlc] scheme %d items %zu..%zu: a wrong item in the sample, no aggregate\n", c.scheme, t.off,
                     t.off + t.cnt);
      if (gf[0] && stage)
        std::fprintf(stderr, "[dsv rlc] scheme %d items %zu..%zu: second stage not needed\n", c.scheme, t.off, t.off + t.cnt);
      for (u32 g = 0; g < pl.groups && !gf[0]; g++) {
        const u32* fl = gf + 4 + 4 * g;
        const size_t lo = t.off + (size_t)g * pl.sub, hi = lo + pl.sub < t.off + t.cnt ? lo + pl.sub : t.off + t.cnt;
        std::fprintf(stderr, "[dsv rlc] scheme %d items %zu..%zu c=%d %ssub-group %u/%u%s (history %u / %u): %s%s%s%s%s\n", c.scheme,
                     lo, hi, pl.c, t.plan2.groups ? (stage ? "second stage, " : "first stage, ") : "", g + 1, pl.groups,
                     t.sampled ? " sampled" : "", history, history_long,
                     fl[1] == 1 ? "" : "chain incomplete ", fl[0] & kRlcOffCurve ? "off-curve " : "",
                     fl[0] & kRlcTorsion ? "subgroup-test " : "", fl[0] & kRlcOverflow ? "bin-overflow " : "",
                     fl[0] & kRlcSum ? "sum " : (fl[0] == 0 && fl[1] == 1 ? "accepted" : ""));
      }
    }
  }
  return DSV_OK;
}
}  // namespace
namespace dsvh {
// Enqueues everything on `s` and returns; *accepted_dev (device-accessible, may be null) = every group was
// decided by its aggregates (and_into: ... AND what the word held: the second kind of a mixed batch).
int verify_rlc_on(Context& ctx, int scheme, const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                  const void* PKp_uv, const void* Gen_uv, const void* m, size_t n, void* ok, void* workspace,
                  hipStream_t s, int window_bits, u32* accepted_dev, bool and_into, bool have_challenges,
                  const uint8_t* valid_in, const RlcStaged* staged) {
  const RlcCall c{ctx, scheme, (const uint8_t*)u, (const uint8_t*)R_uv, (const uint8_t*)Rp_uv, (const uint8_t*)PK_uv,
                  (const uint8_t*)PKp_uv, (const uint8_t*)Gen_uv, (const uint8_t*)m, (uint8_t*)ok, workspace, s};
  if (int r = ensure_rlc_pinned(ctx)) return r;
  const size_t group = rlc_group_items(n);
  static const bool trace = getenv("DSV_RLC_TRACE") != nullptr;  // why a group was (not) accepted: makes the call synchronous
  static const bool sample_on = !(getenv("DSV_RLC_SAMPLE") && atoi(getenv("DSV_RLC_SAMPLE")) == 0);
  static const bool guard_on = !(getenv("DSV_RLC_GUARD") && atoi(getenv("DSV_RLC_GUARD")) == 0);
  const int force_groups = g_rlc_force_groups.load(std::memory_order_relaxed);
  // what the device's recent calls say (written by the verdict kernels; read without waiting for anything)
  const u32 history = *reinterpret_cast<volatile u32*>(ctx.rlc_pinned);
  const u32 history_long = *reinterpret_cast<volatile u32*>(ctx.rlc_pinned + 2);
  RlcVerdictArgs va = {};
  u32* flags_area = nullptr;
  RlcTraced traced[kRlcMaxGroupsPerCall];
  for (size_t off = 0, k = 0; off < n; off += group, k++) {
    const size_t cnt = n - off < group ? n - off : group;
    if (k >= kRlcMaxGroupsPerCall) return fail(DSV_ERR_TOO_LARGE, "more than %zu groups", kRlcMaxGroupsPerCall);
    va.ngroups = (u32)k + 1;
    if (!window_bits && cnt < rlc_min_auto(scheme) && !have_challenges) {
      // too small for an aggregate to pay: the per-signature entry point as it is
      va.subs[k] = 0;
      traced[k] = RlcTraced{off, cnt, RlcPlan{}, RlcPlan{}, false};
      int r;
      const uint8_t* vin = valid_in ? valid_in + off : nullptr;
      if (scheme == 0)
        r = verify_single_on(ctx, c.u + 32 * off, c.R + 64 * off, c.PK + 64 * off, c.m + 32 * off, cnt, c.ok + off, workspace, s, vin);
      else if (scheme == 1)
        r = verify_double_on(ctx, c.u + 32 * off, c.R + 64 * off, c.Rp + 64 * off, c.PK + 64 * off, c.PKp + 64 * off,
                             c.m + 32 * off, cnt, c.ok + off, workspace, s, vin);
      else
        r = verify_vargen_on(ctx, c.u + 32 * off, c.R + 64 * off, c.PK + 64 * off, c.G + 64 * off, c.m + 32 * off, cnt,
                             c.ok + off, workspace, s, vin);
      if (r) return r;
      continue;
    }
    // how this group runs (file header): plain, split (sub-groups + sample) or guarded (two stages)
    int G = 1, G2 = 0;
    if (staged) {
      G = 1;  // (its first range is on the stream already, planned as one sub-group)
    } else if (history == 0 && history_long > 0 && guard_on) {
      G2 = force_groups > 0 ? force_groups : rlc_split_groups(cnt, window_bits);
      if (G2 < 4) G2 = 0;  // (two sub-groups: the second stage costs what it saves — 2^18 items: 4.41 against 4.40 ms)
      if (!G2 && force_groups > 0) G = force_groups;
    } else if (force_groups > 0) {
      G = force_groups;
    } else if (history > 0) {
      G = rlc_split_groups(cnt, window_bits);
    }
    const RlcPlan plan = rlc_group_plan(scheme, cnt, window_bits, G);
    RlcCarve cv = carve_rlc(workspace, group, cnt, plan);
    flags_area = cv.flags_area;
    cv.b.flags = cv.flags_area + (2 * k) * kRlcGroupFlagWords;
    va.subs[k] = (uint8_t)plan.groups;
    va.second[k] = 0;
    ChaChaKey key;
    if (staged) key = staged->key;  // (one group: the bucket pass of its first items is on the stream already)
    else if (int r = rlc_random_key(key)) return r;
    const bool do_sample = !window_bits && sample_on && (ctx.quad || scheme == 2) && history > 0 && !G2;
    traced[k] = RlcTraced{off, cnt, plan, RlcPlan{}, do_sample};
    if (!staged) HIP_TRY(launch_rlc_begin(cv.b, s));  // (staged: the hook did, before the first range)
    // (have_challenges: one group whose c / valid are in the workspace already — the host form hashes
    //  chunk by chunk while the transfers run)
    if (!have_challenges)
      launch_challenge(scheme == 1, c.R + 64 * off, scheme == 1 ? c.Rp + 64 * off : (const uint8_t*)nullptr, c.m + 32 * off, cnt,
                       cv.w.c, cv.w.valid, s, valid_in ? valid_in + off : nullptr);
    if (do_sample) rlc_enqueue_sample(c, cv, key, off, cnt);
    RlcInputs in = {};
    in.u = c.u + 32 * off, in.c = cv.w.c, in.valid = cv.w.valid;
    in.pk[0] = c.PK + 64 * off, in.r[0] = c.R + 64 * off;
    if (scheme == 1) in.pk[1] = c.PKp + 64 * off, in.r[1] = c.Rp + 64 * off;
    if (scheme == 2) in.gen = c.G + 64 * off;
    if (staged) {
      HIP_TRY(launch_rlc_buckets(scheme, rlc_range(plan, staged->boundary, cnt - staged->boundary), cv.b, in, key,
                                 c.ok + off, true, s));
      HIP_TRY(launch_rlc_finish(plan, cv.b, ctx.table[0], ctx.table[1], true, s));
    } else {
      HIP_TRY(launch_rlc_buckets(scheme, plan, cv.b, in, key, c.ok + off, false, s));
      HIP_TRY(launch_rlc_finish(plan, cv.b, ctx.table[0], ctx.table[1], false, s));
    }
    // what decides the per-signature launches: this stage's flag words, or (guarded) those of a second stage
    // of sub-group aggregates over the same challenges, with fresh weights, in the first stage's buffers
    const u32* gflags = cv.b.flags;
    size_t sub = plan.sub;
    bool split_fallback = plan.groups > 1;  // (ONE sub-group — the steady state — expects to be accepted: one launch)
    if (G2) {
      const RlcPlan plan2 = rlc_group_plan(scheme, cnt, window_bits, G2);
      RlcCarve cv2 = carve_rlc(workspace, group, cnt, plan2);
      cv2.b.flags = cv2.flags_area + (2 * k + 1) * kRlcGroupFlagWords;
      ChaChaKey key2;
      if (int r2 = rlc_random_key(key2)) return r2;
      HIP_TRY(launch_rlc_begin(cv2.b, s));
      launch_rlc_chain(cv.b.flags, cv2.b.flags, plan2.groups, s);
      HIP_TRY(launch_rlc_buckets(scheme, plan2, cv2.b, in, key2, c.ok + off, false, s));
      HIP_TRY(launch_rlc_finish(plan2, cv2.b, ctx.table[0], ctx.table[1], false, s));
      gflags = cv2.b.flags;
      sub = plan2.sub;
      split_fallback = false;  // (one gated launch per sub-group on the caller's stream: they are expected to return at once)
      va.subs[k] = (uint8_t)plan2.groups;
      va.second[k] = 1;
      traced[k].plan2 = plan2;
    }
    if (int r = rlc_enqueue_fallback(c, cv.w, gflags, sub, split_fallback, off, cnt)) return r;
  }
  if (!flags_area)
    flags_area = carve_rlc(workspace, group, group, rlc_group_plan(scheme, group, window_bits ? window_bits : 8, 1)).flags_area;
  va.and_into = and_into ? 1u : 0u;
  launch_rlc_verdict(flags_area, va, accepted_dev, ctx.rlc_pinned, s);
  HIP_TRY(hipGetLastError());
  if (trace) return rlc_trace(c, flags_area, va, traced, history, history_long);
  return DSV_OK;
}
}  // namespace dsvh
namespace {
// the largest workspace any plan of a call of n items takes (one group: every sub-group count)
size_t rlc_workspace_for(size_t n, int window_bits) {
  // the size for min(n, 2^22) items, not for the call's own group size: it must not DROP where the group
  // count steps up (a mixed batch sizes ONE workspace for both kinds' item counts, n / 2 each
  // just above 2^22 items where the whole batch runs in groups of n / 2 as well)
  const size_t g = n < kRlcMaxGroup ? n : kRlcMaxGroup;
  size_t most = 0;
  for (int G = 1; G <= kRlcMaxSub; G++) {
    // the double scheme's needs: four points per item, two fixed-base terms (the others fit inside)
    const RlcPlan p = rlc_group_plan(1, g, window_bits, G);
    const size_t b = carve_rlc(reinterpret_cast<void*>((uintptr_t)4096), g, g, p).bytes;
    most = b > most ? b : most;
  }
  return most;
}
}  // namespace
}  // extern "C++"
size_t dsv_rlc_workspace_bytes(size_t n, int window_bits) {
  if (n == 0) return 256;
  if (window_bits && !rlc_bits_ok(window_bits)) return 0;
  return rlc_workspace_for(n, window_bits) + 256;
}
// the geometry of one group's aggregate, for tests and sizing (no GPU needed): `groups` sub-groups
// (0 or 1: one); out[0..23] = c, half, wpk, wr, windows, nseg, nseg2, fine_bits, kmul, lpts, spts, fixed,
// entries, buckets, points of tmp[0], points of tmp[1], coarse_bits, rows, row_stride, bins, bin_cap,
// sub-groups, items per sub-group, workspace bytes of this plan
int dsv_rlc_plan_info(int scheme, size_t n, int window_bits, int groups, uint64_t* out) {
  if (!out || scheme < 0 || scheme > 2 || n == 0 || n > kRlcMaxGroup || groups < 0 || groups > kRlcMaxSub)
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  if (window_bits && !rlc_bits_ok(window_bits))
    return fail(DSV_ERR_INVALID_ARGUMENT, "window_bits must be 0 (automatic) or one of 4, 6, 8, 12, 14, 16");
  const RlcPlan p = rlc_group_plan(scheme, n, window_bits, groups);
  const uint64_t v[24] = {(uint64_t)p.c, (uint64_t)p.half, (uint64_t)p.wpk, (uint64_t)p.wr, (uint64_t)p.windows,
                          (uint64_t)p.nseg, (uint64_t)p.nseg2, (uint64_t)p.fine_bits, p.kmul, (uint64_t)p.lpts,
                          (uint64_t)p.spts, (uint64_t)p.fixed, p.entries, p.buckets, rlc_tmp_points(p, 0),
                          rlc_tmp_points(p, 1), (uint64_t)p.coarse_bits, p.rows, p.row_stride, p.bins, p.bin_cap,
                          p.groups, p.sub, carve_rlc(reinterpret_cast<void*>((uintptr_t)4096), n, n, p).bytes};
  for (int k = 0; k < 24; k++) out[k] = v[k];
  return DSV_OK;
}
// the history counter of a device (tests, tools): > 0 = the next call checks a sample and runs in
// sub-groups; set >= 0 overrides it
int dsv_debug_rlc_history(int device, int set) {
  if (device < 0 || device >= kMaxDevices || !g_ctx[device].ready.load(std::memory_order_acquire))
    return fail(DSV_ERR_NOT_INITIALIZED, "device %d is not initialised", device);
  Context& ctx = g_ctx[device];
  DSV_ON_DEVICE(ctx);
  const int h = rlc_history(ctx);
  if (h < 0) return DSV_ERR_HIP;
  if (set >= 0) *reinterpret_cast<volatile u32*>(ctx.rlc_pinned) = (u32)set;
  return h;
}
int dsv_debug_rlc_history_long(int device, int set) {
  if (device < 0 || device >= kMaxDevices || !g_ctx[device].ready.load(std::memory_order_acquire))
    return fail(DSV_ERR_NOT_INITIALIZED, "device %d is not initialised", device);
  Context& ctx = g_ctx[device];
  DSV_ON_DEVICE(ctx);
  if (rlc_history(ctx) < 0) return DSV_ERR_HIP;
  const int h = (int)*reinterpret_cast<volatile u32*>(ctx.rlc_pinned + 2);
  if (set >= 0) *reinterpret_cast<volatile u32*>(ctx.rlc_pinned + 2) = (u32)set;
  return h;
}
int dsv_debug_rlc_subgroups(int groups) {
  const int before = g_rlc_force_groups.load();
  if (groups >= 0) g_rlc_force_groups.store(groups > kRlcMaxSub ? kRlcMaxSub : groups);
  return before;
}
#define DSV_RLC_PROLOGUE(nullcheck)                                                              \
  if (n == 0) return rlc_clear_accepted(accepted);                                               \
  if (nullcheck) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");                          \
  if (window_bits && !rlc_bits_ok(window_bits))                                                  \
    return fail(DSV_ERR_INVALID_ARGUMENT, "window_bits must be 0 (automatic) or one of 4, 6, 8, 12, 14, 16"); \
  DSV_DEV_PROLOGUE(n, ok);                                                                       \
  RlcVerdictTarget vt;                                                                           \
  if (int r_ = rlc_verdict_target(ctx, accepted, vt)) return r_
int dsv_verify_single_rlc_dev(const void* u, const void* R_uv, const void* PK_uv, const void* m, size_t n,
                              void* ok, void* workspace, void* stream, int window_bits, int* accepted) {
  DSV_RLC_PROLOGUE(!u || !R_uv || !PK_uv || !m || !ok || !workspace);
  if (int r = verify_rlc_on(ctx, 0, u, R_uv, nullptr, PK_uv, nullptr, nullptr, m, n, ok, workspace, (hipStream_t)stream,
                            window_bits, vt.dev))
    return r;
  return rlc_verdict_wait(vt, (hipStream_t)stream);
}
int dsv_verify_double_rlc_dev(const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                              const void* PKp_uv, const void* m, size_t n, void* ok, void* workspace, void* stream,
                              int window_bits, int* accepted) {
  DSV_RLC_PROLOGUE(!u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok || !workspace);
  if (int r = verify_rlc_on(ctx, 1, u, R_uv, Rp_uv, PK_uv, PKp_uv, nullptr, m, n, ok, workspace, (hipStream_t)stream,
                            window_bits, vt.dev))
    return r;
  return rlc_verdict_wait(vt, (hipStream_t)stream);
}
int dsv_verify_vargen_rlc_dev(const void* u, const void* R_uv, const void* PK_uv, const void* Gen_uv, const void* m,
                              size_t n, void* ok, void* workspace, void* stream, int window_bits, int* accepted) {
  DSV_RLC_PROLOGUE(!u || !R_uv || !PK_uv || !Gen_uv || !m || !ok || !workspace);
  if (int r = verify_rlc_on(ctx, 2, u, R_uv, nullptr, PK_uv, nullptr, Gen_uv, m, n, ok, workspace, (hipStream_t)stream,
                            window_bits, vt.dev))
    return r;
  return rlc_verdict_wait(vt, (hipStream_t)stream);
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
      },
      kPipeNoVerdicts | (kind != 0 ? kPipeHeavy : 0u));
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
  // (two ranges only in the steady state: while the history says "batches fail" the group runs in sub-groups,
  //  all of them after the fill)
  const int force_groups = g_rlc_force_groups.load(std::memory_order_relaxed);
  RlcHook hook;
  hook.on = staged_on && n >= ((size_t)1 << 18) && rlc_history(ctx) == 0 && force_groups <= 1;
  if (hook.on) {
    for (auto& e : ar.ev)
      if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hook.ctx = &ctx, hook.kind = kind, hook.n = n, hook.ok = a.ok, hook.stream = ar.stream;
    hook.ev[0] = ar.ev[0], hook.ev[1] = ar.ev[1];
    hook.plan = rlc_group_plan(kind, n, 0, 1);  // (what verify_rlc_on plans for a staged group)
    hook.cv = carve_rlc(a.ws, n, n, hook.plan);
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
  RlcVerdictTarget vt;
  if (int r = rlc_verdict_target(ctx, accepted, vt)) return r;
  if (int r = verify_rlc_on(ctx, kind, a.u, R, Rp, PK, PKp, Gen, /*m: hashed already*/ a.u, n, a.ok, a.ws,
                            ar.stream, 0, vt.dev, false, true, nullptr, hook.fired ? &staged : nullptr))
    return r;
  HIP_TRY(hipMemcpyAsync(ok, a.ok, n, hipMemcpyDeviceToHost, ar.stream));
  HIP_TRY(hipStreamSynchronize(ar.stream));
  if (accepted) *accepted = vt.out ? (int)*reinterpret_cast<volatile u32*>(vt.slot) : *accepted;
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
  if (n == 0) return rlc_clear_accepted(accepted);
  if (!sig || !pk || !m || !ok || !workspace) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
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
  RlcVerdictTarget vt;
  if (int r = rlc_verdict_target(ctx, accepted, vt)) return r;
  if (int r = verify_rlc_on(ctx, kind, w.u, w.R, kind == 1 ? w.Rp : nullptr, w.P0, kind == 1 ? w.P1 : nullptr,
                            kind == 2 ? w.P1 : nullptr, m, n, ok, rws, st, window_bits, vt.dev, false, false, w.valid))
    return r;
  return rlc_verdict_wait(vt, st);
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
    }, kPipeNoVerdicts);
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
