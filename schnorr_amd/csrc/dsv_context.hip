// dsv_context.hip — MI355X (gfx950) batch Schnorr verification engine, host side: the per-device contexts
// (fixed-base tables, streams, staging, pipes), dsv_init / dsv_shutdown and the error channel.
// The kernels live in their own translation units (k_hash.hip, k_verify.hip, k_quad.hip, k_vargen.hip,
// k_misc.hip, k_rlc.hip; launchers declared in launch.h / rlc.h); the other host units are listed in
// dsv_host.h.
//
// Pipeline per batch (one lane = one signature, 64 signatures per wavefront; integer modular
// arithmetic on 29-bit limbs, see fe29.h.  The scalar multiplications use no cross-lane traffic,
// no LDS and no MFMA; the hash runs its constant linear layers as int8 products on the matrix
// cores, the 64 hashes of a wave cooperating — hades_mfma.h):
//
//   k_challenge          c = trunc250(Poseidon(R.u, R.v[, R'.u, R'.v], m))          (~20 % of the work)
//   k_verify_fixed_half  ok &= [ u*G + c*PK == R ], evaluated as                      (~80 %)
//                        (b*u mod r)*G + a*PK - b*R == O  with (a, b) ~ 128 bits, a = b*c mod 8r,
//                        b odd (halfgcd.h: same verdict on the whole curve group):
//                          a*PK - b*R : one per-lane table of the 11 combinations da*PK + db*R
//                                       of signed 2-bit digits (lane-major in a global workspace),
//                                       one Straus chain of ~66 windows (2 doublings + 1 addition)
//                          (b*u)*G    : 16 mixed additions from a signed 16-bit-window table of
//                                       G (or G'), 75.5 MB, built once on the device
//                        <2>: both equations of a double signature in one launch
//   k_verify_fixed_half_oct  the same with eight lanes per signature (batches <= 2^14), its window
//                        tables built by k_prep_var_tables on a side stream beside the hash
//   k_verify_var         both bases variable (PublicKeyVarGen): x*Gen + y*PK - z*R == O with three
//                        ~170-bit scalars (lattice3.h), three per-lane tables, ~44 windows
//   k_normalize_uvz      to_hash_inputs for callers that hold projective points (*_ext entry points)
//   k_decompress         wire-format points (JubJubAffine::from_bytes), decode29.h
//   k_fixed_base_points / k_var_base_points / k_sign_finish : signing and key derivation
//   k_kind_* / k_gather_rows / k_scatter_bytes : device-side split of a mixed batch by kind
//
// HBM traffic per signature is 193 B (single) / 321 B (double) / 257 B (vargen) in, 1 B out, plus
// 33 B of c/valid between the two kernels and the window-table workspace: the path is VALU-bound
// by orders of magnitude, not HBM-bound.
#define DSV_HOST_TABLES 1
#include "dsv_constants.h"
#include "dsv_host.h"

namespace dsvh {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

Context g_ctx[kMaxDevices];
std::mutex g_init_mu;
std::atomic<int> g_primary{-1};
thread_local int t_device = -1;
std::mutex g_jobs_mu;
std::condition_variable g_jobs_cv;
int g_jobs = 0;

int ensure_stage(Context& ctx, size_t bytes) {
  if (ctx.stage_bytes >= bytes) return DSV_OK;
  if (ctx.stage) {
    // what is released may hold secret keys / nonces of an earlier signing call
    HIP_TRY(hipMemset(ctx.stage, 0, ctx.stage_bytes));
    HIP_TRY(hipFree(ctx.stage));
  }
  ctx.stage = nullptr;
  ctx.stage_bytes = 0;
  size_t want = bytes + bytes / 4;
  HIP_TRY(hipMalloc(&ctx.stage, want));
  ctx.stage_bytes = want;
  return DSV_OK;
}

void destroy_pipe_streams(Context& ctx) {
  for (auto& pipe : ctx.pipes)
    for (auto& sl : pipe.slot) {
      hipEvent_t* evs[5] = {&sl.ev_in, &sl.ev_pre, &sl.ev_lane[0], &sl.ev_lane[1], &sl.ev_done};
      for (hipEvent_t* e : evs) {
        if (*e) (void)hipEventDestroy(*e);
        *e = nullptr;
      }
    }
  hipStream_t* streams[6] = {&ctx.pipe_lane[0], &ctx.pipe_lane[1], &ctx.pipe_pre, &ctx.pipe_in, &ctx.pipe_out, &ctx.pipe_small};
  for (hipStream_t* st : streams) {
    if (*st) (void)hipStreamDestroy(*st);
    *st = nullptr;
  }
  ctx.pipe_made = false;
}
// (called under enq_mu)
int ensure_pipe_streams(Context& ctx) {
  if (ctx.pipe_made) return DSV_OK;
  // The two compute lanes MUST sit on different hardware queues: ROCm multiplexes streams onto
  // GPU_MAX_HW_QUEUES (default 4) hardware queues per priority level, and two streams that land on one
  // queue run strictly one after the other — measured: both lanes on queue 4, no overlap at all, every
  // host path 10 - 20 % slower (profiles/r04/host_pipeline_streams.txt).  Which queue a stream gets
  // depends on every stream the process created before at that level (each level hands out its first
  // four queues one per stream, then shares them); the level itself does not: streams of different
  // priorities never share a queue.  So lane 0 is created at the highest priority, lane 1 at the middle
  // one, the two transfer streams (no kernels but a 2 us verdict copy) at the lowest.  The lanes carry
  // alternating sub-batches of equal work, so the priority only decides whose waves are dispatched first.
  // r05: the preprocessing stream (whole-chunk normalisation / limb conversion: few waves, one
  // inversion chain each) also sits at the HIGHEST level — r04 tried it at the lowest, where it came to
  // share a queue with the verdict copies and waited behind them (host_pipeline_streams.txt 6.).  Only
  // the library creates streams at that level (lane 0 here, one sub-batch stream per caller stream of
  // the device-pointer entry points), so it gets a queue of its own unless more than two caller
  // streams are in use; its chunk is needed a whole chunk (~3 ms) later, so its waves slot in at the
  // lanes' next kernel boundary and neither lane runs the latency-bound kernel in its own order.
  const int rc = [&]() -> int {
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&ctx.pipe_lane[0], hipStreamNonBlocking, greatest));
    HIP_TRY(hipStreamCreateWithPriority(&ctx.pipe_pre, hipStreamNonBlocking, greatest));
    HIP_TRY(hipStreamCreateWithPriority(&ctx.pipe_lane[1], hipStreamNonBlocking, (least + greatest) / 2));
    HIP_TRY(hipStreamCreateWithPriority(&ctx.pipe_in, hipStreamNonBlocking, least));
    HIP_TRY(hipStreamCreateWithPriority(&ctx.pipe_out, hipStreamNonBlocking, least));
    HIP_TRY(hipStreamCreateWithFlags(&ctx.pipe_small, hipStreamNonBlocking));
    for (auto& pipe : ctx.pipes)
      for (auto& sl : pipe.slot) {
        hipEvent_t* evs[5] = {&sl.ev_in, &sl.ev_pre, &sl.ev_lane[0], &sl.ev_lane[1], &sl.ev_done};
        for (hipEvent_t* e : evs) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
      }
    return DSV_OK;
  }();
  if (rc != DSV_OK) {  // nothing half-made stays behind: a later call starts from scratch
    const std::string why = g_err;
    destroy_pipe_streams(ctx);
    g_err = why;
    return rc;
  }
  ctx.pipe_made = true;
  return DSV_OK;
}
// (called under enq_mu: no sub-batch is being enqueued; work already in flight on the area is waited
//  for by hipFree itself, which synchronises the device)
int ensure_pipe_work(Context& ctx, int lane, size_t bytes) {
  if (ctx.pipe_work_bytes[lane] >= bytes) return DSV_OK;
  if (ctx.pipe_work[lane]) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(ctx.pipe_work[lane]));
  }
  ctx.pipe_work[lane] = nullptr;
  ctx.pipe_work_bytes[lane] = 0;
  HIP_TRY(hipMalloc(&ctx.pipe_work[lane], bytes));
  ctx.pipe_work_bytes[lane] = bytes;
  return DSV_OK;
}
int ensure_pipe_slot(PipeSlot& sl, size_t dev_bytes, size_t host_bytes, size_t prep_bytes) {
  if (sl.prep_bytes < prep_bytes) {
    if (sl.prep) HIP_TRY(hipFree(sl.prep));
    sl.prep = nullptr;
    sl.prep_bytes = 0;
    HIP_TRY(hipMalloc(&sl.prep, prep_bytes));
    sl.prep_bytes = prep_bytes;
  }
  if (sl.bytes < dev_bytes) {
    if (sl.stage) HIP_TRY(hipFree(sl.stage));
    sl.stage = nullptr;
    sl.bytes = 0;
    HIP_TRY(hipMalloc(&sl.stage, dev_bytes));
    sl.bytes = dev_bytes;
  }
  if (sl.host_bytes < host_bytes) {
    if (sl.host) HIP_TRY(hipHostFree(sl.host));
    sl.host = nullptr;
    sl.host_bytes = 0;
    HIP_TRY(hipHostMalloc(&sl.host, host_bytes, hipHostMallocDefault));
    sl.host_bytes = host_bytes;
  }
  return DSV_OK;
}

int check_n(size_t n) {
  if (n > DSV_MAX_BATCH) return fail(DSV_ERR_TOO_LARGE, "batch of %zu exceeds DSV_MAX_BATCH", n);
  return DSV_OK;
}
// context of the host entry points: this thread's dsv_set_device choice, else the first device
// that was initialised
int host_context(Context*& out) {
  const int d = t_device >= 0 ? t_device : g_primary.load(std::memory_order_acquire);
  if (d < 0 || d >= kMaxDevices || !g_ctx[d].ready.load(std::memory_order_acquire))
    return fail(DSV_ERR_NOT_INITIALIZED, d < 0 ? "dsv_init() has not been called"
                                               : "device %d is not initialised", d);
  out = &g_ctx[d];
  return DSV_OK;
}
// context of a device-pointer entry point: the device that owns `ptr` (one of the call's buffers)
int device_context(const void* ptr, Context*& out) {
  if (g_primary.load(std::memory_order_acquire) < 0)
    return fail(DSV_ERR_NOT_INITIALIZED, "dsv_init() has not been called");
  int d = -1;
  hipPointerAttribute_t attr;
  if (ptr && hipPointerGetAttributes(&attr, ptr) == hipSuccess &&
      (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged)) {
    d = attr.device;
  } else {
    (void)hipGetLastError();  // a failed query leaves a sticky "invalid value"
    if (hipGetDevice(&d) != hipSuccess) d = -1;
  }
  if (d < 0 || d >= kMaxDevices || !g_ctx[d].ready.load(std::memory_order_acquire))
    return fail(DSV_ERR_NOT_INITIALIZED, "device %d (owner of the buffers) is not initialised", d);
  out = &g_ctx[d];
  return DSV_OK;
}

// the dominant kernel: one lane per signature, or eight (small batches); same verdicts
void launch_verify_fixed(const Context& ctx, bool accumulate, const void* u, const void* c,
                         const void* PK_uv, const void* R_uv, int which, const void* valid,
                         size_t n, void* ok, u32* tables, hipStream_t s, bool tables_ready, const u32* gate) {
  const ChainOperands op{(const uint8_t*)PK_uv, (const uint8_t*)R_uv, ctx.table[which]};
  if (ctx.quad && n <= kQuadMaxItems)
    launch_verify_half_quad(1, accumulate, tables_ready, (const uint8_t*)u, (const uint8_t*)c, op, op,
                            (const uint8_t*)valid, n, (uint8_t*)ok, tables, s, gate);
  else
    launch_verify_half(1, accumulate, (const uint8_t*)u, (const uint8_t*)c, op, op,
                       (const uint8_t*)valid, n, (uint8_t*)ok, tables, s, gate);
}
// both equations of a double signature: one fused launch, or two single-equation ones
// (DSV_DOUBLE_FUSED=0: the second pass ANDs into ok[])
void launch_verify_fixed_double(const Context& ctx, const void* u, const void* c, const void* PK_uv,
                                const void* R_uv, const void* PKp_uv, const void* Rp_uv,
                                const void* valid, size_t n, void* ok, u32* tables, hipStream_t s,
                                bool tables_ready, const u32* gate) {
  if (ctx.fuse_double) {
    const ChainOperands op0{(const uint8_t*)PK_uv, (const uint8_t*)R_uv, ctx.table[0]};
    const ChainOperands op1{(const uint8_t*)PKp_uv, (const uint8_t*)Rp_uv, ctx.table[1]};
    if (ctx.quad && n <= kQuadMaxItems)
      launch_verify_half_quad(2, false, tables_ready, (const uint8_t*)u, (const uint8_t*)c, op0, op1,
                              (const uint8_t*)valid, n, (uint8_t*)ok, tables, s, gate);
    else
      launch_verify_half(2, false, (const uint8_t*)u, (const uint8_t*)c, op0, op1,
                         (const uint8_t*)valid, n, (uint8_t*)ok, tables, s, gate);
    return;
  }
  launch_verify_fixed(ctx, false, u, c, PK_uv, R_uv, 0, valid, n, ok, tables, s, tables_ready, gate);
  launch_verify_fixed(ctx, true, u, c, PKp_uv, Rp_uv, 1, valid, n, ok, tables, s, false, gate);
}

int acquire_lane(Context& ctx, hipStream_t user, SplitLane*& out) {
  std::lock_guard<std::mutex> lk(ctx.lane_mu);
  SplitLane* pick = nullptr;
  for (auto& l : ctx.lanes)
    if (l.made && l.owner == user) pick = &l;
  if (!pick)
    for (auto& l : ctx.lanes)
      if (!l.made) {
        pick = &l;
        break;
      }
  if (!pick) pick = &ctx.lanes[((uintptr_t)user >> 6) % kSplitLanes];  // all taken: share one
  if (!pick->made) {
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    // (the side stream FIRST: created behind the two priority streams it came to share a hardware
    //  queue with the caller's stream in the probe process and the small-batch overlap was gone —
    //  0.437 -> 0.539 ms per 1024-signature device call, profiles/r04/host_pipeline_streams.txt 10.)
    HIP_TRY(hipStreamCreateWithFlags(&pick->side, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
      HIP_TRY(hipStreamCreateWithPriority(&pick->stream[k], hipStreamNonBlocking, k == 0 ? greatest : least));
      HIP_TRY(hipEventCreateWithFlags(&pick->join[k], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&pick->side_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&pick->fork, hipEventDisableTiming));
    pick->owner = user;
    pick->made = true;
  }
  out = pick;
  return DSV_OK;
}

void release_context(Context& ctx) {
  // best effort: nothing useful can be done about a failing release
  (void)hipSetDevice(ctx.device);
  (void)hipDeviceSynchronize();
  for (int g = 0; g < 2; g++) {
    if (ctx.table[g]) (void)hipFree(ctx.table[g]);
    ctx.table[g] = nullptr;
  }
  if (ctx.ts_cancel) (void)hipFree(ctx.ts_cancel);
  if (ctx.ts_hash) (void)hipFree(ctx.ts_hash);
  ctx.ts_cancel = nullptr;
  ctx.ts_hash = nullptr;
  if (ctx.stage) {
    (void)hipMemset(ctx.stage, 0, ctx.stage_bytes);  // may hold secret keys / nonces
    (void)hipFree(ctx.stage);
  }
  ctx.stage = nullptr;
  ctx.stage_bytes = 0;
  for (auto& l : ctx.lanes) {
    if (!l.made) continue;
    for (int k = 0; k < 2; k++) {
      (void)hipStreamDestroy(l.stream[k]);
      (void)hipEventDestroy(l.join[k]);
    }
    if (l.side) (void)hipStreamDestroy(l.side);
    if (l.side_join) (void)hipEventDestroy(l.side_join);
    (void)hipEventDestroy(l.fork);
    l = SplitLane();
  }
  for (auto& pipe : ctx.pipes) {
    for (auto& sl : pipe.slot) {
      if (sl.stage) (void)hipFree(sl.stage);
      if (sl.host) (void)hipHostFree(sl.host);
      if (sl.prep) (void)hipFree(sl.prep);
      sl.stage = sl.host = sl.prep = nullptr;
      sl.bytes = sl.host_bytes = sl.prep_bytes = 0;
    }
    pipe.copiers.stop();
  }
  for (int k = 0; k < 3; k++) {
    if (ctx.pipe_work[k]) (void)hipFree(ctx.pipe_work[k]);
    ctx.pipe_work[k] = nullptr;
    ctx.pipe_work_bytes[k] = 0;
  }
  destroy_pipe_streams(ctx);
  ctx.pipe_failed = false;
  if (ctx.rlc_pinned) (void)hipHostFree(ctx.rlc_pinned);
  ctx.rlc_pinned = nullptr;
  for (auto& ar : ctx.rlc_arenas) {
    if (ar.dev) (void)hipFree(ar.dev);
    ar.dev = nullptr;
    ar.bytes = 0;
    if (ar.stream) (void)hipStreamDestroy(ar.stream);
    ar.stream = nullptr;
    for (auto& e : ar.ev) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
}

}  // namespace dsvh

using namespace dsvh;

extern "C" {

const char* dsv_version(void) { return "dsv 0.6.0 (gfx950, fe29)"; }
const char* dsv_last_error(void) { return g_err.c_str(); }

int dsv_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int dsv_init(int device) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail(DSV_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= count || device >= kMaxDevices)
    return fail(DSV_ERR_INVALID_ARGUMENT, "device %d out of range (count %d)", device, count);
  Context& ctx = g_ctx[device];
  if (ctx.ready.load()) return DSV_OK;
  ctx.device = device;
  DSV_ON_DEVICE(ctx);
  // (a failure half-way releases what was allocated: a later dsv_init starts from nothing)
  const int rc = [&]() -> int {
    HIP_TRY(hash_upload_constants());  // this device's __constant__ round constants
    HIP_TRY(hipMalloc(&ctx.ts_cancel, sizeof(DSV_TS_CANCEL_HOST)));
    HIP_TRY(hipMemcpy(ctx.ts_cancel, DSV_TS_CANCEL_HOST, sizeof(DSV_TS_CANCEL_HOST), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&ctx.ts_hash, sizeof(DSV_TS_HASH_HOST)));
    HIP_TRY(hipMemcpy(ctx.ts_hash, DSV_TS_HASH_HOST, sizeof(DSV_TS_HASH_HOST), hipMemcpyHostToDevice));
    for (int g = 0; g < 2; g++) {
      HIP_TRY(hipMalloc(&ctx.table[g], kTableBytes));
      launch_build_fixed_table(ctx.table[g], g, 0);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    return DSV_OK;
  }();
  if (rc != DSV_OK) {
    const std::string why = g_err;
    release_context(ctx);
    g_err = why;
    return rc;
  }
  {
    // NUMA placement of this device's host-side threads (DSV_NUMA=0: none; DSV_SYSFS_ROOT: another sysfs tree)
    char bdf[64] = {0};
    ctx.pci_bdf.clear();
    ctx.numa_node = -1;
    ctx.numa_cpus.clear();
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) == hipSuccess) ctx.pci_bdf = bdf;
    else (void)hipGetLastError();
    const char* off = getenv("DSV_NUMA");
    if (!(off && strcmp(off, "0") == 0) && !ctx.pci_bdf.empty()) {
      const char* root = getenv("DSV_SYSFS_ROOT");
      const std::string sys = root && *root ? root : "/sys";
      ctx.numa_node = numa_node_of_pci(sys, ctx.pci_bdf);
      ctx.numa_cpus = cpus_of_numa_node(sys, ctx.numa_node);
    }
    for (auto& pipe : ctx.pipes) pipe.copiers.set_affinity(ctx.numa_cpus);
  }
  const char* split = getenv("DSV_SPLIT");
  ctx.split = !(split && strcmp(split, "0") == 0);
  const char* quad = getenv("DSV_QUAD");
  ctx.quad = !(quad && strcmp(quad, "0") == 0);
  const char* sov = getenv("DSV_SMALL_OVERLAP");
  ctx.small_overlap = !(sov && strcmp(sov, "0") == 0);
  const char* fused = getenv("DSV_DOUBLE_FUSED");
  ctx.fuse_double = !(fused && strcmp(fused, "0") == 0);
  const char* pre = getenv("DSV_PIPE_PREP_STREAM");
  ctx.prep_stream = pre && strcmp(pre, "1") == 0;
  auto log2_env = [](const char* name, int lo, int hi, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    return (size_t)1 << (v < lo ? lo : (v > hi ? hi : v));
  };
  ctx.plan.chunk = log2_env("DSV_PIPE_CHUNK_LOG2", 16, 20, 18);        // chunk size of the host pipeline
  ctx.plan.first_chunk = log2_env("DSV_PIPE_FIRST_LOG2", 12, 18, 15);  // ... of a call's first chunk (doubling from there)
  if (ctx.plan.first_chunk > ctx.plan.chunk) ctx.plan.first_chunk = ctx.plan.chunk;
  ctx.plan.plan_len = 0;
  if (const char* e = getenv("DSV_PIPE_PLAN")) {
    for (const char* q = e; *q && ctx.plan.plan_len < 16;) {
      char* end = nullptr;
      const long v = strtol(q, &end, 10);
      if (end == q) break;
      ctx.plan.plan[ctx.plan.plan_len++] = v < 12 ? 12 : (v > 20 ? 20 : (int)v);
      q = *end == ',' ? end + 1 : end;
    }
  }
  ctx.pipe_slots = 0;
  ctx.plan.growth = 0;
  // items that share one inversion in the PIPELINE's normalisation: 4 (the device-resident entry points keep
  // 8 - 16: there the kernel runs once, alone; here it sits in a compute lane's order, where its length counts)
  ctx.norm_per_lane = getenv("DSV_NORM_PER_LANE") ? atoi(getenv("DSV_NORM_PER_LANE")) : 4;
  ctx.norm_block = getenv("DSV_NORM_BLOCK") ? atoi(getenv("DSV_NORM_BLOCK")) : 0;
  if (const char* e = getenv("DSV_PIPE_SLOTS")) {
    const int v = atoi(e);
    ctx.pipe_slots = v < 2 ? 2 : (v > kPipeSlots ? kPipeSlots : v);
  }
  if (const char* e = getenv("DSV_PIPE_GROWTH")) {
    const int pct = atoi(e);
    ctx.plan.growth = (pct < 110 ? 110 : (pct > 400 ? 400 : pct)) / 100.0;
  }
  ctx.ready.store(true, std::memory_order_release);
  int none = -1;
  g_primary.compare_exchange_strong(none, device);
  return DSV_OK;
}

// Initialise the devices a process wants to use in one call: the ordinals listed in the
// environment variable DSV_DEVICES (comma-separated, e.g. "0,2,3"), else every visible device.
// Returns the number of initialised devices or a negative dsv_status.  What the language shims call
// from their first verify_batch: an integrator restricts the engine's footprint (151 MB of tables
// and ~50 ms of table construction per device) with the variable, not with code.
int dsv_init_visible(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail(DSV_ERR_NO_DEVICE, "no HIP device visible");
  int done = 0;
  if (const char* e = getenv("DSV_DEVICES")) {
    for (const char* p = e; *p;) {
      char* end = nullptr;
      const long d = strtol(p, &end, 10);
      if (end == p) return fail(DSV_ERR_INVALID_ARGUMENT, "DSV_DEVICES: cannot parse '%s'", e);
      if (int r = dsv_init((int)d)) return r;
      done++;
      p = *end == ',' ? end + 1 : end;
      if (*end && *end != ',') return fail(DSV_ERR_INVALID_ARGUMENT, "DSV_DEVICES: cannot parse '%s'", e);
    }
    if (!done) return fail(DSV_ERR_INVALID_ARGUMENT, "DSV_DEVICES is empty");
    return done;
  }
  for (int d = 0; d < count && d < kMaxDevices; d++) {
    if (int r = dsv_init(d)) return r;
    done++;
  }
  return done;
}

int dsv_shutdown_device(int device) {
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (device < 0 || device >= kMaxDevices) return fail(DSV_ERR_INVALID_ARGUMENT, "bad device %d", device);
  Context& ctx = g_ctx[device];
  if (!ctx.ready.load()) return DSV_OK;
  {
    std::unique_lock<std::mutex> jl(g_jobs_mu);
    g_jobs_cv.wait(jl, [] { return g_jobs == 0; });
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  ctx.ready.store(false);  // new calls are refused from here on
  {
    // host calls in flight finish first: the pipelined ones hold a pipe, the small ones `mu`
    std::unique_lock<std::mutex> pl(ctx.pipe_sync.mu);
    ctx.pipe_sync.cv.wait(pl, [&] { return ctx.pipe_sync.idle(); });
    pl.unlock();
    // fast-accept host calls past their pipeline phase finish first
    std::lock_guard<std::mutex> rlc0(ctx.rlc_arenas[0].mu), rlc1(ctx.rlc_arenas[1].mu);
    std::lock_guard<std::mutex> hold(ctx.mu);
    std::lock_guard<std::mutex> enq(ctx.enq_mu);
    release_context(ctx);
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  if (g_primary.load() == device) {
    int next = -1;
    for (int d = 0; d < kMaxDevices; d++)
      if (g_ctx[d].ready.load()) {
        next = d;
        break;
      }
    g_primary.store(next);
  }
  return DSV_OK;
}
int dsv_shutdown(void) {
  for (int d = 0; d < kMaxDevices; d++)
    if (int r = dsv_shutdown_device(d)) return r;
  return DSV_OK;
}
int dsv_set_device(int device) {
  if (device < 0 || device >= kMaxDevices || !g_ctx[device].ready.load(std::memory_order_acquire))
    return fail(DSV_ERR_NOT_INITIALIZED, "device %d is not initialised", device);
  t_device = device;
  return DSV_OK;
}
int dsv_get_device(void) { return t_device >= 0 ? t_device : g_primary.load(); }
// where a device's host-side work is placed: *node = NUMA node of its PCIe root (-1: unknown / DSV_NUMA=0),
// cpus[0 .. min(count, cap)) = that node's cpus; bdf (may be null): its PCI address, at most 31 characters.
// Returns the cpu count or a negative dsv_status.
int dsv_device_numa(int device, int* node, int* cpus, int cap, char* bdf) {
  if (device < 0 || device >= kMaxDevices || !g_ctx[device].ready.load(std::memory_order_acquire))
    return fail(DSV_ERR_NOT_INITIALIZED, "device %d is not initialised", device);
  const Context& ctx = g_ctx[device];
  if (node) *node = ctx.numa_node;
  for (int k = 0; cpus && k < cap && k < (int)ctx.numa_cpus.size(); k++) cpus[k] = ctx.numa_cpus[k];
  if (bdf) {
    strncpy(bdf, ctx.pci_bdf.c_str(), 31);
    bdf[31] = 0;
  }
  return (int)ctx.numa_cpus.size();
}
// the lookup itself, without a device (tests: a fake sysfs tree): node and cpus of the PCI device `bdf`
// under `sysfs_root`; returns the cpu count
int dsv_debug_numa_lookup(const char* sysfs_root, const char* bdf, int* node, int* cpus, int cap) {
  if (!sysfs_root || !bdf) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  const int nd = numa_node_of_pci(sysfs_root, bdf);
  const std::vector<int> list = cpus_of_numa_node(sysfs_root, nd);
  if (node) *node = nd;
  for (int k = 0; cpus && k < cap && k < (int)list.size(); k++) cpus[k] = list[k];
  return (int)list.size();
}
int dsv_initialized_devices(int* out, int cap) {
  int n = 0;
  for (int d = 0; d < kMaxDevices; d++)
    if (g_ctx[d].ready.load(std::memory_order_acquire)) {
      if (out && n < cap) out[n] = d;
      n++;
    }
  return n;
}

size_t dsv_workspace_bytes(size_t n) {
  // window tables: one launch over n items, or (run_split) two concurrent launches over
  // kSplitItems items each — whichever is larger (they differ when -DDSV_MAX_VERIFY_GRID < 2048)
  size_t tables = var_table_bytes(n, kTablesPerLane);
  if (n >= 2 * kSplitItems && tables < 2 * var_table_bytes(kSplitItems, kTablesPerLane))
    tables = 2 * var_table_bytes(kSplitItems, kTablesPerLane);
  return align_up(n * 32, 256) + align_up(n, 256) + tables + 256;
}

}  // extern "C"
