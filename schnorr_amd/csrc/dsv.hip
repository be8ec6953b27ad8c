// dsv.hip — MI355X (gfx950) batch Schnorr verification engine: host side + C ABI (include/dsv.h).
// The kernels live in their own translation units (k_hash.hip, k_verify.hip, k_quad.hip,
// k_vargen.hip, k_misc.hip; launchers declared in launch.h); this file owns the per-device
// contexts, the sub-batch / pipeline scheduling and every extern "C" entry point.
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
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <errno.h>
#include <sys/random.h>

#include "../../include/dsv.h"
#define DSV_HOST_TABLES 1
#include "dsv_constants.h"
#include "launch.h"
#include "rlc.h"
#include "host_sync.h"


// ==========================================================================================
// host side: per-device contexts + C ABI
// ==========================================================================================
namespace {

using namespace dsv;
typedef uint32_t u32;

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
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(DSV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                  \
  } while (0)

constexpr int kPipeSlots = 8;   // chunks in flight per host call: at most (Context::pipe_slots are used)
constexpr int kMaxDevices = 16;
constexpr int kSplitLanes = 8;

// Two internal streams + the events that fork them from / join them to ONE caller stream.  A lane
// is bound to the caller stream that first used it, so callers on different streams get different
// internal streams and really run concurrently (r01 had two process-wide streams: every large
// batch of every caller queued on them).  With more than kSplitLanes distinct caller streams in
// flight lanes are shared by hashing — still correct (work is ordered by the events), only
// serialised.
// The two sub-batch streams sit on DIFFERENT PRIORITY LEVELS (highest / lowest): ROCm multiplexes
// streams onto four hardware queues per priority level, two streams that land on one queue run
// strictly one after the other, and which queue a plain stream gets depends on every stream the
// process — torch, RCCL, the caller — created before; streams of different levels never share a
// queue (profiles/r04/host_pipeline_streams.txt: the 2^20 headline is the same with plain streams
// when they happen not to collide, 6 - 10 % lower when they do).  The side stream of the small-batch
// table preparation stays a plain one: on a priority stream a 1024-signature call takes 25 % longer.
struct SplitLane {
  hipStream_t owner = nullptr;
  bool made = false;
  hipStream_t stream[2] = {nullptr, nullptr};
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr}, side_join = nullptr;
};

// One chunk in flight of a host call: staging on both sides, the events that chain its stages.
struct PipeSlot {
  hipEvent_t ev_in = nullptr;            // the chunk's transfer to the device is done
  hipEvent_t ev_pre = nullptr;           // its whole-chunk preprocessing is done
  hipEvent_t ev_lane[2] = {nullptr, nullptr};  // its sub-batches on compute lane k are done
  hipEvent_t ev_done = nullptr;          // its verdicts are in the pinned block
  uint8_t* prep = nullptr;               // what the preprocessing produced
  size_t prep_bytes = 0;
  uint8_t* stage = nullptr;              // device side: the input block, then the verdicts
  size_t bytes = 0;
  uint8_t* host = nullptr;               // pinned host side (inputs, then verdicts)
  size_t host_bytes = 0;
};
// What ONE host call in flight owns.
struct Pipe {
  PipeSlot slot[kPipeSlots];
  CopyPool copiers;
};

// Everything the library owns on one GPU.  One Context per device ordinal; several devices can be
// initialised in one process (dsv_init(d) for each) and used concurrently from different host
// threads: no state is shared between contexts.
struct Context {
  std::atomic<bool> ready{false};
  int device = -1;
  bool split = true;         // DSV_SPLIT=0: one stream, whole batch per launch
  bool fuse_double = true;   // DSV_DOUBLE_FUSED=0: two single-equation launches per double batch (r01; A/B)
  bool quad = true;          // DSV_QUAD=0: batches of <= 2^14 items also take the one-lane-per-signature kernel
  bool small_overlap = true; // DSV_SMALL_OVERLAP=0: small batches build their window tables inside the verify kernel
  u32* table[2] = {nullptr, nullptr};  // fixed-base tables for G, G'
  u32* ts_cancel = nullptr;            // square-root tables (decode29.h)
  uint8_t* ts_hash = nullptr;
  // host-pointer entry points of this device serialise here (they share the staging below)
  std::mutex mu;
  uint8_t* stage = nullptr;
  size_t stage_bytes = 0;
  // sub-batch lanes of the device-pointer entry points (run_split)
  std::mutex lane_mu;
  SplitLane lanes[kSplitLanes];
  // Pipeline of the host verify entry points (run_pipelined).  SIX streams per device, shared by every
  // call: one for the transfers in, one for whole-chunk preprocessing, two compute lanes that all
  // sub-batches of all calls alternate between, one for the verdicts out, one for small one-chunk
  // calls.  What a call owns while it runs is a Pipe: kPipeSlots chunks in flight, each with its own
  // device and pinned host staging, its events, and the call's copy threads.  kPipes calls can be in
  // flight per device (r05; r01 - r04 held `mu` for the whole call): the second call's ramp — small
  // first chunks, an idle GPU waiting for the first transfer — runs under the first call's tail.
  PipeSync pipe_sync;                     // pipe acquisition (FIFO by ticket), compute turns, shutdown (host_sync.h)
  Pipe pipes[kPipes];
  std::mutex enq_mu;                      // one chunk's enqueue onto the shared streams is atomic: the
                                          // lanes' work areas below belong to the sub-batch being enqueued
  bool pipe_made = false;                 // streams + events exist
  bool pipe_failed = false;               // ... could not be created (reported on every later call)
  hipStream_t pipe_in = nullptr, pipe_out = nullptr, pipe_lane[2] = {nullptr, nullptr}, pipe_pre = nullptr;
  hipStream_t pipe_small = nullptr;       // a call of one small chunk runs on this stream alone
  uint8_t* pipe_work[3] = {};             // per compute lane (+ [2]: pipe_small): verify workspace + scratch of one sub-batch
  size_t pipe_work_bytes[3] = {};
  uint64_t pipe_parts = 0;                // sub-batches enqueued so far (ties between the lanes alternate by it)
  long lane_load[2] = {0, 0};             // sub-batches enqueued on a lane and not yet known to be done (under enq_mu)
  bool prep_stream = false;               // DSV_PIPE_PREP_STREAM=1: whole-chunk preprocessing on a stream of its own (A/B)
  PlanParams plan;                        // chunk plan of a call (host_sync.h; DSV_PIPE_CHUNK_LOG2 / _FIRST_LOG2 / _GROWTH / _PLAN)
  int pipe_slots = 0;                     // DSV_PIPE_SLOTS: chunks in flight per call (<= kPipeSlots); 0 = three
  int norm_per_lane = 0, norm_block = 0;  // DSV_NORM_PER_LANE / DSV_NORM_BLOCK: shape of the pipeline's normalisation kernels
  // batch fast accept from host memory (dsv_verify_*_mont_cols_rlc): the whole group's normalised inputs,
  // its verdict bytes and the aggregate's workspace stay resident in an arena for the duration of a
  // call.  Two arenas per device: while one call's aggregate runs (a latency-bound tail on a stream of
  // its own), the next call's pipeline already fills the other
  struct RlcHostArena {
    std::mutex mu;
    uint8_t* dev = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};  // "the compute lanes have reached the staged range's end"
  } rlc_arenas[2];
  // the fast accept's sample check (kRlcSample items through the per-signature kernel before an aggregate
  // is paid for) runs while this is > 0: a rejected group sets it to 8, an accepted one takes 1 off — a
  // caller whose batches are valid pays for it on the first call only, one whose batches are tampered
  // with pays an aggregate once and 0.26 ms per group from then on
  std::atomic<int> rlc_suspicion{1};
  std::mutex rlc_sample_mu;
  uint8_t* rlc_sample_host = nullptr;  // pinned: the sample's verdicts and what tells a wrong item from a malformed one
};
Context g_ctx[kMaxDevices];
std::mutex g_init_mu;               // dsv_init / dsv_shutdown
std::atomic<int> g_primary{-1};     // first device initialised: default of the host entry points
thread_local int t_device = -1;     // dsv_set_device: this thread's choice for host entry points
// jobs submitted and not finished (dsv_*_submit): dsv_shutdown lets them run to their verdicts first —
// a job's driver may not have queued for its pipe yet when the shutdown arrives
std::mutex g_jobs_mu;
std::condition_variable g_jobs_cv;
int g_jobs = 0;

// the calling thread's current HIP device is restored on scope exit: the library must not leave a
// caller's thread on another GPU
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) {
      err = hipSetDevice(device);
      switched = err == hipSuccess;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};
#define DSV_ON_DEVICE(ctx)                     \
  DeviceGuard guard_((ctx).device);            \
  if (guard_.err != hipSuccess)                \
  return fail(DSV_ERR_HIP, "cannot select device %d: %s", (ctx).device, hipGetErrorString(guard_.err))

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

// workspace layout for the *_dev verify entry points: c[n][32] | valid[n]
// + the per-lane window tables of the verify kernels (fixed grid, see kMaxVerifyGrid)
struct Workspace {
  uint8_t* c;
  uint8_t* valid;
  u32* tables;
};
constexpr int kTablesPerLane = 3;  // the var-generator kernel keeps three (Gen, PK, R), the others two
size_t var_table_bytes(size_t n, int tables_per_lane) {
  return (size_t)verify_grid(n) * kVerifyBlock * kVarLaneWords * 4 * (size_t)tables_per_lane;
}
Workspace carve(void* ws, size_t n) {
  Workspace w;
  w.c = static_cast<uint8_t*>(ws);
  w.valid = w.c + align_up(n * 32, 256);
  w.tables = reinterpret_cast<u32*>(w.valid + align_up(n, 256));
  return w;
}

struct Stager {
  uint8_t* base;
  size_t off = 0;
  explicit Stager(uint8_t* b) : base(b) {}
  uint8_t* take(size_t bytes) {
    uint8_t* p = base + off;
    off += align_up(bytes, 256);
    return p;
  }
};

// the dominant kernel: one lane per signature, or eight (small batches); same verdicts
void launch_verify_fixed(const Context& ctx, bool accumulate, const void* u, const void* c,
                         const void* PK_uv, const void* R_uv, int which, const void* valid,
                         size_t n, void* ok, u32* tables, hipStream_t s, bool tables_ready = false) {
  const ChainOperands op{(const uint8_t*)PK_uv, (const uint8_t*)R_uv, ctx.table[which]};
  if (ctx.quad && n <= kQuadMaxItems)
    launch_verify_half_quad(1, accumulate, tables_ready, (const uint8_t*)u, (const uint8_t*)c, op, op,
                            (const uint8_t*)valid, n, (uint8_t*)ok, tables, s);
  else
    launch_verify_half(1, accumulate, (const uint8_t*)u, (const uint8_t*)c, op, op,
                       (const uint8_t*)valid, n, (uint8_t*)ok, tables, s);
}
// both equations of a double signature: one fused launch, or two single-equation ones
// (DSV_DOUBLE_FUSED=0: the second pass ANDs into ok[])
void launch_verify_fixed_double(const Context& ctx, const void* u, const void* c, const void* PK_uv,
                                const void* R_uv, const void* PKp_uv, const void* Rp_uv,
                                const void* valid, size_t n, void* ok, u32* tables, hipStream_t s,
                                bool tables_ready = false) {
  if (ctx.fuse_double) {
    const ChainOperands op0{(const uint8_t*)PK_uv, (const uint8_t*)R_uv, ctx.table[0]};
    const ChainOperands op1{(const uint8_t*)PKp_uv, (const uint8_t*)Rp_uv, ctx.table[1]};
    if (ctx.quad && n <= kQuadMaxItems)
      launch_verify_half_quad(2, false, tables_ready, (const uint8_t*)u, (const uint8_t*)c, op0, op1,
                              (const uint8_t*)valid, n, (uint8_t*)ok, tables, s);
    else
      launch_verify_half(2, false, (const uint8_t*)u, (const uint8_t*)c, op0, op1,
                         (const uint8_t*)valid, n, (uint8_t*)ok, tables, s);
    return;
  }
  launch_verify_fixed(ctx, false, u, c, PK_uv, R_uv, 0, valid, n, ok, tables, s, tables_ready);
  launch_verify_fixed(ctx, true, u, c, PKp_uv, Rp_uv, 1, valid, n, ok, tables, s);
}

// Sub-batch scheduling of the device-pointer verify entry points.
// One launch over 2^20 signatures pays a fill and a drain phase per kernel (~0.5 ms + ~0.8 ms of
// 18 ms, tools/scaling_probe.py) and runs the two waves of every SIMD through the same phase of
// the same kernel.  Cutting the batch into sub-batches of 2^16 signatures (1024 waves: ONE wave
// per SIMD) that alternate between two internal streams keeps two different kernels co-resident
// on every SIMD — hash next to scalar multiplication, table build next to window loop — and
// leaves no gap between kernels: +6 % on 2^20, same-box A/B (probe: tools/overlap_probe.py).  The caller's stream is
// forked / joined with events, so the call still behaves as one enqueue on that stream.

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
// part(offset, count, workspace-for-this-part, stream)
template <class Part>
int run_split(Context& ctx, size_t n, void* workspace, hipStream_t user, Part part) {
  Workspace w = carve(workspace, n);
  if (!ctx.split || n < 2 * kSplitItems) {
    part((size_t)0, n, w, user);
    HIP_TRY(hipGetLastError());
    return DSV_OK;
  }
  SplitLane* lane = nullptr;
  if (int r = acquire_lane(ctx, user, lane)) return r;
  // a shared lane's events may be re-recorded by another caller between our record and our
  // wait; record + wait pairs are therefore issued under the lane lock
  std::lock_guard<std::mutex> lk(ctx.lane_mu);
  HIP_TRY(hipEventRecord(lane->fork, user));
  const size_t tbl_words = var_table_bytes(kSplitItems, kTablesPerLane) / 4;  // per internal stream
  for (int k = 0; k < 2; k++) HIP_TRY(hipStreamWaitEvent(lane->stream[k], lane->fork, 0));
  size_t off = 0;
  for (size_t p = 0; off < n; p++) {
    const size_t cnt = n - off < kSplitItems ? n - off : kSplitItems;
    const int k = (int)(p & 1);
    Workspace wp;
    wp.c = w.c + off * 32;
    wp.valid = w.valid + off;
    wp.tables = w.tables + (size_t)k * tbl_words;
    part(off, cnt, wp, lane->stream[k]);
    off += cnt;
  }
  HIP_TRY(hipGetLastError());
  for (int k = 0; k < 2; k++) {
    HIP_TRY(hipEventRecord(lane->join[k], lane->stream[k]));
    HIP_TRY(hipStreamWaitEvent(user, lane->join[k], 0));
  }
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
  ctx.rlc_suspicion.store(1);
  if (ctx.rlc_sample_host) (void)hipHostFree(ctx.rlc_sample_host);
  ctx.rlc_sample_host = nullptr;
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

}  // namespace

extern "C" {

const char* dsv_version(void) { return "dsv 0.5.1 (gfx950, fe29)"; }
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

// ---- device-pointer entry points --------------------------------------------------------
// The context is the one of the device that owns the output buffer; the calling thread's current
// device is switched for the duration of the call and restored.
#define DSV_DEV_PROLOGUE(n, owner_ptr)              \
  if (int r_ = check_n(n)) return r_;               \
  if ((n) == 0) return DSV_OK;                      \
  Context* ctxp_ = nullptr;                         \
  if (int r_ = device_context(owner_ptr, ctxp_)) return r_; \
  Context& ctx = *ctxp_;                            \
  DSV_ON_DEVICE(ctx)

int dsv_challenge_single_dev(const void* R_uv, const void* m, size_t n, void* c, void* valid,
                             void* stream) {
  if (n && (!R_uv || !m || !c)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, c);
  launch_challenge(false, (const uint8_t*)R_uv, (const uint8_t*)nullptr, (const uint8_t*)m, n, (uint8_t*)c, (uint8_t*)valid, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_challenge_double_dev(const void* R_uv, const void* Rp_uv, const void* m, size_t n, void* c,
                             void* valid, void* stream) {
  if (n && (!R_uv || !Rp_uv || !m || !c)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, c);
  launch_challenge(true, (const uint8_t*)R_uv, (const uint8_t*)Rp_uv, (const uint8_t*)m, n, (uint8_t*)c, (uint8_t*)valid, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

namespace {
// bodies shared by the device-pointer entry points and the host pipeline (context resolved)
// Small batches (eight-lane kernel): the window tables of (PK, R) do not depend on the challenge, so
// they are built on one of the lane's internal streams WHILE k_challenge runs on the caller's: fork
// by an event here; the caller enqueues the hash, then waits for the returned join event on its own
// stream in front of the verify kernel.  nullptr: not a small batch (or the overlap is off, or no
// lane could be had) — the verify kernel then builds its tables itself.
thread_local bool t_pipeline_part = false;  // run_pipelined, several chunks: its four streams are all there is
thread_local size_t t_chunk_first = 0;      // run_pipelined: first item of the chunk being enqueued (prep / part
                                            // callbacks that place their output by item number: the fast accept)
hipEvent_t prep_tables_beside_hash(Context& ctx, const void* PK_uv, const void* R_uv, size_t n,
                                   u32* tables, hipStream_t user) {
  if (!(ctx.quad && ctx.small_overlap && n <= kQuadMaxItems) || t_pipeline_part) return nullptr;
  SplitLane* lane = nullptr;
  if (acquire_lane(ctx, user, lane) != DSV_OK) return nullptr;
  std::lock_guard<std::mutex> lk(ctx.lane_mu);
  if (hipEventRecord(lane->fork, user) != hipSuccess ||
      hipStreamWaitEvent(lane->side, lane->fork, 0) != hipSuccess)
    return nullptr;
  launch_prep_var_tables((const uint8_t*)PK_uv, (const uint8_t*)R_uv, n, tables, lane->side);
  if (hipEventRecord(lane->side_join, lane->side) != hipSuccess) {
    // the prep kernel is already writing this call's table slots: it must have finished before the
    // verify kernel, told to build its tables itself, writes the same slots from the caller's stream
    (void)hipStreamSynchronize(lane->side);
    return nullptr;
  }
  return lane->side_join;
}
// valid_in (may be null): per-item validity found by an earlier stage (normalisation, decompression);
// the hash kernel folds it into the validity the verify kernel starts from
int verify_single_on(Context& ctx, const void* u, const void* R_uv, const void* PK_uv, const void* m,
                     size_t n, void* ok, void* workspace, hipStream_t stream, const uint8_t* valid_in = nullptr) {
  const uint8_t *pu = (const uint8_t*)u, *pR = (const uint8_t*)R_uv, *pPK = (const uint8_t*)PK_uv,
                *pm = (const uint8_t*)m;
  uint8_t* pok = (uint8_t*)ok;
  Context* cp = &ctx;
  return run_split(ctx, n, workspace, stream,
                   [=](size_t off, size_t cnt, const Workspace& w, hipStream_t s) {
    // (only for an unsplit call: inside run_split's loop the lane lock is held and cnt > 2^14 anyway,
    //  except for a short last part, which simply builds its tables in the kernel)
    hipEvent_t ready = cnt == n ? prep_tables_beside_hash(*cp, pPK + 64 * off, pR + 64 * off, cnt, w.tables, s)
                                : nullptr;
    launch_challenge(false, pR + 64 * off, (const uint8_t*)nullptr, pm + 32 * off, cnt, w.c, w.valid, s,
                     valid_in ? valid_in + off : nullptr);
    if (ready && hipStreamWaitEvent(s, ready, 0) != hipSuccess) return;  // (surfaces through hipGetLastError)
    launch_verify_fixed(*cp, false, pu + 32 * off, w.c, pPK + 64 * off, pR + 64 * off, 0, w.valid,
                        cnt, pok + off, w.tables, s, ready != nullptr);
  });
}
int verify_double_on(Context& ctx, const void* u, const void* R_uv, const void* Rp_uv,
                     const void* PK_uv, const void* PKp_uv, const void* m, size_t n, void* ok,
                     void* workspace, hipStream_t stream, const uint8_t* valid_in = nullptr) {
  const uint8_t *pu = (const uint8_t*)u, *pR = (const uint8_t*)R_uv, *pRp = (const uint8_t*)Rp_uv,
                *pPK = (const uint8_t*)PK_uv, *pPKp = (const uint8_t*)PKp_uv, *pm = (const uint8_t*)m;
  uint8_t* pok = (uint8_t*)ok;
  Context* cp = &ctx;
  return run_split(ctx, n, workspace, stream,
                   [=](size_t off, size_t cnt, const Workspace& w, hipStream_t s) {
    // (only for an unsplit call: inside run_split's loop the lane lock is held and cnt > 2^14 anyway,
    //  except for a short last part, which simply builds its tables in the kernel)
    hipEvent_t ready = cnt == n ? prep_tables_beside_hash(*cp, pPK + 64 * off, pR + 64 * off, cnt, w.tables, s)
                                : nullptr;
    launch_challenge(true, pR + 64 * off, pRp + 64 * off, pm + 32 * off, cnt, w.c, w.valid, s,
                     valid_in ? valid_in + off : nullptr);
    if (ready && hipStreamWaitEvent(s, ready, 0) != hipSuccess) return;
    launch_verify_fixed_double(*cp, pu + 32 * off, w.c, pPK + 64 * off, pR + 64 * off,
                               pPKp + 64 * off, pRp + 64 * off, w.valid, cnt, pok + off, w.tables, s,
                               ready != nullptr);
  });
}
int verify_vargen_on(Context& ctx, const void* u, const void* R_uv, const void* PK_uv,
                     const void* Gen_uv, const void* m, size_t n, void* ok, void* workspace,
                     hipStream_t stream, const uint8_t* valid_in = nullptr) {
  const uint8_t *pu = (const uint8_t*)u, *pR = (const uint8_t*)R_uv, *pPK = (const uint8_t*)PK_uv,
                *pG = (const uint8_t*)Gen_uv, *pm = (const uint8_t*)m;
  uint8_t* pok = (uint8_t*)ok;
  return run_split(ctx, n, workspace, stream,
                   [=](size_t off, size_t cnt, const Workspace& w, hipStream_t s) {
    launch_challenge(false, pR + 64 * off, (const uint8_t*)nullptr, pm + 32 * off, cnt, w.c, w.valid, s,
                     valid_in ? valid_in + off : nullptr);
    launch_verify_var(pu + 32 * off, (const uint8_t*)w.c, pPK + 64 * off, pG + 64 * off, pR + 64 * off, (const uint8_t*)w.valid, cnt, pok + off, w.tables, s);
  });
}
int decompress_on(Context& ctx, const void* in, size_t in_stride, size_t n, void* out_uv, void* ok,
                  int accumulate, hipStream_t stream) {
  if (!in || !out_uv || !ok || in_stride < 32 || (in_stride & 15) || ((uintptr_t)in & 15))
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad pointer / stride (need 16-byte alignment)");
  launch_decompress((const uint8_t*)in, in_stride, n, (uint8_t*)out_uv, (uint8_t*)ok, accumulate, ctx.ts_cancel, ctx.ts_hash, stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
}  // namespace

int dsv_verify_single_dev(const void* u, const void* R_uv, const void* PK_uv, const void* m,
                          size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uv || !PK_uv || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  return verify_single_on(ctx, u, R_uv, PK_uv, m, n, ok, workspace, (hipStream_t)stream);
}

// ---- random-linear-combination fast accept in front of the per-signature path ----
// Per group of at most kRlcMaxGroup items: hash, then one aggregate test (k_rlc.hip: every key and
// nonce point in the prime-order subgroup AND the z-weighted sum of the equations is the identity,
// z_i secret, fresh per call).  Accepted: every well-formed item's verdict is `true`, as the
// reference's (error <= 2^-112).  Rejected — one wrong signature, one point with a small-order
// component or off the curve — the group goes through dsv_verify_single_dev's kernels and gets
// THEIR verdicts.  The call blocks on `stream` once per group (the decision is taken on the host).
extern "C++" {
namespace {
struct RlcStaged {  // a host call whose bucket pass over items [0, boundary) was enqueued while the rest was still on the bus
  ChaChaKey key;
  size_t boundary;
};
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
// scheme 0 single (R, PK), 1 double (R, R', PK, PK'), 2 var-generator (R, PK, Gen): unused pointers null
int verify_rlc_on(Context& ctx, int scheme, const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                  const void* PKp_uv, const void* Gen_uv, const void* m, size_t n, void* ok, void* workspace,
                  hipStream_t s, int window_bits, int* accepted, bool have_challenges = false,
                  const uint8_t* valid_in = nullptr, const RlcStaged* staged = nullptr) {
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

// second stage alone (c and valid already computed): lets callers time / profile the dominant
// kernel separately, and re-use one challenge for several key pairs
int dsv_verify_core_dev(const void* u, const void* c, const void* valid, const void* PK_uv,
                        const void* R_uv, int which, int accumulate, size_t n, void* ok,
                        void* workspace, void* stream) {
  if (n && (!u || !c || !valid || !PK_uv || !R_uv || !ok || !workspace || which < 0 || which > 1))
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  DSV_DEV_PROLOGUE(n, ok);
  Workspace w = carve(workspace, n);
  launch_verify_fixed(ctx, accumulate != 0, u, c, PK_uv, R_uv, which, valid, n, ok, w.tables,
                      (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// both equations of a double signature from a precomputed challenge, one launch
int dsv_verify_core_double_dev(const void* u, const void* c, const void* valid, const void* PK_uv,
                               const void* R_uv, const void* PKp_uv, const void* Rp_uv, size_t n,
                               void* ok, void* workspace, void* stream) {
  if (n && (!u || !c || !valid || !PK_uv || !R_uv || !PKp_uv || !Rp_uv || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  DSV_DEV_PROLOGUE(n, ok);
  Workspace w = carve(workspace, n);
  launch_verify_fixed_double(ctx, u, c, PK_uv, R_uv, PKp_uv, Rp_uv, valid, n, ok, w.tables,
                             (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

int dsv_verify_double_dev(const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                          const void* PKp_uv, const void* m, size_t n, void* ok, void* workspace,
                          void* stream) {
  if (n && (!u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  return verify_double_on(ctx, u, R_uv, Rp_uv, PK_uv, PKp_uv, m, n, ok, workspace, (hipStream_t)stream);
}

int dsv_verify_vargen_dev(const void* u, const void* R_uv, const void* PK_uv, const void* Gen_uv,
                          const void* m, size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uv || !PK_uv || !Gen_uv || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  return verify_vargen_on(ctx, u, R_uv, PK_uv, Gen_uv, m, n, ok, workspace, (hipStream_t)stream);
}

// ---- mixed batches: device-side split by kind ---------------------------------------------
namespace {
struct SplitScratch {
  u32* tile_counts;
  u32* totals;
};
size_t split_scratch_bytes(size_t n) {
  const size_t tiles = (n + kSplitTile - 1) / kSplitTile;
  return align_up(tiles * 8, 256) + 256;
}
SplitScratch carve_split(void* p, size_t n) {
  const size_t tiles = (n + kSplitTile - 1) / kSplitTile;
  SplitScratch s;
  s.tile_counts = static_cast<u32*>(p);
  s.totals = reinterpret_cast<u32*>(static_cast<uint8_t*>(p) + align_up(tiles * 8, 256));
  return s;
}
int split_on(const void* kinds, size_t n, void* idx_single, size_t cap_single, void* idx_double,
             size_t cap_double, void* scratch, hipStream_t s) {
  if ((uintptr_t)kinds & 15) return fail(DSV_ERR_INVALID_ARGUMENT, "kinds must be 16-byte aligned");
  SplitScratch sc = carve_split(scratch, n);
  launch_split_kinds((const uint8_t*)kinds, n, sc.tile_counts, sc.totals, (u32*)idx_single, cap_single,
                     (u32*)idx_double, cap_double, s);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// limit (device u32, may be null): only min(count, *limit) index entries are dereferenced
int gather_on(const void* src, size_t src_rows, size_t row_bytes, const void* idx, size_t count,
              const void* limit, void* dst, hipStream_t s) {
  if (row_bytes == 0 || (row_bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
    return fail(DSV_ERR_INVALID_ARGUMENT, "rows must be multiples of 16 bytes, 16-byte aligned");
  if (count == 0) return DSV_OK;
  launch_gather_rows(src, src_rows, (u32)(row_bytes / 16), (const u32*)idx, count, (const u32*)limit, dst, s);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
}  // namespace

size_t dsv_split_scratch_bytes(size_t n) { return split_scratch_bytes(n); }

int dsv_split_kinds_dev(const void* kinds, size_t n, void* idx_single, size_t cap_single,
                        void* idx_double, size_t cap_double, void* scratch, void* stream) {
  if (n && (!kinds || !idx_single || !idx_double || !scratch))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, scratch);
  return split_on(kinds, n, idx_single, cap_single, idx_double, cap_double, scratch, (hipStream_t)stream);
}
int dsv_gather_rows_dev(const void* src, size_t src_rows, size_t row_bytes, const void* idx,
                        size_t count, const void* count_limit, void* dst, void* stream) {
  if (count && (!src || !idx || !dst)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(count, dst);
  return gather_on(src, src_rows, row_bytes, idx, count, count_limit, dst, (hipStream_t)stream);
}
int dsv_scatter_verdicts_dev(const void* src, const void* idx, size_t count, const void* count_limit,
                             void* dst, size_t dst_len, void* stream) {
  if (count && (!src || !idx || !dst)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(count, dst);
  launch_scatter_bytes((const uint8_t*)src, (const u32*)idx, count, (const u32*)count_limit, (uint8_t*)dst,
                       dst_len, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

size_t dsv_mixed_workspace_bytes(size_t n) {
  // split scratch | idx_single[n] | idx_double[n] | compacted rows (<= 320 B per item) |
  // per-kind verdicts | verify workspace
  return split_scratch_bytes(n) + 2 * align_up(n * 4, 256) + 6 * align_up(n * 64, 256) +
         2 * align_up(n, 256) + align_up(dsv_workspace_bytes(n), 256) + 256;
}

// One batch holding single (kind 0) and double (kind 1) signatures in any interleaving, as a
// structure of arrays over ALL n items (Rp_uv / PKp_uv rows of single items are ignored).
// n_double = number of kind-1 items (the caller knows its batch); every other item must be kind 0.
extern "C++" {
namespace {
// fast: both kinds' groups through the batch fast accept (verify_rlc_on; blocks on the stream); *accepted =
// every group of both kinds was decided by its aggregate
int verify_mixed_dev(const void* kinds, const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                     const void* PKp_uv, const void* m, size_t n, size_t n_double, void* ok, void* workspace,
                     void* stream, bool fast, int* accepted) {
  if (accepted) *accepted = 0;
  if (n && (!kinds || !u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  if (n_double > n) return fail(DSV_ERR_INVALID_ARGUMENT, "n_double exceeds n");
  DSV_DEV_PROLOGUE(n, ok);
  hipStream_t s = (hipStream_t)stream;
  const size_t ns = n - n_double, nd = n_double;
  Stager st(static_cast<uint8_t*>(workspace));
  void* scratch = st.take(split_scratch_bytes(n));
  u32* idx_s = reinterpret_cast<u32*>(st.take(n * 4));
  u32* idx_d = reinterpret_cast<u32*>(st.take(n * 4));
  uint8_t *cu = st.take(n * 32), *cm = st.take(n * 32);   // singles first, doubles behind them
  uint8_t *cR = st.take(n * 64), *cPK = st.take(n * 64);
  uint8_t *cRp = st.take(n * 64), *cPKp = st.take(n * 64);  // doubles only
  uint8_t *oks = st.take(n), *okd = st.take(n);
  void* vws = st.take(fast ? dsv_rlc_workspace_bytes(n, 0) : dsv_workspace_bytes(n));
  HIP_TRY(hipMemsetAsync(ok, 0, n, s));  // items of an invalid kind keep verdict 0
  if (int r = split_on(kinds, n, idx_s, ns, idx_d, nd, scratch, s)) return r;
  // Only index entries the split really wrote are dereferenced: every gather / scatter is bounded
  // on the device by the split's own totals (and skips an index >= n), whatever the caller
  // declared.  With a wrong n_double the compacted rows are partly stale workspace bytes — the
  // verify kernels take any bytes (out-of-contract inputs never fault) and k_mixed_check zeroes
  // the whole verdict vector at the end.
  const u32* totals = carve_split(scratch, n).totals;
  struct Col { const void* src; size_t bytes; uint8_t* dst; };
  const Col single_cols[4] = {{u, 32, cu}, {m, 32, cm}, {R_uv, 64, cR}, {PK_uv, 64, cPK}};
  for (const Col& c : single_cols)
    if (int r = gather_on(c.src, n, c.bytes, idx_s, ns, totals, c.dst, s)) return r;
  const Col double_cols[6] = {{u, 32, cu + ns * 32},      {m, 32, cm + ns * 32},
                              {R_uv, 64, cR + ns * 64},   {PK_uv, 64, cPK + ns * 64},
                              {Rp_uv, 64, cRp},           {PKp_uv, 64, cPKp}};
  for (const Col& c : double_cols)
    if (int r = gather_on(c.src, n, c.bytes, idx_d, nd, totals + 1, c.dst, s)) return r;
  int acc_s = 1, acc_d = 1;
  if (ns) {
    if (int r = fast ? verify_rlc_on(ctx, 0, cu, cR, nullptr, cPK, nullptr, nullptr, cm, ns, oks, vws, s, 0, &acc_s)
                     : verify_single_on(ctx, cu, cR, cPK, cm, ns, oks, vws, s))
      return r;
    launch_scatter_bytes(oks, idx_s, ns, totals, (uint8_t*)ok, n, s);
  }
  if (nd) {
    if (int r = fast ? verify_rlc_on(ctx, 1, cu + ns * 32, cR + ns * 64, cRp, cPK + ns * 64, cPKp, nullptr,
                                     cm + ns * 32, nd, okd, vws, s, 0, &acc_d)
                     : verify_double_on(ctx, cu + ns * 32, cR + ns * 64, cRp, cPK + ns * 64, cPKp, cm + ns * 32, nd,
                                        okd, vws, s))
      return r;
    launch_scatter_bytes(okd, idx_d, nd, totals + 1, (uint8_t*)ok, n, s);
  }
  launch_mixed_check(totals, (u32)ns, (u32)nd, (uint8_t*)ok, n, s);
  HIP_TRY(hipGetLastError());
  if (fast && accepted) *accepted = (acc_s && acc_d) ? 1 : 0;
  return DSV_OK;
}
}  // namespace
}  // extern "C++"
int dsv_verify_mixed_dev(const void* kinds, const void* u, const void* R_uv, const void* Rp_uv,
                         const void* PK_uv, const void* PKp_uv, const void* m, size_t n,
                         size_t n_double, void* ok, void* workspace, void* stream) {
  return verify_mixed_dev(kinds, u, R_uv, Rp_uv, PK_uv, PKp_uv, m, n, n_double, ok, workspace, stream, false, nullptr);
}
// the same with each kind's items through the batch fast accept (a wrong n_double: every verdict 0, as above)
size_t dsv_mixed_rlc_workspace_bytes(size_t n) {
  return dsv_mixed_workspace_bytes(n) - align_up(dsv_workspace_bytes(n), 256) + align_up(dsv_rlc_workspace_bytes(n, 0), 256);
}
int dsv_verify_mixed_rlc_dev(const void* kinds, const void* u, const void* R_uv, const void* Rp_uv,
                             const void* PK_uv, const void* PKp_uv, const void* m, size_t n, size_t n_double,
                             void* ok, void* workspace, void* stream, int* accepted) {
  return verify_mixed_dev(kinds, u, R_uv, Rp_uv, PK_uv, PKp_uv, m, n, n_double, ok, workspace, stream, true, accepted);
}

// ---- host-pointer entry points ----------------------------------------------------------

#define H2D(dst, src, bytes) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, 0))
#define D2H(dst, src, bytes) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, 0))

extern "C++" {
namespace {
// Chunked host path shared by the verify entry points.
//   ins[k] = {host array, bytes per item[, stride]}; launch(dev_ptrs, count, dok, ws, extra, stream)
//   enqueues the kernels for `count` items on `stream` (it is handed sub-batches, not chunks).
// Per chunk c (slot c % kPipeSlots): the copy threads gather the caller's arrays (pageable in general,
// or one field out of every typed object) into the slot's pinned staging, ONE asynchronous DMA moves
// the block to the device, the chunk's sub-batches of 2^16 items run, the verdict bytes come back into
// the pinned block; they are handed to the caller when the slot is recycled.  While the GPU works
// on chunk c the host is already gathering chunk c + 1.
//
// Streams (r04): FOUR per device for the whole pipeline —
//   pipe_in      every chunk's transfer to the device
//   pipe_lane[2] the two compute lanes: sub-batch p of the CALL (not of the chunk) runs on lane p & 1
//                with everything it needs — normalisation / decompression / limb conversion of ITS
//                items, hash, verify, validity AND — as one in-order chain, so a lane never waits for
//                another chunk's preprocessing and two sub-batches are co-resident at any time, as in
//                the device-resident entry points
//   pipe_out     every chunk's verdicts back to the host
// chained by events (transfer done -> lanes; lanes done -> verdicts out -> slot free).  r01 - r03 gave
// every slot its own stream plus a pair of sub-batch streams: nine streams on the four hardware
// queues ROCm multiplexes streams onto by default, i.e. kernels of one chunk queued behind another
// chunk's on the same hardware queue although nothing ordered them — the wire path, with the most
// kernels per chunk, lost 22 % against the device-resident rate and gained 12 - 15 % from
// GPU_MAX_HW_QUEUES=8 alone (profiles/r04/host_pipeline_streams.txt).
struct HostIn {
  const uint8_t* p;
  size_t bytes;       // per item
  size_t stride = 0;  // distance between items in the caller's memory; 0 = `bytes` (a dense array)
};
// Chunk sizes double from 2^15 up to 2^18 items: the GPU starts after ~0.3 ms of staging, every
// gather runs under the previous (half as long) chunk's kernels, and from the fourth chunk on the
// transfers are long enough to run near the link rate (r03, same box, 2^20 items: chunks
// capped at 2^17 as in r02: 68.0 M/s affine / 59.6 projective; 2^18: 72.4 / 63.6; two, three or
// four slots: equal; profiles/r03/host_paths.txt; r04: first chunk 2^16 / 2^17, a merged last chunk:
// equal, profiles/r04/ab_host_chunk_policy.txt)

std::atomic<int> g_host_threads{0};  // dsv_set_host_threads; 0 = $DSV_HOST_THREADS, else 4
inline int clamp_host_threads(int v) {
  const int hw = (int)std::thread::hardware_concurrency();
  if (hw > 0 && v > hw) v = hw;
  return v < 1 ? 1 : (v > 16 ? 16 : v);
}
inline int host_copy_threads() {
  static const int from_env = [] {
    const char* e = getenv("DSV_HOST_THREADS");
    return clamp_host_threads(e ? atoi(e) : 4);
  }();
  const int set = g_host_threads.load(std::memory_order_relaxed);
  return set > 0 ? set : from_env;
}

// What a chunk looks like to its sub-batches: arrays in the argument order of the entry point that
// handles a sub-batch, element k of item i at p[k] + i * bytes[k]; `valid` (may be null): per-item
// bytes that a chunk-level preprocessing wants AND-ed into the verdicts.
struct Staged {
  const uint8_t* p[8] = {};
  size_t bytes[8] = {};
  const uint8_t* valid = nullptr;
};
struct NoPrep {};
//   prep(dev_ptrs, count, scratch, stream, staged)  [or NoPrep{}: the chunk is used as transferred]:
//     preprocessing of the WHOLE chunk — projective -> affine, limb conversion: kernels with one
//     inversion chain per 8 - 16 items, i.e. few waves — enqueued once per chunk on the lane of the
//     chunk's first sub-batch (the other lane waits for its event), results in the slot's own
//     scratch (prep_item_bytes per item).  Per SUB-BATCH these kernels would be 128 waves each in
//     front of every hash: 16 low-occupancy phases per 2^20 items instead of 4.
//   part(staged, offset, count, dok, ws, extra, stream): one sub-batch; `extra`: scratch of
//     extra_item_bytes per item behind the lane's verify workspace (the wire path decompresses per
//     sub-batch: full-occupancy kernels, no reason to serialise a chunk's worth on one lane).
template <size_t NIN, class Prep, class Part>
int run_pipelined(Context& ctx, const HostIn (&ins)[NIN], uint8_t* ok, size_t n, size_t prep_item_bytes,
                  size_t extra_item_bytes, Prep prep, Part part) {
  constexpr bool has_prep = !std::is_same<Prep, NoPrep>::value;
  const bool small = n <= kPipeSmallCall;  // transfer, kernels and verdicts on ONE stream, a work area of its own
  TurnTicket turn(ctx.pipe_sync);          // (on an error path its destructor still passes the turn on, in order)
  PipeLease lease(ctx.pipe_sync, small ? nullptr : &turn);  // blocks while kPipes calls are in flight on this device
  Pipe& pipe = ctx.pipes[lease.index];
  if (!ctx.ready.load(std::memory_order_acquire))
    return fail(DSV_ERR_NOT_INITIALIZED, "device %d was shut down", ctx.device);
  DSV_ON_DEVICE(ctx);
  // nobody else in flight: the GPU is idle, so start small and grow; behind another call (it holds its
  // pipe until its last verdicts are out): full chunks at once
  const size_t part_cap = kSplitItems;
  const std::vector<size_t> chunks = plan_chunks(ctx.plan, n, small || lease.alone, part_cap);
  const size_t nchunks = chunks.size();
  size_t chunk = 0;  // the largest chunk: slot capacity
  for (size_t c : chunks) chunk = c > chunk ? c : chunk;
  // slot capacity: offsets of the LARGEST chunk (a shorter chunk packs its arrays tighter, see below)
  size_t cap_off = 0;
  for (size_t k = 0; k < NIN; k++) cap_off += align_up(chunk * ins[k].bytes, 256);
  const size_t host_need = cap_off + align_up(chunk, 256);
  // chunks in flight: three (r05, one-shot calls of 2^16-item chunks: six slots are ~1 ms per 2^20
  // SLOWER than three — with the host far ahead every chunk's preprocessing kernel is resident at the
  // lanes' kernel boundaries; profiles/r05/ab_chunk_plans.txt)
  const int kSlots = ctx.pipe_slots ? ctx.pipe_slots : 3;
  const int nslots = nchunks < (size_t)kSlots ? (int)nchunks : kSlots;
  const size_t prep_need = has_prep ? chunk * prep_item_bytes + 64 * 256 : 0;
  // a sub-batch: at most kSplitItems items (run_split's unit), or the whole chunk when the split is off
  // or the call is one small chunk; each compute lane owns the workspace + scratch of one sub-batch
  const bool one_part = !ctx.split || small;
  const size_t part_max = one_part ? chunk : (chunk < kSplitItems ? chunk : kSplitItems);
  const size_t ws_bytes = align_up(dsv_workspace_bytes(part_max), 256);
  const size_t work_need = ws_bytes + part_max * extra_item_bytes + 16 * 256;
  {
    std::lock_guard<std::mutex> enq(ctx.enq_mu);
    if (int r = ensure_pipe_streams(ctx)) return r;
    if (small) {
      if (int r = ensure_pipe_work(ctx, 2, work_need)) return r;
    } else {
      for (int k = 0; k < 2; k++)
        if (int r = ensure_pipe_work(ctx, k, work_need)) return r;
    }
  }
  for (int sl = 0; sl < nslots; sl++)
    if (int r = ensure_pipe_slot(pipe.slot[sl], host_need, host_need, prep_need)) return r;
  // Whole-chunk preprocessing (normalisation / limb conversion: few waves, one inversion chain each) runs
  // ON the lane that takes the chunk's first sub-batch, in that lane's order; the other lane waits for its
  // event.  r05 tried it on a high-priority stream of its own (DSV_PIPE_PREP_STREAM=1): the two lane
  // kernels resident at any time fill every wave slot of the chip exactly (1024 + 1024 waves, two slots on
  // each of 1024 SIMDs), so a third kernel's waves only ever start where a lane kernel has just ended, the
  // lane's next kernel starts that many waves short, and an in-order lane cannot go on before those
  // stragglers are done: a small kernel beside the lanes costs ~10 x its work (profiles/r05/squat_probe.txt).
  // With random z the side stream still came out 0.7 ms ahead for a one-shot call (its inversions are long
  // enough to be worth taking off the lane), with z = 1 everywhere — deserialised keys and signatures — it
  // fell into a mode 2 - 3 ms slower every other call; in a lane's order the preprocessing gets that lane's
  // own slots, nothing depends on when the host issued it, and two calls in flight are faster as well
  // (0.95 against 0.92 x the device-resident rate).  profiles/r05/ab_prep_placement.txt.
  const bool pre_stream = has_prep && !small && ctx.prep_stream;
  // what a slot currently holds: a chunk that is staged (gathered + on its way to the device), then
  // enqueued (its kernels and verdict copy are in the streams), then drained (verdicts delivered)
  struct Held {
    size_t first = 0, cnt = 0, ok_off = 0;
    size_t in_off[NIN + 1] = {};
    bool enqueued = false;
    int parts_on[2] = {0, 0};  // its sub-batches per compute lane
  } held[kPipeSlots];
  int slot_of[kPipeSlots + 1] = {};  // ring: slot of chunk c at slot_of[c % (kPipeSlots + 1)] (staged .. enqueued)
  auto drain = [&](int sl) -> int {
    Held& h = held[sl];
    if (!h.cnt) return DSV_OK;
    if (small) HIP_TRY(hipStreamSynchronize(ctx.pipe_small));  // (the one-stream path of a small call)
    else HIP_TRY(hipEventSynchronize(pipe.slot[sl].ev_done));
    memcpy(ok + h.first, pipe.slot[sl].host + h.ok_off, h.cnt);
    h.cnt = 0;
    if (h.parts_on[0] | h.parts_on[1]) {
      std::lock_guard<std::mutex> enq(ctx.enq_mu);
      for (int k = 0; k < 2; k++) {
        ctx.lane_load[k] -= h.parts_on[k];
        h.parts_on[k] = 0;
      }
    }
    return DSV_OK;
  };
  // A slot for the next chunk: a free one, else one whose chunk is DONE — in completion order, not in
  // chunk order: when one lane runs slower than the other (its kernels shared the SIMDs with the
  // preprocessing, or simply the higher-priority lane's), the chunk the host would wait for in order is the
  // slow lane's while the fast lane's later chunk finished long ago — and the fast lane starves behind a
  // slot that is free (r05: host calls whose preprocessing is quick, all points with z = 1, fell into that
  // mode every other call: 15.8 against 18.5 ms).  Else wait for the oldest enqueued chunk.
  auto free_slot = [&](int& out) -> int {
    for (int sl = 0; sl < nslots; sl++)
      if (!held[sl].cnt) {
        out = sl;
        return DSV_OK;
      }
    if (!small)
      for (int sl = 0; sl < nslots; sl++)
        if (held[sl].enqueued) {
          const hipError_t q = hipEventQuery(pipe.slot[sl].ev_done);
          if (q == hipSuccess) {
            out = sl;
            return drain(sl);
          }
          if (q != hipErrorNotReady) return fail(DSV_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        }
    int oldest = -1;
    for (int sl = 0; sl < nslots; sl++)
      if (held[sl].enqueued && (oldest < 0 || held[sl].first < held[oldest].first)) oldest = sl;
    if (oldest < 0) return fail(DSV_ERR_HIP, "host pipeline: no slot to wait for");  // (cannot happen: see the loop)
    out = oldest;
    return drain(oldest);
  };
  // an error half-way: nothing of this call may still be in flight when the caller's buffers go away
  // (the streams are shared: this waits for the other call in flight as well — errors are rare)
  auto bail = [&](int rc) {
    const std::string why = g_err;
    (void)hipStreamSynchronize(ctx.pipe_in);
    (void)hipStreamSynchronize(ctx.pipe_pre);
    (void)hipStreamSynchronize(ctx.pipe_small);
    for (int k = 0; k < 2; k++) (void)hipStreamSynchronize(ctx.pipe_lane[k]);
    (void)hipStreamSynchronize(ctx.pipe_out);
    {  // what this call had outstanding on the lanes is gone
      std::lock_guard<std::mutex> enq(ctx.enq_mu);
      for (auto& h : held)
        for (int k = 0; k < 2; k++) {
          ctx.lane_load[k] -= h.parts_on[k];
          h.parts_on[k] = 0;
        }
    }
    g_err = why;
    return rc;
  };
  struct PartScope {  // a short last sub-batch of a long call builds its window tables in its own kernel
    explicit PartScope(bool on) { t_pipeline_part = on; }
    ~PartScope() { t_pipeline_part = false; }
  } part_scope(nchunks > 1);
  // DSV_PIPE_TRACE=1: per call, where the host thread's time went (stderr)
  static const bool trace = getenv("DSV_PIPE_TRACE") != nullptr;
  double t_drain = 0, t_copy = 0, t_enq = 0, t_turn = 0;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();

  // ---- stage chunk c into its slot: gather the caller's arrays into pinned memory, start the transfer ----
  auto stage = [&](size_t c, size_t first) -> int {
    const size_t cnt = chunks[c];
    const double t0 = now();
    int sl = -1;
    if (int r = free_slot(sl)) return r;  // a slot that is free (again), its verdicts delivered
    slot_of[c % (size_t)(kPipeSlots + 1)] = sl;
    PipeSlot& slot = pipe.slot[sl];
    Held& h = held[sl];
    const double t1 = now();
    t_drain += t1 - t0;
    uint8_t* host = slot.host;
    h.in_off[0] = 0;  // offsets inside the slot for THIS chunk, the same on both sides
    for (size_t k = 0; k < NIN; k++) h.in_off[k + 1] = h.in_off[k] + align_up(cnt * ins[k].bytes, 256);
    h.ok_off = h.in_off[NIN];
    size_t bytes = 0;
    for (size_t k = 0; k < NIN; k++) bytes += cnt * ins[k].bytes;
    const int T = bytes >= ((size_t)1 << 20) ? host_copy_threads() : 1;
    pipe.copiers.run(T, [&](int t, int nt) {
      for (size_t k = 0; k < NIN; k++) {
        if (ins[k].stride && ins[k].stride != ins[k].bytes) {  // one field out of every object
          const size_t lo = cnt * (size_t)t / (size_t)nt, hi = cnt * (size_t)(t + 1) / (size_t)nt;
          copy_strided(host + h.in_off[k] + lo * ins[k].bytes, ins[k].p + (first + lo) * ins[k].stride,
                       ins[k].stride, ins[k].bytes, hi - lo);
          continue;
        }
        const size_t len = cnt * ins[k].bytes;
        const size_t lo = len * (size_t)t / (size_t)nt / 64 * 64;
        const size_t hi = t + 1 == nt ? len : len * (size_t)(t + 1) / (size_t)nt / 64 * 64;
        memcpy(host + h.in_off[k] + lo, ins[k].p + first * ins[k].bytes + lo, hi - lo);
      }
    });
    t_copy += now() - t1;
    h.first = first;
    h.cnt = cnt;
    h.enqueued = false;
    // the input block is contiguous on both sides (pad bytes ride along); a single small chunk goes
    // with its kernels and verdicts on ONE stream (no event hop on the latency path of a
    // 1024-signature call), everything else on the shared transfer stream
    hipStream_t s_in = small ? ctx.pipe_small : ctx.pipe_in;
    std::lock_guard<std::mutex> enq(ctx.enq_mu);  // (record + later waits on ev_in stay paired)
    if (hipMemcpyAsync(slot.stage, host, h.in_off[NIN - 1] + cnt * ins[NIN - 1].bytes, hipMemcpyHostToDevice, s_in) != hipSuccess ||
        (!small && hipEventRecord(slot.ev_in, s_in) != hipSuccess))
      return fail(DSV_ERR_HIP, "transfer to the device failed: %s", hipGetErrorString(hipGetLastError()));
    return DSV_OK;
  };

  // ---- enqueue chunk c's preprocessing, sub-batches and verdict copy ----
  // From here to the verdict copy the chunk is enqueued as one unit: a lane's work area belongs to one
  // sub-batch at a time (the lanes are in-order, so enqueue order = use order).
  auto enqueue = [&](size_t c) -> int {
    const int sl = slot_of[c % (size_t)(kPipeSlots + 1)];
    PipeSlot& slot = pipe.slot[sl];
    Held& h = held[sl];
    const size_t cnt = h.cnt;
    t_chunk_first = h.first;
    const double t2 = now();
    uint8_t* dev = slot.stage;
    uint8_t* dok = dev + h.ok_off;
    std::lock_guard<std::mutex> enq(ctx.enq_mu);
    hipStream_t s_out = small ? ctx.pipe_small : ctx.pipe_out;
    bool used[2] = {false, false};
    // first use of a lane by this chunk: its inputs must have arrived (and, with the preprocessing on
    // its own stream, been preprocessed: ev_pre implies ev_in)
    auto lane_for = [&](int k) -> int {
      if (!small && !used[k]) {
        if (hipStreamWaitEvent(ctx.pipe_lane[k], pre_stream ? slot.ev_pre : slot.ev_in, 0) != hipSuccess)
          return fail(DSV_ERR_HIP, "hipStreamWaitEvent failed");
        used[k] = true;
      }
      return DSV_OK;
    };
    // the lane with fewer sub-batches outstanding; a tie alternates
    auto next_lane = [&]() -> int {
      if (ctx.lane_load[0] != ctx.lane_load[1]) return ctx.lane_load[0] < ctx.lane_load[1] ? 0 : 1;
      return (int)(ctx.pipe_parts & 1);
    };
    Staged sg;
    int prep_lane = -1;
    if constexpr (has_prep) {
      hipStream_t sp;
      if (small) {
        sp = ctx.pipe_small;
      } else {
        if (!pre_stream) prep_lane = next_lane();  // the lane of the chunk's first sub-batch (nothing advanced)
        sp = pre_stream ? ctx.pipe_pre : ctx.pipe_lane[prep_lane];
        if (hipStreamWaitEvent(sp, slot.ev_in, 0) != hipSuccess) return fail(DSV_ERR_HIP, "hipStreamWaitEvent failed");
      }
      const void* dptr[NIN];
      for (size_t j = 0; j < NIN; j++) dptr[j] = dev + h.in_off[j];
      Stager scratch(slot.prep);
      if (int r = prep(dptr, cnt, scratch, sp, sg)) return r;
      if (!small && hipEventRecord(slot.ev_pre, sp) != hipSuccess) return fail(DSV_ERR_HIP, "hipEventRecord failed");
      if (prep_lane >= 0) used[prep_lane] = true;  // (in order behind the transfer already)
    } else {
      (void)prep;
      for (size_t j = 0; j < NIN; j++) {
        sg.p[j] = dev + h.in_off[j];
        sg.bytes[j] = ins[j].bytes;
      }
    }
    bool waited_pre[2] = {false, false};
    size_t part_items = 0;
    (void)plan_parts(cnt, one_part, part_cap, part_items);
    for (size_t off = 0; off < cnt;) {
      const size_t pc = cnt - off < part_items ? cnt - off : part_items;
      const int k = small ? 0 : next_lane();
      if (!small) {
        ctx.pipe_parts++;
        ctx.lane_load[k]++;
        h.parts_on[k]++;
      }
      hipStream_t st = small ? ctx.pipe_small : ctx.pipe_lane[k];
      if (int r = lane_for(k)) return r;
      if (prep_lane >= 0 && k != prep_lane && !waited_pre[k]) {
        if (hipStreamWaitEvent(st, slot.ev_pre, 0) != hipSuccess) return fail(DSV_ERR_HIP, "hipStreamWaitEvent failed");
        waited_pre[k] = true;
      }
      uint8_t* ws = ctx.pipe_work[small ? 2 : k];
      Stager extra(ws + ws_bytes);
      if (int r = part(sg, off, pc, dok + off, ws, extra, st)) return r;
      off += pc;
    }
    if (!small) {
      for (int k = 0; k < 2; k++)
        if (used[k]) {
          if (hipEventRecord(slot.ev_lane[k], ctx.pipe_lane[k]) != hipSuccess ||
              hipStreamWaitEvent(s_out, slot.ev_lane[k], 0) != hipSuccess)
            return fail(DSV_ERR_HIP, "event record / wait failed");
        }
    }
    if (hipMemcpyAsync(slot.host + h.ok_off, dok, cnt, hipMemcpyDeviceToHost, s_out) != hipSuccess ||
        (!small && hipEventRecord(slot.ev_done, s_out) != hipSuccess))
      return fail(DSV_ERR_HIP, "verdict copy failed: %s", hipGetErrorString(hipGetLastError()));
    h.enqueued = true;
    t_enq += now() - t2;
    return DSV_OK;
  };

  size_t staged = 0, enqueued = 0, first = 0;
  while (enqueued < nchunks) {
    // stage the next chunk if its slot can be had: free, or holding a chunk whose kernels are enqueued
    // (drain then waits for them) — never one that is itself still waiting for the turn
    if (staged < nchunks && staged - enqueued < (size_t)kSlots) {
      if (int r = stage(staged, first)) return bail(r);
      first += chunks[staged];
      staged++;
    }
    if (!small && !turn.held) {
      // not our turn yet: keep staging while slots are free, then wait
      if (!turn.try_acquire()) {
        if (staged < nchunks && staged - enqueued < (size_t)kSlots) continue;
        const double tw = now();
        turn.acquire();
        t_turn += now() - tw;
      }
    }
    while (enqueued < staged) {
      if (int r = enqueue(enqueued)) return bail(r);
      enqueued++;
    }
    if (enqueued == nchunks) turn.release();  // the next call's chunks go behind ours
  }
  const double t3 = now();
  for (int sl = 0; sl < nslots; sl++)
    if (int r = drain(sl)) return bail(r);
  if (trace)
    fprintf(stderr, "[dsv pipe] n=%zu chunks=%zu total %.2f ms: gather %.2f, enqueue %.2f, waiting for slots %.2f, for the turn %.2f, final drain %.2f\n",
            n, nchunks, now() - t_begin, t_copy, t_enq, t_drain, t_turn, now() - t3);
  return DSV_OK;
}
}  // namespace
}  // extern "C++"


#define DSV_HOST_PROLOGUE(n)                                  \
  if (int r_ = check_n(n)) return r_;                         \
  if ((n) == 0) return DSV_OK;                                \
  Context* ctxp_ = nullptr;                                   \
  if (int r_ = host_context(ctxp_)) return r_;                \
  Context& ctx = *ctxp_
// small host calls: one lock, the context's staging buffer, the device's null stream
#define DSV_HOST_LOCK()                                                                \
  std::lock_guard<std::mutex> lk(ctx.mu);                                              \
  if (!ctx.ready.load(std::memory_order_acquire))                                      \
    return fail(DSV_ERR_NOT_INITIALIZED, "device %d was shut down", ctx.device);       \
  DSV_ON_DEVICE(ctx)

extern "C++" {
namespace {
// one sub-batch of a staged chunk through the affine path of its scheme (0 single, 1 double, 2 vargen)
int part_verify(Context& ctx, int kind, const Staged& g, size_t off, size_t cnt, void* dok, void* ws,
                hipStream_t st) {
  auto at = [&](int k) { return g.p[k] + off * g.bytes[k]; };
  int rc;
  const uint8_t* vin = g.valid ? g.valid + off : nullptr;  // (the chunk-level preprocessing's verdict on the items)
  if (kind == 0) rc = verify_single_on(ctx, at(0), at(1), at(2), at(3), cnt, dok, ws, st, vin);
  else if (kind == 1) rc = verify_double_on(ctx, at(0), at(1), at(2), at(3), at(4), at(5), cnt, dok, ws, st, vin);
  else rc = verify_vargen_on(ctx, at(0), at(1), at(2), at(3), at(4), cnt, dok, ws, st, vin);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
#define DSV_PART(kind_)                                                                                     \
  [=](const Staged& g, size_t off, size_t cnt, void* dok, void* ws, Stager&, hipStream_t st) {              \
    return part_verify(*cp, kind_, g, off, cnt, dok, ws, st);                                               \
  }
int verify_single_host(Context& ctx, const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                       const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[4] = {{u, 32}, {R_uv, 64}, {PK_uv, 64}, {m, 32}};
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, 0, 0, NoPrep{}, DSV_PART(0));
}
int verify_double_host(Context& ctx, const uint8_t* u, const uint8_t* R_uv, const uint8_t* Rp_uv,
                       const uint8_t* PK_uv, const uint8_t* PKp_uv, const uint8_t* m, size_t n,
                       uint8_t* ok) {
  const HostIn ins[6] = {{u, 32}, {R_uv, 64}, {Rp_uv, 64}, {PK_uv, 64}, {PKp_uv, 64}, {m, 32}};
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, 0, 0, NoPrep{}, DSV_PART(1));
}
int verify_vargen_host(Context& ctx, const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                       const uint8_t* Gen_uv, const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[5] = {{u, 32}, {R_uv, 64}, {PK_uv, 64}, {Gen_uv, 64}, {m, 32}};
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, 0, 0, NoPrep{}, DSV_PART(2));
}

// One host batch over every initialised device: contiguous shards, one host thread per device
// (the calling thread takes the first shard), no collective — each context stages, computes and
// returns its own slice of ok[].  part(ctx, offset, count) -> dsv_status.
template <class Part>
int run_multi(size_t n, Part part) {
  Context* devs[kMaxDevices];
  int nd = 0;
  for (int d = 0; d < kMaxDevices; d++)
    if (g_ctx[d].ready.load(std::memory_order_acquire)) devs[nd++] = &g_ctx[d];
  if (nd == 0) return fail(DSV_ERR_NOT_INITIALIZED, "dsv_init() has not been called");
  // DSV_MULTI_SHARDS=k (read per call): at least k shards, wrapping over the devices — lets a
  // one-GPU box exercise the sharding, the worker threads and their error path (shards of one
  // device then serialise on that device's lock)
  if (const char* e = getenv("DSV_MULTI_SHARDS")) {
    const int want = atoi(e);
    const int have = nd;
    while (nd < want && nd < kMaxDevices) {
      devs[nd] = devs[nd % have];
      nd++;
    }
  }
  if (nd == 1 || n < (size_t)nd * 1024) return part(*devs[0], (size_t)0, n);
  int rc[kMaxDevices] = {};
  std::string msg[kMaxDevices];
  std::vector<std::thread> th;
  auto work = [&](int k) {
    const size_t lo = n * (size_t)k / (size_t)nd, hi = n * (size_t)(k + 1) / (size_t)nd;
    rc[k] = hi > lo ? part(*devs[k], lo, hi - lo) : (int)DSV_OK;
    if (rc[k]) msg[k] = g_err;  // the error text lives in the worker's thread-local
  };
  for (int k = 1; k < nd; k++) th.emplace_back(work, k);
  work(0);
  for (auto& t : th) t.join();
  for (int k = 0; k < nd; k++)
    if (rc[k]) return fail(rc[k], "device %d: %s", devs[k]->device, msg[k].c_str());
  return DSV_OK;
}
}  // namespace
}  // extern "C++"

int dsv_verify_single(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                      const uint8_t* m, size_t n, uint8_t* ok) {
  if (n && (!u || !R_uv || !PK_uv || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_single_host(ctx, u, R_uv, PK_uv, m, n, ok);
}
int dsv_verify_double(const uint8_t* u, const uint8_t* R_uv, const uint8_t* Rp_uv,
                      const uint8_t* PK_uv, const uint8_t* PKp_uv, const uint8_t* m, size_t n,
                      uint8_t* ok) {
  if (n && (!u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_double_host(ctx, u, R_uv, Rp_uv, PK_uv, PKp_uv, m, n, ok);
}
int dsv_verify_vargen(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                      const uint8_t* Gen_uv, const uint8_t* m, size_t n, uint8_t* ok) {
  if (n && (!u || !R_uv || !PK_uv || !Gen_uv || !m || !ok))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_vargen_host(ctx, u, R_uv, PK_uv, Gen_uv, m, n, ok);
}

// ---- the same over ALL initialised devices (what a Rust verify_batch on an 8-GPU node calls) ----
int dsv_verify_single_multi(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                            const uint8_t* m, size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uv || !PK_uv || !m || !ok) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_single_host(ctx, u + 32 * off, R_uv + 64 * off, PK_uv + 64 * off, m + 32 * off, cnt,
                              ok + off);
  });
}
int dsv_verify_double_multi(const uint8_t* u, const uint8_t* R_uv, const uint8_t* Rp_uv,
                            const uint8_t* PK_uv, const uint8_t* PKp_uv, const uint8_t* m, size_t n,
                            uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uv || !Rp_uv || !PK_uv || !PKp_uv || !m || !ok)
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_double_host(ctx, u + 32 * off, R_uv + 64 * off, Rp_uv + 64 * off, PK_uv + 64 * off,
                              PKp_uv + 64 * off, m + 32 * off, cnt, ok + off);
  });
}
int dsv_verify_vargen_multi(const uint8_t* u, const uint8_t* R_uv, const uint8_t* PK_uv,
                            const uint8_t* Gen_uv, const uint8_t* m, size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uv || !PK_uv || !Gen_uv || !m || !ok)
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_vargen_host(ctx, u + 32 * off, R_uv + 64 * off, PK_uv + 64 * off, Gen_uv + 64 * off,
                              m + 32 * off, cnt, ok + off);
  });
}

// ---- projective inputs: the reference's in-memory types ------------------------------------
// `PublicKey::from(&sk)` = GENERATOR_EXTENDED * sk and R = GENERATOR_EXTENDED * r are JubJubExtended
// values with z != 1 (/root/reference/src/keys/public.rs:61-67, src/keys/secret.rs:159), and the
// reference's verify starts with `to_hash_inputs` (src/signatures.rs:131, :280-281): one field
// inversion per point.  The *_ext entry points take (u, v, z) and do that step on the device —
// Montgomery's trick over all points of an item and over the items of a lane (k_normalize_uvz) —
// so a caller (the Rust verify_batch) does no field arithmetic on the host at all.
extern "C++" {
namespace {
struct ExtWs {
  uint8_t* pts[4];
  uint8_t* valid;
  u32* prefix;
  void* vws;
};
size_t ext_workspace_bytes(size_t n) {
  return 4 * align_up(n * 64, 256) + align_up(n, 256) + align_up(normalize_prefix_bytes(n, 4), 256) +
         align_up(dsv_workspace_bytes(n), 256) + 256;
}
ExtWs carve_ext(void* ws, size_t n) {
  Stager st(static_cast<uint8_t*>(ws));
  ExtWs w;
  for (int k = 0; k < 4; k++) w.pts[k] = st.take(n * 64);
  w.valid = st.take(n);
  w.prefix = reinterpret_cast<u32*>(st.take(normalize_prefix_bytes(n, 4)));
  w.vws = st.take(dsv_workspace_bytes(n));
  return w;
}
// kind 0: pts = {R, PK}; 1: {R, R', PK, PK'}; 2: {R, PK, Gen} — each n x 96 B (u || v || z)
// u_mont / m_mont (both or neither): the scalars as Montgomery limbs, converted INTO u / m by the
// normalisation kernel (dsv_verify_*_mont_dev)
int verify_ext_on(Context& ctx, int kind, const void* u, const void* const* pts_uvz, const void* m,
                  size_t n, void* ok, void* workspace, hipStream_t s, const uint8_t* u_mont = nullptr,
                  const uint8_t* m_mont = nullptr) {
  const int np = kind == 0 ? 2 : (kind == 1 ? 4 : 3);
  const ExtWs w = carve_ext(workspace, n);
  NormalizeArgs a = {};
  for (int k = 0; k < np; k++) {
    a.in[k] = (const uint8_t*)pts_uvz[k];
    a.out[k] = w.pts[k];
  }
  if (u_mont) {
    a.u_mont = u_mont;
    a.m_mont = m_mont;
    a.u_out = (uint8_t*)const_cast<void*>(u);
    a.m_out = (uint8_t*)const_cast<void*>(m);
  }
  launch_normalize_uvz(a, np, n, w.valid, w.prefix, s);
  int rc;
  if (kind == 0) rc = verify_single_on(ctx, u, w.pts[0], w.pts[1], m, n, ok, w.vws, s, w.valid);
  else if (kind == 1) rc = verify_double_on(ctx, u, w.pts[0], w.pts[1], w.pts[2], w.pts[3], m, n, ok, w.vws, s, w.valid);
  else rc = verify_vargen_on(ctx, u, w.pts[0], w.pts[1], w.pts[2], m, n, ok, w.vws, s, w.valid);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// host buffers: the same through the chunked pipeline; the ext workspace of a chunk sits in the
// slot's `extra` area and the pipeline's own verify workspace is not used
constexpr size_t kExtItemBytes = 4 * 64 + 1 + 4 * kLimbs * 4 + 1;
// whole-chunk `to_hash_inputs` of the scheme's points (d: u, points..., m; [u_alt, m_alt]: converted
// scalars to use instead of d's); fills the staged view of the affine path
// mont: the scalars d[0] / d[1 + np] are Montgomery limbs — the same kernel converts them into (u_alt, m_alt)
int prep_normalize(const Context& ctx, int kind, const void* const* d, size_t cnt, Stager& x, hipStream_t st, Staged& g,
                   uint8_t* u_alt = nullptr, uint8_t* m_alt = nullptr, bool mont = false) {
  const int np = kind == 0 ? 2 : (kind == 1 ? 4 : 3);
  NormalizeArgs a = {};
  for (int k = 0; k < np; k++) {
    a.in[k] = (const uint8_t*)d[1 + k];
    a.out[k] = x.take(cnt * 64);
    g.p[1 + k] = a.out[k];
    g.bytes[1 + k] = 64;
  }
  uint8_t* valid = x.take(cnt);
  u32* prefix = reinterpret_cast<u32*>(x.take(normalize_prefix_bytes(cnt, np)));
  if (mont) {
    a.u_mont = (const uint8_t*)d[0];
    a.m_mont = (const uint8_t*)d[1 + np];
    a.u_out = u_alt;
    a.m_out = m_alt;
  }
  launch_normalize_uvz(a, np, cnt, valid, prefix, st, ctx.norm_per_lane, ctx.norm_block);
  HIP_TRY(hipGetLastError());
  g.p[0] = u_alt ? u_alt : (const uint8_t*)d[0];
  g.p[1 + np] = m_alt ? m_alt : (const uint8_t*)d[1 + np];
  g.bytes[0] = g.bytes[1 + np] = 32;
  g.valid = valid;
  return DSV_OK;
}
template <size_t NIN>
int verify_ext_host(Context& ctx, int kind, const HostIn (&ins)[NIN], size_t n, uint8_t* ok) {
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, kExtItemBytes, 0,
                       [kind, cp](const void* const* d, size_t cnt, Stager& x, hipStream_t st, Staged& g) {
                         return prep_normalize(*cp, kind, d, cnt, x, st, g);
                       },
                       DSV_PART(kind));
}
int verify_single_ext_host(Context& ctx, const uint8_t* u, const uint8_t* R, const uint8_t* PK,
                           const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[4] = {{u, 32}, {R, 96}, {PK, 96}, {m, 32}};
  return verify_ext_host(ctx, 0, ins, n, ok);
}
int verify_double_ext_host(Context& ctx, const uint8_t* u, const uint8_t* R, const uint8_t* Rp,
                           const uint8_t* PK, const uint8_t* PKp, const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[6] = {{u, 32}, {R, 96}, {Rp, 96}, {PK, 96}, {PKp, 96}, {m, 32}};
  return verify_ext_host(ctx, 1, ins, n, ok);
}
int verify_vargen_ext_host(Context& ctx, const uint8_t* u, const uint8_t* R, const uint8_t* PK,
                           const uint8_t* Gen, const uint8_t* m, size_t n, uint8_t* ok) {
  const HostIn ins[5] = {{u, 32}, {R, 96}, {PK, 96}, {Gen, 96}, {m, 32}};
  return verify_ext_host(ctx, 2, ins, n, ok);
}
}  // namespace
}  // extern "C++"

size_t dsv_ext_workspace_bytes(size_t n) { return ext_workspace_bytes(n); }

// JubJubExtended::to_hash_inputs for n points: (u, v, z) -> (u/z, v/z); ok[i] = 0 for z = 0 or a
// non-canonical coordinate (the reference would panic / cannot hold such a value)
int dsv_to_hash_inputs(const uint8_t* in_uvz, size_t n, uint8_t* out_uv, uint8_t* ok) {
  if (n && (!in_uvz || !out_uv || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  const size_t pre = normalize_prefix_bytes(n, 1);
  if (int r = ensure_stage(ctx, align_up(n * 96, 256) + align_up(n * 64, 256) + align_up(n, 256) +
                                    align_up(pre, 256)))
    return r;
  Stager st(ctx.stage);
  uint8_t *din = st.take(n * 96), *dout = st.take(n * 64), *dok = st.take(n);
  u32* dpre = reinterpret_cast<u32*>(st.take(pre));
  H2D(din, in_uvz, n * 96);
  NormalizeArgs a = {};
  a.in[0] = din;
  a.out[0] = dout;
  launch_normalize_uvz(a, 1, n, dok, dpre, 0);
  HIP_TRY(hipGetLastError());
  D2H(out_uv, dout, n * 64);
  D2H(ok, dok, n);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

int dsv_verify_single_ext(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                          const uint8_t* m, size_t n, uint8_t* ok) {
  if (n && (!u || !R_uvz || !PK_uvz || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_single_ext_host(ctx, u, R_uvz, PK_uvz, m, n, ok);
}
int dsv_verify_double_ext(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* Rp_uvz,
                          const uint8_t* PK_uvz, const uint8_t* PKp_uvz, const uint8_t* m, size_t n,
                          uint8_t* ok) {
  if (n && (!u || !R_uvz || !Rp_uvz || !PK_uvz || !PKp_uvz || !m || !ok))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_double_ext_host(ctx, u, R_uvz, Rp_uvz, PK_uvz, PKp_uvz, m, n, ok);
}
int dsv_verify_vargen_ext(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                          const uint8_t* Gen_uvz, const uint8_t* m, size_t n, uint8_t* ok) {
  if (n && (!u || !R_uvz || !PK_uvz || !Gen_uvz || !m || !ok))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_vargen_ext_host(ctx, u, R_uvz, PK_uvz, Gen_uvz, m, n, ok);
}
int dsv_verify_single_ext_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                                const uint8_t* m, size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uvz || !PK_uvz || !m || !ok) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_single_ext_host(ctx, u + 32 * off, R_uvz + 96 * off, PK_uvz + 96 * off, m + 32 * off,
                                  cnt, ok + off);
  });
}
int dsv_verify_double_ext_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* Rp_uvz,
                                const uint8_t* PK_uvz, const uint8_t* PKp_uvz, const uint8_t* m,
                                size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uvz || !Rp_uvz || !PK_uvz || !PKp_uvz || !m || !ok)
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_double_ext_host(ctx, u + 32 * off, R_uvz + 96 * off, Rp_uvz + 96 * off,
                                  PK_uvz + 96 * off, PKp_uvz + 96 * off, m + 32 * off, cnt, ok + off);
  });
}
int dsv_verify_vargen_ext_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                                const uint8_t* Gen_uvz, const uint8_t* m, size_t n, uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!u || !R_uvz || !PK_uvz || !Gen_uvz || !m || !ok)
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
    return verify_vargen_ext_host(ctx, u + 32 * off, R_uvz + 96 * off, PK_uvz + 96 * off,
                                  Gen_uvz + 96 * off, m + 32 * off, cnt, ok + off);
  });
}
int dsv_verify_single_ext_dev(const void* u, const void* R_uvz, const void* PK_uvz, const void* m,
                              size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uvz || !PK_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[2] = {R_uvz, PK_uvz};
  return verify_ext_on(ctx, 0, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}
int dsv_verify_double_ext_dev(const void* u, const void* R_uvz, const void* Rp_uvz, const void* PK_uvz,
                              const void* PKp_uvz, const void* m, size_t n, void* ok, void* workspace,
                              void* stream) {
  if (n && (!u || !R_uvz || !Rp_uvz || !PK_uvz || !PKp_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[4] = {R_uvz, Rp_uvz, PK_uvz, PKp_uvz};
  return verify_ext_on(ctx, 1, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}
int dsv_verify_vargen_ext_dev(const void* u, const void* R_uvz, const void* PK_uvz, const void* Gen_uvz,
                              const void* m, size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uvz || !PK_uvz || !Gen_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[3] = {R_uvz, PK_uvz, Gen_uvz};
  return verify_ext_on(ctx, 2, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}

// ---- the reference's in-memory representation: Montgomery limbs --------------------------------
// The Rust types hold every field element as `[u64; 4]` Montgomery limbs with R = 2^256
// (`BlsScalar(pub [u64; 4])`, dusk-bls12_381 0.13; `JubJubScalar`, the coordinates of
// `JubJubExtended`, dusk-jubjub 0.14 — /root/reference/Cargo.toml:25-26; the fields:
// src/signatures.rs:58-61, src/keys/public.rs:59).  `to_bytes()` is one Montgomery reduction per
// element — eight per single signature, fourteen per double one — on ONE host thread: ~30x below the
// engine.  The *_mont entry points take the limbs as they lie in memory:
//   points  (u R, v R, z R) : straight into k_normalize_uvz — a quotient does not see the common factor
//   u, m                    : two reductions per signature inside the same kernel (scalars_from_mont_item)
// so a binding copies bytes and nothing else; the *_mont_cols forms even take the typed objects
// where they lie (one strided column per field) and gather them into the pinned staging with the
// pipeline's copy threads — no intermediate structure of arrays on the host.
extern "C++" {
namespace {
size_t mont_workspace_bytes(size_t n) { return 2 * align_up(n * 32, 256) + ext_workspace_bytes(n); }
int verify_mont_on(Context& ctx, int kind, const void* u, const void* const* pts_uvz, const void* m,
                   size_t n, void* ok, void* workspace, hipStream_t s) {
  Stager st(static_cast<uint8_t*>(workspace));
  uint8_t *cu = st.take(n * 32), *cm = st.take(n * 32);
  return verify_ext_on(ctx, kind, cu, pts_uvz, cm, n, ok, st.take(0), s, (const uint8_t*)u, (const uint8_t*)m);
}
constexpr size_t kMontItemBytes = kExtItemBytes + 64;
// ins: u, points..., m — dense arrays or strided columns of typed objects
template <size_t NIN>
int verify_mont_host(Context& ctx, int kind, const HostIn (&ins)[NIN], size_t n, uint8_t* ok) {
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, kMontItemBytes, 0,
                       [kind, cp](const void* const* d, size_t cnt, Stager& x, hipStream_t st, Staged& g) {
                         const int np = kind == 0 ? 2 : (kind == 1 ? 4 : 3);
                         uint8_t *cu = x.take(cnt * 32), *cm = x.take(cnt * 32);
                         (void)np;
                         return prep_normalize(*cp, kind, d, cnt, x, st, g, cu, cm, true);  // ONE launch
                       },
                       DSV_PART(kind));
}
// columns of one scheme: u (32 B), its points (96 B each), m (32 B)
constexpr int kMontCols[3] = {4, 6, 5};
int check_cols(int kind, const dsv_column* cols, size_t n, const uint8_t* ok) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!cols || !ok) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  const int nc = kMontCols[kind];
  for (int k = 0; k < nc; k++) {
    const size_t width = (k == 0 || k == nc - 1) ? 32 : 96;
    if (!cols[k].base) return fail(DSV_ERR_INVALID_ARGUMENT, "column %d: null pointer", k);
    if (cols[k].stride < width) return fail(DSV_ERR_INVALID_ARGUMENT, "column %d: stride %zu < %zu", k, cols[k].stride, width);
  }
  return DSV_OK;
}
// one shard [off, off + cnt) of a column batch on one device
int verify_mont_cols_shard(Context& ctx, int kind, const dsv_column* cols, size_t off, size_t cnt, uint8_t* ok) {
  auto in = [&](int k, size_t width) {
    return HostIn{static_cast<const uint8_t*>(cols[k].base) + off * cols[k].stride, width, cols[k].stride};
  };
  if (kind == 0) {
    const HostIn ins[4] = {in(0, 32), in(1, 96), in(2, 96), in(3, 32)};
    return verify_mont_host(ctx, 0, ins, cnt, ok + off);
  }
  if (kind == 1) {
    const HostIn ins[6] = {in(0, 32), in(1, 96), in(2, 96), in(3, 96), in(4, 96), in(5, 32)};
    return verify_mont_host(ctx, 1, ins, cnt, ok + off);
  }
  const HostIn ins[5] = {in(0, 32), in(1, 96), in(2, 96), in(3, 96), in(4, 32)};
  return verify_mont_host(ctx, 2, ins, cnt, ok + off);
}
int verify_mont_cols(int kind, const dsv_column* cols, size_t n, uint8_t* ok, bool multi) {
  if (int r = check_cols(kind, cols, n, ok)) return r;
  if (n == 0) return DSV_OK;
  if (multi)
    return run_multi(n, [=](Context& ctx, size_t off, size_t cnt) {
      return verify_mont_cols_shard(ctx, kind, cols, off, cnt, ok);
    });
  Context* ctxp = nullptr;
  if (int r = host_context(ctxp)) return r;
  return verify_mont_cols_shard(*ctxp, kind, cols, 0, n, ok);
}
}  // namespace
}  // extern "C++"

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

size_t dsv_mont_workspace_bytes(size_t n) { return mont_workspace_bytes(n); }

// copy threads of the host entry points (per process): n >= 1 sets, 0 restores the default
// ($DSV_HOST_THREADS, else 4); returns the value now in force
int dsv_set_host_threads(int n) {
  g_host_threads.store(n > 0 ? clamp_host_threads(n) : 0, std::memory_order_relaxed);
  return host_copy_threads();
}

// ---- asynchronous form: submit returns at once, the batch runs on a library-owned driver thread ----
// What a caller with a stream of batches uses to keep the GPU busy across calls: while batch k's last
// chunks are on the GPU, batch k + 1's driver already gathers, transfers and enqueues its first ones
// (each call in flight owns a Pipe; the compute lanes are shared, so the GPU sees one FIFO of
// sub-batches).  A third submit simply waits for a pipe inside its driver thread.
struct dsv_job {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  bool started = false;
  std::atomic<bool> finished{false};
  int rc = DSV_OK;
  std::string err;
  int kind = 0;
  dsv_column cols[6] = {};
  size_t n = 0;
  uint8_t* ok = nullptr;
};
namespace {
int submit_mont_cols(int kind, const dsv_column* cols, size_t n, uint8_t* ok, dsv_job** out) {
  if (!out) return fail(DSV_ERR_INVALID_ARGUMENT, "null job pointer");
  *out = nullptr;
  if (int r = check_cols(kind, cols, n, ok)) return r;
  if (g_primary.load(std::memory_order_acquire) < 0)
    return fail(DSV_ERR_NOT_INITIALIZED, "dsv_init() has not been called");
  dsv_job* j = new (std::nothrow) dsv_job;
  if (!j) return fail(DSV_ERR_HIP, "out of host memory");
  j->kind = kind;
  j->n = n;
  j->ok = ok;
  for (int k = 0; n && k < kMontCols[kind]; k++) j->cols[k] = cols[k];
  auto job_count = [](int d) {
    std::lock_guard<std::mutex> lk(g_jobs_mu);
    g_jobs += d;
    if (g_jobs == 0) g_jobs_cv.notify_all();
  };
  job_count(+1);
  try {
    j->th = std::thread([j, job_count] {
      {
        std::lock_guard<std::mutex> lk(j->m);
        j->started = true;
      }
      j->cv.notify_all();
      j->rc = verify_mont_cols(j->kind, j->cols, j->n, j->ok, true);
      if (j->rc) j->err = g_err;  // the text lives in this thread's thread-local
      j->finished.store(true, std::memory_order_release);
      job_count(-1);
    });
  } catch (...) {
    job_count(-1);
    delete j;
    return fail(DSV_ERR_HIP, "could not start the driver thread of the batch");
  }
  {
    // jobs take their place in the device's queue in submission order: return once the driver runs
    // (it queues for its pipe within microseconds; the next submit has a thread to start first)
    std::unique_lock<std::mutex> lk(j->m);
    j->cv.wait(lk, [j] { return j->started; });
  }
  *out = j;
  return DSV_OK;
}
}  // namespace
int dsv_verify_single_mont_cols_submit(const dsv_column* cols, size_t n, uint8_t* ok, dsv_job** job) { return submit_mont_cols(0, cols, n, ok, job); }
int dsv_verify_double_mont_cols_submit(const dsv_column* cols, size_t n, uint8_t* ok, dsv_job** job) { return submit_mont_cols(1, cols, n, ok, job); }
int dsv_verify_vargen_mont_cols_submit(const dsv_column* cols, size_t n, uint8_t* ok, dsv_job** job) { return submit_mont_cols(2, cols, n, ok, job); }
int dsv_job_done(const dsv_job* job) {
  if (!job) return fail(DSV_ERR_INVALID_ARGUMENT, "null job");
  return job->finished.load(std::memory_order_acquire) ? 1 : 0;
}
int dsv_job_wait(dsv_job* job) {
  if (!job) return fail(DSV_ERR_INVALID_ARGUMENT, "null job");
  if (job->th.joinable()) job->th.join();
  const int rc = job->rc;
  if (rc) g_err = job->err;
  delete job;
  return rc;
}
int dsv_max_in_flight(void) { return kPipes; }

int dsv_verify_single_mont_cols(const dsv_column* cols, size_t n, uint8_t* ok) { return verify_mont_cols(0, cols, n, ok, true); }
int dsv_verify_double_mont_cols(const dsv_column* cols, size_t n, uint8_t* ok) { return verify_mont_cols(1, cols, n, ok, true); }
int dsv_verify_vargen_mont_cols(const dsv_column* cols, size_t n, uint8_t* ok) { return verify_mont_cols(2, cols, n, ok, true); }

#define DSV_DENSE_COLS_SINGLE {{u, 32}, {R_uvz, 96}, {PK_uvz, 96}, {m, 32}}
#define DSV_DENSE_COLS_DOUBLE {{u, 32}, {R_uvz, 96}, {Rp_uvz, 96}, {PK_uvz, 96}, {PKp_uvz, 96}, {m, 32}}
#define DSV_DENSE_COLS_VARGEN {{u, 32}, {R_uvz, 96}, {PK_uvz, 96}, {Gen_uvz, 96}, {m, 32}}
int dsv_verify_single_mont(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz, const uint8_t* m,
                           size_t n, uint8_t* ok) {
  const dsv_column cols[4] = DSV_DENSE_COLS_SINGLE;
  return verify_mont_cols(0, cols, n, ok, false);
}
int dsv_verify_double_mont(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* Rp_uvz,
                           const uint8_t* PK_uvz, const uint8_t* PKp_uvz, const uint8_t* m, size_t n,
                           uint8_t* ok) {
  const dsv_column cols[6] = DSV_DENSE_COLS_DOUBLE;
  return verify_mont_cols(1, cols, n, ok, false);
}
int dsv_verify_vargen_mont(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                           const uint8_t* Gen_uvz, const uint8_t* m, size_t n, uint8_t* ok) {
  const dsv_column cols[5] = DSV_DENSE_COLS_VARGEN;
  return verify_mont_cols(2, cols, n, ok, false);
}
int dsv_verify_single_mont_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                                 const uint8_t* m, size_t n, uint8_t* ok) {
  const dsv_column cols[4] = DSV_DENSE_COLS_SINGLE;
  return verify_mont_cols(0, cols, n, ok, true);
}
int dsv_verify_double_mont_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* Rp_uvz,
                                 const uint8_t* PK_uvz, const uint8_t* PKp_uvz, const uint8_t* m, size_t n,
                                 uint8_t* ok) {
  const dsv_column cols[6] = DSV_DENSE_COLS_DOUBLE;
  return verify_mont_cols(1, cols, n, ok, true);
}
int dsv_verify_vargen_mont_multi(const uint8_t* u, const uint8_t* R_uvz, const uint8_t* PK_uvz,
                                 const uint8_t* Gen_uvz, const uint8_t* m, size_t n, uint8_t* ok) {
  const dsv_column cols[5] = DSV_DENSE_COLS_VARGEN;
  return verify_mont_cols(2, cols, n, ok, true);
}
int dsv_verify_single_mont_dev(const void* u, const void* R_uvz, const void* PK_uvz, const void* m,
                               size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uvz || !PK_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[2] = {R_uvz, PK_uvz};
  return verify_mont_on(ctx, 0, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}
int dsv_verify_double_mont_dev(const void* u, const void* R_uvz, const void* Rp_uvz, const void* PK_uvz,
                               const void* PKp_uvz, const void* m, size_t n, void* ok, void* workspace,
                               void* stream) {
  if (n && (!u || !R_uvz || !Rp_uvz || !PK_uvz || !PKp_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[4] = {R_uvz, Rp_uvz, PK_uvz, PKp_uvz};
  return verify_mont_on(ctx, 1, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}
int dsv_verify_vargen_mont_dev(const void* u, const void* R_uvz, const void* PK_uvz, const void* Gen_uvz,
                               const void* m, size_t n, void* ok, void* workspace, void* stream) {
  if (n && (!u || !R_uvz || !PK_uvz || !Gen_uvz || !m || !ok || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, ok);
  const void* pts[3] = {R_uvz, PK_uvz, Gen_uvz};
  return verify_mont_on(ctx, 2, u, pts, m, n, ok, workspace, (hipStream_t)stream);
}

int dsv_challenge_single(const uint8_t* R_uv, const uint8_t* m, size_t n, uint8_t* c) {
  if (n && (!R_uv || !m || !c)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, align_up(n * 64, 256) + 2 * align_up(n * 32, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *dR = st.take(n * 64), *dm = st.take(n * 32), *dc = st.take(n * 32);
  H2D(dR, R_uv, n * 64);
  H2D(dm, m, n * 32);
  launch_challenge(false, (const uint8_t*)dR, (const uint8_t*)nullptr, (const uint8_t*)dm, n, dc, (uint8_t*)nullptr, 0);
  HIP_TRY(hipGetLastError());
  D2H(c, dc, n * 32);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_challenge_double(const uint8_t* R_uv, const uint8_t* Rp_uv, const uint8_t* m, size_t n,
                         uint8_t* c) {
  if (n && (!R_uv || !Rp_uv || !m || !c)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, 2 * align_up(n * 64, 256) + 2 * align_up(n * 32, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *dR = st.take(n * 64), *dRp = st.take(n * 64), *dm = st.take(n * 32),
          *dc = st.take(n * 32);
  H2D(dR, R_uv, n * 64);
  H2D(dRp, Rp_uv, n * 64);
  H2D(dm, m, n * 32);
  launch_challenge(true, (const uint8_t*)dR, (const uint8_t*)dRp, (const uint8_t*)dm, n, dc, (uint8_t*)nullptr, 0);
  HIP_TRY(hipGetLastError());
  D2H(c, dc, n * 32);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

// ---- signing / key derivation: INPUT GENERATION for tests and benchmarks ----------------------
// NOT a replacement for SecretKey::sign in production: the fixed- and variable-base multiplications
// index tables in global memory with digits of the secret scalar (addresses depend on secrets;
// dusk-jubjub's multiplication is constant-time), and secrets pass through library-owned staging.
// The host entry points scrub that staging before they return; the *_dev ones never own secrets.
namespace {
void launch_sign_single(Context& ctx, const void* sk, const void* m, const void* r, size_t n, void* u,
                        void* R_uv, hipStream_t s) {
  launch_fixed_base_points((const uint8_t*)r, (const u32*)ctx.table[0], n, (uint8_t*)R_uv, s);
  // scratch use: c is written to u (32 B per item) before k_sign_finish overwrites it in place
  launch_challenge(false, (const uint8_t*)R_uv, (const uint8_t*)nullptr, (const uint8_t*)m, n, (uint8_t*)u, (uint8_t*)nullptr, s);
  launch_sign_finish((const uint8_t*)r, (const uint8_t*)u, (const uint8_t*)sk, n, (uint8_t*)u, s);
}
void launch_sign_double(Context& ctx, const void* sk, const void* m, const void* r, size_t n, void* u,
                        void* R_uv, void* Rp_uv, hipStream_t s) {
  launch_fixed_base_points((const uint8_t*)r, (const u32*)ctx.table[0], n, (uint8_t*)R_uv, s);
  launch_fixed_base_points((const uint8_t*)r, (const u32*)ctx.table[1], n, (uint8_t*)Rp_uv, s);
  launch_challenge(true, (const uint8_t*)R_uv, (const uint8_t*)Rp_uv, (const uint8_t*)m, n, (uint8_t*)u, (uint8_t*)nullptr, s);
  launch_sign_finish((const uint8_t*)r, (const uint8_t*)u, (const uint8_t*)sk, n, (uint8_t*)u, s);
}
// host-side check of what the kernels would otherwise poison: scalars must be < r
int check_canonical_scalars(const uint8_t* s, size_t n, const char* what) {
  static const uint32_t kR[8] = DSV_R32;
  for (size_t i = 0; i < n; i++) {
    uint32_t w[8];
    memcpy(w, s + 32 * i, 32);
    bool lt = false;
    for (int k = 7; k >= 0; k--) {
      if (w[k] != kR[k]) {
        lt = w[k] < kR[k];
        break;
      }
    }
    if (!lt) return fail(DSV_ERR_INVALID_ARGUMENT, "%s[%zu] is not a canonical JubJubScalar (>= r)", what, i);
  }
  return DSV_OK;
}
}  // namespace

int dsv_public_keys_dev(const void* sk, int which, size_t n, void* PK_uv, void* stream) {
  if (n && (!sk || !PK_uv || which < 0 || which > 1)) return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  DSV_DEV_PROLOGUE(n, PK_uv);
  launch_fixed_base_points((const uint8_t*)sk, (const u32*)ctx.table[which], n, (uint8_t*)PK_uv, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_sign_single_dev(const void* sk, const void* m, const void* r, size_t n, void* u, void* R_uv,
                        void* stream) {
  if (n && (!sk || !m || !r || !u || !R_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, u);
  launch_sign_single(ctx, sk, m, r, n, u, R_uv, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_sign_double_dev(const void* sk, const void* m, const void* r, size_t n, void* u, void* R_uv,
                        void* Rp_uv, void* stream) {
  if (n && (!sk || !m || !r || !u || !R_uv || !Rp_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, u);
  launch_sign_double(ctx, sk, m, r, n, u, R_uv, Rp_uv, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
// variable-base forms (var-generator scheme): PK = sk * Gen; R = r * Gen, c, u.  `workspace`:
// dsv_workspace_bytes(n) device bytes (the per-lane window tables live there)
int dsv_public_keys_vargen_dev(const void* sk, const void* Gen_uv, size_t n, void* PK_uv,
                               void* workspace, void* stream) {
  if (n && (!sk || !Gen_uv || !PK_uv || !workspace)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, PK_uv);
  launch_var_base_points((const uint8_t*)sk, (const uint8_t*)Gen_uv, n, (uint8_t*)PK_uv, reinterpret_cast<u32*>(workspace), (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_sign_vargen_dev(const void* sk, const void* Gen_uv, const void* m, const void* r, size_t n,
                        void* u, void* R_uv, void* workspace, void* stream) {
  if (n && (!sk || !Gen_uv || !m || !r || !u || !R_uv || !workspace))
    return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, u);
  hipStream_t s = (hipStream_t)stream;
  launch_var_base_points((const uint8_t*)r, (const uint8_t*)Gen_uv, n, (uint8_t*)R_uv, reinterpret_cast<u32*>(workspace), s);
  launch_challenge(false, (const uint8_t*)R_uv, (const uint8_t*)nullptr, (const uint8_t*)m, n, (uint8_t*)u, (uint8_t*)nullptr, s);
  launch_sign_finish((const uint8_t*)r, (const uint8_t*)u, (const uint8_t*)sk, n, (uint8_t*)u, s);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

int dsv_public_keys(const uint8_t* sk, int which, const uint8_t* gen_uv, size_t n, uint8_t* PK_uv) {
  if (n && (!sk || !PK_uv || which < 0 || which > 1)) return fail(DSV_ERR_INVALID_ARGUMENT, "bad argument");
  DSV_HOST_PROLOGUE(n);
  if (int r = check_canonical_scalars(sk, n, "sk")) return r;
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, align_up(n * 32, 256) + 2 * align_up(n * 64, 256) +
                                    var_table_bytes(n, 1) + 256))
    return r;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dg = st.take(n * 64), *dpk = st.take(n * 64),
          *dtab = st.take(var_table_bytes(n, 1));
  H2D(dsk, sk, n * 32);
  if (gen_uv) {
    H2D(dg, gen_uv, n * 64);
    launch_var_base_points((const uint8_t*)dsk, (const uint8_t*)dg, n, dpk, reinterpret_cast<u32*>(dtab), 0);
  } else {
    launch_fixed_base_points((const uint8_t*)dsk, (const u32*)ctx.table[which], n, dpk, 0);
  }
  HIP_TRY(hipGetLastError());
  D2H(PK_uv, dpk, n * 64);
  // scrub the secret keys and the window tables derived from them
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));
  if (gen_uv) HIP_TRY(hipMemsetAsync(dtab, 0, var_table_bytes(n, 1), 0));
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_sign_single(const uint8_t* sk, const uint8_t* m, const uint8_t* r, size_t n, uint8_t* u,
                    uint8_t* R_uv) {
  if (n && (!sk || !m || !r || !u || !R_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  if (int rc = check_canonical_scalars(sk, n, "sk")) return rc;
  if (int rc = check_canonical_scalars(r, n, "nonce")) return rc;
  DSV_HOST_LOCK();
  if (int rc = ensure_stage(ctx, 4 * align_up(n * 32, 256) + align_up(n * 64, 256))) return rc;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dm = st.take(n * 32), *dr = st.take(n * 32),
          *du = st.take(n * 32), *dR = st.take(n * 64);
  H2D(dsk, sk, n * 32);
  H2D(dm, m, n * 32);
  H2D(dr, r, n * 32);
  launch_sign_single(ctx, dsk, dm, dr, n, du, dR, 0);
  HIP_TRY(hipGetLastError());
  D2H(u, du, n * 32);
  D2H(R_uv, dR, n * 64);
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));  // scrub key and nonce
  HIP_TRY(hipMemsetAsync(dr, 0, n * 32, 0));
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_sign_double(const uint8_t* sk, const uint8_t* m, const uint8_t* r, size_t n, uint8_t* u,
                    uint8_t* R_uv, uint8_t* Rp_uv) {
  if (n && (!sk || !m || !r || !u || !R_uv || !Rp_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  if (int rc = check_canonical_scalars(sk, n, "sk")) return rc;
  if (int rc = check_canonical_scalars(r, n, "nonce")) return rc;
  DSV_HOST_LOCK();
  if (int rc = ensure_stage(ctx, 4 * align_up(n * 32, 256) + 2 * align_up(n * 64, 256))) return rc;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dm = st.take(n * 32), *dr = st.take(n * 32),
          *du = st.take(n * 32), *dR = st.take(n * 64), *dRp = st.take(n * 64);
  H2D(dsk, sk, n * 32);
  H2D(dm, m, n * 32);
  H2D(dr, r, n * 32);
  launch_sign_double(ctx, dsk, dm, dr, n, du, dR, dRp, 0);
  HIP_TRY(hipGetLastError());
  D2H(u, du, n * 32);
  D2H(R_uv, dR, n * 64);
  D2H(Rp_uv, dRp, n * 64);
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));
  HIP_TRY(hipMemsetAsync(dr, 0, n * 32, 0));
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_sign_vargen(const uint8_t* sk, const uint8_t* Gen_uv, const uint8_t* m, const uint8_t* r,
                    size_t n, uint8_t* u, uint8_t* R_uv) {
  if (n && (!sk || !Gen_uv || !m || !r || !u || !R_uv)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  if (int rc = check_canonical_scalars(sk, n, "sk")) return rc;
  if (int rc = check_canonical_scalars(r, n, "nonce")) return rc;
  DSV_HOST_LOCK();
  if (int rc = ensure_stage(ctx, 4 * align_up(n * 32, 256) + 2 * align_up(n * 64, 256) +
                                     var_table_bytes(n, 1) + 256))
    return rc;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dm = st.take(n * 32), *dr = st.take(n * 32),
          *du = st.take(n * 32), *dG = st.take(n * 64), *dR = st.take(n * 64),
          *dtab = st.take(var_table_bytes(n, 1));
  H2D(dsk, sk, n * 32);
  H2D(dm, m, n * 32);
  H2D(dr, r, n * 32);
  H2D(dG, Gen_uv, n * 64);
  launch_var_base_points((const uint8_t*)dr, (const uint8_t*)dG, n, dR, reinterpret_cast<u32*>(dtab), 0);
  launch_challenge(false, (const uint8_t*)dR, (const uint8_t*)nullptr, (const uint8_t*)dm, n, du, (uint8_t*)nullptr, 0);
  launch_sign_finish((const uint8_t*)dr, (const uint8_t*)du, (const uint8_t*)dsk, n, du, 0);
  HIP_TRY(hipGetLastError());
  D2H(u, du, n * 32);
  D2H(R_uv, dR, n * 64);
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));
  HIP_TRY(hipMemsetAsync(dr, 0, n * 32, 0));
  HIP_TRY(hipMemsetAsync(dtab, 0, var_table_bytes(n, 1), 0));  // multiples of Gen, not secret — cheap anyway
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

// ---- wire formats ---------------------------------------------------------------------------
int dsv_decompress_points_dev(const void* in, size_t in_stride, size_t n, void* out_uv, void* ok,
                              int accumulate, void* stream) {
  DSV_DEV_PROLOGUE(n, out_uv);
  return decompress_on(ctx, in, in_stride, n, out_uv, ok, accumulate, (hipStream_t)stream);
}
// JubJubAffine::to_bytes: canonical v with bit 255 = lowest bit of canonical u.  Pure byte
// shuffling on affine input, so it runs on the host.
int dsv_compress_points(const uint8_t* in_uv, size_t n, uint8_t* out32) {
  if (int r = check_n(n)) return r;
  if (n == 0) return DSV_OK;
  if (!in_uv || !out32) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  for (size_t i = 0; i < n; i++) {
    memcpy(out32 + 32 * i, in_uv + 64 * i + 32, 32);
    out32[32 * i + 31] |= (uint8_t)((in_uv[64 * i] & 1) << 7);
  }
  return DSV_OK;
}

int dsv_decompress_points(const uint8_t* in32, size_t n, uint8_t* out_uv, uint8_t* ok) {
  if (n && (!in32 || !out_uv || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, align_up(n * 32, 256) + align_up(n * 64, 256) + align_up(n, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *din = st.take(n * 32), *dout = st.take(n * 64), *dok = st.take(n);
  H2D(din, in32, n * 32);
  if (int r = decompress_on(ctx, din, 32, n, dout, dok, 0, 0)) return r;
  D2H(out_uv, dout, n * 64);
  D2H(ok, dok, n);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

namespace {
// shared body of the *_wire entry points.  sig: n records of sig_bytes = 32 (u) + 32*n_sig_points;
// pk: n records of 32*n_pk_points compressed points.  kind: 0 single, 1 double, 2 vargen.
struct WireWs {
  uint8_t *u, *R, *Rp, *P0, *P1, *valid;
};
constexpr size_t kWireItemBytes = 32 + 4 * 64 + 1;
int verify_wire_on(Context& ctx, int kind, const uint8_t* dsig, const uint8_t* dpk, const void* dm,
                   size_t cnt, void* dok, const WireWs& x, void* vws, hipStream_t st) {
  const size_t sig_bytes = kind == 1 ? 96 : 64, pk_bytes = kind == 0 ? 32 : 64;
  launch_gather32(dsig, sig_bytes, cnt, x.u, st);
  if (int r = decompress_on(ctx, dsig + 32, sig_bytes, cnt, x.R, x.valid, 0, st)) return r;
  if (kind == 1)
    if (int r = decompress_on(ctx, dsig + 64, sig_bytes, cnt, x.Rp, x.valid, 1, st)) return r;
  if (int r = decompress_on(ctx, dpk, pk_bytes, cnt, x.P0, x.valid, 1, st)) return r;
  if (kind != 0)
    if (int r = decompress_on(ctx, dpk + 32, pk_bytes, cnt, x.P1, x.valid, 1, st)) return r;
  int rc;
  if (kind == 0) rc = verify_single_on(ctx, x.u, x.R, x.P0, dm, cnt, dok, vws, st, x.valid);
  else if (kind == 1) rc = verify_double_on(ctx, x.u, x.R, x.Rp, x.P0, x.P1, dm, cnt, dok, vws, st, x.valid);
  else rc = verify_vargen_on(ctx, x.u, x.R, x.P0, x.P1, dm, cnt, dok, vws, st, x.valid);
  if (rc) return rc;
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int verify_wire(Context& ctx, int kind, const uint8_t* sig, const uint8_t* pk, const uint8_t* m,
                size_t n, uint8_t* ok) {
  const size_t sig_bytes = kind == 1 ? 96 : 64, pk_bytes = kind == 0 ? 32 : 64;
  const HostIn ins[3] = {{sig, sig_bytes}, {pk, pk_bytes}, {m, 32}};
  Context* cp = &ctx;
  return run_pipelined(ctx, ins, ok, n, 0, kWireItemBytes, NoPrep{},
                       [=](const Staged& g, size_t off, size_t cnt, void* dok, void* ws, Stager& x, hipStream_t st) {
    WireWs w;
    w.u = x.take(cnt * 32);
    w.R = x.take(cnt * 64);
    w.Rp = x.take(cnt * 64);
    w.P0 = x.take(cnt * 64);
    w.P1 = x.take(cnt * 64);
    w.valid = x.take(cnt);
    return verify_wire_on(*cp, kind, g.p[0] + off * g.bytes[0], g.p[1] + off * g.bytes[1],
                          g.p[2] + off * g.bytes[2], cnt, dok, w, ws, st);
  });
}
// device-pointer form: serialized records already resident in HBM
int verify_wire_dev(int kind, const void* sig, const void* pk, const void* m, size_t n, void* ok,
                    void* workspace, void* stream) {
  if (n && (!sig || !pk || !m || !ok || !workspace)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  if (((uintptr_t)sig | (uintptr_t)pk) & 15) return fail(DSV_ERR_INVALID_ARGUMENT, "records must be 16-byte aligned");
  DSV_DEV_PROLOGUE(n, ok);
  Stager x(static_cast<uint8_t*>(workspace));
  WireWs w;
  w.u = x.take(n * 32);
  w.R = x.take(n * 64);
  w.Rp = x.take(n * 64);
  w.P0 = x.take(n * 64);
  w.P1 = x.take(n * 64);
  w.valid = x.take(n);
  void* vws = x.take(dsv_workspace_bytes(n));
  return verify_wire_on(ctx, kind, (const uint8_t*)sig, (const uint8_t*)pk, m, n, ok, w, vws,
                        (hipStream_t)stream);
}
}  // namespace

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

size_t dsv_wire_workspace_bytes(size_t n) {
  return align_up(n * 32, 256) + 4 * align_up(n * 64, 256) + align_up(n, 256) +
         align_up(dsv_workspace_bytes(n), 256) + 256;
}
int dsv_verify_single_wire_dev(const void* sig64, const void* pk32, const void* m, size_t n, void* ok,
                               void* workspace, void* stream) {
  return verify_wire_dev(0, sig64, pk32, m, n, ok, workspace, stream);
}
int dsv_verify_double_wire_dev(const void* sig96, const void* pk64, const void* m, size_t n, void* ok,
                               void* workspace, void* stream) {
  return verify_wire_dev(1, sig96, pk64, m, n, ok, workspace, stream);
}
int dsv_verify_vargen_wire_dev(const void* sig64, const void* pk64, const void* m, size_t n, void* ok,
                               void* workspace, void* stream) {
  return verify_wire_dev(2, sig64, pk64, m, n, ok, workspace, stream);
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

int dsv_verify_single_wire(const uint8_t* sig64, const uint8_t* pk32, const uint8_t* m, size_t n,
                           uint8_t* ok) {
  if (n && (!sig64 || !pk32 || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_wire(ctx, 0, sig64, pk32, m, n, ok);
}
int dsv_verify_double_wire(const uint8_t* sig96, const uint8_t* pk64, const uint8_t* m, size_t n,
                           uint8_t* ok) {
  if (n && (!sig96 || !pk64 || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_wire(ctx, 1, sig96, pk64, m, n, ok);
}
int dsv_verify_vargen_wire(const uint8_t* sig64, const uint8_t* pk64, const uint8_t* m, size_t n,
                           uint8_t* ok) {
  if (n && (!sig64 || !pk64 || !m || !ok)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  return verify_wire(ctx, 2, sig64, pk64, m, n, ok);
}

// ---- reference-harness inputs ---------------------------------------------------------------
namespace {
// rand_core 0.6 SeedableRng::seed_from_u64: eight PCG32 outputs = the ChaCha key
ChaChaKey stdrng_key(uint64_t state) {
  ChaChaKey k;
  for (int i = 0; i < 8; i++) {
    state = state * 6364136223846793005ULL + 11634580027462260723ULL;
    const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    const uint32_t rot = (uint32_t)(state >> 59);
    k.w[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
  }
  return k;
}
}  // namespace

int dsv_stdrng_sign_inputs_dev(uint64_t seed, size_t first_item, size_t n, void* sk, void* m,
                               void* r, void* stream) {
  if (n && (!sk || !m || !r)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, sk);
  launch_stdrng_triples(stdrng_key(seed), first_item, n, (uint8_t*)sk, (uint8_t*)m, (uint8_t*)r, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}
int dsv_stdrng_sign_inputs(uint64_t seed, size_t first_item, size_t n, uint8_t* sk, uint8_t* m,
                           uint8_t* r) {
  if (n && (!sk || !m || !r)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int rc = ensure_stage(ctx, 3 * align_up(n * 32, 256))) return rc;
  Stager st(ctx.stage);
  uint8_t *dsk = st.take(n * 32), *dm = st.take(n * 32), *dr = st.take(n * 32);
  launch_stdrng_triples(stdrng_key(seed), first_item, n, dsk, dm, dr, 0);
  HIP_TRY(hipGetLastError());
  D2H(sk, dsk, n * 32);
  D2H(m, dm, n * 32);
  D2H(r, dr, n * 32);
  HIP_TRY(hipMemsetAsync(dsk, 0, n * 32, 0));
  HIP_TRY(hipMemsetAsync(dr, 0, n * 32, 0));
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}
int dsv_stdrng_vargen_inputs_dev(uint64_t seed, size_t first_item, size_t n, void* sk, void* g,
                                 void* m, void* r, void* stream) {
  if (n && (!sk || !g || !m || !r)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_DEV_PROLOGUE(n, sk);
  launch_stdrng_quads(stdrng_key(seed), first_item, n, (uint8_t*)sk, (uint8_t*)g, (uint8_t*)m, (uint8_t*)r, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return DSV_OK;
}

int dsv_debug_table_entry(int which, int window, int digit, uint8_t out96[96]) {
  if (which < 0 || which > 1 || window < 0 || window >= kFixedWindows || digit < 0 ||
      digit >= kFixedEntries || !out96)
    return fail(DSV_ERR_INVALID_ARGUMENT, "bad table coordinates");
  Context* ctxp = nullptr;
  if (int r = host_context(ctxp)) return r;
  Context& ctx = *ctxp;
  DSV_HOST_LOCK();
  u32 e[kEntryWords];
  HIP_TRY(hipMemcpy(e, ctx.table[which] + ((size_t)window * kFixedEntries + digit) * kEntryWords,
                    sizeof e, hipMemcpyDeviceToHost));
  // entries are Montgomery (R = 2^261) canonical limbs; hand back the raw limbs as 3 x 9 x 29-bit
  // packed LE integers so the test can undo the Montgomery factor with Python integers.
  for (int f = 0; f < 3; f++) {
    unsigned __int128 acc = 0;
    int bits = 0, o = 0;
    uint8_t* dst = out96 + 32 * f;
    memset(dst, 0, 32);
    for (int i = 0; i < kLimbs; i++) {
      acc |= (unsigned __int128)e[f * kLimbs + i] << bits;
      bits += 29;
      while (bits >= 8 && o < 32) {
        dst[o++] = (uint8_t)acc;
        acc >>= 8;
        bits -= 8;
      }
    }
    while (o < 32) {
      dst[o++] = (uint8_t)acc;
      acc >>= 8;
    }
  }
  return DSV_OK;
}

int dsv_fixed_window_bits(void) { return kFixedBits; }

int dsv_debug_lattice3(const uint8_t* u, const uint8_t* c, size_t n, uint8_t* out128) {
  if (n && (!u || !c || !out128)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, 2 * align_up(n * 32, 256) + align_up(n * 128, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *du = st.take(n * 32), *dc = st.take(n * 32), *dout = st.take(n * 128);
  H2D(du, u, n * 32);
  H2D(dc, c, n * 32);
  launch_debug_lattice3(du, dc, n, dout, 0);
  HIP_TRY(hipGetLastError());
  D2H(out128, dout, n * 128);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

int dsv_debug_half_scalars(const uint8_t* c, size_t n, uint8_t* out96) {
  if (n && (!c || !out96)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, align_up(n * 32, 256) + align_up(n * 96, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *dc = st.take(n * 32), *dout = st.take(n * 96);
  H2D(dc, c, n * 32);
  launch_debug_half_scalars(dc, n, dout, 0);
  HIP_TRY(hipGetLastError());
  D2H(out96, dout, n * 96);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

int dsv_debug_fq_mul(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
  if (n && (!a || !b || !out)) return fail(DSV_ERR_INVALID_ARGUMENT, "null pointer");
  DSV_HOST_PROLOGUE(n);
  DSV_HOST_LOCK();
  if (int r = ensure_stage(ctx, 3 * align_up(n * 32, 256))) return r;
  Stager st(ctx.stage);
  uint8_t *da = st.take(n * 32), *db = st.take(n * 32), *dout = st.take(n * 32);
  H2D(da, a, n * 32);
  H2D(db, b, n * 32);
  launch_debug_fq_mul((const uint8_t*)da, (const uint8_t*)db, n, dout, 0);
  HIP_TRY(hipGetLastError());
  D2H(out, dout, n * 32);
  HIP_TRY(hipStreamSynchronize(0));
  return DSV_OK;
}

}  // extern "C"
