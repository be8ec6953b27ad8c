// dsv_host.h — what the host-side translation units of libdsv.so share: the per-device contexts, the
// error channel, workspace carving, the sub-batch scheduler of the device-pointer entry points and the
// bodies that more than one unit calls.  The units (one hipcc job each, build.py: UNITS):
//   dsv_context.hip   contexts, dsv_init / dsv_shutdown, streams and staging of a device
//   dsv_device.hip    device-pointer entry points: challenge, verify (affine), core, mixed batches
//   dsv_host.hip      host-pointer entry points: affine / projective / Montgomery-limb / typed-object
//                     columns, *_multi, submit / wait
//   dsv_wire.hip      serialized records (decode + verify), compress / decompress
//   dsv_rlc.hip       the batch fast accept: control of k_rlc.hip, every *_rlc entry point
//   dsv_inputs.hip    signing / key derivation / StdRng inputs (input generation), debug probes
// The host pipeline itself (run_pipelined, run_multi) is a template: dsv_pipeline.h.
// Nothing here is part of the C ABI (include/dsv.h); everything lives in namespace dsvh.
#pragma once
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
#include "launch.h"
#include "rlc.h"
#include "host_sync.h"

namespace dsvh {

using namespace dsv;
typedef uint32_t u32;

extern thread_local std::string g_err;
int fail(int code, const char* fmt, ...);
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
  // where the device hangs off the host (r06): PCI address, NUMA node of its root (-1: unknown or DSV_NUMA=0)
  // and that node's cpus — the copy threads of this device's pipes and the library's own per-device worker
  // threads run there (host_sync.h: "NUMA placement")
  std::string pci_bdf;
  int numa_node = -1;
  std::vector<int> numa_cpus;
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
  // the fast accept's words in pinned host memory, written by its verdict kernels (dsv_rlc.hip): [0] the
  // history counter — a call with a rejected sub-group sets it to 8, one whose aggregates all accepted takes 1
  // off; while it is > 0 calls check a sample first and run in sub-groups —, [1] calls completed, then a ring
  // of verdict slots for callers whose `accepted` is pageable memory
  std::mutex rlc_pinned_mu;
  u32* rlc_pinned = nullptr;
  std::atomic<u32> rlc_slot{0};
};
extern Context g_ctx[kMaxDevices];
extern std::mutex g_init_mu;               // dsv_init / dsv_shutdown
extern std::atomic<int> g_primary;         // first device initialised: default of the host entry points
extern thread_local int t_device;          // dsv_set_device: this thread's choice for host entry points
// jobs submitted and not finished (dsv_*_submit): dsv_shutdown lets them run to their verdicts first —
// a job's driver may not have queued for its pipe yet when the shutdown arrives
extern std::mutex g_jobs_mu;
extern std::condition_variable g_jobs_cv;
extern int g_jobs;

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

int ensure_stage(Context& ctx, size_t bytes);
void destroy_pipe_streams(Context& ctx);
int ensure_pipe_streams(Context& ctx);   // (called under enq_mu)
int ensure_pipe_work(Context& ctx, int lane, size_t bytes);
int ensure_pipe_slot(PipeSlot& sl, size_t dev_bytes, size_t host_bytes, size_t prep_bytes);
int check_n(size_t n);
// context of the host entry points: this thread's dsv_set_device choice, else the first device
// that was initialised
int host_context(Context*& out);
// context of a device-pointer entry point: the device that owns `ptr` (one of the call's buffers)
int device_context(const void* ptr, Context*& out);
void release_context(Context& ctx);

// workspace layout for the *_dev verify entry points: c[n][32] | valid[n]
// + the per-lane window tables of the verify kernels (fixed grid, see kMaxVerifyGrid)
struct Workspace {
  uint8_t* c;
  uint8_t* valid;
  u32* tables;
};
constexpr int kTablesPerLane = 3;  // the var-generator kernel keeps three (Gen, PK, R), the others two
inline size_t var_table_bytes(size_t n, int tables_per_lane) {
  return (size_t)verify_grid(n) * kVerifyBlock * kVarLaneWords * 4 * (size_t)tables_per_lane;
}
inline Workspace carve(void* ws, size_t n) {
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
                         size_t n, void* ok, u32* tables, hipStream_t s, bool tables_ready = false,
                         const u32* gate = nullptr);
// both equations of a double signature: one fused launch, or two single-equation ones
// (DSV_DOUBLE_FUSED=0: the second pass ANDs into ok[])
void launch_verify_fixed_double(const Context& ctx, const void* u, const void* c, const void* PK_uv,
                                const void* R_uv, const void* PKp_uv, const void* Rp_uv,
                                const void* valid, size_t n, void* ok, u32* tables, hipStream_t s,
                                bool tables_ready = false, const u32* gate = nullptr);

// Sub-batch scheduling of the device-pointer verify entry points.
// One launch over 2^20 signatures pays a fill and a drain phase per kernel (~0.5 ms + ~0.8 ms of
// 18 ms, tools/scaling_probe.py) and runs the two waves of every SIMD through the same phase of
// the same kernel.  Cutting the batch into sub-batches of 2^16 signatures (1024 waves: ONE wave
// per SIMD) that alternate between two internal streams keeps two different kernels co-resident
// on every SIMD — hash next to scalar multiplication, table build next to window loop — and
// leaves no gap between kernels: +6 % on 2^20, same-box A/B (probe: tools/overlap_probe.py).  The caller's stream is
// forked / joined with events, so the call still behaves as one enqueue on that stream.

int acquire_lane(Context& ctx, hipStream_t user, SplitLane*& out);
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


// bodies shared by the device-pointer entry points and the host pipeline (context resolved); dsv_device.hip
extern thread_local bool t_pipeline_part;  // run_pipelined, several chunks: its four streams are all there is
extern thread_local size_t t_chunk_first;  // run_pipelined: first item of the chunk being enqueued (prep / part
                                           // callbacks that place their output by item number: the fast accept)
// valid_in (may be null): per-item validity found by an earlier stage (normalisation, decompression);
// the hash kernel folds it into the validity the verify kernel starts from
int verify_single_on(Context& ctx, const void* u, const void* R_uv, const void* PK_uv, const void* m,
                     size_t n, void* ok, void* workspace, hipStream_t stream, const uint8_t* valid_in = nullptr);
int verify_double_on(Context& ctx, const void* u, const void* R_uv, const void* Rp_uv,
                     const void* PK_uv, const void* PKp_uv, const void* m, size_t n, void* ok,
                     void* workspace, hipStream_t stream, const uint8_t* valid_in = nullptr);
int verify_vargen_on(Context& ctx, const void* u, const void* R_uv, const void* PK_uv,
                     const void* Gen_uv, const void* m, size_t n, void* ok, void* workspace,
                     hipStream_t stream, const uint8_t* valid_in = nullptr);
int decompress_on(Context& ctx, const void* in, size_t in_stride, size_t n, void* out_uv, void* ok,
                  int accumulate, hipStream_t stream);

// the batch fast accept over one call's items (dsv_rlc.hip)
struct RlcStaged {  // a host call whose bucket pass over items [0, boundary) was enqueued while the rest was still on the bus
  ChaChaKey key;
  size_t boundary;
};
// scheme 0 single (R, PK), 1 double (R, R', PK, PK'), 2 var-generator (R, PK, Gen): unused pointers null
// Enqueue-only; *accepted_dev (device-accessible memory, may be null) is written by a kernel at the end.
int verify_rlc_on(Context& ctx, int scheme, const void* u, const void* R_uv, const void* Rp_uv, const void* PK_uv,
                  const void* PKp_uv, const void* Gen_uv, const void* m, size_t n, void* ok, void* workspace,
                  hipStream_t s, int window_bits, u32* accepted_dev, bool and_into = false, bool have_challenges = false,
                  const uint8_t* valid_in = nullptr, const RlcStaged* staged = nullptr);
// where a call's verdict goes: `dev` is what the verdict kernel writes; a caller's pageable `int* accepted`
// gets a pinned slot of the context and the call waits for the stream at its end (rlc_verdict_wait)
struct RlcVerdictTarget {
  u32* dev = nullptr;
  u32* slot = nullptr;
  int* out = nullptr;
};
int rlc_verdict_target(Context& ctx, int* accepted, RlcVerdictTarget& t);
int rlc_verdict_wait(const RlcVerdictTarget& t, hipStream_t s);
int rlc_clear_accepted(int* accepted);
int rlc_history(Context& ctx);  // the device's history counter (-1: no pinned memory to be had)

#define H2D(dst, src, bytes) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, 0))
#define D2H(dst, src, bytes) HIP_TRY(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, 0))
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


// per-item scratch of the host pipeline's whole-chunk preprocessing: projective inputs / Montgomery limbs
constexpr size_t kExtItemBytes = 4 * 64 + 1 + 4 * kLimbs * 4 + 1;
constexpr size_t kMontItemBytes = kExtItemBytes + 64;
// columns of typed objects (dsv_host.hip)
int check_cols(int kind, const dsv_column* cols, size_t n, const uint8_t* ok);
int verify_mont_cols(int kind, const dsv_column* cols, size_t n, uint8_t* ok, bool multi);

// serialized records (dsv_wire.hip)
struct WireWs {
  uint8_t *u, *R, *Rp, *P0, *P1, *valid;
};
int verify_wire(Context& ctx, int kind, const uint8_t* sig, const uint8_t* pk, const uint8_t* m,
                size_t n, uint8_t* ok);

}  // namespace dsvh
