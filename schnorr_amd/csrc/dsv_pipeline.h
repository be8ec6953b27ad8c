// dsv_pipeline.h — the chunked host pipeline shared by every host-pointer verify entry point
// (run_pipelined) and the sharding of one host batch over the initialised devices (run_multi).
// Templates over the chunk / sub-batch callbacks, hence a header; included by dsv_host.hip,
// dsv_wire.hip and dsv_rlc.hip.
#pragma once
#include "dsv_host.h"

namespace dsvh {

enum : unsigned { kPipeHeavy = 1, kPipeNoVerdicts = 2 };  // run_pipelined's flags

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

extern std::atomic<int> g_host_threads;  // dsv_set_host_threads; 0 = $DSV_HOST_THREADS, else 4 (dsv_host.hip)
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
//   flags: kPipeHeavy — the scheme does about twice a single signature's device work per item (double,
//     var-generator): the ramp of a call that finds the GPU idle then uses larger chunks (host_sync.h:
//     plan_chunks); kPipeNoVerdicts — the parts write no verdict bytes (the fast accept's arena fills): no
//     verdict copy per chunk, `ok` is left alone (it used to receive stale device bytes)
//   part(staged, offset, count, dok, ws, extra, stream): one sub-batch; `extra`: scratch of
//     extra_item_bytes per item behind the lane's verify workspace (the wire path decompresses per
//     sub-batch: full-occupancy kernels, no reason to serialise a chunk's worth on one lane).
template <size_t NIN, class Prep, class Part>
int run_pipelined(Context& ctx, const HostIn (&ins)[NIN], uint8_t* ok, size_t n, size_t prep_item_bytes,
                  size_t extra_item_bytes, Prep prep, Part part, unsigned flags = 0) {
  const bool heavy = (flags & kPipeHeavy) != 0, no_verdicts = (flags & kPipeNoVerdicts) != 0;
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
  const std::vector<size_t> chunks = plan_chunks(ctx.plan, n, small || lease.alone, part_cap, heavy);
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
    if (!no_verdicts) memcpy(ok + h.first, pipe.slot[sl].host + h.ok_off, h.cnt);
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
    // (r06 tried the verify kernels writing their verdicts straight into the slot's pinned host block — no copy,
    //  one stream hop less: equal on every scheme and on the 1024-item call, same box, alternating runs)
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
    if ((!no_verdicts && hipMemcpyAsync(slot.host + h.ok_off, dok, cnt, hipMemcpyDeviceToHost, s_out) != hipSuccess) ||
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
    if (k) (void)pin_this_thread(devs[k]->numa_cpus);  // (the library's own threads; shard 0 runs on the caller's)
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

}  // namespace dsvh
