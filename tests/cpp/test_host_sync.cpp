// The HIP-free part of the host pipeline (schnorr_amd/csrc/host_sync.h) under ThreadSanitizer — no GPU.
// What a verify call does around its kernels, with the kernels replaced by bookkeeping:
//   * CopyPool: many jobs of changing width, two pools at once (two calls in flight gather concurrently);
//   * PipeLease / TurnTicket: at most kPipes calls in flight, pipes and turns handed out in ONE order,
//     a call's chunks enqueued as one uninterrupted run, early exits (error paths) still pass the turn on;
//   * plan_chunks / plan_parts: every item exactly once, bounds, balanced sub-batches.
// Built and run by tests/test_host_sync.py (g++ -fsanitize=thread); exit code 0 = all checks passed and
// the sanitizer saw no race.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../schnorr_amd/csrc/host_sync.h"

using namespace dsv;

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      std::fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                     \
    }                                                                   \
  } while (0)

static void copy_pool() {
  auto worker = [](unsigned seed) {
    CopyPool pool;
    std::mt19937 rng(seed);
    std::vector<int> buf(1 << 14);
    for (int job = 0; job < 300; job++) {
      const int T = 1 + (int)(rng() % 6);
      std::fill(buf.begin(), buf.end(), -1);
      pool.run(T, [&](int t, int nt) {
        CHECK(nt == T && t >= 0 && t < nt);
        const size_t lo = buf.size() * (size_t)t / (size_t)nt, hi = buf.size() * (size_t)(t + 1) / (size_t)nt;
        for (size_t i = lo; i < hi; i++) buf[i] = job * 8 + t;
      });
      for (size_t i = 0; i < buf.size(); i++) {  // every slice written by exactly the thread that owns it
        const int t = buf[i] - job * 8;
        CHECK(t >= 0 && t < T && i >= buf.size() * (size_t)t / (size_t)T && i < buf.size() * (size_t)(t + 1) / (size_t)T);
      }
      if (job % 64 == 63) std::this_thread::sleep_for(std::chrono::microseconds(700));  // workers go to sleep
    }
  };
  std::thread a(worker, 1u), b(worker, 2u);  // two calls in flight: two pools
  a.join();
  b.join();
}

struct Log {
  std::mutex m;
  std::vector<std::pair<uint64_t, int>> enq;  // (turn of the call, chunk)
  std::vector<uint64_t> lease_order;          // turns in the order the leases were granted
};

static void leases_and_turns() {
  PipeSync sync;
  Log log;
  std::atomic<int> in_flight{0}, max_in_flight{0}, calls{0};
  std::mutex enq_mu;  // dsv_pipeline.h: one chunk's enqueue is atomic
  auto caller = [&](unsigned seed) {
    std::mt19937 rng(seed);
    for (int call = 0; call < 40; call++) {
      const bool small = rng() % 4 == 0;
      const bool fails = rng() % 7 == 0;  // an error half-way: leave without having enqueued everything
      const int nchunks = 1 + (int)(rng() % 6), slots = 3;
      TurnTicket turn(sync);
      PipeLease lease(sync, small ? nullptr : &turn);
      CHECK(lease.index >= 0 && lease.index < kPipes);
      const int now = ++in_flight;
      int seen = max_in_flight.load();
      while (now > seen && !max_in_flight.compare_exchange_weak(seen, now)) {
      }
      CHECK(now <= kPipes);
      if (!small) {
        std::lock_guard<std::mutex> lk(log.m);
        log.lease_order.push_back(turn.mine);
      }
      int staged = 0, enqueued = 0;
      while (enqueued < nchunks) {
        if (staged < nchunks && staged - enqueued < slots) {
          std::this_thread::sleep_for(std::chrono::microseconds(rng() % 200));  // gather + transfer
          staged++;
        }
        if (fails && staged == 2) break;
        if (!small && !turn.held) {
          if (!turn.try_acquire()) {
            if (staged < nchunks && staged - enqueued < slots) continue;
            turn.acquire();
          }
        }
        while (enqueued < staged) {
          std::lock_guard<std::mutex> e(enq_mu);
          if (!small) {
            std::lock_guard<std::mutex> lk(log.m);
            log.enq.push_back({turn.mine, enqueued});
          }
          enqueued++;
        }
        if (enqueued == nchunks) turn.release();
      }
      std::this_thread::sleep_for(std::chrono::microseconds(rng() % 300));  // final drain
      --in_flight;
      ++calls;
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 0; t < 6; t++) th.emplace_back(caller, 100 + t);
  for (auto& t : th) t.join();
  CHECK(calls.load() == 6 * 40 && max_in_flight.load() == kPipes);
  {
    std::lock_guard<std::mutex> lk(sync.mu);
    CHECK(sync.idle() && sync.turn_next == sync.turn_serving);
  }
  // every multi-chunk call got its own turn, none was skipped ...
  std::sort(log.lease_order.begin(), log.lease_order.end());
  for (size_t i = 0; i < log.lease_order.size(); i++) CHECK(log.lease_order[i] == i);
  // ... and the chunks of a call went out as one run, calls in turn order, chunks in order
  uint64_t last_turn = 0;
  int last_chunk = -1;
  for (auto& e : log.enq) {
    CHECK(e.first >= last_turn);
    if (e.first != last_turn) last_chunk = -1;
    CHECK(e.second == last_chunk + 1);
    last_turn = e.first;
    last_chunk = e.second;
  }
}

static void plans() {
  std::mt19937_64 rng(7);
  PlanParams p;
  for (int iter = 0; iter < 4000; iter++) {
    if (iter % 500 == 0) {
      p = PlanParams();
      if (iter >= 2000) p.chunk = (size_t)1 << (16 + rng() % 4);
      if (iter >= 3000) {
        p.plan_len = 3;
        p.plan[0] = 15, p.plan[1] = 16, p.plan[2] = 17;
      }
      if (p.first_chunk > p.chunk) p.first_chunk = p.chunk;
    }
    const size_t n = 1 + rng() % (iter % 3 ? ((size_t)1 << 18) : ((size_t)1 << 23));
    for (int ramp = 0; ramp < 4; ramp++) {  // (2, 3: the plan of a heavy scheme — double, var-generator)
      const bool heavy = ramp >= 2;
      const std::vector<size_t> chunks = plan_chunks(p, n, (ramp & 1) != 0, kSplitItems, heavy);
      size_t sum = 0;
      for (size_t c : chunks) {
        CHECK(c > 0 && (n <= kPipeSmallCall || c <= p.chunk + p.chunk / 4 + kSplitItems / 2));
        sum += c;
        for (int one = 0; one < 2; one++) {
          size_t part = 0;
          const size_t parts = plan_parts(c, one != 0, kSplitItems, part);
          CHECK(parts >= 1 && part > 0 && part * parts >= c && part * (parts - 1) < c);
          if (one || c <= kSplitItems) CHECK(parts == 1 && part == c);
          else CHECK(parts % 2 == 0 || part * parts - c >= part || parts == (c + part - 1) / part);
          if (!one) CHECK(part <= kSplitItems);
        }
      }
      CHECK(sum == n);
      if (n <= kPipeSmallCall) CHECK(chunks.size() == 1);
      if (!(ramp & 1) && n > kPipeSmallCall)  // behind another call: full chunks from the start
        for (size_t k = 0; k + 1 < chunks.size(); k++) CHECK(chunks[k] == p.chunk);
      if ((ramp & 1) && !heavy && n > 4 * kSplitItems && !p.plan_len)
        CHECK(chunks[0] == std::min(p.first_chunk, kSplitItems) && chunks[1] == chunks[0]);
      if ((ramp & 1) && heavy && n > 6 * kSplitItems && !p.plan_len)
        CHECK(chunks[0] == kSplitItems && chunks[1] == kSplitItems && chunks[2] == std::min(2 * kSplitItems, p.chunk));
    }
  }
}

// the strided gather: non-temporal and plain stores give the same bytes for every width, stride,
// source misalignment and count the pipeline uses (and for the widths that fall back to plain stores)
static void gather() {
  std::mt19937 rng(11);
  std::vector<uint8_t> src(1 << 20), a(1 << 19), b(1 << 19);
  for (auto& x : src) x = (uint8_t)rng();
  const size_t widths[] = {32, 96, 64, 128, 48, 1};
  for (size_t bytes : widths)
    for (int iter = 0; iter < 40; iter++) {
      const size_t stride = bytes + (rng() % 5) * 32 + (iter & 1 ? rng() % 17 : 0);  // 160, 192, 320, 352 ... and odd ones
      const size_t count = 1 + rng() % 1500, off = rng() % 64;
      if (off + stride * count > src.size() || bytes * count + 64 > a.size()) continue;
      uint8_t* da = a.data() + ((64 - (uintptr_t)a.data() % 64) % 64);  // 64-byte aligned, as the slots are
      uint8_t* db = b.data() + ((64 - (uintptr_t)b.data() % 64) % 64);
      std::fill(a.begin(), a.end(), 0xAA);
      std::fill(b.begin(), b.end(), 0xAA);
      copy_strided(da, src.data() + off, stride, bytes, count);
      copy_strided_plain(db, src.data() + off, stride, bytes, count);
      CHECK(std::memcmp(da, db, bytes * count + 64) == 0);  // (incl. nothing written behind the block)
      for (size_t i = 0; i < count; i += 97) CHECK(std::memcmp(da + bytes * i, src.data() + off + stride * i, bytes) == 0);
    }
}

int main() {
  gather();
  plans();
  copy_pool();
  leases_and_turns();
  std::printf("ok\n");
  return 0;
}
