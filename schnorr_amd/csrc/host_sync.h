// host_sync.h — the HIP-free part of the host pipeline (dsv_pipeline.h: run_pipelined): the copy-thread pool,
// the order in which calls in flight get a pipe and their turn on the compute lanes, and the chunk /
// sub-batch plan of a call.  Plain C++17 so that it can be exercised WITHOUT a GPU under
// ThreadSanitizer (tests/cpp/test_host_sync.cpp, run by tests/test_host_sync.py): GPU sanitizers are
// not available on this pool, and this is exactly the code where a data race would hide.
#pragma once
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace dsv {

constexpr int kPipes = 2;  // host calls in flight per device (each owns a Pipe)
constexpr size_t kSplitItems = (size_t)1 << 16;      // one sub-batch: 1024 waves, ONE wave per SIMD
constexpr size_t kPipeSmallCall = (size_t)1 << 16;   // up to here a call is ONE chunk on one stream

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- NUMA placement of a device's host-side work (r06) --------------------------------------------
// An 8-GPU host has two sockets; a GPU's copy threads should run — and its pinned staging lie — on the
// socket its PCIe root hangs off, or every gathered byte crosses the socket interconnect once more on
// its way to the DMA engine.  The mapping comes from sysfs: bus/pci/devices/<bdf>/numa_node and
// devices/system/node/node<N>/cpulist under `root` (normally /sys; tests point it at a fake tree).
// Plain file reads, no libnuma.  Pinned staging needs no code here: hipHostMalloc already allocates on
// the node nearest the current device unless hipHostMallocNumaUser is passed.
inline bool read_small_file(const std::string& path, std::string& out) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096];
  const size_t got = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[got] = 0;
  out = buf;
  return true;
}
// NUMA node of a PCI device ("0000:c1:00.0"); -1: unknown (no such file, or the kernel says -1)
inline int numa_node_of_pci(const std::string& root, const std::string& bdf) {
  std::string lower = bdf, text;
  for (auto& ch : lower) ch = (char)tolower((unsigned char)ch);
  if (!read_small_file(root + "/bus/pci/devices/" + lower + "/numa_node", text)) return -1;
  char* end = nullptr;
  const long v = strtol(text.c_str(), &end, 10);
  return end == text.c_str() || v < 0 ? -1 : (int)v;
}
// "0-15,64-79" -> cpu numbers (empty on any parse error: the caller then pins nothing)
inline std::vector<int> parse_cpulist(const std::string& text) {
  std::vector<int> cpus;
  const char* p = text.c_str();
  while (*p && *p != '\n') {
    char* end = nullptr;
    const long lo = strtol(p, &end, 10);
    if (end == p || lo < 0) return {};
    long hi = lo;
    p = end;
    if (*p == '-') {
      hi = strtol(p + 1, &end, 10);
      if (end == p + 1 || hi < lo) return {};
      p = end;
    }
    if (hi - lo > 4096) return {};
    for (long c = lo; c <= hi; c++) cpus.push_back((int)c);
    if (*p == ',') p++;
    else if (*p && *p != '\n') return {};
  }
  return cpus;
}
inline std::vector<int> cpus_of_numa_node(const std::string& root, int node) {
  std::string text;
  if (node < 0 || !read_small_file(root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", text)) return {};
  return parse_cpulist(text);
}
// restrict the CALLING thread to `cpus` (those of them the process may use at all); false: nothing changed
inline bool pin_this_thread(const std::vector<int>& cpus) {
  if (cpus.empty()) return false;
  cpu_set_t allowed, want;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
  CPU_ZERO(&want);
  int n = 0;
  for (int c : cpus)
    if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {
      CPU_SET(c, &want);
      n++;
    }
  if (n == 0) return false;  // (a container's cpuset on the other socket: leave the thread where it may run)
  return pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
}

// Copy threads of the host path: the caller's (pageable) arrays are gathered into pinned staging
// by several threads at once, so the DMA engine is fed faster than one memcpy stream can.
class CopyPool {
 public:
  ~CopyPool() { stop(); }
  // fn(t, T) runs on T threads (t = 0 is the caller); returns when all are done.
  // Between the chunks of one call the workers SPIN for a short while before they block: a chunk
  // arrives every 0.3 - 1 ms, and a condition-variable wake-up (30 - 100 us, twice per chunk: start
  // and completion) is a tenth of a first chunk's whole gather — on the path the GPU is waiting for.
  void run(int T, const std::function<void(int, int)>& fn) {
    if (T <= 1) {
      fn(0, 1);
      return;
    }
    {
      std::unique_lock<std::mutex> lk(m_);
      while ((int)th_.size() < T - 1) {
        const int id = (int)th_.size() + 1;
        th_.emplace_back([this, id] { loop(id); });
      }
      job_ = &fn;
      job_threads_ = T;
      pending_.store(T - 1, std::memory_order_relaxed);
      gen_.fetch_add(1, std::memory_order_release);
    }
    go_.notify_all();
    fn(0, T);
    for (int spin = 0; spin < kSpin && pending_.load(std::memory_order_acquire) != 0; spin++) cpu_relax();
    std::unique_lock<std::mutex> lk(m_);
    if (pending_.load(std::memory_order_acquire) != 0)
      done_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) == 0; });
    job_ = nullptr;  // (under the lock: a worker that does not take part in this job may be reading it)
  }
  // the workers (not the caller, thread 0 of a job: its placement is the application's) run on these cpus
  // from their next start on; call before the first run()
  void set_affinity(const std::vector<int>& cpus) {
    std::unique_lock<std::mutex> lk(m_);
    cpus_ = cpus;
  }
  void stop() {
    {
      std::unique_lock<std::mutex> lk(m_);
      quit_ = true;
      gen_.fetch_add(1, std::memory_order_release);
    }
    go_.notify_all();
    for (auto& t : th_) t.join();
    th_.clear();
    quit_ = false;
  }

 private:
  static constexpr int kSpin = 20000;  // ~0.2 - 0.4 ms of pause instructions
  static void cpu_relax() { __builtin_ia32_pause(); }
  void loop(int id) {
    {
      std::vector<int> cpus;
      {
        std::unique_lock<std::mutex> lk(m_);
        cpus = cpus_;
      }
      (void)pin_this_thread(cpus);
    }
    uint64_t seen = gen_.load(std::memory_order_acquire) - 1;  // started while a job is being posted: take it
    for (;;) {
      for (int spin = 0; spin < kSpin && gen_.load(std::memory_order_acquire) == seen; spin++) cpu_relax();
      const std::function<void(int, int)>* job;
      int T;
      {
        std::unique_lock<std::mutex> lk(m_);
        go_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
        seen = gen_.load(std::memory_order_acquire);
        if (quit_) return;
        job = job_;
        T = job_threads_;
      }
      if (job && id < T) {
        (*job)(id, T);
        if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
          std::unique_lock<std::mutex> lk(m_);  // (the waiter may be between its check and its wait)
          done_.notify_one();
        }
      }
    }
  }
  std::vector<std::thread> th_;
  std::vector<int> cpus_;
  std::mutex m_;
  std::condition_variable go_, done_;
  const std::function<void(int, int)>* job_ = nullptr;
  int job_threads_ = 0;
  std::atomic<int> pending_{0};
  std::atomic<uint64_t> gen_{0};
  bool quit_ = false;
};


// ---- the strided gather ----------------------------------------------------------------------
// `count` items of `bytes` bytes, `stride` apart, packed densely into dst (the typed objects of a
// language binding: one field out of every struct)
inline void copy_strided_plain(uint8_t* dst, const uint8_t* src, size_t stride, size_t bytes, size_t count) {
  switch (bytes) {  // constant sizes: the copies are inlined vector moves
    case 32:
      for (size_t i = 0; i < count; i++) memcpy(dst + 32 * i, src + stride * i, 32);
      break;
    case 96:
      for (size_t i = 0; i < count; i++) memcpy(dst + 96 * i, src + stride * i, 96);
      break;
    default:
      for (size_t i = 0; i < count; i++) memcpy(dst + bytes * i, src + stride * i, bytes);
  }
}
// The same with non-temporal stores (dst 32-byte aligned, bytes a multiple of 32): the pinned staging
// block is written once and read by the DMA engine only — ordinary stores first READ every line they
// are about to overwrite (read-for-ownership: 40 % of the gather's memory traffic) and push the
// caller's objects out of the cache.
__attribute__((target("avx2"))) inline void copy_strided_nt(uint8_t* dst, const uint8_t* src, size_t stride,
                                                           size_t bytes, size_t count) {
  if (bytes == 32) {
    for (size_t i = 0; i < count; i++)
      _mm256_stream_si256((__m256i*)(dst + 32 * i), _mm256_loadu_si256((const __m256i*)(src + stride * i)));
  } else if (bytes == 96) {
    for (size_t i = 0; i < count; i++) {
      const __m256i a = _mm256_loadu_si256((const __m256i*)(src + stride * i));
      const __m256i b = _mm256_loadu_si256((const __m256i*)(src + stride * i + 32));
      const __m256i c = _mm256_loadu_si256((const __m256i*)(src + stride * i + 64));
      _mm256_stream_si256((__m256i*)(dst + 96 * i), a);
      _mm256_stream_si256((__m256i*)(dst + 96 * i + 32), b);
      _mm256_stream_si256((__m256i*)(dst + 96 * i + 64), c);
    }
  } else {
    for (size_t i = 0; i < count; i++)
      for (size_t o = 0; o < bytes; o += 32)
        _mm256_stream_si256((__m256i*)(dst + bytes * i + o), _mm256_loadu_si256((const __m256i*)(src + stride * i + o)));
  }
  _mm_sfence();  // the stores are globally visible before the transfer is enqueued
}
inline void copy_strided(uint8_t* dst, const uint8_t* src, size_t stride, size_t bytes, size_t count) {
  static const bool nt = [] {
    const char* e = getenv("DSV_GATHER_NT");  // 0: ordinary stores (A/B)
    return !(e && strcmp(e, "0") == 0) && __builtin_cpu_supports("avx2");
  }();
  if (nt && (bytes & 31) == 0 && ((uintptr_t)dst & 31) == 0) copy_strided_nt(dst, src, stride, bytes, count);
  else copy_strided_plain(dst, src, stride, bytes, count);
}

// ---- who runs when ---------------------------------------------------------------------------
// One per device.  Calls take a ticket; a pipe is handed out first come first served while fewer than
// kPipes calls are in flight; multi-chunk calls also take their place in the order in which calls
// enqueue compute (their TURN), in the same critical section, so pipes and turns follow one order.
struct PipeSync {
  std::mutex mu;
  std::condition_variable cv;
  uint64_t ticket_next = 0, ticket_serving = 0;
  uint64_t turn_next = 0, turn_serving = 0;
  int busy_count = 0;
  bool busy[kPipes] = {};
  // (shutdown) nobody holds a pipe and nobody waits for one
  bool idle() const { return busy_count == 0 && ticket_next == ticket_serving; }
};

// Whose turn it is to enqueue compute on the device's lanes.  The holder enqueues ALL its chunks, then
// passes the turn on; the next call meanwhile gathers and transfers its first chunks (up to its slots)
// and enqueues them the moment the turn arrives — behind the holder's last chunks in the lanes' queues,
// so its ramp runs under the holder's tail.  (Without turns two calls in flight share the lanes chunk by
// chunk, advance in lock-step and finish together: both ramps and both tails coincide — measured, r05.)
struct TurnTicket {
  PipeSync& sync;
  uint64_t mine = 0;
  bool taken = false, held = false, released = false;
  explicit TurnTicket(PipeSync& s) : sync(s) {}
  void take(uint64_t ticket) {  // (by PipeLease, under sync.mu)
    mine = ticket;
    taken = true;
  }
  bool try_acquire() {
    if (held) return true;
    std::lock_guard<std::mutex> lk(sync.mu);
    held = sync.turn_serving == mine;
    return held;
  }
  void acquire() {
    if (held) return;
    std::unique_lock<std::mutex> lk(sync.mu);
    sync.cv.wait(lk, [&] { return sync.turn_serving == mine; });
    held = true;
  }
  void release() {  // (also on error paths: the turn must reach the calls behind this one)
    if (!taken || released) return;
    acquire();
    {
      std::lock_guard<std::mutex> lk(sync.mu);
      sync.turn_serving++;
    }
    released = true;
    sync.cv.notify_all();
  }
  ~TurnTicket() { release(); }
  TurnTicket(const TurnTicket&) = delete;
  TurnTicket& operator=(const TurnTicket&) = delete;
};

// A host call's lease on one of the device's pipes: FIFO by ticket, blocks while kPipes calls are in
// flight; released (and the next waiter woken) on scope exit.
struct PipeLease {
  PipeSync& sync;
  int index = -1;     // which pipe
  bool alone = true;  // no other call held a pipe of this device when this one got its own
  // want_turn (may be null): the call also takes its turn ticket
  PipeLease(PipeSync& s, TurnTicket* want_turn) : sync(s) {
    std::unique_lock<std::mutex> lk(sync.mu);
    const uint64_t mine = sync.ticket_next++;
    sync.cv.wait(lk, [&] { return sync.ticket_serving == mine && sync.busy_count < kPipes; });
    sync.ticket_serving++;
    if (want_turn) want_turn->take(sync.turn_next++);
    for (int k = 0; k < kPipes; k++)
      if (!sync.busy[k]) {
        index = k;
        break;
      }
    sync.busy[index] = true;
    alone = sync.busy_count == 0;
    sync.busy_count++;
    lk.unlock();
    sync.cv.notify_all();  // the next ticket may find the other pipe free
  }
  ~PipeLease() {
    {
      std::lock_guard<std::mutex> lk(sync.mu);
      sync.busy[index] = false;
      sync.busy_count--;
    }
    sync.cv.notify_all();
  }
  PipeLease(const PipeLease&) = delete;
  PipeLease& operator=(const PipeLease&) = delete;
};

// ---- the chunks and sub-batches of one call ------------------------------------------------------
struct PlanParams {
  size_t chunk = (size_t)1 << 18;        // pipeline chunk (DSV_PIPE_CHUNK_LOG2)
  size_t first_chunk = (size_t)1 << 15;  // first chunk of a call that finds the GPU idle (DSV_PIPE_FIRST_LOG2)
  double growth = 0;                     // DSV_PIPE_GROWTH (A/B): geometric ramp factor, 0 = the backlog rule
  int plan[16] = {};                     // DSV_PIPE_PLAN (A/B): explicit log2 sizes, the last one repeats
  int plan_len = 0;
};
// The chunks of one call.
//   ramp (the call found the GPU idle): two first chunks of 2^15 items (the GPU starts after ~0.5 ms of
//     staging, the second lane half a millisecond later), then chunks of ONE sub-batch (2^16 items) — growing only with what is already staged
//     (an eighth of it: 2^17 from 2^20 items on, 2^18 from 2^21).  A chunk is gathered, transferred and
//     preprocessed as a unit, ~12 ns per item before its first kernel can start against ~12 ns per item
//     of GPU work, so a chunk must stay well below the backlog the GPU still has: with the r01 - r04
//     doubling (2^15 .. 2^18) the fourth chunk arrived ~1 ms after the GPU had run dry
//     (profiles/r05/ab_chunk_plans.txt; the same box, one-shot calls: 16.6 -> 15.8 ms per 2^20).
//   flat (behind another call in flight): full chunks at once — there is no idle GPU to feed quickly
//     and whole chunks cost fewer launches (two in flight: 14.2 against 14.9 ms per 2^20 with 2^16).
// A remainder of at most a quarter chunk (or half a sub-batch) is merged into the last chunk instead of
// trailing behind it as a part of its own on ONE lane.
inline std::vector<size_t> plan_chunks(const PlanParams& p, size_t n, bool ramp, size_t unit = kSplitItems,
                                       bool heavy = false) {
  std::vector<size_t> out;
  if (n <= kPipeSmallCall) {  // one small call: a single chunk
    out.push_back(n);
    return out;
  }
  size_t left = n, staged = 0;
  double want_f = (double)p.first_chunk;
  for (size_t c = 0; left; c++) {
    size_t want = p.chunk;
    if (ramp && p.plan_len) {  // DSV_PIPE_PLAN: explicit sizes, the last one repeats
      want = (size_t)1 << p.plan[c < (size_t)p.plan_len ? c : (size_t)p.plan_len - 1];
    } else if (ramp && p.growth > 0) {  // DSV_PIPE_GROWTH: geometric from the first chunk
      want = align_up((size_t)want_f, 4096);
      want_f *= p.growth;
    } else if (ramp && heavy) {
      // double / var-generator items: two chunks of one sub-batch, then chunks of two.  Their kernels are
      // latency-bound below a sub-batch (k_verify_var takes 1.2 ms for 2^15 items and 1.4 ms for 2^16), so
      // the small first chunks that pay for single signatures only add launches here.  r06, same box, one-shot
      // from typed objects: var-generator 2^18 items 7.01 -> 6.23 ms (0.69 -> 0.78 x the device-resident rate),
      // double 2^20 items 28.98 -> 27.88 ms (0.835 -> 0.868 x); single signatures: equal or worse (15.4 -> 15.9)
      want = c < 2 ? unit : 2 * unit;
    } else if (ramp) {
      // (chunks of TWO sub-batches, one per lane at a time, from a first chunk of 2 x 2^15: 17.3 against
      //  15.8 ms per 2^20 one-shot, same box — the staging latency of the larger chunks outweighs it)
      // two first chunks of 2^15, one per lane, then one sub-batch per chunk
      want = c < 2 ? (p.first_chunk < unit ? p.first_chunk : unit) : unit;
      while (want * 2 <= staged / 8) want *= 2;
    }
    if (want > p.chunk) want = p.chunk;
    if (left <= want + unit / 2 || left <= want + want / 4) want = left;  // the rest rides along
    out.push_back(want);
    left -= want;
    staged += want;
  }
  return out;
}
// A chunk is cut into sub-batches of at most kSplitItems items — an EVEN number of equal ones once it
// holds more than one, so that both compute lanes get the same work from every chunk and finish the
// call together.
inline size_t plan_parts(size_t cnt, bool one_part, size_t cap, size_t& part_items) {
  if (one_part || cnt <= cap) {
    part_items = cnt;
    return 1;
  }
  size_t parts = (cnt + cap - 1) / cap;
  parts += parts & 1;
  part_items = align_up((cnt + parts - 1) / parts, 256);
  return (cnt + part_items - 1) / part_items;
}

}  // namespace dsv
