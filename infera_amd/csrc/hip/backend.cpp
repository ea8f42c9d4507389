#include "backend.hpp"

#include <sched.h>
#include <sys/prctl.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <sstream>
#include <thread>

#include "../host/common.hpp"

namespace infera_hip {

namespace {

// A failed HIP call.  The text is what a caller sees (error.rs:24-25 "ONNX error: ..."); the code lets the host path tell a device fault
// (the slot is taken out of service and the call re-dealt to another GPU) from an allocation failure (the call fails, the GPU stays).
struct HipFault : InferaError {
  hipError_t code;
  HipFault(hipError_t e, const char *what) : InferaError(InferaError::onnx(std::string("HIP: ") + what + ": " + hipGetErrorString(e))), code(e) {}
};
[[noreturn]] void hip_fail(hipError_t e, const char *what) { throw HipFault(e, what); }
#define HIP_TRY(expr)                               \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) hip_fail(_e, #expr);      \
  } while (0)

// hipGraph mode only: allocation / free / device-wide synchronisation from ANY thread invalidates an
// open stream capture on ROCm 7.2 even in ThreadLocal capture mode ("operation failed due to a previous
// error during capture").  Captures therefore hold this lock shared, and the operations that would
// break them hold it exclusively.  In the default direct-enqueue mode nobody captures and the guards
// are not taken.
std::shared_mutex g_capture_mu;
struct UnsafeOpGuard {
  std::unique_lock<std::shared_mutex> lk;
  UnsafeOpGuard() {
    if (Config::get().use_hipgraph) lk = std::unique_lock<std::shared_mutex>(g_capture_mu);
  }
};

constexpr size_t kHostPassBytes = 64ull << 20;     // pinned staging per direction per thread
constexpr size_t kPipePassBytes = 16ull << 20;     // pass size of the two-slot pipeline used for larger host inputs
constexpr size_t kScratchBudgetBytes = 8ull << 30; // activation scratch per thread for unfused plans
// Pinned staging a GPU slot's contexts may hold for big-row (BLOB) batches, all of them together: contexts are pooled for the life of the
// process and never shrink, so without a bound a host with many worker threads serving image batches locks RAM in proportion to its thread
// count (24 contexts x 308 MB per GPU).  Each context gets an equal share (budget / INFERA_HOST_CONTEXTS); the pipeline pass
// shrinks to fit it.  6 GiB / 24 = 256 MB = 221 ResNet-sized images per pass and direction (256 was the measured optimum: -1 %).
constexpr size_t kPinnedBudgetPerSlot = 6ull << 30;
std::atomic<uint64_t> g_pinned_bytes[64];  // per device slot: pinned staging held by its contexts right now (infera_hip_get_devices)

// ---------------------------------------------------------------------------------------------
// per-thread, per-device execution context (stream + staging + scratch)
// ---------------------------------------------------------------------------------------------
struct ThreadCtx {
  int device = -1;
  int slot = 0;
  hipStream_t stream = nullptr;
  float *pin_in = nullptr, *pin_out = nullptr, *dev_in = nullptr, *dev_out = nullptr, *scratch = nullptr, *dev_cm = nullptr;
  size_t pin_in_cap = 0, pin_out_cap = 0, dev_in_cap = 0, dev_out_cap = 0, scratch_cap = 0, dev_cm_cap = 0;  // bytes
  // hipGraph per (model uid, rows): {H2D memcpy, kernels, D2H memcpy} captured once on this context's
  // stream and buffers, replayed for every later DataChunk of that shape (one API call per chunk
  // instead of one per node).  Any reallocation of the buffers the graph points at drops the cache.
  struct GraphEntry {
    uint64_t uid;
    int64_t rows;  // (bit 62 set: the graph was captured for a column-major chunk)
    hipGraphExec_t exec;
    uint64_t last_use;
  };
  std::vector<GraphEntry> graphs;
  uint64_t graph_clock = 0;
  hipEvent_t pipe_ev[2] = {nullptr, nullptr};  // completion of the pass that last used staging slot 0 / 1
  // the big-row pipeline's H2D copies run on the slot's shared copy stream (big_copy_stream); these mark a pass's copy on it
  hipEvent_t h2d_ev[2] = {nullptr, nullptr};
  // second lane of a long convolutional pass (exec_plan): its own stream, forked from / joined into `stream` by events
  static constexpr int kMaxLanes = 2;  // (three and four lanes: no better than one, profiles/r04_conv_lanes_ab.txt)
  hipStream_t lane_stream[kMaxLanes - 1] = {nullptr};
  hipEvent_t lane_ev[kMaxLanes] = {nullptr, nullptr};  // [0]: fork; [i]: lane i done
  hipEvent_t poll_ev = nullptr;                // completion marker of a host-ABI call, queried between naps
  // How a host-ABI call waits for its chunk: it NAPS.  ROCm 7.2's "blocking" event wait (hipEventSynchronize on a hipEventBlockingSync event)
  // burns the core for the whole wait, and so does hipStreamSynchronize: 277 us of CPU per chunk at 16 callers against 85 with naps at the same
  // rows/s (profiles/r03_host_cpu_ab_wait_gather.txt) -- under a CPU quota (16 CPUs feeding 8 GPUs) CPU time per chunk is what bounds the
  // scan.  Nap for most of what this context's recent waits OF THE SAME KIND took (`key`: model and row count -- a context that served a
  // 30 ms image batch must not sleep 2 ms on the 50 us table chunk that follows it), then query between short naps: one or two
  // clock_nanosleep calls and a few queries per chunk.  The first nap is bounded by the SHORTEST of the recent waits as well as by their
  // average (a call that finishes faster than the average must not oversleep).  With no estimate (first wait of a kind on this context)
  // the naps grow with the time already waited (a quarter of it, 3..200 us).
  struct WaitEstimate {
    double ema_ns = 0.0, min_ns = 0.0;  // average and (slowly rising) minimum of the recent waits
    uint64_t key = 0;
  };
  WaitEstimate wait_est, pipe_est;
  static constexpr double kPollFirst = 0.75, kPollNext = 0.1;  // shares of the expected wait (0.6-0.9 / 0.05-0.2 measured equal)
  // The naps need a timer slack of ~1 us (the default 50 us would turn a 20 us nap into 70).  The slack belongs to the CALLER's thread -- a
  // DuckDB worker -- so it is set for the duration of the wait only and restored afterwards.
  struct TimerSlack {
    long saved = -1;
    TimerSlack() {
      saved = prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL);
      if (saved > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
      else saved = -1;
    }
    ~TimerSlack() {
      if (saved > 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)saved, 0UL, 0UL, 0UL);
    }
  };
  template <class Query>
  static void poll_until(Query &&query, WaitEstimate &est, uint64_t key) {
    if (key != est.key || key == 0) {
      est.key = key;
      est.ema_ns = est.min_ns = 0.0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto waited_ns = [&] { return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count(); };
    auto nap = [](double ns) {
      if (ns < 1500.0) return;
      const long long n = (long long)ns;
      timespec ts{time_t(n / 1000000000LL), long(n % 1000000000LL)};
      (void)clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, nullptr);
    };
    const bool known = est.ema_ns > 0.0;
    {
      TimerSlack slack;
      if (known) nap(std::min({est.ema_ns * kPollFirst, est.min_ns * 0.9, 2.0e6}));
      for (;;) {
        const hipError_t e = query();
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) hip_fail(e, "hipEventQuery");
        nap(known ? std::max(3000.0, std::min(est.ema_ns * kPollNext, 50000.0)) : std::max(3000.0, std::min(waited_ns() * 0.25, 200000.0)));
      }
    }
    (void)hipGetLastError();  // hipErrorNotReady from the queries must not surface at the next launch check
    const double waited = waited_ns();
    est.ema_ns = known ? 0.75 * est.ema_ns + 0.25 * waited : waited;
    est.min_ns = known ? std::min(waited, est.min_ns * 1.05) : waited;
  }
  // waits for `ev` (already recorded)
  void wait_event(hipEvent_t ev, WaitEstimate &est, uint64_t key) {
    poll_until([&] { return hipEventQuery(ev); }, est, key);
  }
  // waits for everything enqueued on `stream` so far.  `key` identifies the kind of work (0 = unknown)
  void wait_stream(uint64_t key = 0) {
    if (!poll_ev) HIP_TRY(hipEventCreateWithFlags(&poll_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(poll_ev, stream));
    poll_until([&] { return hipEventQuery(poll_ev); }, wait_est, key);
  }
  void drop_graphs() {
    for (auto &g : graphs) (void)hipGraphExecDestroy(g.exec);
    graphs.clear();
  }

  void ensure_pinned(float *&p, size_t &cap, size_t bytes) {
    if (bytes <= cap) return;
    UnsafeOpGuard guard;
    drop_graphs();
    if (p) HIP_TRY(hipHostFree(p));
    p = nullptr;
    g_pinned_bytes[size_t(slot) % 64].fetch_sub(cap, std::memory_order_relaxed);
    cap = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p), bytes, hipHostMallocDefault));  // (host-coherent: kernels may store results into it)
    cap = bytes;
    g_pinned_bytes[size_t(slot) % 64].fetch_add(bytes, std::memory_order_relaxed);
  }
  void ensure_dev(float *&p, size_t &cap, size_t bytes) {
    if (bytes <= cap) return;
    UnsafeOpGuard guard;
    drop_graphs();
    if (p) {
      HIP_TRY(hipStreamSynchronize(stream));
      HIP_TRY(hipFree(p));
    }
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), bytes));
    cap = bytes;
  }
};

// Contexts are owned by a process-lifetime pool (never destroyed: tearing HIP objects down from
// thread-exit / atexit handlers races the runtime's own shutdown).  A thread returns its contexts
// to the pool when it exits so short-lived threads do not grow it.
std::mutex g_pool_mu;
std::vector<std::vector<ThreadCtx *>> g_pool;  // [device slot] -> free contexts

extern std::atomic<int> g_slot_threads[64];
struct ThreadHolder {
  std::vector<ThreadCtx *> by_slot;
  int home_slot = -1;
  ~ThreadHolder() {
    if (home_slot >= 0) g_slot_threads[size_t(home_slot) % 64].fetch_sub(1, std::memory_order_relaxed);
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t s = 0; s < by_slot.size(); s++)
      if (by_slot[s]) g_pool[s].push_back(by_slot[s]);
  }
};
thread_local ThreadHolder t_holder;
std::atomic<unsigned> g_next_home{0};

int slot_of_ordinal(int ordinal) {
  const auto &ds = devices();
  for (size_t i = 0; i < ds.ids.size(); i++)
    if (ds.ids[i] == ordinal) return int(i);
  throw InferaError::onnx("HIP device " + std::to_string(ordinal) + " is not among the selected devices");
}

ThreadCtx &ctx_for_slot(int slot) {
  const auto &ds = devices();
  if (t_holder.by_slot.size() < ds.ids.size()) t_holder.by_slot.resize(ds.ids.size(), nullptr);
  ThreadCtx *&c = t_holder.by_slot[size_t(slot)];
  HIP_TRY(hipSetDevice(ds.ids[size_t(slot)]));
  if (!c) {
    {
      std::lock_guard<std::mutex> lk(g_pool_mu);
      if (g_pool.size() < ds.ids.size()) g_pool.resize(ds.ids.size());
      if (!g_pool[size_t(slot)].empty()) {
        c = g_pool[size_t(slot)].back();
        g_pool[size_t(slot)].pop_back();
      }
    }
    if (!c) {
      auto *n = new ThreadCtx();
      n->device = ds.ids[size_t(slot)];
      n->slot = slot;
      hipError_t e = hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking);
      if (e != hipSuccess) {
        delete n;
        hip_fail(e, "hipStreamCreateWithFlags");
      }
      c = n;
    }
  }
  return *c;
}

// Host-ABI calls do not own a context per caller thread: a DuckDB scan on a 256-thread host would pin 256 sets of
// staging buffers and per-model scratch (ResNet-18: ~1 GB each) and create 256 streams, for no throughput -- the path
// saturates at ~16 callers per GPU.  They lease one of at most INFERA_HOST_CONTEXTS (default 24) contexts per GPU for
// the duration of the call; the rest wait.  (The device-resident entry points keep the caller thread's own stream.)
struct HostPool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<ThreadCtx *> free;
  int created = 0;
};
HostPool &host_pool(int slot) {
  static HostPool pools[64];
  return pools[size_t(slot) % 64];
}
thread_local ThreadCtx *t_last_ctx[64];  // per device slot: the staging context this thread leased last (never dereferenced: an identity)
struct HostLease {
  HostPool &pool;
  ThreadCtx *c = nullptr;
  explicit HostLease(int slot) : pool(host_pool(slot)) {
    const auto &ds = devices();
    HIP_TRY(hipSetDevice(ds.ids[size_t(slot)]));
    const int cap = Config::get().host_contexts;
    {
      std::unique_lock<std::mutex> lk(pool.mu);
      pool.cv.wait(lk, [&] { return !pool.free.empty() || pool.created < cap; });
      if (!pool.free.empty()) {
        // the context this thread used last, when it is free: its pinned staging lines are (still) in THIS core's caches -- a chunk
        // gathered into the buffer another core wrote last pays a cache-to-cache transfer per line (gather 42 -> 50 us per chunk
        // already at 2 caller threads with plain LIFO reuse; CPU per chunk 88.9 -> 76.9 us at 16 callers, profiles/r03_host_cpu_ab_ctx_affinity.txt).
        size_t pick = pool.free.size() - 1;
        if (size_t(slot) < 64 && t_last_ctx[slot])
          for (size_t i = pool.free.size(); i-- > 0;)
            if (pool.free[i] == t_last_ctx[slot]) {
              pick = i;
              break;
            }
        c = pool.free[pick];
        pool.free.erase(pool.free.begin() + long(pick));
        if (size_t(slot) < 64) t_last_ctx[slot] = c;
        return;
      }
      pool.created++;
    }
    auto *n = new ThreadCtx();
    n->device = ds.ids[size_t(slot)];
    n->slot = slot;
    const hipError_t e = hipStreamCreateWithFlags(&n->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete n;
      {
        std::lock_guard<std::mutex> lk(pool.mu);
        pool.created--;
      }
      pool.cv.notify_one();
      hip_fail(e, "hipStreamCreateWithFlags");
    }
    c = n;
    if (size_t(slot) < 64) t_last_ctx[slot] = c;
  }
  ~HostLease() {
    if (!c) return;
    {
      std::lock_guard<std::mutex> lk(pool.mu);
      pool.free.push_back(c);
    }
    pool.cv.notify_one();
  }
  HostLease(const HostLease &) = delete;
  HostLease &operator=(const HostLease &) = delete;
};

// Host-ABI work served per device slot (calls, rows): lets a scan report how DuckDB's worker threads were dealt over
// the GPUs (infera_hip_get_devices), and lets the tests see that a second slot really took its share.
std::atomic<uint64_t> g_slot_calls[64], g_slot_rows[64];

// Where a host-ABI call's wall time goes (single-pass path = one DataChunk per call), summed over all calls, in ns:
// lease (waiting for a staging context), gather (caller's buffer -> pinned), gate (waiting for admission), enqueue
// (H2D + kernels [+ D2H] API calls), wait (until the device is done), copy_out (pinned -> result buffer).
// Seven steady_clock reads per call (~0.2 us) -- always on, reported by infera_hip_get_devices.
enum HostPhase { kPhLease, kPhGather, kPhGate, kPhEnqueue, kPhWait, kPhCopyOut, kPhCount };
std::atomic<uint64_t> g_phase_ns[kPhCount], g_phase_calls;
inline uint64_t now_ns() { return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count()); }

// NUMA node of the CPU the calling thread runs on right now (-1 = unknown), from /sys/devices/system/node/node*/cpulist
int current_numa_node() {
  static const std::vector<int> node_of_cpu = [] {
    std::vector<int> map;
    for (int node = 0; node < 64; node++) {
      FILE *fp = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
      if (!fp) continue;
      char buf[4096];
      const size_t n = std::fread(buf, 1, sizeof buf - 1, fp);
      std::fclose(fp);
      buf[n] = 0;
      for (char *p = buf; *p;) {  // "0-63,128-191"
        char *end;
        const long a = std::strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        if (*end == '-') b = std::strtol(end + 1, &end, 10);
        for (long c = a; c <= b && c < 4096; c++) {
          if (size_t(c) >= map.size()) map.resize(size_t(c) + 1, -1);
          map[size_t(c)] = node;
        }
        p = *end == ',' ? end + 1 : end;
        if (*end != ',') break;
      }
    }
    return map;
  }();
  const int cpu = sched_getcpu();
  return cpu >= 0 && size_t(cpu) < node_of_cpu.size() ? node_of_cpu[size_t(cpu)] : -1;
}

// caller threads whose home is slot i right now (a thread leaves when it exits)
std::atomic<int> g_slot_threads[64];

// Device-fault handling (SURVEY 5 "failure detection"): a HIP error on the host path other than an allocation failure takes the slot out
// of service -- its callers are re-dealt to the remaining slots and the failed chunk is run again there (the caller's buffers are only
// read, so a chunk can be staged twice); infera_hip_get_devices reports the slot as unhealthy with the error text.  With no healthy slot
// left the error goes to the caller as before (status -1 + last error, error.rs:13-61).
std::atomic<bool> g_slot_unhealthy[64];
std::mutex g_fault_mu;
std::string g_slot_fault[64];
int healthy_slots() {
  int n = 0;
  for (size_t i = 0; i < devices().ids.size() && i < 64; i++) n += !g_slot_unhealthy[i].load(std::memory_order_acquire);
  return n;
}
void mark_slot_unhealthy(int slot, const std::string &why) {
  {
    std::lock_guard<std::mutex> lk(g_fault_mu);
    if (g_slot_fault[size_t(slot) % 64].empty()) g_slot_fault[size_t(slot) % 64] = why;
  }
  if (!g_slot_unhealthy[size_t(slot) % 64].exchange(true, std::memory_order_acq_rel))
    log_msg(0, "device slot " + std::to_string(slot) + " (HIP device " + std::to_string(devices().ids[size_t(slot)]) + ") taken out of service: " + why);
}
// Which HIP errors mean "this GPU is gone or wedged" rather than "this call was refused": the sticky execution faults, a lost device, a
// dead context.  (An allocation failure or an invalid argument fails the call and leaves the GPU in service.)
bool is_device_fault(hipError_t e) {
  switch (e) {
    case hipErrorLaunchFailure: case hipErrorIllegalAddress: case hipErrorLaunchTimeOut: case hipErrorECCNotCorrectable: case hipErrorNoDevice:
    case hipErrorContextIsDestroyed: case hipErrorDeinitialized: case hipErrorUnknown: case hipErrorAssert:
      return true;
    default: return false;
  }
}
// TEST HOOK (tests/test_multi_device_gpu.py): INFERA_FAULT_INJECT=<slot>:<n> makes every host-ABI call on that slot after its n-th fail as a
// launch failure would.  Read once; unset (always, outside that test) it costs one relaxed load per call.
bool fault_injected(int slot) {
  static const std::pair<int, long> inj = [] {
    const char *e = getenv("INFERA_FAULT_INJECT");
    int sl = -1;
    long n = 0;
    if (e && std::sscanf(e, "%d:%ld", &sl, &n) == 2) return std::make_pair(sl, n);
    return std::make_pair(-1, 0L);
  }();
  if (inj.first != slot) return false;
  static std::atomic<long> calls{0};
  return calls.fetch_add(1, std::memory_order_relaxed) >= inj.second;
}

int home_slot() {
  if (t_holder.home_slot >= 0 && g_slot_unhealthy[size_t(t_holder.home_slot) % 64].load(std::memory_order_acquire)) {  // re-deal
    g_slot_threads[size_t(t_holder.home_slot) % 64].fetch_sub(1, std::memory_order_relaxed);
    t_holder.home_slot = -1;
  }
  if (t_holder.home_slot < 0) {
    const auto &ds = devices();
    const size_t n = ds.ids.size();
    // INFERA_NUMA_SLOTS=0: no NUMA preference at all (unpinned workers that all START on one socket, or a
    // taskset'ed process, must not leave the other socket's GPUs idle -- the balanced policy below already bounds that
    // imbalance to one thread per slot; the knob switches the preference off altogether)
    const int node = n > 1 && Config::get().numa_slots ? current_numa_node() : -1;
    std::vector<int> load(n);
    for (size_t i = 0; i < n; i++)  // (a slot that is out of service is never anybody's home while another one works)
      load[i] = g_slot_unhealthy[i % 64].load(std::memory_order_acquire) ? (1 << 28) : g_slot_threads[i % 64].load(std::memory_order_relaxed);
    t_holder.home_slot = choose_slot_balanced(ds.numa, load, node);
    g_slot_threads[size_t(t_holder.home_slot) % 64].fetch_add(1, std::memory_order_relaxed);
  }
  return t_holder.home_slot;
}

kern::ActParam act_of(const Step &s) { return kern::ActParam{int(s.act), s.act_a, s.act_b}; }
// A Dense layer is a 1x1 convolution over 1x1 "images": with H = W = 1 the channel-quad layout IS the row-major
// [rows, K] matrix, so the tiled conv kernel (packed weights through LDS, unit-pipelined MFMA stream) serves it.
// K and M are padded to multiples of 32 with zero weights; the real row lengths travel in kvalid / mvalid.
kern::ConvGeom dense_as_conv(const Step &s) {
  const int kp = int((s.K + 31) / 32 * 32), mp = int((s.M + 31) / 32 * 32);
  return kern::ConvGeom{kp, 1, 1, mp, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, int(s.K), int(s.M)};
}

// Weight upload on an explicit (non-blocking) stream: a legacy-stream hipMemcpy would try to
// synchronise with every blocking stream of the device, which is illegal while another thread is
// capturing a hipGraph ("would make the legacy stream depend on a capturing blocking stream").
float *upload(const std::vector<float> &v, hipStream_t stream) {
  if (v.empty()) return nullptr;
  float *d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), v.size() * sizeof(float)));
  hipError_t e = hipMemcpyAsync(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) {
    (void)hipFree(d);
    hip_fail(e, "hipMemcpy(weights)");
  }
  return d;
}

// Which buffers an executed step reads / writes once fusion decisions are applied.
struct EffStep {
  int idx;
  std::vector<int> reads;
  int writes;
};

std::vector<EffStep> effective_steps(const LoadedModel &m) {
  std::vector<EffStep> out;
  const auto &st = m.plan.steps;
  for (size_t i = 0; i < st.size(); i++) {
    switch (m.exec[i]) {
      case ExecKind::Skipped: break;
      case ExecKind::Mlp3Head: out.push_back({int(i), {st[i].in0}, st[i + 2].out}); break;
      case ExecKind::DenseSoftmax: out.push_back({int(i), {st[i].in0}, st[i + 1].out}); break;
      case ExecKind::ChainHead: out.push_back({int(i), {st[i].in0}, st[i + size_t(m.chain_at(i)->nsteps) - 1].out}); break;
      case ExecKind::ConvPatch:
        out.push_back({int(i), {st[i].in0}, i < m.conv_fused_pool.size() && m.conv_fused_pool[i] >= 0 ? st[size_t(m.conv_fused_pool[i])].out : st[i].out});
        break;
      case ExecKind::ConvTiled:
        if (i < m.conv_fold.size() && m.conv_fold[i] >= 0)  // the block's projection shortcut rides in this convolution: it reads that layer's input too
          out.push_back({int(i), {st[i].in0, st[size_t(m.conv_fold[i])].in0}, st[size_t(m.conv_fused_add[i])].out});
        else if (i < m.conv_fused_add.size() && m.conv_fused_add[i] >= 0)
          out.push_back({int(i), {st[i].in0, m.conv_residual_buf[i]}, st[size_t(m.conv_fused_add[i])].out});
        else
          out.push_back({int(i), {st[i].in0}, st[i].out});
        break;
      default: {
        EffStep e{int(i), {st[i].in0}, st[i].out};
        if (st[i].in1 >= 0) e.reads.push_back(st[i].in1);
        out.push_back(e);
      }
    }
  }
  return out;
}

void schedule(LoadedModel &m) {
  const auto &st = m.plan.steps;
  const size_t n = st.size();
  m.exec.assign(n, ExecKind::Normal);
  std::vector<int> uses(m.plan.buf_per_row.size(), 0);
  for (const auto &s : st) {
    if (s.in0 >= 0) uses[size_t(s.in0)]++;
    if (s.in1 >= 0) uses[size_t(s.in1)]++;
  }
  uses[size_t(m.plan.out_buf)]++;
  const Config &cfg = Config::get();
  bool have_mlp3 = false;
  for (size_t i = 0; i < n; i++) {
    if (m.exec[i] != ExecKind::Normal) continue;
    // Dense -> Dense -> Dense with private intermediates: whole-chain fused kernel
    if (cfg.fused_mlp && !have_mlp3 && i + 2 < n && st[i].kind == StepKind::Dense && st[i + 1].kind == StepKind::Dense &&
        st[i + 2].kind == StepKind::Dense && st[i + 1].in0 == st[i].out && st[i + 2].in0 == st[i + 1].out &&
        uses[size_t(st[i].out)] == 1 && uses[size_t(st[i + 1].out)] == 1) {
      kern::Mlp3Shape sh{int(st[i].K), int(st[i].M), int(st[i + 1].M), int(st[i + 2].M), int(st[i].act), int(st[i + 1].act),
                         int(st[i + 2].act)};
      std::string why;
      // only parameter-free activations can be baked into the fused chain (LeakyRelu/Clip carry arguments)
      const bool acts_ok = sh.act1 <= 3 && sh.act2 <= 3 && sh.act3 <= 3;
      if (acts_ok && kern::mlp3_supported(sh, &why)) {
        m.exec[i] = ExecKind::Mlp3Head;
        m.exec[i + 1] = m.exec[i + 2] = ExecKind::Skipped;
        m.mlp3_shape = sh;
        have_mlp3 = true;
        i += 2;
        continue;
      }
      if (!why.empty()) log_msg(2, "model '" + m.name + "': Dense x3 chain stays layer-by-layer: " + why);
    }
    // A run of small Dense layers over a table of any width (optionally behind the PadCols the lowering put in front of
    // a wide first layer, optionally ending in Softmax / ArgMax): one load-time specialised kernel reads the table once
    // and writes only the last layer (chain_device.inc).  Single layers stay with the ahead-of-time kernels unless the
    // chain also saves them the padding pass.
    if (cfg.fused_mlp) {
      size_t j = i;
      int pad = 0;
      if (st[j].kind == StepKind::PadCols && j + 1 < n && st[j + 1].kind == StepKind::Dense && st[j + 1].in0 == st[j].out &&
          uses[size_t(st[j].out)] == 1 && m.exec[j + 1] == ExecKind::Normal) {
        pad = 1;
        j++;
      }
      kern::ChainShape sh;
      sh.k0 = pad ? int(st[i].K) : int(st[j].K);
      const size_t d0 = j;
      while (j < n && st[j].kind == StepKind::Dense && m.exec[j] == ExecKind::Normal && int(st[j].act) <= kMaxMfmaFusedAct &&
             st[j].K <= 128 && st[j].M <= 128 && (j == d0 || (st[j].in0 == st[j - 1].out && uses[size_t(st[j - 1].out)] == 1))) {
        sh.dims.push_back(int(st[j].M));
        sh.acts.push_back(int(st[j].act));
        sh.pa.push_back(st[j].act_a);
        sh.pb.push_back(st[j].act_b);
        j++;
      }
      const size_t layers = j - d0;
      // a single layer with 17..32 outputs over rows the aligned kernels cannot read would fall to the generic kernel
      const bool tail_next = layers >= 1 && j < n && st[j].in0 == st[j - 1].out && uses[size_t(st[j - 1].out)] == 1 &&
                             (st[j].kind == StepKind::Softmax || st[j].kind == StepKind::ArgMax);
      // ... and a single 17..128-wide layer whose Softmax / ArgMax would otherwise cost two more passes over its scores
      const bool lone_gap = layers == 1 && !pad && st[d0].M > 16 && ((st[d0].M <= 32 && st[d0].K % 8 != 0) || tail_next);  // (layers == 1: d0 is a Dense step)
      if (layers >= 2 || (layers == 1 && pad) || lone_gap) {
        if (j < n && st[j].in0 == st[j - 1].out && uses[size_t(st[j - 1].out)] == 1 && m.exec[j] == ExecKind::Normal) {
          if (st[j].kind == StepKind::Softmax && st[j].sm_norm == 0 && st[j].sm_outer == 1 && st[j].sm_inner == 1 && st[j].sm_len == st[j - 1].M) {
            sh.sm = st[j].log_softmax ? 2 : 1;
            j++;
          } else if (st[j].kind == StepKind::ArgMax && st[j].K == st[j - 1].M) {
            sh.sm = 3;
            j++;
          }
        }
        std::string why;
        if (kern::chain_supported(sh, &why)) {
          LoadedModel::ChainRun run;
          run.first = int(i);
          run.nsteps = int(j - i);
          run.pad = pad;
          run.shape = sh;
          m.chains.push_back(run);
          m.exec[i] = ExecKind::ChainHead;
          for (size_t k = i + 1; k < j; k++) m.exec[k] = ExecKind::Skipped;
          i = j - 1;
          continue;
        }
        log_msg(2, "model '" + m.name + "': Dense chain at step " + std::to_string(i) + " stays layer-by-layer: " + why);
      }
    }
    // Dense + row Softmax over exactly its M outputs: softmax in the GEMM epilogue
    if (i + 1 < n && st[i].kind == StepKind::Dense && st[i + 1].kind == StepKind::Softmax && st[i + 1].in0 == st[i].out &&
        uses[size_t(st[i].out)] == 1 && st[i + 1].sm_outer == 1 && st[i + 1].sm_inner == 1 && st[i + 1].sm_len == st[i].M &&
        kern::dense_can_fuse_softmax(int(st[i].K), int(st[i].M)) && st[i + 1].sm_norm == 0) {
      m.exec[i] = ExecKind::DenseSoftmax;
      m.exec[i + 1] = ExecKind::Skipped;
      i += 1;
      continue;
    }
    // Dense + ArgMax over exactly its M scores (a classifier's label): the label is picked in the GEMM epilogue and
    // the scores never reach memory.  Both buffers stay planned: the launch falls back to the two kernels when the
    // input pointer it meets at run time cannot feed a kernel with that epilogue (dense_can_fuse_argmax).
    if (i + 1 < n && st[i].kind == StepKind::Dense && st[i + 1].kind == StepKind::ArgMax && st[i + 1].in0 == st[i].out &&
        uses[size_t(st[i].out)] == 1 && st[i + 1].K == st[i].M && st[i].M <= 64 &&
        kern::dense_can_fuse_argmax(nullptr, int(st[i].K), int(st[i].M))) {
      m.exec[i] = ExecKind::DenseArgMax;
      i += 1;
    }
  }
  // Remaining Dense layers with K % 32 == 0 and M % 32 == 0 -> the tiled kernel (the generic dense kernel fetches
  // one weight per lane per MFMA from L2 and measured 13-16 TFLOP/s; narrow heads keep their streaming kernels)
  for (size_t i = 0; i < n; i++)
    if (m.exec[i] == ExecKind::Normal && st[i].kind == StepKind::Dense && st[i].M > 32 && st[i].K % 4 == 0 &&
        mfma_fusable(st[i].act) && kern::conv2d_tiled_supported(dense_as_conv(st[i])))
      m.exec[i] = ExecKind::DenseTiled;
  // ---- layout decision for convolutional plans ----
  auto is4d = [&](int b) { return b >= 0 && m.plan.buf_shape[size_t(b)].size() == 4; };
  auto spatial = [&](int b) { return is4d(b) ? m.plan.buf_shape[size_t(b)][2] * m.plan.buf_shape[size_t(b)][3] : int64_t(1); };
  bool any_conv = false, ok = true;
  std::string nchw_reason;  // first thing that keeps a convolutional plan out of the channel-quad layout (logged: it costs ~10x)
  auto refuse = [&](const std::string &why) {
    if (ok) nchw_reason = why;
    ok = false;
  };
  std::vector<size_t> flat_dense;  // Dense layers fed by a flattened [C,H,W] activation
  // buffers that keep the caller's NCHW order: the input, and elementwise preprocessing of it (in-graph normalisation)
  m.nchw_buf.assign(m.plan.buf_shape.size(), 0);
  m.nchw_buf[0] = 1;
  auto elementwise = [](const Step &s) { return s.kind == StepKind::Unary || s.kind == StepKind::BinaryConst || s.kind == StepKind::AffineChannel; };
  for (const auto &s : st)
    if (elementwise(s) && s.in0 >= 0 && m.nchw_buf[size_t(s.in0)] && is4d(s.out) && s.out != m.plan.out_buf) m.nchw_buf[size_t(s.out)] = 1;
  for (const auto &s : st) {
    any_conv = any_conv || s.kind == StepKind::Conv2d;
    // (CopyCols = channel concat: a contiguous per-row block in NCHW and in channel-quad planes alike)
    const bool layout_free = s.kind == StepKind::Conv2d || s.kind == StepKind::Pool2d || s.kind == StepKind::GlobalAvgPool ||
                             s.kind == StepKind::BinaryAct || s.kind == StepKind::Unary || s.kind == StepKind::AffineChannel ||
                             s.kind == StepKind::CopyCols || s.kind == StepKind::SliceCols || s.kind == StepKind::LRN ||
                             s.kind == StepKind::ChannelShuffle ||
                             s.kind == StepKind::BinaryConst;  // (its per-row constant is permuted to channel-quad order below)
    // A channel slice is one contiguous block per sample in channel-quad planes only when it starts on a quad
    // boundary (its length is covered by the whole-quads check on the output tensor below): channels 2..5 of an
    // 8-channel tensor are NOT floats [2HW, 6HW) of the interleaved buffer.
    if (s.kind == StepKind::SliceCols && is4d(s.in0) && spatial(s.in0) > 1 && !m.nchw_buf[size_t(s.in0)] &&
        s.col_off % (4 * spatial(s.in0)) != 0)
      refuse("'" + s.origin + "' slices channels from an offset that is not a whole quad");
    for (int b : {s.in0, s.in1}) {
      if (b < 0) continue;
      if (m.nchw_buf[size_t(b)] && is4d(b) && spatial(b) > 1) {  // NCHW tensors are read by convolutions and by their own elementwise chain only
        if (!(s.kind == StepKind::Conv2d || (elementwise(s) && b == s.in0 && m.nchw_buf[size_t(s.out)])))
          refuse("'" + s.origin + "' reads the NCHW input tensor and is neither a convolution nor elementwise preprocessing");
        continue;
      }
      if (!layout_free && spatial(b) > 1) {
        // Flatten(C,H,W) -> Gemm (VGG / AlexNet heads): the layer reads the channel-quad tensor as it lies and its
        // weight rows are permuted to that order once, below.  Anything else that looks at flattened features in
        // NCHW order keeps the whole plan NCHW.
        if (s.kind == StepKind::Dense && b == s.in0 && b != 0 && s.K == m.plan.buf_per_row[size_t(b)]) flat_dense.push_back(&s - st.data());
        else refuse("'" + s.origin + "' looks at a [C,H,W] tensor in NCHW element order");
      }
    }
  }
  if (spatial(m.plan.out_buf) > 1) refuse("the served output is a [C,H,W] tensor (results leave in the caller's NCHW order)");
  // channel-quad planes need whole quads in every internal 4-D tensor (the caller's input stays NCHW)
  for (size_t b = 1; b < m.plan.buf_shape.size(); b++)
    if (m.plan.buf_shape[b].size() == 4 && m.plan.buf_shape[b][1] % 4 != 0 && !m.nchw_buf[b]) {
      // a [N,C,1,1] tensor has no layout to speak of (a conv head with 10 classes behind the global pool): plain order
      if (spatial(int(b)) == 1) m.nchw_buf[b] = 1;
      else refuse("an internal [N,C,H,W] tensor has " + std::to_string(m.plan.buf_shape[b][1]) + " channels (not whole quads)");
    }
  m.cq_mode = any_conv && ok;
  if (any_conv && !ok)
    log_msg(1, "model '" + m.name + "': convolutional plan stays in NCHW (generic kernels, roughly 10x slower): " + nchw_reason);
  if (m.cq_mode)
    for (size_t i : flat_dense) {  // W rows: NCHW feature c*HW + p  ->  channel-quad feature ((c/4)*HW + p)*4 + c%4
      Step &d = m.plan.steps[i];
      const auto &bs = m.plan.buf_shape[size_t(d.in0)];
      const int64_t C = bs[1], HW = bs[2] * bs[3], M = d.M;
      std::vector<float> w(d.W.size());
      for (int64_t c = 0; c < C; c++)
        for (int64_t p = 0; p < HW; p++)
          std::copy_n(d.W.begin() + (c * HW + p) * M, M, w.begin() + (((c >> 2) * HW + p) * 4 + (c & 3)) * M);
      d.W = std::move(w);
      d.origin += "[rows in channel-quad order]";
    }
  if (m.cq_mode)
    for (size_t i = 0; i < n; i++) {  // PRelu slopes, Min / Max / Pow constants on channel-quad tensors: same permutation
      Step &b = m.plan.steps[i];
      if (b.kind != StepKind::BinaryConst || !is4d(b.in0) || m.nchw_buf[size_t(b.in0)] || spatial(b.in0) <= 1) continue;
      const auto &bs = m.plan.buf_shape[size_t(b.in0)];
      const int64_t C = bs[1], HW = bs[2] * bs[3];
      if (int64_t(b.cst.size()) != C * HW) continue;
      std::vector<float> c2(b.cst.size());
      for (int64_t c = 0; c < C; c++)
        for (int64_t p = 0; p < HW; p++) c2[size_t((((c >> 2) * HW + p) << 2) + (c & 3))] = b.cst[size_t(c * HW + p)];
      b.cst = std::move(c2);
    }
  m.conv_fused_pool.assign(n, -1);
  if (m.cq_mode)
    for (size_t i = 0; i < n; i++) {
      const Step &s = st[i];
      if (m.exec[i] != ExecKind::Normal || s.kind != StepKind::Conv2d) continue;
      kern::ConvGeom g{int(s.C), int(s.H), int(s.Wd), int(s.Mo), int(s.OH), int(s.OW), int(s.kh), int(s.kw),
                       int(s.sh), int(s.sw), int(s.pt), int(s.pl), int(s.dh), int(s.dw), int(s.groups)};
      if (m.nchw_buf[size_t(s.in0)]) {  // the caller's NCHW blob (or its normalised copy): few channels -> LDS patch kernel
        if (!kern::conv2d_patch_supported(kern::conv2d_patch_geom(g))) continue;
        m.exec[i] = ExecKind::ConvPatch;
        // stem -> MaxPool 3x3 / 2 (ResNet, DenseNet, SqueezeNet): pooled in the stem's kernel, the stem's own output is never stored
        // (INFERA_STEM_POOL=0, read at load time: the two kernels)
        if (ScheduleKnobs::read().stem_pool && uses[size_t(s.out)] == 1 && s.out != m.plan.out_buf)
          for (size_t j = i + 1; j < n; j++) {
            const Step &q = st[j];
            if (q.in0 != s.out && q.in1 != s.out) continue;
            const kern::PoolTail tail{int(q.OH), int(q.OW), int(q.pt), int(q.pl)};
            if (q.kind == StepKind::Pool2d && m.exec[j] == ExecKind::Normal && q.is_max && q.kh == 3 && q.kw == 3 && q.sh == 2 && q.sw == 2 &&
                q.dh == 1 && q.dw == 1 && kern::conv2d_patch_pool_supported(kern::conv2d_patch_geom(g), tail)) {
              m.conv_fused_pool[i] = int(j);
              m.exec[j] = ExecKind::Skipped;
            }
            break;
          }
        continue;
      }
      if (kern::conv2d_tiled_supported(g)) m.exec[i] = ExecKind::ConvTiled;
      else if (kern::conv2d_depthwise_supported(g)) m.exec[i] = ExecKind::ConvDepthwise;
    }
  // Residual Add (+ activation) of a ResNet block -> epilogue of whichever of its two producers runs LAST
  // (conv2, or the 1x1 downsample conv when the block has one), the other operand being the skip tensor.
  m.conv_fused_add.assign(n, -1);
  m.conv_residual_buf.assign(n, -1);
  if (m.cq_mode) {
    std::vector<int> prod(m.plan.buf_per_row.size(), -1);
    for (size_t i = 0; i < n; i++)
      if (m.exec[i] != ExecKind::Skipped) prod[size_t(st[i].out)] = int(i);
    for (size_t j = 0; j < n; j++) {
      const Step &a = st[j];
      if (m.exec[j] != ExecKind::Normal || a.kind != StepKind::BinaryAct || a.bop != '+' || !is4d(a.out) || a.S > 1) continue;
      if (!mfma_fusable(a.act)) continue;  // the conv epilogue resolves only the MFMA-fusable kinds
      const int pa = prod[size_t(a.in0)], pb = prod[size_t(a.in1)];
      const int late = std::max(pa, pb);
      if (late < 0 || a.in0 == a.in1) continue;
      const int fused_in = late == pa ? a.in0 : a.in1, skip = late == pa ? a.in1 : a.in0;
      const Step &c = st[size_t(late)];
      if (m.exec[size_t(late)] != ExecKind::ConvTiled || c.act != Act::None || uses[size_t(fused_in)] != 1) continue;
      if (m.conv_fused_add[size_t(late)] >= 0) continue;
      m.conv_fused_add[size_t(late)] = int(j);
      m.conv_residual_buf[size_t(late)] = skip;
      m.exec[j] = ExecKind::Skipped;
    }
  }

  // the default arithmetic of the tiled convolutions and the 7x7 / stride-2 stem: bf16 x three exact parts (conv_split.hip)
  m.conv_split6.assign(n, 0);
  m.stem_split6.assign(n, 0);
  if (ScheduleKnobs::read().conv_bf16x6 && m.cq_mode) {
    for (size_t i = 0; i < n; i++) {
      if (m.exec[i] != ExecKind::ConvTiled) continue;
      const Step &c = st[i];
      const kern::ConvGeom g{int(c.C), int(c.H), int(c.Wd), int(c.Mo), int(c.OH), int(c.OW), int(c.kh), int(c.kw),
                             int(c.sh), int(c.sw), int(c.pt), int(c.pl), int(c.dh), int(c.dw), int(c.groups)};
      if (kern::conv2d_split6_supported(kern::conv2d_tiled_geom(g)) && !m.nchw_buf[size_t(c.in0)]) m.conv_split6[i] = 1;
    }
    for (size_t i = 0; i < n; i++) {  // the 7x7 / stride-2 stem + max-pool in the same arithmetic
      if (m.exec[i] != ExecKind::ConvPatch || m.conv_fused_pool[i] < 0) continue;
      const Step &c = st[i], &q = st[size_t(m.conv_fused_pool[i])];
      const kern::ConvGeom g{int(c.C), int(c.H), int(c.Wd), int(c.Mo), int(c.OH), int(c.OW), int(c.kh), int(c.kw),
                             int(c.sh), int(c.sw), int(c.pt), int(c.pl), int(c.dh), int(c.dw), int(c.groups)};
      const kern::ConvGeom gp = kern::conv2d_patch_geom(g);
      if (gp.mvalid == 0 && kern::conv2d_stem_split6_supported(gp, kern::PoolTail{int(q.OH), int(q.OW), int(q.pt), int(q.pl)})) m.stem_split6[i] = 1;
    }
  }
  // A ResNet block's PROJECTION SHORTCUT folded into the block's second convolution (round 4): out = act(conv_kxk(A) + conv_1x1/s(P)) where the
  // 1x1 layer runs last and carries the fused Add today -- its result tensor is written, and the other operand read back, only to be added.  As
  // x2.C / 32 more K stages of the second convolution's kernel (conv_split.hip SecondInput) the sum forms in one accumulator: one launch and two
  // tensor passes less per block.  Conditions: both layers on the split form, the 1x1 layer unpadded and undilated, the other one a
  // 128-feature launch with no activation and no residual of its own, its output read by the Add alone.  INFERA_CONV_FOLD_SHORTCUT=0 (read when
  // a model is scheduled): two launches (tests, A/B).
  m.conv_fold.assign(n, -1);
  if (m.cq_mode && ScheduleKnobs::read().conv_fold_shortcut) {
    std::vector<int> prod(m.plan.buf_per_row.size(), -1);
    for (size_t i = 0; i < n; i++)
      if (m.exec[i] != ExecKind::Skipped) prod[size_t(st[i].out)] = int(i);
    // which EXECUTED step writes each buffer once the fusions so far are applied: the output of a fused epilogue (conv + Add, stem + pool, a
    // fused head) belongs to the kernel that carries it, not to its Skipped Add / Pool step -- `prod` above knows nothing of those buffers
    // (the ordering check below was vacuous for them).  Rebuilt after every fold: a folded block's output moves to its second conv.
    std::vector<int> writer;
    auto rebuild_writers = [&] {
      writer.assign(m.plan.buf_per_row.size(), -1);
      for (const auto &e : effective_steps(m)) writer[size_t(e.writes)] = e.idx;
    };
    rebuild_writers();
    for (size_t late = 0; late < n; late++) {
      const int j = m.conv_fused_add[late];
      if (j < 0 || !m.conv_split6[late]) continue;
      const Step &d = st[late];
      if (d.kh != 1 || d.kw != 1 || d.pt != 0 || d.pl != 0 || d.groups != 1 || d.C % 32 != 0 || d.act != Act::None) continue;
      const int skip = m.conv_residual_buf[late], early = skip >= 0 ? prod[size_t(skip)] : -1;
      if (early < 0 || size_t(early) >= late || !m.conv_split6[size_t(early)] || m.conv_fused_add[size_t(early)] >= 0) continue;
      const Step &c = st[size_t(early)];
      const kern::ConvGeom gc{int(c.C), int(c.H), int(c.Wd), int(c.Mo), int(c.OH), int(c.OW), int(c.kh), int(c.kw),
                              int(c.sh), int(c.sw), int(c.pt), int(c.pl), int(c.dh), int(c.dw), int(c.groups)};
      if (!kern::conv2d_split6_takes_second_input(kern::conv2d_tiled_geom(gc)) || c.act != Act::None || uses[size_t(c.out)] != 1) continue;
      if (c.Mo != d.Mo || c.OH != d.OH || c.OW != d.OW || c.bias.empty() != d.bias.empty()) continue;
      if ((d.OH - 1) * d.sh >= d.H || (d.OW - 1) * d.sw >= d.Wd) continue;  // (every output pixel's source pixel lies inside the shortcut's input)
      // the shortcut's input must EXIST when the second convolution runs in its place: written by a step executed before `early` (the caller's
      // tensor, buffer 0, is excluded above as NCHW).  y = Conv3x3(A); P = Relu(Conv(B) + C); out = Relu(y + Conv1x1(P)) is a valid ONNX order
      // in which P is produced BETWEEN the two convolutions: no fold.
      if (m.nchw_buf[size_t(d.in0)] || writer[size_t(d.in0)] < 0 || writer[size_t(d.in0)] >= early) continue;
      m.conv_fold[size_t(early)] = int(late);
      m.conv_fused_add[size_t(early)] = j;   // the Add's activation and output now belong to the second convolution ...
      m.conv_residual_buf[size_t(early)] = -1;  // ... which has no residual to read
      m.conv_fused_add[late] = -1;
      m.conv_residual_buf[late] = -1;
      m.exec[late] = ExecKind::Skipped;
      rebuild_writers();
    }
  }

  // scratch slots by liveness: a slot is reused once its buffer has been read for the last time
  auto eff = effective_steps(m);
  {  // the served output: one writer (a fused streaming kernel that only stores it), no reader
    int writers = 0, readers = 0;
    bool streaming = false;
    for (const auto &e : eff) {
      if (e.writes == m.plan.out_buf) {
        writers++;
        // every tabular head qualifies: fused chains, Dense with or without a fused Softmax / ArgMax epilogue, a row Softmax or
        // ArgMax kernel -- each stores every result element exactly once (convolutional plans keep the D2H copy: their
        // outputs leave in strided channel-quad order)
        const ExecKind k = m.exec[size_t(e.idx)];
        const StepKind sk = st[size_t(e.idx)].kind;
        streaming = k == ExecKind::Mlp3Head || k == ExecKind::ChainHead || k == ExecKind::DenseSoftmax || k == ExecKind::DenseArgMax ||
                    k == ExecKind::DenseTiled || (k == ExecKind::Normal && (sk == StepKind::Dense || sk == StepKind::Softmax || sk == StepKind::ArgMax));
      }
      for (int b : e.reads) readers += b == m.plan.out_buf;
    }
    m.out_write_once = writers == 1 && readers == 0 && streaming && m.plan.out_buf != 0;
    int in_readers = 0;
    for (const auto &e : eff)
      for (int b : e.reads) in_readers += b == 0;
    m.in_single_reader = false;
    if (in_readers == 1)
      for (const auto &e : eff) {
        if (std::find(e.reads.begin(), e.reads.end(), 0) == e.reads.end()) continue;
        // ... and that kernel streams it about once (a windowed kernel would fetch a small image over PCIe once per tap)
        const StepKind sk = st[size_t(e.idx)].kind;
        const bool windowed = (sk == StepKind::Conv2d && m.exec[size_t(e.idx)] != ExecKind::ConvPatch) || sk == StepKind::Pool2d || sk == StepKind::LRN;
        m.in_single_reader = !windowed;
        break;
      }
    m.in_colmajor_max_rows = INT64_MAX;
    m.in_colmajor_ok = !eff.empty() && in_readers == 1 && m.exec[size_t(eff[0].idx)] == ExecKind::Mlp3Head && eff[0].reads[0] == 0 &&
                       kern::mlp3_colmajor_supported(m.mlp3_shape);
    if (m.in_colmajor_ok) m.in_colmajor_max_rows = kern::mlp3_colmajor_max_rows(m.mlp3_shape);
    // the fused small-MLP chain reads a column-major chunk too (a run-time flag of the same kernel); INFERA_CHAIN_XCM=0: transpose first
    const ScheduleKnobs knobs = ScheduleKnobs::read();
    if (knobs.chain_xcm && !eff.empty() && in_readers == 1 && m.exec[size_t(eff[0].idx)] == ExecKind::ChainHead && eff[0].reads[0] == 0)
      m.in_colmajor_ok = true;
    // ... and so do the two as-it-lies streaming kernels of single narrow layers (linear / logistic regression, with or without the
    // softmax / label epilogue); INFERA_DENSE_XCM=0: transpose first
    if (knobs.dense_xcm && !eff.empty() && in_readers == 1 && eff[0].reads[0] == 0) {
      const size_t i0 = size_t(eff[0].idx);
      const ExecKind k0 = m.exec[i0];
      if (st[i0].kind == StepKind::Dense && (k0 == ExecKind::Normal || k0 == ExecKind::DenseSoftmax || k0 == ExecKind::DenseArgMax) &&
          kern::dense_colmajor_supported(int(st[i0].K), int(st[i0].M)))
        m.in_colmajor_ok = true;
    }
  }
  const size_t nb = m.plan.buf_per_row.size();
  std::vector<int> last_read(nb, -1);
  for (size_t e = 0; e < eff.size(); e++)
    for (int b : eff[e].reads) last_read[size_t(b)] = int(e);
  m.slot_of_buf.assign(nb, -1);
  m.slot_per_row.clear();
  std::vector<int> slot_free_after;  // per slot: effective-step index after which it is free (-2 = free now)
  for (size_t e = 0; e < eff.size(); e++) {
    const int b = eff[e].writes;
    if (b == m.plan.out_buf || b == 0) continue;
    if (m.slot_of_buf[size_t(b)] >= 0) continue;  // a Concat output: its first CopyCols piece already placed it
    int chosen = -1;
    for (size_t s = 0; s < slot_free_after.size(); s++)
      if (slot_free_after[s] < int(e)) {  // strictly before this step: in-place reuse is not allowed
        chosen = int(s);
        break;
      }
    if (chosen < 0) {
      chosen = int(slot_free_after.size());
      slot_free_after.push_back(0);
      m.slot_per_row.push_back(0);
    }
    m.slot_of_buf[size_t(b)] = chosen;
    m.slot_per_row[size_t(chosen)] = std::max(m.slot_per_row[size_t(chosen)], m.plan.buf_per_row[size_t(b)]);
    slot_free_after[size_t(chosen)] = last_read[size_t(b)] < 0 ? int(e) : last_read[size_t(b)];
  }
  m.scratch_per_row = 0;
  for (auto v : m.slot_per_row) m.scratch_per_row += v;
}

void upload_to_device(const LoadedModel &m, DeviceModel &dm) {
  UnsafeOpGuard guard;
  hipStream_t us = ctx_for_slot(slot_of_ordinal(dm.device)).stream;  // also does hipSetDevice
  const auto &st = m.plan.steps;
  dm.steps.resize(st.size());
  for (size_t i = 0; i < st.size(); i++) {
    if (m.exec[i] == ExecKind::Skipped) continue;
    const Step &s = st[i];
    DeviceStep &d = dm.steps[i];
    if (m.exec[i] == ExecKind::Mlp3Head) {
      std::vector<float> packed(kern::mlp3_packed_floats(m.mlp3_shape));
      const Step &s1 = st[i], &s2 = st[i + 1], &s3 = st[i + 2];
      kern::mlp3_pack(m.mlp3_shape, s1.W.data(), s1.bias.empty() ? nullptr : s1.bias.data(), s2.W.data(),
                      s2.bias.empty() ? nullptr : s2.bias.data(), s3.W.data(), s3.bias.empty() ? nullptr : s3.bias.data(),
                      packed.data());
      dm.mlp3_packed = upload(packed, us);
      continue;
    }
    if (m.exec[i] == ExecKind::ChainHead) {
      const LoadedModel::ChainRun &run = *m.chain_at(i);
      std::vector<const float *> W, B;
      for (size_t l = 0; l < run.shape.dims.size(); l++) {
        const Step &ls = st[i + size_t(run.pad) + l];
        W.push_back(ls.W.data());
        B.push_back(ls.bias.empty() ? nullptr : ls.bias.data());
      }
      std::vector<float> packed(kern::chain_packed_floats(run.shape));
      kern::chain_pack(run.shape, W, B, packed.data());
      dm.chain_packed.resize(m.chains.size(), nullptr);
      dm.chain_packed[size_t(&run - m.chains.data())] = upload(packed, us);
      continue;
    }
    if (m.exec[i] == ExecKind::ConvTiled) {
      kern::ConvGeom g{int(s.C), int(s.H), int(s.Wd), int(s.Mo), int(s.OH), int(s.OW), int(s.kh), int(s.kw),
                       int(s.sh), int(s.sw), int(s.pt), int(s.pl), int(s.dh), int(s.dw), int(s.groups)};
      const kern::ConvGeom gp = kern::conv2d_tiled_geom(g);
      std::vector<float> packed(kern::conv2d_tiled_packed_floats(gp));
      if (gp.padc) {  // channel counts padded to 32: zero weights and zero bias beyond the real ones
        const size_t taps = size_t(g.kh) * g.kw;
        std::vector<float> wt(size_t(gp.M) * gp.C * taps, 0.f);
        for (int mo = 0; mo < g.M; mo++)
          std::copy_n(s.W.begin() + size_t(mo) * g.C * taps, size_t(g.C) * taps, wt.begin() + size_t(mo) * gp.C * taps);
        kern::conv2d_tiled_pack(gp, wt.data(), packed.data());
        d.W = upload(packed, us);
        if (!s.bias.empty()) {
          std::vector<float> bp(size_t(gp.M), 0.f);
          std::copy(s.bias.begin(), s.bias.end(), bp.begin());
          d.bias = upload(bp, us);
        }
        continue;
      }
      if (m.conv_split6[i]) {
        const size_t main_floats = kern::conv2d_split6_packed_floats(g);
        packed.resize(main_floats);
        kern::conv2d_split6_pack(g, s.W.data(), packed.data());
        if (const int fl = m.conv_fold[i]; fl >= 0) {  // the folded 1x1 shortcut: its chunks behind the main filter's, its bias added to this layer's
          const Step &q = st[size_t(fl)];
          const kern::ConvGeom gq{int(q.C), int(q.H), int(q.Wd), int(q.Mo), int(q.OH), int(q.OW), 1, 1, int(q.sh), int(q.sw), 0, 0, 1, 1, 1};
          packed.resize(main_floats + kern::conv2d_split6_packed_floats(gq));
          kern::conv2d_split6_pack(gq, q.W.data(), packed.data() + main_floats);
          d.W = upload(packed, us);
          std::vector<float> b(s.bias);
          for (size_t k = 0; k < b.size() && k < q.bias.size(); k++) b[k] += q.bias[k];
          d.bias = upload(b, us);
          continue;
        }
      } else {
        kern::conv2d_tiled_pack(g, s.W.data(), packed.data());
      }
      d.W = upload(packed, us);
    } else if (m.exec[i] == ExecKind::ConvDepthwise) {
      kern::ConvGeom g{int(s.C), int(s.H), int(s.Wd), int(s.Mo), int(s.OH), int(s.OW), int(s.kh), int(s.kw),
                       int(s.sh), int(s.sw), int(s.pt), int(s.pl), int(s.dh), int(s.dw), int(s.groups)};
      std::vector<float> packed(s.W.size());
      kern::conv2d_depthwise_pack(g, s.W.data(), packed.data());
      d.W = upload(packed, us);
    } else if (m.exec[i] == ExecKind::DenseTiled) {
      const kern::ConvGeom g = dense_as_conv(s);
      std::vector<float> wt(size_t(g.C) * g.M, 0.f), packed(kern::conv2d_tiled_packed_floats(g));
      for (int64_t k = 0; k < s.K; k++)
        for (int64_t j = 0; j < s.M; j++) wt[size_t(j) * g.C + size_t(k)] = s.W[size_t(k * s.M + j)];  // [K][M] -> conv's [Mp][Cp]
      kern::conv2d_tiled_pack(g, wt.data(), packed.data());
      d.W = upload(packed, us);
      if (!s.bias.empty()) {
        std::vector<float> bp(size_t(g.M), 0.f);
        std::copy(s.bias.begin(), s.bias.end(), bp.begin());
        d.bias = upload(bp, us);
      }
      d.cst = upload(s.cst, us);
      d.scale = upload(s.scale, us);
      d.shift = upload(s.shift, us);
      continue;
    } else if (m.exec[i] == ExecKind::ConvPatch) {
      kern::ConvGeom g{int(s.C), int(s.H), int(s.Wd), int(s.Mo), int(s.OH), int(s.OW), int(s.kh), int(s.kw),
                       int(s.sh), int(s.sw), int(s.pt), int(s.pl), int(s.dh), int(s.dw), int(s.groups)};
      const kern::ConvGeom gp = kern::conv2d_patch_geom(g);
      std::vector<float> packed(kern::conv2d_patch_packed_floats(gp));
      if (gp.mvalid > 0) {  // output features padded to whole tiles: zero weights and bias beyond the real ones
        std::vector<float> wt(size_t(gp.M) * g.C * g.kh * g.kw, 0.f);
        std::copy(s.W.begin(), s.W.end(), wt.begin());
        kern::conv2d_patch_pack(gp, wt.data(), packed.data());
        d.W = upload(packed, us);
        if (!s.bias.empty()) {
          std::vector<float> bp(size_t(gp.M), 0.f);
          std::copy(s.bias.begin(), s.bias.end(), bp.begin());
          d.bias = upload(bp, us);
        }
        continue;
      }
      if (const int fj = m.conv_fused_pool[i]; fj >= 0) {
        const Step &q = st[size_t(fj)];
        const kern::PoolTail tail{int(q.OH), int(q.OW), int(q.pt), int(q.pl)};
        kern::conv2d_patch_pack(g, s.W.data(), packed.data(), &tail);
        if (m.stem_split6[i]) {  // (the exact-fp32 blob above stays: INFERA_STEM_SPLIT=0 at run time compares the two)
          std::vector<float> sp(kern::conv2d_stem_split6_packed_floats());
          kern::conv2d_stem_split6_pack(g, s.W.data(), sp.data());
          d.cst = upload(sp, us);
        }
      } else {
        kern::conv2d_patch_pack(g, s.W.data(), packed.data());
      }
      d.W = upload(packed, us);
    } else if (s.kind == StepKind::Conv2d) {
      kern::ConvGeom g{int(s.C), int(s.H), int(s.Wd), int(s.Mo), int(s.OH), int(s.OW), int(s.kh), int(s.kw),
                       int(s.sh), int(s.sw), int(s.pt), int(s.pl), int(s.dh), int(s.dw), int(s.groups)};
      if (!kern::conv2d_generic_supported(g))
        throw InferaError::onnx("Conv with (C/group)*kh*kw = " + std::to_string(s.K) + " > 8192 is not supported by the generic kernel");
      std::vector<float> packed(s.W.size());
      kern::conv2d_generic_pack(g, s.W.data(), packed.data());
      d.W = upload(packed, us);
    } else {
      d.W = upload(s.W, us);
    }
    d.bias = upload(s.bias, us);
    if (!d.cst) d.cst = upload(s.cst, us);  // (a split stem keeps its bf16 blob there: convolutions have no constants)
    d.scale = upload(s.scale, us);
    d.shift = upload(s.shift, us);
  }
}

// One copy stream per device slot for the big-row pipeline's H2D copies (run_host: why), created on first use; `mu` serialises
// {copy, event record} pairs of different callers.
hipStream_t big_copy_stream(int slot, std::mutex *&mu) {
  static std::mutex create_mu, pair_mu[64];
  static hipStream_t streams[64] = {};
  const size_t i = size_t(slot) % 64;
  mu = &pair_mu[i];
  std::lock_guard<std::mutex> lk(create_mu);
  if (!streams[i]) HIP_TRY(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));  // (the caller has made the slot's device current)
  return streams[i];
}

// Rows per device pass for plans that need activation scratch (pure: no allocation).
int64_t rows_per_pass(const LoadedModel &m, int64_t rows) {
  if (m.scratch_per_row <= 0 || m.plan.out_buf == 0) return rows;
  int64_t by_budget = int64_t(kScratchBudgetBytes / (size_t(m.scratch_per_row) * 4));
  // INFERA_MAX_ROWS_PER_PASS (2^18) bounds the scratch of plans with wide intermediates; plans whose intermediates are a
  // few floats per row (a linear model + Softmax + Normalizer) take passes of up to 256 MB of scratch instead -- 77 passes
  // of 262k rows over a 20M-row table were launch-bound (three ~10 us kernels each)
  const int64_t by_size = int64_t((256ull << 20) / (size_t(m.scratch_per_row) * 4));
  const int64_t cap = std::max<int64_t>(int64_t(Config::get().max_rows_per_pass), by_size);
  int64_t rows_pass = std::min<int64_t>(rows, std::max<int64_t>(1, std::min<int64_t>(by_budget, cap)));
  // equal passes (2000 images at 1783 per pass run as 1000 + 1000, not 1783 + 217: the short tail would leave the chip half empty)
  const int64_t npass = (rows + rows_pass - 1) / rows_pass;
  return (rows + npass - 1) / npass;
}

// ... and grows the scratch for it (never inside a stream capture: callers that capture call this first).
int64_t prepare_scratch(const LoadedModel &m, ThreadCtx &ctx, int64_t rows) {
  const int64_t rows_pass = rows_per_pass(m, rows);
  // (+ 1 row: a pass cut into two lanes of ceil(n / 2) rows each)
  if (m.scratch_per_row > 0 && m.plan.out_buf != 0)
    ctx.ensure_dev(ctx.scratch, ctx.scratch_cap, size_t(rows_pass + ThreadCtx::kMaxLanes) * size_t(m.scratch_per_row) * 4);
  return rows_pass;
}

// A long pass of a convolutional plan runs as TWO LANES: its rows in two halves, each through all the plan's kernels on its own stream, with
// its own half of the scratch.  Every launch of such a plan ends in a partial round of workgroups (ResNet-18 at 1024 images: 12.25 / 6.125 /
// 3.06 rounds for its 128 / 256 / 512-channel layers -- 3.2 % of the pass, profiles/r04_tail_rounds.txt); with two independent kernel
// sequences in flight the other lane's workgroups fill those rounds (and the stem of one lane runs beside the matrix-bound layers of the
// other).  Same kernels, same per-row arithmetic: results are bit-identical (tests/test_conv_split_gpu.py).  ResNet-18, 1024 images: 18.42 ->
// 17.94 ms (-2.6 %); three or four lanes: no gain (profiles/r04_conv_lanes_ab.txt).  Not under a stream capture (INFERA_HIPGRAPH=1), not
// for the short passes of the host path (many contexts already overlap there).
constexpr int64_t kLaneMinRows = 512;
int lanes_of(const LoadedModel &m, int64_t nr) {
  // (INFERA_CONV_LANES=1: one lane -- for counter passes, which serialise kernels: per-kernel figures of full-size launches; tools/profile_bench.sh)
  static const bool one = getenv("INFERA_CONV_LANES") && atoi(getenv("INFERA_CONV_LANES")) == 1;
  if (one || nr < kLaneMinRows || Config::get().use_hipgraph || m.scratch_per_row <= 0) return 1;
  for (const ExecKind k : m.exec)
    if (k == ExecKind::ConvTiled) return ThreadCtx::kMaxLanes;
  return 1;
}

// in_colmajor: d_in is one column-major chunk [in_per_row][rows] (only with m.in_colmajor_ok, which implies a single pass)
void exec_plan(const LoadedModel &m, const DeviceModel &dm, ThreadCtx &ctx, const float *d_in, float *d_out, int64_t rows,
               bool in_colmajor = false) {
  const Plan &p = m.plan;
  if (rows <= 0) return;
  hipStream_t s = ctx.stream;
  if (p.out_buf == 0) {  // pure alias / Identity graph
    HIP_TRY(hipMemcpyAsync(d_out, d_in, size_t(rows) * size_t(p.in_per_row()) * 4, hipMemcpyDeviceToDevice, s));
    return;
  }
  const int64_t rows_pass = prepare_scratch(m, ctx, rows);
  // a column-major chunk [K][rows] cannot be cut into row passes (pass r0 would start at a row-major offset with stride nr):
  // callers check single_pass() first and stage such calls row-major instead; this guards every kernel family at once
  if (in_colmajor && rows_pass != rows) throw InferaError::onnx("internal: column-major input needs a single pass");
  std::vector<int64_t> slot_base(m.slot_per_row.size(), 0);
  const auto &st = p.steps;
  // rows r0 .. r0 + nr - 1 through every step on stream s; scratch slots sized for slot_rows rows, from scratch_off floats into the scratch
  auto run_rows = [&](hipStream_t s, int64_t r0, int64_t nr, int64_t slot_rows, int64_t scratch_off) {
    {
      int64_t off = scratch_off;
      for (size_t i = 0; i < m.slot_per_row.size(); i++) {
        slot_base[i] = off;
        off += m.slot_per_row[i] * slot_rows;
      }
    }
    auto buf = [&](int b) -> float * {
      if (b == 0) return const_cast<float *>(d_in) + r0 * p.in_per_row();
      if (b == p.out_buf) return d_out + r0 * p.out_per_row();
      return ctx.scratch + slot_base[size_t(m.slot_of_buf[size_t(b)])];
    };
    for (size_t i = 0; i < st.size(); i++) {
      const Step &x = st[i];
      const DeviceStep &d = dm.steps[i];
      switch (m.exec[i]) {
        case ExecKind::Skipped: continue;
        case ExecKind::Mlp3Head:
        {
          std::string why;
          const bool cm = in_colmajor && x.in0 == 0;
          if (!kern::mlp3(s, m.mlp3_shape, buf(x.in0), dm.mlp3_packed, buf(st[i + 2].out), nr, dm.num_cus, &why, cm))
            throw InferaError::onnx("fused MLP kernel launch failed: " + why);
          continue;
        }
        case ExecKind::ChainHead: {
          const LoadedModel::ChainRun &run = *m.chain_at(i);
          std::string why;
          if (!kern::chain(s, run.shape, buf(x.in0), dm.chain_packed[size_t(&run - m.chains.data())],
                           buf(st[i + size_t(run.nsteps) - 1].out), nr, dm.num_cus, &why, in_colmajor && x.in0 == 0))
            throw InferaError::onnx("fused chain kernel launch failed: " + why);
          continue;
        }
        case ExecKind::DenseArgMax:
          if ((in_colmajor && x.in0 == 0) || kern::dense_can_fuse_argmax(buf(x.in0), int(x.K), int(x.M))) {  // (both column-major kernels have the epilogue)
            kern::dense(s, buf(x.in0), d.W, d.bias, buf(st[i + 1].out), nr, int(x.K), int(x.M), act_of(x), 3, in_colmajor && x.in0 == 0);
            i += 1;  // the ArgMax step is done
            continue;
          }
          break;  // as two kernels
        case ExecKind::DenseSoftmax:
          kern::dense(s, buf(x.in0), d.W, d.bias, buf(st[i + 1].out), nr, int(x.K), int(x.M), act_of(x),
                      st[i + 1].log_softmax ? 2 : 1, in_colmajor && x.in0 == 0);
          continue;
        case ExecKind::ConvTiled: {
          kern::ConvGeom g{int(x.C), int(x.H), int(x.Wd), int(x.Mo), int(x.OH), int(x.OW), int(x.kh), int(x.kw),
                           int(x.sh), int(x.sw), int(x.pt), int(x.pl), int(x.dh), int(x.dw), int(x.groups)};
          const int fj = m.conv_fused_add[i];
          const kern::ConvGeom gp = kern::conv2d_tiled_geom(g);
          if (m.conv_split6[i] && m.conv_fold[i] >= 0) {
            const Step &q = st[size_t(m.conv_fold[i])];
            const kern::SecondInput x2{buf(q.in0), int(q.C), int(q.H), int(q.Wd), int(q.sh), int(q.sw)};
            kern::conv2d_split6(s, buf(x.in0), d.W, d.bias, nullptr, buf(st[size_t(fj)].out), nr, gp, act_of(st[size_t(fj)]), x2);
            continue;
          }
          if (m.conv_split6[i]) {
            if (fj >= 0) kern::conv2d_split6(s, buf(x.in0), d.W, d.bias, buf(m.conv_residual_buf[i]), buf(st[size_t(fj)].out), nr, gp, act_of(st[size_t(fj)]));
            else kern::conv2d_split6(s, buf(x.in0), d.W, d.bias, nullptr, buf(x.out), nr, gp, act_of(x));
            continue;
          }
          if (fj >= 0) kern::conv2d_tiled(s, buf(x.in0), d.W, d.bias, buf(m.conv_residual_buf[i]), buf(st[size_t(fj)].out), nr, gp, act_of(st[size_t(fj)]));
          else kern::conv2d_tiled(s, buf(x.in0), d.W, d.bias, nullptr, buf(x.out), nr, gp, act_of(x));
          continue;
        }
        case ExecKind::DenseTiled:
          kern::conv2d_tiled(s, buf(x.in0), d.W, d.bias, nullptr, buf(x.out), nr, dense_as_conv(x), act_of(x));
          continue;
        case ExecKind::ConvDepthwise: {
          kern::ConvGeom g{int(x.C), int(x.H), int(x.Wd), int(x.Mo), int(x.OH), int(x.OW), int(x.kh), int(x.kw),
                           int(x.sh), int(x.sw), int(x.pt), int(x.pl), int(x.dh), int(x.dw), int(x.groups)};
          kern::conv2d_depthwise(s, buf(x.in0), d.W, d.bias, buf(x.out), nr, g, act_of(x));
          continue;
        }
        case ExecKind::ConvPatch: {
          kern::ConvGeom g{int(x.C), int(x.H), int(x.Wd), int(x.Mo), int(x.OH), int(x.OW), int(x.kh), int(x.kw),
                           int(x.sh), int(x.sw), int(x.pt), int(x.pl), int(x.dh), int(x.dw), int(x.groups)};
          if (const int fj = m.conv_fused_pool[i]; fj >= 0) {
            const Step &q = st[size_t(fj)];
            const char *sse = getenv("INFERA_STEM_SPLIT");  // 0: the exact-fp32 stem kernels under a split plan (read per launch: tests, A/B)
            if (m.stem_split6[i] && d.cst && !(sse && atoi(sse) == 0))
              kern::conv2d_stem_split6(s, buf(x.in0), d.cst, d.bias, buf(q.out), nr, kern::conv2d_patch_geom(g), act_of(x),
                                       kern::PoolTail{int(q.OH), int(q.OW), int(q.pt), int(q.pl)}, dm.num_cus);
            else
              kern::conv2d_patch_pool(s, buf(x.in0), d.W, d.bias, buf(q.out), nr, kern::conv2d_patch_geom(g), act_of(x),
                                      kern::PoolTail{int(q.OH), int(q.OW), int(q.pt), int(q.pl)}, dm.num_cus);
            continue;
          }
          kern::conv2d_patch(s, buf(x.in0), d.W, d.bias, buf(x.out), nr, kern::conv2d_patch_geom(g), act_of(x), dm.num_cus);
          continue;
        }
        default: break;
      }
      switch (x.kind) {
        case StepKind::Dense: kern::dense(s, buf(x.in0), d.W, d.bias, buf(x.out), nr, int(x.K), int(x.M), act_of(x), 0, in_colmajor && x.in0 == 0); break;
        case StepKind::Unary: kern::unary(s, buf(x.in0), buf(x.out), nr * p.buf_per_row[size_t(x.out)], act_of(x)); break;
        case StepKind::AffineChannel:
          kern::affine_channel(s, buf(x.in0), d.scale, d.shift, buf(x.out), nr, x.C, x.S, act_of(x), m.cq_mode && !m.nchw_buf[size_t(x.in0)]);
          break;
        case StepKind::BinaryConst:
          kern::binary_const(s, buf(x.in0), d.cst, buf(x.out), nr, p.buf_per_row[size_t(x.out)], x.bop, x.const_left, act_of(x));
          break;
        case StepKind::BinaryAct:
          if (x.S > 1) kern::binary_gate(s, buf(x.in0), buf(x.in1), buf(x.out), nr, x.C, x.S, x.bop, act_of(x), m.cq_mode && !m.nchw_buf[size_t(x.in0)]);
          else kern::binary_act(s, buf(x.in0), buf(x.in1), buf(x.out), nr * p.buf_per_row[size_t(x.out)], x.bop, act_of(x));
          break;
        case StepKind::Softmax: kern::softmax(s, buf(x.in0), buf(x.out), nr, x.sm_outer, x.sm_len, x.sm_inner, x.sm_norm ? 1 + x.sm_norm : int(x.log_softmax)); break;
        case StepKind::Conv2d: {
          kern::ConvGeom g{int(x.C), int(x.H), int(x.Wd), int(x.Mo), int(x.OH), int(x.OW), int(x.kh), int(x.kw),
                           int(x.sh), int(x.sw), int(x.pt), int(x.pl), int(x.dh), int(x.dw), int(x.groups)};
          kern::conv2d(s, buf(x.in0), d.W, d.bias, buf(x.out), nr, g, act_of(x), m.cq_mode && !m.nchw_buf[size_t(x.in0)],
                       m.cq_mode && !m.nchw_buf[size_t(x.out)]);
          break;
        }
        case StepKind::Pool2d:
          kern::pool2d(s, buf(x.in0), buf(x.out), nr, int(x.C), int(x.H), int(x.Wd), int(x.OH), int(x.OW), int(x.kh), int(x.kw),
                       int(x.sh), int(x.sw), int(x.pt), int(x.pl), int(x.dh), int(x.dw), x.is_max, x.count_pad,
                       m.cq_mode && !m.nchw_buf[size_t(x.in0)]);
          break;
        case StepKind::GlobalAvgPool:
          kern::global_avgpool(s, buf(x.in0), buf(x.out), nr, int(x.C), int(x.S), m.cq_mode && !m.nchw_buf[size_t(x.in0)], x.is_max);
          break;
        case StepKind::CopyCols:
          kern::copy_cols(s, buf(x.in0), buf(x.out), nr, p.buf_per_row[size_t(x.in0)], p.buf_per_row[size_t(x.in0)], 0,
                          p.buf_per_row[size_t(x.out)], x.col_off);
          break;
        case StepKind::PadCols: kern::pad_cols(s, buf(x.in0), buf(x.out), nr, x.K, x.M); break;
        case StepKind::LRN:
          kern::lrn(s, buf(x.in0), buf(x.out), nr, int(x.C), int(x.S), int(x.lrn_size), x.lrn_alpha, x.lrn_beta, x.lrn_bias,
                    m.cq_mode && !m.nchw_buf[size_t(x.in0)]);
          break;
        case StepKind::ChannelShuffle:
          kern::channel_shuffle(s, buf(x.in0), buf(x.out), nr, int(x.C), int(x.S), int(x.groups), m.cq_mode && !m.nchw_buf[size_t(x.in0)]);
          break;
        case StepKind::SliceCols:
          kern::copy_cols(s, buf(x.in0), buf(x.out), nr, x.K, p.buf_per_row[size_t(x.in0)], x.col_off, x.K, 0);
          break;
        case StepKind::ArgMax: kern::argmax_rows(s, buf(x.in0), buf(x.out), nr, x.K); break;
      }
    }
    HIP_TRY(hipGetLastError());
  };
  for (int64_t r0 = 0; r0 < rows; r0 += rows_pass) {
    const int64_t nr = std::min(rows_pass, rows - r0);
    const int nl = in_colmajor ? 1 : lanes_of(m, nr);
    if (nl == 1) {
      run_rows(ctx.stream, r0, nr, rows_pass, 0);
      continue;
    }
    if (!ctx.lane_ev[0]) {
      for (hipStream_t &ls : ctx.lane_stream) HIP_TRY(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
      for (hipEvent_t &e : ctx.lane_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int64_t nlane = (nr + nl - 1) / nl;  // rows per lane (the last lane: what is left)
    HIP_TRY(hipEventRecord(ctx.lane_ev[0], ctx.stream));  // (the input is on the device, the previous pass has left the scratch)
    try {
      for (int l = 0; l < nl; l++) {
        const int64_t l0 = l * nlane, ln = std::min(nlane, nr - l0);
        if (ln <= 0) break;
        hipStream_t ls = l == 0 ? ctx.stream : ctx.lane_stream[l - 1];
        if (l > 0) HIP_TRY(hipStreamWaitEvent(ls, ctx.lane_ev[0], 0));
        run_rows(ls, r0 + l0, ln, nlane, int64_t(m.scratch_per_row) * nlane * l);
        if (l > 0) {
          HIP_TRY(hipEventRecord(ctx.lane_ev[l], ls));
          HIP_TRY(hipStreamWaitEvent(ctx.stream, ctx.lane_ev[l], 0));
        }
      }
    } catch (...) {
      // a lane that was not joined must not still be writing the scratch / the result when the caller unwinds and the context is reused
      for (hipStream_t ls : ctx.lane_stream)
        if (ls) (void)hipStreamSynchronize(ls);
      throw;
    }
  }
}

const DeviceModel &device_model(const LoadedModel &m, int slot) {
  if (m.dev.empty()) throw InferaError::onnx("HIP backend unavailable: " + m.device_error);
  return *m.dev[size_t(slot)];
}

}  // namespace

// -------------------------------------------------------------------------------------------------

// What the host link really delivers on this box: `threads` threads, each looping {hipMemcpyAsync(bytes) from its own
// pinned buffer on its own stream; wait} -- the ceiling the host path's "fraction of PCIe" is honestly compared with
// (measured 46-48 GB/s on the round-2 MI355X boxes against 64 GB/s raw Gen5 x16).
double h2d_probe_gbs(int device_ordinal, size_t bytes, int iters, int threads) {
  if (threads < 1) threads = 1;
  if (iters < 1) iters = 1;
  std::atomic<int> failed{0};
  auto worker = [&] {
    hipStream_t s = nullptr;
    char *pin = nullptr, *dev = nullptr;
    hipEvent_t ev = nullptr;
    bool ok = hipSetDevice(device_ordinal) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void **>(&pin), bytes, hipHostMallocDefault) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&dev), bytes) == hipSuccess &&
              hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
    if (ok) std::memset(pin, 1, bytes);
    for (int i = 0; ok && i < iters; i++)
      ok = hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s) == hipSuccess && hipEventRecord(ev, s) == hipSuccess &&
           hipEventSynchronize(ev) == hipSuccess;
    if (!ok) failed = 1;
    if (ev) (void)hipEventDestroy(ev);
    if (dev) (void)hipFree(dev);
    if (pin) (void)hipHostFree(pin);
    if (s) (void)hipStreamDestroy(s);
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(worker);
  for (auto &x : th) x.join();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  (void)hipGetLastError();
  return failed ? -1.0 : double(bytes) * iters * threads / sec / 1e9;
}

std::string host_phase_json() {
  static const char *names[kPhCount] = {"lease", "gather", "gate", "enqueue", "wait", "copy_out"};
  std::string o = "{\"passes\":" + std::to_string(g_phase_calls.load(std::memory_order_relaxed));
  for (int i = 0; i < kPhCount; i++) o += std::string(",\"") + names[i] + "_ns\":" + std::to_string(g_phase_ns[i].load(std::memory_order_relaxed));
  return o + "}";
}

int choose_slot(const std::vector<int> &slot_numa, int thread_node, uint64_t ticket_on_node, uint64_t ticket_global) {
  const size_t n = slot_numa.size();
  if (n <= 1) return 0;
  if (thread_node >= 0) {
    size_t local = 0;
    for (int nd : slot_numa) local += nd == thread_node;
    if (local > 0) {
      size_t want = size_t(ticket_on_node % local);
      for (size_t i = 0; i < n; i++)
        if (slot_numa[i] == thread_node && want-- == 0) return int(i);
    }
  }
  return int(ticket_global % n);
}

// Load-aware dealing (what home_slot() uses): the least-loaded slot on the thread's own NUMA node, unless it already carries more
// than ONE thread above the least-loaded slot of the whole set -- then that one.  A node whose workers all start on one socket thus
// fills its local GPUs first and spills to the other socket's GPUs one round later; nobody stays idle.  Ties: lowest index.
int choose_slot_balanced(const std::vector<int> &slot_numa, const std::vector<int> &slot_threads, int thread_node) {
  const size_t n = std::min(slot_numa.size(), slot_threads.size());
  if (n <= 1) return 0;
  size_t g = 0;
  long l = -1;
  for (size_t i = 0; i < n; i++) {
    if (slot_threads[i] < slot_threads[g]) g = i;
    if (thread_node >= 0 && slot_numa[i] == thread_node && (l < 0 || slot_threads[i] < slot_threads[size_t(l)])) l = long(i);
  }
  if (l >= 0 && slot_threads[size_t(l)] <= slot_threads[g] + 1) return int(l);
  return int(g);
}

uint64_t slot_pinned_bytes(int slot) { return g_pinned_bytes[size_t(slot) % 64].load(std::memory_order_relaxed); }

void slot_counters(int slot, uint64_t *calls, uint64_t *rows) {
  *calls = g_slot_calls[size_t(slot) % 64].load(std::memory_order_relaxed);
  *rows = g_slot_rows[size_t(slot) % 64].load(std::memory_order_relaxed);
}

const DeviceSet &devices() {
  static const DeviceSet ds = [] {
    DeviceSet d;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
      d.why = std::string("no HIP device visible (hipGetDeviceCount: ") + (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) + ")";
      (void)hipGetLastError();
      return d;
    }
    std::vector<int> want = Config::get().devices;
    if (want.empty())
      for (int i = 0; i < n; i++) want.push_back(i);
    for (int id : want) {
      if (id < 0 || id >= n) continue;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, id) != hipSuccess) continue;
      d.ids.push_back(id);
      d.cus.push_back(prop.multiProcessorCount);
      d.arch.push_back(prop.gcnArchName);
      int node = -1;
      char bdf[64] = {0};
      if (hipDeviceGetPCIBusId(bdf, int(sizeof bdf), id) == hipSuccess) {
        std::string b = bdf;
        for (auto &ch : b) ch = char(std::tolower(static_cast<unsigned char>(ch)));
        if (FILE *fp = std::fopen(("/sys/bus/pci/devices/" + b + "/numa_node").c_str(), "r")) {
          if (std::fscanf(fp, "%d", &node) != 1) node = -1;
          std::fclose(fp);
        }
      }
      (void)hipGetLastError();
      d.numa.push_back(node);
    }
    if (d.ids.empty()) d.why = "INFERA_DEVICES selects no usable HIP device";
    return d;
  }();
  return ds;
}

DeviceModel::~DeviceModel() {
  if (device < 0) return;
  UnsafeOpGuard guard;
  if (hipSetDevice(device) != hipSuccess) return;
  (void)hipDeviceSynchronize();
  for (auto &d : steps) {
    for (float *p : {d.W, d.bias, d.cst, d.scale, d.shift})
      if (p) (void)hipFree(p);
  }
  if (mlp3_packed) (void)hipFree(mlp3_packed);
  for (float *p : chain_packed)
    if (p) (void)hipFree(p);
}

std::shared_ptr<LoadedModel> build_model(const std::string &name, const std::string &path, const std::string &output_select) {
  static std::atomic<uint64_t> next_uid{1};
  auto m = std::make_shared<LoadedModel>();
  m->uid = next_uid.fetch_add(1);
  m->name = name;
  onnx::Model om = onnx::parse_file(path);
  m->plan = lower_model(om, output_select);
  schedule(*m);
  const DeviceSet &ds = devices();
  if (ds.ids.empty()) {
    // No GPU: the model is registered (metadata, shape validation and the error paths above the
    // compute call keep working) but cannot execute; see run_host/run_device.
    m->device_error = ds.why;
    log_msg(1, "model '" + name + "' loaded without a GPU: " + ds.why + "; predictions will fail");
    return m;
  }
  for (size_t i = 0; i < ds.ids.size(); i++) {
    auto dm = std::make_unique<DeviceModel>();
    dm->device = ds.ids[i];
    dm->num_cus = ds.cus[i];
    upload_to_device(*m, *dm);
    m->dev.push_back(std::move(dm));
  }
  return m;
}

// A DuckDB scan on a 256-thread host calls the C ABI from every worker at once.  Dozens of streams each pushing one
// small H2D + kernel + D2H per 2048-row chunk collapse the HIP submission path (measured: 229 M rows/s with 16 threads,
// 71 M with 48 on a 30-column model), so each GPU admits INFERA_MAX_INFLIGHT calls (default 12) between their first
// H2D and their sync; the others finish gathering their chunk into pinned memory and wait their turn.
class SubmitGate {
 public:
  int acquire(int limit) {  // returns the number of calls in flight on this GPU, this one included
    if (limit <= 0) return 1;
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return in_flight_ < limit; });
    approx_.store(in_flight_ + 1, std::memory_order_relaxed);
    return ++in_flight_;
  }
  void release(int limit) {
    if (limit <= 0) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      in_flight_--;
      approx_.store(in_flight_, std::memory_order_relaxed);
    }
    cv_.notify_one();
  }
  int peek() const { return approx_.load(std::memory_order_relaxed); }  // calls in flight right now (unlocked read: a hint)

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int in_flight_ = 0;
  std::atomic<int> approx_{0};
};
// One gate per PHYSICAL GPU, not per device slot: two slots on one GPU (INFERA_DEVICES=0,0) used to admit 2 x 12 calls onto the
// same submission path -- the 2-slot scan fell from 101 to 66 M rows/s between 16 and 32 caller threads where the 1-slot scan
// held 94-110.  INFERA_MAX_INFLIGHT_TOTAL adds a process-wide cap on top (all GPUs share one HIP runtime).
SubmitGate &gate_for_slot(int slot) {
  static SubmitGate gates[64];
  return gates[size_t(devices().ids[size_t(slot)]) % 64];
}
SubmitGate &total_gate() {
  static SubmitGate g;
  return g;
}
struct GateHold {
  SubmitGate &g;
  int limit, total_limit;
  int in_flight = 1;
  GateHold(SubmitGate &gate, int lim, int total_lim) : g(gate), limit(lim), total_limit(total_lim) {
    if (total_limit > 0) total_gate().acquire(total_limit);  // (order: process-wide first, then the GPU's -- released in reverse)
    in_flight = g.acquire(limit);
  }
  ~GateHold() {
    g.release(limit);
    if (total_limit > 0) total_gate().release(total_limit);
  }
};

// Can the model's first kernel take a host call of `rows` rows as column-major chunks as they lie?  It must be able to read one
// (in_colmajor_ok, up to in_colmajor_max_rows) AND every host pass must be ONE device pass: a plan with activation scratch cuts long
// calls into row passes, which a [K][rows] chunk cannot be cut into (300k rows x 16 columns through a narrow Dense +
// unfused tail read the chunk with the wrong stride).
bool colmajor_direct_ok(const LoadedModel &m, int64_t rows) {
  if (!m.in_colmajor_ok || rows <= 0) return false;
  const size_t widest = std::max(size_t(m.plan.in_per_row()), size_t(m.plan.out_per_row())) * 4;
  const int64_t nr = std::min<int64_t>(rows, std::max<int64_t>(1, int64_t(kHostPassBytes / widest)));
  return nr <= m.in_colmajor_max_rows && rows_per_pass(m, nr) == nr;
}

namespace {
bool run_host_impl(const LoadedModel &m, const FillFn &fill, const DeviceFillFn *dfill, float *h_out, int64_t rows, bool col_major);
// the call on the thread's home slot; a device fault there takes the slot out of service and the call goes to the next healthy one
bool run_host_redealt(const LoadedModel &m, const FillFn &fill, const DeviceFillFn *dfill, float *h_out, int64_t rows, bool col_major) {
  for (;;) {
    const int slot = home_slot();
    try {
      return run_host_impl(m, fill, dfill, h_out, rows, col_major);
    } catch (const HipFault &f) {
      (void)hipGetLastError();
      if (!is_device_fault(f.code)) throw;  // the GPU is fine: the call itself was refused (too big for what is free, a bad argument)
      mark_slot_unhealthy(slot, f.what());
      if (healthy_slots() == 0) throw;
    }
  }
}
}
void run_host_fill(const LoadedModel &m, const FillFn &fill, float *h_out, int64_t rows, bool col_major) {
  (void)run_host_redealt(m, fill, nullptr, h_out, rows, col_major);
}
// Zero-copy fetches a GPU has in flight right now.  GPU-initiated reads of host memory top out at ~42 GB/s on this link whoever issues them, and
// two to four fetches in flight already reach that; the copy engines, which the STAGED path uses, read host memory at 56.  So with more callers
// than INFERA_ZERO_COPY_MAX_INFLIGHT (per GPU) the surplus chunks take the staged path -- the two mechanisms share the link instead of queueing
// on the slower one (false = "stage it", exactly as for a chunk outside the registered ranges).
std::atomic<int> g_zc_fetches[64];
bool run_host_device_fill(const LoadedModel &m, const DeviceFillFn &dfill, float *h_out, int64_t rows) {
  const int limit = Config::get().zero_copy_max_inflight;
  std::atomic<int> &n = g_zc_fetches[size_t(home_slot()) % 64];
  if (limit > 0 && n.fetch_add(1, std::memory_order_relaxed) >= limit) {
    n.fetch_sub(1, std::memory_order_relaxed);
    return false;
  }
  struct Leave {
    std::atomic<int> *n;
    ~Leave() {
      if (n) n->fetch_sub(1, std::memory_order_relaxed);
    }
  } leave{limit > 0 ? &n : nullptr};
  return run_host_redealt(m, FillFn(), &dfill, h_out, rows, /*col_major=*/true);
}
bool slot_health(int slot, std::string *fault) {
  std::lock_guard<std::mutex> lk(g_fault_mu);
  if (fault) *fault = g_slot_fault[size_t(slot) % 64];
  return !g_slot_unhealthy[size_t(slot) % 64].load(std::memory_order_acquire);
}

namespace {
bool run_host_impl(const LoadedModel &m, const FillFn &fill, const DeviceFillFn *dfill, float *h_out, int64_t rows, bool col_major) {
  if (m.dev.empty()) throw InferaError::onnx("HIP backend unavailable: " + m.device_error);
  if (rows <= 0) return true;
  if (dfill) {  // zero-copy: one host pass, direct (non-graph) enqueue only -- anything else goes the staging way
    const size_t widest_row = std::max(size_t(m.plan.in_per_row()), size_t(m.plan.out_per_row())) * 4;
    if (Config::get().use_hipgraph || size_t(rows) * widest_row > kPipePassBytes + kPipePassBytes / 2) return false;
  }
  const int slot = home_slot();
  const uint64_t t_entry = now_ns();
  HostLease lease(slot);
  const uint64_t t_leased = now_ns();
  ThreadCtx &ctx = *lease.c;
  if (fault_injected(slot)) hip_fail(hipErrorLaunchFailure, "injected fault (INFERA_FAULT_INJECT)");
  const DeviceModel &dm = device_model(m, slot);
  g_slot_calls[size_t(slot) % 64].fetch_add(1, std::memory_order_relaxed);
  g_slot_rows[size_t(slot) % 64].fetch_add(uint64_t(rows), std::memory_order_relaxed);
  const size_t in_row = size_t(m.plan.in_per_row()) * 4, out_row = size_t(m.plan.out_per_row()) * 4;
  const size_t widest = std::max(in_row, out_row);
  const bool use_graph = Config::get().use_hipgraph;
  // hipGraph mode captures {H2D, kernels[, D2H]}: a column-major chunk only when the model's first kernel reads it itself (the
  // transposing path allocates per pass, which a capture cannot contain)
  if (col_major && use_graph && !m.in_colmajor_ok) throw InferaError::onnx("internal: column-major staging is not captured in hipGraph mode for this plan");
  // H2D of one pass; a column-major pass lands in dev_cm first and is transposed into the row-major table on the GPU
  auto h2d_bytes = [&](int64_t nr) { return size_t(nr) * in_row; };
  auto upload_pass = [&](const float *pin, float *din, int64_t nr) {
    if (col_major) {
      ctx.ensure_dev(ctx.dev_cm, ctx.dev_cm_cap, size_t(nr) * in_row);
      HIP_TRY(hipMemcpyAsync(ctx.dev_cm, pin, h2d_bytes(nr), hipMemcpyHostToDevice, ctx.stream));
      kern::transpose_cm(ctx.stream, ctx.dev_cm, din, nr, int64_t(in_row / 4));
    } else {
      HIP_TRY(hipMemcpyAsync(din, pin, h2d_bytes(nr), hipMemcpyHostToDevice, ctx.stream));
    }
  };
  if (!use_graph && size_t(rows) * widest > kPipePassBytes + kPipePassBytes / 2 && size_t(rows) > 1) {
    // Larger host inputs (a whole BLOB batch, a big infera_predict call): two staging slots.  The CPU copy of
    // pass i+1 into pinned memory -- the slowest stage, the caller's buffer is only borrowed -- overlaps the
    // H2D / kernels / D2H of pass i instead of following them.
    // Pass size: 16 MB keeps the CPU copy and the H2D of table rows overlapped best; rows as big as images (602 KB) get
    // many more of them per pass, because a 27-image pass leaves the conv kernels half empty
    // (ResNet-18, 16 threads x 256-image calls: 18.5k img/s with 16 MB passes, 29.7k -- the resident rate -- with 64 MB).
    const int64_t by_bytes = std::max<int64_t>(1, int64_t(kPipePassBytes / widest));
    // (round 3: up to 256 such rows per pass -- 96-image passes left C5 at 0.90 of its resident rate end to end, 256-image passes reach 0.96:
    // 31.97k -> 34.0k img/s at 16 callers, 33.6k at 192, 34.1k at 384.  The pinned staging this costs -- two passes of inputs and two of
    // results per context -- is bounded by the context's share of kPinnedBudgetPerSlot: 221 images of 602 KB per pass.)
    const int64_t share_rows = int64_t(kPinnedBudgetPerSlot / size_t(std::max(1, Config::get().host_contexts)) / (2 * (in_row + out_row)));
    const int64_t by_rows = std::min<int64_t>(std::min<int64_t>(256, std::max<int64_t>(16, share_rows)), std::max<int64_t>(1, int64_t(4 * kHostPassBytes / widest)));
    // equal passes, at least two when the call is worth cutting: the CPU copy of pass 2 must overlap the GPU's pass 1 also when ONE
    // caller brings one batch (256 images as 128 + 128, not as a single pass with nothing to overlap)
    const int64_t p0 = std::min<int64_t>(rows, std::max(by_bytes, by_rows));
    const int64_t npass = std::max<int64_t>(rows >= 64 ? 2 : 1, (rows + p0 - 1) / p0);
    const int64_t P = (rows + npass - 1) / npass;
    ctx.ensure_pinned(ctx.pin_in, ctx.pin_in_cap, 2 * size_t(P) * in_row);
    ctx.ensure_pinned(ctx.pin_out, ctx.pin_out_cap, 2 * size_t(P) * out_row);
    ctx.ensure_dev(ctx.dev_in, ctx.dev_in_cap, 2 * size_t(P) * in_row);
    ctx.ensure_dev(ctx.dev_out, ctx.dev_out_cap, 2 * size_t(P) * out_row);
    for (auto &e : ctx.pipe_ev)
      if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // Row-major passes are copied to the device on ONE copy stream per GPU slot, shared by all its contexts, and a pass's kernels wait for
    // the copy's event.  (1) On the context's own stream the copy of pass i + 1 queues behind the kernels of pass i.  (2) Copies of
    // several callers on several streams SHARE the link: sixteen 133 MB copies issued together all arrive after 38 ms, where one at a time
    // the first arrives after 2.4 ms and its kernels start -- a kernel + copy trace of the C5 scan at 16 callers showed the matrix cores
    // idle 8 % of the time, always with copies running (profiles/r04_e2e_timeline.txt).  One stream is first-come-first-served at full
    // link speed.  (A column-major pass lands in the single dev_cm buffer first: it stays on the context's stream.)
    hipStream_t copy_stream = nullptr;
    std::mutex *copy_mu = nullptr;
    if (!col_major) {
      copy_stream = big_copy_stream(ctx.slot, copy_mu);
      for (auto &e : ctx.h2d_ev)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    int64_t pend_r0[2] = {0, 0}, pend_nr[2] = {0, 0};
    auto slot_ptr = [&](float *base, int k, size_t row_bytes) { return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + size_t(k) * size_t(P) * row_bytes); };
    auto drain = [&](int k) {
      if (!pend_nr[k]) return;
      const uint64_t t0 = now_ns();
      // (poll mode: nap instead of hipEventSynchronize's spin -- a BLOB batch is milliseconds of GPU time per pass)
      ctx.wait_event(ctx.pipe_ev[k], ctx.pipe_est, m.uid * 0x9E3779B97F4A7C15ull ^ uint64_t(pend_nr[k]) ^ (uint64_t(1) << 63));
      const uint64_t t1 = now_ns();
      std::memcpy(h_out + size_t(pend_r0[k]) * (out_row / 4), slot_ptr(ctx.pin_out, k, out_row), size_t(pend_nr[k]) * out_row);
      g_phase_ns[kPhWait].fetch_add(t1 - t0, std::memory_order_relaxed);
      g_phase_ns[kPhCopyOut].fetch_add(now_ns() - t1, std::memory_order_relaxed);
      pend_nr[k] = 0;
    };
    int k = 0;
    try {
      for (int64_t r0 = 0; r0 < rows; r0 += P, k ^= 1) {
        const int64_t nr = std::min(P, rows - r0);
        drain(k);
        float *pin = slot_ptr(ctx.pin_in, k, in_row), *din = slot_ptr(ctx.dev_in, k, in_row), *dout = slot_ptr(ctx.dev_out, k, out_row);
        const uint64_t t_f0 = now_ns();
        fill(pin, r0, nr);
        const uint64_t t_f1 = now_ns();
        if (col_major) {
          upload_pass(pin, din, nr);
        } else {  // (slot k's device buffer is free: drain(k) waited for the pass that used it last)
          {
            std::lock_guard<std::mutex> lk(*copy_mu);  // (copy + its event as one unit: the event must not cover a later caller's copy)
            HIP_TRY(hipMemcpyAsync(din, pin, h2d_bytes(nr), hipMemcpyHostToDevice, copy_stream));
            HIP_TRY(hipEventRecord(ctx.h2d_ev[k], copy_stream));
          }
          HIP_TRY(hipStreamWaitEvent(ctx.stream, ctx.h2d_ev[k], 0));
        }
        exec_plan(m, dm, ctx, din, dout, nr);
        HIP_TRY(hipMemcpyAsync(slot_ptr(ctx.pin_out, k, out_row), dout, size_t(nr) * out_row, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipEventRecord(ctx.pipe_ev[k], ctx.stream));
        g_phase_ns[kPhGather].fetch_add(t_f1 - t_f0, std::memory_order_relaxed);  // (the per-phase counters of infera_hip_get_devices)
        g_phase_ns[kPhEnqueue].fetch_add(now_ns() - t_f1, std::memory_order_relaxed);
        g_phase_calls.fetch_add(1, std::memory_order_relaxed);
        pend_r0[k] = r0;
        pend_nr[k] = nr;
      }
      drain(k);
      drain(k ^ 1);
    } catch (...) {
      if (copy_stream) (void)hipStreamSynchronize(copy_stream);
      (void)hipStreamSynchronize(ctx.stream);  // nothing may still be reading the staging slots when we unwind
      throw;
    }
    return true;
  }
  int64_t rows_pass = std::max<int64_t>(1, int64_t(kHostPassBytes / widest));
  rows_pass = std::min(rows_pass, rows);
  const bool direct_out = m.out_write_once && size_t(rows_pass) * out_row <= (1u << 20);
  const bool cm_graph = use_graph && col_major;  // (implies m.in_colmajor_ok)
  if (!dfill) ctx.ensure_pinned(ctx.pin_in, ctx.pin_in_cap, size_t(rows_pass) * in_row);
  ctx.ensure_pinned(ctx.pin_out, ctx.pin_out_cap, size_t(rows_pass) * out_row);
  ctx.ensure_dev(ctx.dev_in, ctx.dev_in_cap, size_t(rows_pass) * in_row);
  ctx.ensure_dev(ctx.dev_out, ctx.dev_out_cap, size_t(rows_pass) * out_row);
  // H2D (or not: small inputs are read from pinned memory by the kernel itself) + the plan's kernels [+ D2H] of ONE pass whose staged
  // input is at `pin`, on ctx.stream; results land at `pout` (a position in the pinned result buffer).  Non-graph mode.
  int64_t dfill_r0 = 0;  // (zero-copy) first row of the pass being enqueued
  auto enqueue_pass = [&](const float *pin, float *din, float *dout, float *pout, int64_t nr, bool single_pass, int in_flight) {
    // column-major chunk straight into the model's first kernel when it can read one (no transpose launch)
    const bool cm_direct = col_major && m.in_colmajor_ok && nr <= m.in_colmajor_max_rows && single_pass;
    // Small inputs (a narrow table's chunk, a point query; INFERA_HOST_DIRECT_IN bytes, default 128 KB) are not copied to HBM first:
    // the kernel that reads them -- the plan's first kernel, or the transpose in front of it -- loads them from the pinned
    // (host-coherent) buffer over PCIe itself.  One submission less per call, and a DMA engine's start-up latency is as long as
    // such a transfer: 13-column table +6..17 % at 2..64 threads.  (1 MB chunks read this way reach 34 GB/s against the copy
    // engines' 50: C2 and every larger chunk keep the H2D copy.)
    // (a quiet GPU reads larger chunks this way -- twice that size with at most four calls in flight, four times with at most two: a
    // few kernels pulling over PCIe do not yet compete with each other, and the copy engine's latency is the larger part of such a
    // call.  30 -> 100 -> 2, 245 KB chunks: +17 / +8 / +6 / +4 % at 1 / 2 / 4 / 8 threads; 64 -> 128 -> 64 -> 1, 512 KB: +13 / +6 % at 1 / 2)
    const int quiet_mult = in_flight <= 2 ? 4 : in_flight <= 4 ? 2 : 1;
    const int64_t din_limit = int64_t(Config::get().host_direct_in_bytes) * quiet_mult;
    const bool small_in = int64_t(nr) * int64_t(in_row) <= din_limit;
    const float *kin = din;
    if (dfill) {
      // zero-copy: the GPU pulls the caller's (registered) column runs itself -- straight into the chunk the first kernel reads when it
      // reads column-major chunks, else into the transpose's source
      if (cm_direct) {
        (*dfill)(ctx.stream, din, dfill_r0, nr);
      } else {
        ctx.ensure_dev(ctx.dev_cm, ctx.dev_cm_cap, size_t(nr) * in_row);
        (*dfill)(ctx.stream, ctx.dev_cm, dfill_r0, nr);
        kern::transpose_cm(ctx.stream, ctx.dev_cm, din, nr, int64_t(in_row / 4));
      }
    } else if (cm_direct) {
      if (small_in) kin = pin;
      else HIP_TRY(hipMemcpyAsync(din, pin, h2d_bytes(nr), hipMemcpyHostToDevice, ctx.stream));
    } else if (col_major && small_in) {
      kern::transpose_cm(ctx.stream, pin, din, nr, int64_t(in_row / 4));
    } else if (!col_major && small_in && m.in_single_reader) {
      kin = pin;
    } else {
      upload_pass(pin, din, nr);
    }
    if (direct_out) {
      // the plan's only writer of the result stores it straight into the pinned (host-coherent) buffer: a few KB per
      // chunk over PCIe from the kernel's epilogue instead of one more enqueue + blit kernel + dependency per chunk
      exec_plan(m, dm, ctx, kin, pout, nr, cm_direct);
    } else {
      exec_plan(m, dm, ctx, kin, dout, nr, cm_direct);
      HIP_TRY(hipMemcpyAsync(pout, dout, size_t(nr) * out_row, hipMemcpyDeviceToHost, ctx.stream));
    }
  };
  // (Measured and dropped in round 3: one chunk as two sub-passes on the call's stream, the gather of the second overlapping the H2D + kernels
  // of the first -- a loss at every caller count: a chunk's 45-50 us in flight are fixed latencies, not its 20 us of transfer, and halves pay
  // them twice; profiles/r03_host_cpu_ab_split_pollq.txt.)
  for (int64_t r0 = 0; r0 < rows; r0 += rows_pass) {
    const int64_t nr = std::min(rows_pass, rows - r0);
    // The caller's buffer is only borrowed for the call (SURVEY.md 8b "Ownership"): stage it.
    const uint64_t t_f0 = now_ns();
    if (!dfill) fill(ctx.pin_in, r0, nr);
    dfill_r0 = r0;
    const uint64_t t_f1 = now_ns();
    GateHold admitted(gate_for_slot(slot), Config::get().max_inflight, Config::get().max_inflight_total);  // until this pass has been synchronised
    const uint64_t t_g = now_ns();
    hipGraphExec_t exec = nullptr;
    // one device pass for the whole chunk?  (plans with activation scratch split long calls; may reallocate -- and drop graphs --
    // so before the lookup).  A column-major chunk is only handed to the first kernel as it lies when it is.
    const bool single_pass = prepare_scratch(m, ctx, nr) == nr;
    if (cm_graph && !single_pass) throw InferaError::onnx("internal: column-major chunk of " + std::to_string(nr) + " rows needs several device passes; not captured in hipGraph mode");
    if (use_graph) {
      for (auto &g : ctx.graphs)
        if (g.uid == m.uid && g.rows == (nr | (cm_graph ? int64_t(1) << 62 : 0))) {
          g.last_use = ++ctx.graph_clock;
          exec = g.exec;
        }
      if (exec) {
        HIP_TRY(hipGraphLaunch(exec, ctx.stream));
      } else {
        // First chunk of this (model, rows) on this context: run it directly -- one-time work such as
        // hipFuncSetAttribute / code-object loading must not happen inside a capture -- and then record
        // the graph (capturing does not execute anything) for the chunks that follow.
        HIP_TRY(hipMemcpyAsync(ctx.dev_in, ctx.pin_in, size_t(nr) * in_row, hipMemcpyHostToDevice, ctx.stream));
        exec_plan(m, dm, ctx, ctx.dev_in, direct_out ? ctx.pin_out : ctx.dev_out, nr, cm_graph);
        if (!direct_out) HIP_TRY(hipMemcpyAsync(ctx.pin_out, ctx.dev_out, size_t(nr) * out_row, hipMemcpyDeviceToHost, ctx.stream));
        HIP_TRY(hipStreamSynchronize(ctx.stream));
        std::memcpy(h_out + size_t(r0) * (out_row / 4), ctx.pin_out, size_t(nr) * out_row);
        hipGraph_t graph = nullptr;
        std::shared_lock<std::shared_mutex> capture_lock(g_capture_mu);
        HIP_TRY(hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeThreadLocal));
        hipError_t e = hipMemcpyAsync(ctx.dev_in, ctx.pin_in, size_t(nr) * in_row, hipMemcpyHostToDevice, ctx.stream);
        try {
          if (e == hipSuccess) exec_plan(m, dm, ctx, ctx.dev_in, direct_out ? ctx.pin_out : ctx.dev_out, nr, cm_graph);
        } catch (...) {
          (void)hipStreamEndCapture(ctx.stream, &graph);
          if (graph) (void)hipGraphDestroy(graph);
          throw;
        }
        if (e == hipSuccess && !direct_out) e = hipMemcpyAsync(ctx.pin_out, ctx.dev_out, size_t(nr) * out_row, hipMemcpyDeviceToHost, ctx.stream);
        hipError_t e2 = hipStreamEndCapture(ctx.stream, &graph);
        if (e != hipSuccess) hip_fail(e, "stream capture");
        if (e2 != hipSuccess) hip_fail(e2, "hipStreamEndCapture");
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) hip_fail(e, "hipGraphInstantiate");
        if (ctx.graphs.size() >= 16) {  // evict the least recently used entry
          size_t victim = 0;
          for (size_t i = 1; i < ctx.graphs.size(); i++)
            if (ctx.graphs[i].last_use < ctx.graphs[victim].last_use) victim = i;
          (void)hipGraphExecDestroy(ctx.graphs[victim].exec);
          ctx.graphs.erase(ctx.graphs.begin() + long(victim));
        }
        ctx.graphs.push_back({m.uid, nr | (cm_graph ? int64_t(1) << 62 : 0), exec, ++ctx.graph_clock});
        continue;  // this chunk's result is already in h_out
      }
    } else {
      enqueue_pass(ctx.pin_in, ctx.dev_in, ctx.dev_out, ctx.pin_out, nr, single_pass, admitted.in_flight);
    }
    const uint64_t t_e = now_ns();
    ctx.wait_stream(m.uid * 0x9E3779B97F4A7C15ull ^ uint64_t(nr));  // (spinning on hipStreamQuery instead measured slower: 41 vs 69 M rows/s at 16 threads)
    const uint64_t t_w = now_ns();
    std::memcpy(h_out + size_t(r0) * (out_row / 4), ctx.pin_out, size_t(nr) * out_row);
    const uint64_t t_c = now_ns();
    if (r0 == 0) g_phase_ns[kPhLease].fetch_add(t_leased - t_entry, std::memory_order_relaxed);
    g_phase_ns[kPhGather].fetch_add(t_f1 - t_f0, std::memory_order_relaxed);
    g_phase_ns[kPhGate].fetch_add(t_g - t_f1, std::memory_order_relaxed);
    g_phase_ns[kPhEnqueue].fetch_add(t_e - t_g, std::memory_order_relaxed);
    g_phase_ns[kPhWait].fetch_add(t_w - t_e, std::memory_order_relaxed);
    g_phase_ns[kPhCopyOut].fetch_add(t_c - t_w, std::memory_order_relaxed);
    g_phase_calls.fetch_add(1, std::memory_order_relaxed);
  }
  return true;
}
}  // namespace

// ---- registered host memory (zero-copy host path) -----------------------------------------------------------------------
// The runtime pins whole pages, callers register byte ranges (numpy arrays, malloc'ed buffers, a database allocator's blocks: neighbours on
// the heap share pages).  So a registered RANGE (what lookups test against) is covered by one or more page BLOCKS (what hipHostRegister
// was called on): registering a range pins only the pages no earlier block covers -- a block is never replaced or re-registered once it
// exists (round 3 merged neighbours into one new registration, which unmapped memory under running calls and forced every (un)registration
// to drain ALL zero-copy calls in flight).  A block lives until the last range on its pages is unregistered; then it leaves the index at
// once and is unmapped as soon as the calls that PINNED it (found it in a lookup and have not finished) are done -- only those calls are
// waited for, every other call and every registration proceeds.  Lookups are O(log n) under a shared lock; the writers serialise among
// themselves on a separate mutex and hold the index lock only for the map update, not for the hipHostRegister / hipHostUnregister calls.
namespace {
struct PageBlock {
  uintptr_t pb, pe;              // page-aligned span handed to hipHostRegister
  intptr_t dev_delta;            // device-visible address = host address + dev_delta (the same on every selected GPU: checked)
  int refs = 0;                  // registered ranges that touch these pages (under g_reg_mu)
  std::atomic<int> readers{0};   // zero-copy calls in flight that resolved an address inside this block
};
struct HostRange {
  uintptr_t end;
  bool usable;  // all covering blocks share one device delta (always, on the systems seen so far)
  intptr_t dev_delta;
};
std::mutex g_reg_mu;                                            // serialises register / unregister
std::shared_mutex g_index_mu;                                   // g_ranges / g_blocks (readers: lookups)
std::map<uintptr_t, HostRange> g_ranges;                        // by base; non-overlapping
std::map<uintptr_t, std::shared_ptr<PageBlock>> g_blocks;       // by pb; non-overlapping
std::atomic<size_t> g_nranges{0};

// pins [pb, pe) and returns its device delta, the same on every selected GPU or an error
// (un)registration is called from the APPLICATION's threads (an allocator hook): their current HIP device is put back afterwards
struct DeviceRestore {
  int dev = -1;
  DeviceRestore() {
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  }
  ~DeviceRestore() {
    if (dev >= 0) (void)hipSetDevice(dev);
  }
};

intptr_t hip_register_span(uintptr_t pb, uintptr_t pe) {
  const auto &ds = devices();
  DeviceRestore restore;
  // portable + mapped: visible to every selected GPU; the pages stay where they are (no copy), pinned until unregistered
  HIP_TRY(hipSetDevice(ds.ids[0]));
  HIP_TRY(hipHostRegister(reinterpret_cast<void *>(pb), pe - pb, hipHostRegisterPortable | hipHostRegisterMapped));
  intptr_t delta = 0;
  for (size_t i = 0; i < ds.ids.size(); i++) {
    void *dptr = nullptr;
    hipError_t ge = hipSetDevice(ds.ids[i]);
    if (ge == hipSuccess) ge = hipHostGetDevicePointer(&dptr, reinterpret_cast<void *>(pb), 0);
    const intptr_t d = intptr_t(reinterpret_cast<uintptr_t>(dptr)) - intptr_t(pb);
    if (ge != hipSuccess || (i > 0 && d != delta)) {
      (void)hipHostUnregister(reinterpret_cast<void *>(pb));
      if (ge != hipSuccess) hip_fail(ge, "hipHostGetDevicePointer");
      throw InferaError::onnx("registered host memory has different device addresses on different GPUs");
    }
    delta = d;
  }
  return delta;
}
}  // namespace

void register_host_memory(const void *base, size_t bytes) {
  if (!base || !bytes) throw InferaError::null_pointer();
  const auto &ds = devices();
  if (ds.ids.empty()) throw InferaError::onnx("HIP backend unavailable: " + ds.why);
  const uintptr_t b = reinterpret_cast<uintptr_t>(base), e = b + bytes;
  const uintptr_t pb = b & ~uintptr_t(4095), pe = (e + 4095) & ~uintptr_t(4095);
  std::lock_guard<std::mutex> writer(g_reg_mu);
  {  // (nobody else mutates the index while g_reg_mu is held: reading it unlocked is safe here)
    auto it = g_ranges.upper_bound(b);
    if (it != g_ranges.end() && it->first < e) throw InferaError::onnx("host memory range overlaps a registered range");
    if (it != g_ranges.begin() && std::prev(it)->second.end > b) throw InferaError::onnx("host memory range overlaps a registered range");
  }
  // pages of [pb, pe) that no block covers yet -> new blocks, pinned OUTSIDE the index lock
  std::vector<std::shared_ptr<PageBlock>> fresh, covering;
  UnsafeOpGuard guard;
  try {
    uintptr_t at = pb;
    auto it = g_blocks.upper_bound(pb);
    if (it != g_blocks.begin() && std::prev(it)->second->pe > pb) --it;
    for (; at < pe; ++it) {
      const uintptr_t gap_end = it == g_blocks.end() || it->second->pb >= pe ? pe : it->second->pb;
      if (gap_end > at) {
        auto blk = std::make_shared<PageBlock>();
        blk->pb = at;
        blk->pe = gap_end;
        blk->dev_delta = hip_register_span(at, gap_end);
        fresh.push_back(blk);
        covering.push_back(blk);
      }
      if (it == g_blocks.end() || it->second->pb >= pe) break;
      covering.push_back(it->second);
      at = it->second->pe;
    }
  } catch (...) {
    for (auto &blk : fresh) (void)hipHostUnregister(reinterpret_cast<void *>(blk->pb));  // nothing else changed: the index is as it was
    throw;
  }
  HostRange r{e, true, covering.front()->dev_delta};
  for (auto &blk : covering) r.usable = r.usable && blk->dev_delta == r.dev_delta;
  {
    std::unique_lock<std::shared_mutex> lk(g_index_mu);
    for (auto &blk : fresh) g_blocks.emplace(blk->pb, blk);
    for (auto &blk : covering) blk->refs++;
    g_ranges.emplace(b, r);
    g_nranges.store(g_ranges.size(), std::memory_order_release);
  }
  if (!r.usable) log_msg(1, "registered host range is covered by blocks with different device addresses: its chunks take the staged path");
}

bool unregister_host_memory(const void *base) {
  const uintptr_t b = reinterpret_cast<uintptr_t>(base);
  std::lock_guard<std::mutex> writer(g_reg_mu);
  std::vector<std::shared_ptr<PageBlock>> dead;
  {
    std::unique_lock<std::shared_mutex> lk(g_index_mu);
    auto rit = g_ranges.find(b);
    if (rit == g_ranges.end()) return false;
    const uintptr_t pb = b & ~uintptr_t(4095), pe = (rit->second.end + 4095) & ~uintptr_t(4095);
    g_ranges.erase(rit);
    g_nranges.store(g_ranges.size(), std::memory_order_release);
    auto it = g_blocks.upper_bound(pb);
    if (it != g_blocks.begin() && std::prev(it)->second->pe > pb) --it;
    while (it != g_blocks.end() && it->second->pb < pe) {
      if (--it->second->refs == 0) {  // (a block shared with neighbours stays pinned until the last of them goes)
        dead.push_back(it->second);
        it = g_blocks.erase(it);     // no later lookup can find it
      } else {
        ++it;
      }
    }
  }
  // unmap the dead blocks once the calls that pinned them have finished (a chunk's time, ~100 us); nobody else waits for anything
  UnsafeOpGuard guard;
  for (auto &blk : dead) {
    for (int spin = 0; blk->readers.load(std::memory_order_acquire) > 0; spin++) {
      if (spin < 64) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    (void)hipHostUnregister(reinterpret_cast<void *>(blk->pb));
  }
  return true;
}

// Resolves n host runs to device-visible addresses; every run must lie inside ONE registered range.  The blocks under the runs are PINNED
// (reader count) until the returned guard dies -- hold it until the GPU has finished reading.  O(log n) per run, one shared lock.
ZeroCopyPins::~ZeroCopyPins() {
  for (size_t i = 0; i < count; i++) static_cast<PageBlock *>(blocks[i].get())->readers.fetch_sub(1, std::memory_order_release);
}

bool lookup_host_memory_many(size_t n, const void *const *ptrs, const size_t *bytes, const void **out, ZeroCopyPins &pins) {
  if (g_nranges.load(std::memory_order_acquire) == 0) return false;
  std::shared_lock<std::shared_mutex> lk(g_index_mu);
  uintptr_t hb = 0, he = 0;  // the range the previous run lay in (columns of one table usually share it)
  intptr_t hd = 0;
  PageBlock *last = nullptr;  // the block pinned last (consecutive runs usually share it too)
  for (size_t i = 0; i < n; i++) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptrs[i]), ae = a + bytes[i];
    if (!(a >= hb && ae <= he)) {
      auto it = g_ranges.upper_bound(a);
      if (it == g_ranges.begin()) return false;
      --it;
      if (ae > it->second.end || !it->second.usable) return false;
      hb = it->first;
      he = it->second.end;
      hd = it->second.dev_delta;
    }
    out[i] = reinterpret_cast<const void *>(intptr_t(a) + hd);
    // pin every block under [a, ae) (one, unless the run straddles a block border)
    for (uintptr_t at = a; at < ae;) {
      if (last && at >= last->pb && at < last->pe) {
        at = last->pe;
        continue;
      }
      auto bit = g_blocks.upper_bound(at);
      if (bit == g_blocks.begin()) return false;
      --bit;
      if (at >= bit->second->pe) return false;  // (cannot happen for a registered range)
      if (pins.count == ZeroCopyPins::kMax) return false;  // more blocks than a chunk is expected to touch: take the staged path
      bit->second->readers.fetch_add(1, std::memory_order_acquire);
      pins.blocks[pins.count++] = bit->second;
      last = bit->second.get();
      at = last->pe;
    }
  }
  return true;
}

const void *lookup_host_memory(const void *p, size_t bytes) {
  const void *out = nullptr;
  ZeroCopyPins pins;
  return lookup_host_memory_many(1, &p, &bytes, &out, pins) ? out : nullptr;  // (address only: the pin ends with this call)
}

size_t registered_host_ranges() { return g_nranges.load(std::memory_order_acquire); }

void copy_rect_to_device(hipStream_t stream, float *dst, const void *src, size_t src_pitch, size_t width, size_t height) {
  HIP_TRY(hipMemcpy2DAsync(dst, width, src, src_pitch, width, height, hipMemcpyHostToDevice, stream));
}
bool zero_copy_rect_enabled() { return Config::get().zero_copy_rect; }

void run_host(const LoadedModel &m, const float *h_in, float *h_out, int64_t rows) {
  const size_t in_per_row = size_t(m.plan.in_per_row());
  run_host_fill(m, [&](float *dst, int64_t r0, int64_t nr) { std::memcpy(dst, h_in + size_t(r0) * in_per_row, size_t(nr) * in_per_row * 4); },
                h_out, rows);
}

void run_device(const LoadedModel &m, int device_ordinal, const float *d_in, float *d_out, int64_t rows) {
  if (m.dev.empty()) throw InferaError::onnx("HIP backend unavailable: " + m.device_error);
  const int slot = slot_of_ordinal(device_ordinal);
  ThreadCtx &ctx = ctx_for_slot(slot);
  exec_plan(m, device_model(m, slot), ctx, d_in, d_out, rows);
}

void sync_device(int device_ordinal) {
  ThreadCtx &ctx = ctx_for_slot(slot_of_ordinal(device_ordinal));
  HIP_TRY(hipStreamSynchronize(ctx.stream));
}

hipStream_t thread_stream(int device_ordinal) { return ctx_for_slot(slot_of_ordinal(device_ordinal)).stream; }

std::string LoadedModel::describe_json() const {
  static const char *ek[] = {"normal", "skipped", "mlp3_fused", "dense_softmax", "conv_tiled_cq", "conv_patch", "conv_depthwise", "dense_tiled", "dense_argmax", "chain_fused"};
  std::ostringstream o;
  o << "{\"name\":" << json_str(name) << ",\"plan\":" << plan.describe_json() << ",\"exec\":[";
  for (size_t i = 0; i < exec.size(); i++)
    o << (i ? "," : "") << "\"" << (exec[i] == ExecKind::Skipped ? "skipped" : i < stem_split6.size() && stem_split6[i] ? "conv_patch_pool_bf16x6" : i < conv_fused_pool.size() && conv_fused_pool[i] >= 0 ? "conv_patch_pool" : i < conv_split6.size() && conv_split6[i] ? "conv_split_bf16x6" : ek[int(exec[i])]) << "\"";
  o << "],\"activation_layout\":\"" << (cq_mode ? "NC/4HW4" : "NCHW") << "\",\"scratch_floats_per_row\":" << scratch_per_row << ",\"devices\":[";
  for (size_t i = 0; i < dev.size(); i++) o << (i ? "," : "") << dev[i]->device;
  o << "]";
  if (std::find(conv_split6.begin(), conv_split6.end(), char(1)) != conv_split6.end())
    o << ",\"conv_precision\":\"bf16x6 (bf16 matrix cores, operands cut exactly into three parts, six partial products, fp32 accumulate)\"";
  {
    std::string folded;
    for (size_t i = 0; i < conv_fold.size(); i++)
      if (conv_fold[i] >= 0) folded += std::string(folded.empty() ? "" : ",") + std::to_string(conv_fold[i]);
    if (!folded.empty()) o << ",\"folded_shortcuts\":[" << folded << "]";
  }
  for (size_t i = 0; i < exec.size(); i++)
    if (exec[i] == ExecKind::Mlp3Head)
      o << ",\"fused_kernel\":" << json_str(kern::mlp3_kernel_name(mlp3_shape)) << ",\"precision\":\"fp32\"";
  {  // kernel family of every Dense layer that runs on the streaming / generic Dense kernels, for a large aligned device-resident scan
    std::string dk;
    for (size_t i = 0; i < exec.size(); i++) {
      const Step &x = plan.steps[i];
      if (x.kind != StepKind::Dense) continue;
      int sm = -1;
      if (exec[i] == ExecKind::DenseSoftmax) sm = plan.steps[i + 1].log_softmax ? 2 : 1;
      else if (exec[i] == ExecKind::DenseArgMax) sm = 3;
      else if (exec[i] == ExecKind::Normal) sm = 0;
      if (sm < 0) continue;
      dk += std::string(dk.empty() ? "" : ",") + json_str(kern::dense_kernel_family(int64_t(1) << 20, int(x.K), int(x.M), sm, false, true));
    }
    if (!dk.empty()) o << ",\"dense_kernels\":[" << dk << "]";
  }
  if (!chains.empty()) {
    o << ",\"chain_kernels\":[";
    for (size_t i = 0; i < chains.size(); i++) o << (i ? "," : "") << json_str(kern::chain_kernel_name(chains[i].shape));
    o << "]";
  }
  if (!device_error.empty()) o << ",\"device_error\":" << json_str(device_error);
  o << "}";
  return o.str();
}

}  // namespace infera_hip
