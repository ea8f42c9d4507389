// runtime.hpp -- what the translation units of the HIP backend share (internal; the public face is backend.hpp):
//   context.cpp    device discovery, per-thread / leased execution contexts, slot dealing, slot health, counters
//   schedule.cpp   plan -> kernels: fusion decisions, activation layout, scratch slots (device independent)
//   model.cpp      weight packing + upload, build_model
//   exec.cpp       the plan executor (device passes, lanes), the device-resident entry points
//   host_path.cpp  the host ABI: staging, admission gate, waits, the big-row pipeline, hipGraph replay, fault re-deal
//   zero_copy.cpp  registered host memory
#pragma once

#include <sys/prctl.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../host/common.hpp"
#include <optional>

#include "backend.hpp"
#include "profile.hpp"

namespace infera_hip {
namespace rt {

// A failed HIP call.  The text is what a caller sees (error.rs:24-25 "ONNX error: ..."); the code lets the host path tell a device fault
// (the slot is taken out of service and the call re-dealt to another GPU) from an allocation failure (the call fails, the GPU stays).
struct HipFault : InferaError {
  hipError_t code;
  HipFault(hipError_t e, const char *what) : InferaError(InferaError::onnx(std::string("HIP: ") + what + ": " + hipGetErrorString(e))), code(e) {}
};
[[noreturn]] inline void hip_fail(hipError_t e, const char *what) { throw HipFault(e, what); }
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
extern std::shared_mutex g_capture_mu;
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
extern std::atomic<uint64_t> g_pinned_bytes[64];  // per device slot: pinned staging held by its contexts right now (infera_hip_get_devices)

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
      std::optional<prof::Section> sec_slack(std::in_place, 13, "wait: timer slack set");
      TimerSlack slack;
      sec_slack.reset();
      if (known) {
        prof::Section sec(11, "wait: first nap (asleep)");
        nap(std::min({est.ema_ns * kPollFirst, est.min_ns * 0.9, 2.0e6}));
      }
      for (;;) {
        hipError_t e;
        {
          prof::Section sec(12, "wait: event queries");
          e = query();
        }
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) hip_fail(e, "hipEventQuery");
        prof::Section sec(16, "wait: later naps (asleep)");
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
  // waits for everything enqueued on `stream` so far.  `key` identifies the kind of work (0 = unknown).  (Querying the STREAM between the naps
  // instead of a marker event -- no record call, no marker packet -- measured 20-45 % slower at 5-45 us more CPU per chunk: profiles/r05_wait_stream_query_ab.txt.)
  void wait_stream(uint64_t key = 0) {
    if (!poll_ev) HIP_TRY(hipEventCreateWithFlags(&poll_ev, hipEventDisableTiming));
    {
      prof::Section sec(10, "wait: event record");
      HIP_TRY(hipEventRecord(poll_ev, stream));
    }
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

// ---- context.cpp ------------------------------------------------------------------------------------------------------------
int slot_of_ordinal(int ordinal);
// the calling thread's own context on device slot `slot` (device-resident entry points, weight uploads); makes the slot's device current
ThreadCtx &ctx_for_slot(int slot);

// Host-ABI calls do not own a context per caller thread: a DuckDB scan on a 256-thread host would pin 256 sets of
// staging buffers and per-model scratch (ResNet-18: ~1 GB each) and create 256 streams, for no throughput -- the path
// saturates at ~16 callers per GPU.  They lease one of at most INFERA_HOST_CONTEXTS (default 24) contexts per GPU for
// the duration of the call; the rest wait.  (The device-resident entry points keep the caller thread's own stream.)
struct HostPool;
struct HostLease {
  HostPool &pool;
  ThreadCtx *c = nullptr;
  explicit HostLease(int slot);
  ~HostLease();
  HostLease(const HostLease &) = delete;
  HostLease &operator=(const HostLease &) = delete;
};

// Host-ABI work served per device slot (calls, rows): lets a scan report how DuckDB's worker threads were dealt over
// the GPUs (infera_hip_get_devices), and lets the tests see that a second slot really took its share.
extern std::atomic<uint64_t> g_slot_calls[64], g_slot_rows[64];

// Where a host-ABI call's wall time goes (single-pass path = one DataChunk per call), summed over all calls, in ns:
// lease (waiting for a staging context), gather (caller's buffer -> pinned), gate (waiting for admission), enqueue
// (H2D + kernels [+ D2H] API calls), wait (until the device is done), copy_out (pinned -> result buffer).
// Seven steady_clock reads per call (~0.2 us) -- always on, reported by infera_hip_get_devices.
enum HostPhase { kPhLease, kPhGather, kPhGate, kPhEnqueue, kPhWait, kPhCopyOut, kPhCount };
extern std::atomic<uint64_t> g_phase_ns[kPhCount], g_phase_calls;
inline uint64_t now_ns() { return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count()); }

// Device-fault handling (SURVEY 5 "failure detection"): a HIP error on the host path other than an allocation failure takes the slot out
// of service -- its callers are re-dealt to the remaining slots and the failed chunk is run again there (the caller's buffers are only
// read, so a chunk can be staged twice); infera_hip_get_devices reports the slot as unhealthy with the error text.  With no healthy slot
// left the error goes to the caller as before (status -1 + last error, error.rs:13-61).  A slot stays out of service until the process
// restarts: nothing ever clears the flag (a sticky HIP fault does not clear either).
int healthy_slots();
bool slot_is_unhealthy(int slot);
void mark_slot_unhealthy(int slot, const std::string &why);
bool is_device_fault(hipError_t e);
bool fault_injected(int slot);  // TEST HOOK: INFERA_FAULT_INJECT=<slot>:<n>
// the device slot the calling thread's host-ABI calls go to (dealt on its first call; re-dealt when that slot went out of service)
int home_slot();

// ---- schedule.cpp -----------------------------------------------------------------------------------------------------------
inline kern::ActParam act_of(const Step &s) { return kern::ActParam{int(s.act), s.act_a, s.act_b}; }
// the geometry of a Conv2d step as the kernels take it
inline kern::ConvGeom conv_geom(const Step &s) {
  return kern::ConvGeom{int(s.C), int(s.H), int(s.Wd), int(s.Mo), int(s.OH), int(s.OW), int(s.kh), int(s.kw),
                        int(s.sh), int(s.sw), int(s.pt), int(s.pl), int(s.dh), int(s.dw), int(s.groups)};
}
inline kern::PoolTail pool_tail(const Step &q) { return kern::PoolTail{int(q.OH), int(q.OW), int(q.pt), int(q.pl)}; }
// A Dense layer is a 1x1 convolution over 1x1 "images": with H = W = 1 the channel-quad layout IS the row-major
// [rows, K] matrix, so the tiled conv kernel (packed weights through LDS, unit-pipelined MFMA stream) serves it.
// K and M are padded to multiples of 32 with zero weights; the real row lengths travel in kvalid / mvalid.
inline kern::ConvGeom dense_as_conv(const Step &s) {
  const int kp = int((s.K + 31) / 32 * 32), mp = int((s.M + 31) / 32 * 32);
  return kern::ConvGeom{kp, 1, 1, mp, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, int(s.K), int(s.M)};
}
// Which buffers an executed step reads / writes once fusion decisions are applied.
struct EffStep {
  int idx;
  std::vector<int> reads;
  int writes;
};
std::vector<EffStep> effective_steps(const LoadedModel &m);
// fills m.exec and every fusion / layout / scratch decision from m.plan (may permute weights of m.plan for the chosen layout)
void schedule(LoadedModel &m);

// ---- model.cpp --------------------------------------------------------------------------------------------------------------
void upload_to_device(const LoadedModel &m, DeviceModel &dm);

// ---- exec.cpp ---------------------------------------------------------------------------------------------------------------
// Rows per device pass for plans that need activation scratch (pure: no allocation).
int64_t rows_per_pass(const LoadedModel &m, int64_t rows);
// ... and grows the scratch for it (never inside a stream capture: callers that capture call this first).
int64_t prepare_scratch(const LoadedModel &m, ThreadCtx &ctx, int64_t rows);
// in_colmajor: d_in is one column-major chunk [in_per_row][rows] (only with m.in_colmajor_ok, which implies a single pass)
void exec_plan(const LoadedModel &m, const DeviceModel &dm, ThreadCtx &ctx, const float *d_in, float *d_out, int64_t rows, bool in_colmajor = false);
const DeviceModel &device_model(const LoadedModel &m, int slot);

}  // namespace rt
}  // namespace infera_hip
