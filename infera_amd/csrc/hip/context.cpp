// context.cpp -- device discovery, execution contexts (per thread and leased), how caller threads are dealt over the device slots,
// slot health, and the counters infera_hip_get_devices reports.  See runtime.hpp.
#include <sched.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "runtime.hpp"

namespace infera_hip {
namespace rt {

std::shared_mutex g_capture_mu;
std::atomic<uint64_t> g_pinned_bytes[64];
std::atomic<uint64_t> g_slot_calls[64], g_slot_rows[64];
std::atomic<uint64_t> g_phase_ns[kPhCount], g_phase_calls;

namespace {
// Contexts are owned by a process-lifetime pool (never destroyed: tearing HIP objects down from
// thread-exit / atexit handlers races the runtime's own shutdown).  A thread returns its contexts
// to the pool when it exits so short-lived threads do not grow it.
std::mutex g_pool_mu;
std::vector<std::vector<ThreadCtx *>> g_pool;  // [device slot] -> free contexts

std::atomic<int> g_slot_threads[64];  // caller threads whose home is slot i right now (a thread leaves when it exits)
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
}  // namespace

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

struct HostPool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<ThreadCtx *> free;
  int created = 0;
};
namespace {
HostPool &host_pool(int slot) {
  static HostPool pools[64];
  return pools[size_t(slot) % 64];
}
thread_local ThreadCtx *t_last_ctx[64];  // per device slot: the staging context this thread leased last (never dereferenced: an identity)
}  // namespace
HostLease::HostLease(int slot) : pool(host_pool(slot)) {
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
HostLease::~HostLease() {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(pool.mu);
    pool.free.push_back(c);
  }
  pool.cv.notify_one();
}

namespace {
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
std::atomic<bool> g_slot_unhealthy[64];
std::mutex g_fault_mu;
std::string g_slot_fault[64];
}  // namespace
int healthy_slots() {
  int n = 0;
  for (size_t i = 0; i < devices().ids.size() && i < 64; i++) n += !g_slot_unhealthy[i].load(std::memory_order_acquire);
  return n;
}
bool slot_is_unhealthy(int slot) { return g_slot_unhealthy[size_t(slot) % 64].load(std::memory_order_acquire); }
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


}  // namespace rt
using namespace rt;

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

bool slot_health(int slot, std::string *fault) {
  std::lock_guard<std::mutex> lk(g_fault_mu);
  if (fault) *fault = g_slot_fault[size_t(slot) % 64];
  return !g_slot_unhealthy[size_t(slot) % 64].load(std::memory_order_acquire);
}

}  // namespace infera_hip
