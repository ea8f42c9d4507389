// zero_copy.cpp -- registered host memory: the GPU reads an application's column storage in place (infera_hip_register_host_memory,
// the reference's ROADMAP.md:44 "zero-copy").  Ranges -> page blocks -> reader pins; see the comment below and DESIGN.md 6.2.
#include <exception>
#include <map>
#include <thread>

#include "runtime.hpp"

namespace infera_hip {
using namespace rt;

// ---- registered host memory (zero-copy host path) -----------------------------------------------------------------------
// The runtime pins whole pages, callers register byte ranges (numpy arrays, malloc'ed buffers, a database allocator's blocks: neighbours on
// the heap share pages).  So a registered RANGE (what lookups test against) is covered by one or more page BLOCKS (what hipHostRegister
// was called on): registering a range pins only the pages no earlier block covers -- a block is never replaced or re-registered once it
// exists (round 3 merged neighbours into one new registration, which unmapped memory under running calls and forced every (un)registration
// to drain ALL zero-copy calls in flight).  A block lives until the last range on its pages is unregistered; then it leaves the index at
// once and is unmapped as soon as the calls that PINNED it (found it in a lookup and have not finished) are done -- only those calls are
// waited for, every other call and every registration proceeds.  Lookups are O(log n) under a shared lock; the writers hold their own
// mutex (g_reg_mu) and the index lock only for the planning and the map updates -- NOT across hipHostRegister / hipHostUnregister or the wait
// for a dead block's readers (g_busy below), so an allocator hook calling in from many threads never queues behind a pin or a chunk.
namespace {
struct PageBlock {
  uintptr_t pb, pe;              // page-aligned span handed to hipHostRegister
  intptr_t dev_delta;            // device-visible address = host address + dev_delta (the same on every selected GPU: checked)
  int refs = 0;                  // registered ranges that touch these pages (under g_reg_mu)
  std::atomic<int> readers{0};   // zero-copy calls in flight that resolved an address inside this block
};
struct HostRange {
  uintptr_t end;
  bool usable;  // all covering blocks share one device delta (always, on the systems seen so far); false while `pending`
  intptr_t dev_delta;
  bool pending = false;  // its pages are being pinned right now: lookups skip it, unregister does not see it
};
std::mutex g_reg_mu;                                            // serialises register / unregister
std::shared_mutex g_index_mu;                                   // g_ranges / g_blocks (readers: lookups)
std::map<uintptr_t, HostRange> g_ranges;                        // by base; non-overlapping
std::map<uintptr_t, std::shared_ptr<PageBlock>> g_blocks;       // by pb; non-overlapping
std::atomic<size_t> g_nranges{0};
// Bumped (with g_index_mu held exclusively) whenever a range or a block LEAVES the index: what a caller thread's cache of resolved runs is
// valid against (additions never invalidate a hit).
std::atomic<uint64_t> g_index_gen{0};

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

// Page spans being pinned or unmapped right now (under g_reg_mu).  hipHostRegister / hipHostUnregister take up to milliseconds and an unmap
// first waits for the calls reading the block: none of that happens with g_reg_mu held (ADVICE r4 -- with the registering DuckDB allocator
// every Allocate / Free of a large block from any thread would otherwise queue behind one in-flight chunk or pin).  A writer whose pages
// overlap a busy span waits for THAT span only; everybody else proceeds.
namespace {
std::condition_variable g_reg_cv;
std::vector<std::pair<uintptr_t, uintptr_t>> g_busy;
bool busy_overlaps(uintptr_t pb, uintptr_t pe) {
  for (const auto &s : g_busy)
    if (s.first < pe && pb < s.second) return true;
  return false;
}
void busy_remove(uintptr_t pb, uintptr_t pe) {
  auto it = std::find(g_busy.begin(), g_busy.end(), std::make_pair(pb, pe));
  if (it != g_busy.end()) g_busy.erase(it);
}
// Unmaps blocks that have left the index, once the calls that pinned them are done (a chunk's time, ~100 us).  Bounded: a call stuck on a
// wedged GPU must not block the application's allocator for ever -- after kUnmapWaitSeconds the pages stay pinned (leaked until the process
// exits; registering them again fails and their chunks are staged) and the error is logged.  Called WITHOUT g_reg_mu; the spans are in g_busy.
constexpr int kUnmapWaitSeconds = 5;
// false: at least one block is still being read after the wait -- its pages stay pinned AND MAY STILL BE READ, the caller must be told
// (the header's contract is "free once unregister returns 0").
bool unmap_dead_blocks(const std::vector<std::shared_ptr<PageBlock>> &dead) {
  if (dead.empty()) return true;
  bool all_drained = true;
  for (const auto &blk : dead) {
    // the reader wait holds NOTHING (in hipGraph mode UnsafeOpGuard is the capture lock, exclusively: five seconds of it per stuck block would
    // stall every capture and allocation of the process); only the unmap itself is an operation a capture must not see
    const auto t0 = std::chrono::steady_clock::now();
    bool drained = true;
    for (int spin = 0; blk->readers.load(std::memory_order_acquire) > 0; spin++) {
      if (spin < 64) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(20));
      if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kUnmapWaitSeconds)) {
        drained = false;
        break;
      }
    }
    if (drained) {
      UnsafeOpGuard guard;
      (void)hipHostUnregister(reinterpret_cast<void *>(blk->pb));
    } else {
      all_drained = false;
      // the stuck readers hold RAW pointers to the block object (ZeroCopyPins) and will decrement its reader count if they ever return: the
      // object must outlive them -> parked for the life of the process (a handful of bytes per wedged block)
      static std::mutex graveyard_mu;
      static std::vector<std::shared_ptr<PageBlock>> *graveyard = new std::vector<std::shared_ptr<PageBlock>>();
      {
        std::lock_guard<std::mutex> lk(graveyard_mu);
        graveyard->push_back(blk);
      }
      log_msg(0, "unregister_host_memory: calls still read a " + std::to_string((blk->pe - blk->pb) >> 10) + " KiB block after " +
                     std::to_string(kUnmapWaitSeconds) + " s (GPU wedged?): its pages stay pinned");
    }
  }
  std::lock_guard<std::mutex> writer(g_reg_mu);
  for (const auto &blk : dead) busy_remove(blk->pb, blk->pe);
  g_reg_cv.notify_all();
  return all_drained;
}
// (under g_reg_mu + g_index_mu) one reference less on every block under [pb, pe); blocks nobody refers to any more leave the index -> `dead`
void release_blocks(uintptr_t pb, uintptr_t pe, std::vector<std::shared_ptr<PageBlock>> &dead) {
  auto it = g_blocks.upper_bound(pb);
  if (it != g_blocks.begin() && std::prev(it)->second->pe > pb) --it;
  while (it != g_blocks.end() && it->second->pb < pe) {
    if (--it->second->refs == 0) {  // (a block shared with neighbours stays pinned until the last of them goes)
      dead.push_back(it->second);
      g_busy.emplace_back(it->second->pb, it->second->pe);
      it = g_blocks.erase(it);  // no later lookup can find it
      g_index_gen.fetch_add(1, std::memory_order_release);
    } else {
      ++it;
    }
  }
}
}  // namespace

void register_host_memory(const void *base, size_t bytes) {
  if (!base || !bytes) throw InferaError::null_pointer();
  const auto &ds = devices();
  if (ds.ids.empty()) throw InferaError::onnx("HIP backend unavailable: " + ds.why);
  const uintptr_t b = reinterpret_cast<uintptr_t>(base), e = b + bytes;
  const uintptr_t pb = b & ~uintptr_t(4095), pe = (e + 4095) & ~uintptr_t(4095);
  // ---- plan (under g_reg_mu): the range enters the index as PENDING (lookups skip it), the existing blocks under it are referenced, the
  // ---- uncovered page gaps become busy spans
  std::vector<std::shared_ptr<PageBlock>> fresh, covering;
  std::unique_lock<std::mutex> writer(g_reg_mu);
  g_reg_cv.wait(writer, [&] { return !busy_overlaps(pb, pe); });
  {  // (the index is mutated only with g_reg_mu held: reading it here without g_index_mu is safe)
    auto it = g_ranges.upper_bound(b);
    if (it != g_ranges.end() && it->first < e) throw InferaError::onnx("host memory range overlaps a registered range");
    if (it != g_ranges.begin() && std::prev(it)->second.end > b) throw InferaError::onnx("host memory range overlaps a registered range");
  }
  {
    std::unique_lock<std::shared_mutex> lk(g_index_mu);
    uintptr_t at = pb;
    auto it = g_blocks.upper_bound(pb);
    if (it != g_blocks.begin() && std::prev(it)->second->pe > pb) --it;
    for (; at < pe; ++it) {
      const uintptr_t gap_end = it == g_blocks.end() || it->second->pb >= pe ? pe : it->second->pb;
      if (gap_end > at) {
        auto blk = std::make_shared<PageBlock>();
        blk->pb = at;
        blk->pe = gap_end;
        fresh.push_back(blk);
        covering.push_back(blk);
        g_busy.emplace_back(at, gap_end);
      }
      if (it == g_blocks.end() || it->second->pb >= pe) break;
      it->second->refs++;
      covering.push_back(it->second);
      at = it->second->pe;
    }
    g_ranges.emplace(b, HostRange{e, /*usable=*/false, 0, /*pending=*/true});
    g_nranges.store(g_ranges.size(), std::memory_order_release);
  }
  writer.unlock();
  // ---- pin the gaps: no lock held
  size_t pinned = 0;
  std::exception_ptr failure;
  {
    UnsafeOpGuard guard;
    try {
      for (; pinned < fresh.size(); pinned++) fresh[pinned]->dev_delta = hip_register_span(fresh[pinned]->pb, fresh[pinned]->pe);
    } catch (...) {
      failure = std::current_exception();
      for (size_t i = 0; i < pinned; i++) (void)hipHostUnregister(reinterpret_cast<void *>(fresh[i]->pb));
    }
  }
  // ---- publish (or take the plan back)
  std::vector<std::shared_ptr<PageBlock>> dead;
  bool usable = true;
  writer.lock();
  {
    std::unique_lock<std::shared_mutex> lk(g_index_mu);
    for (auto &blk : fresh) busy_remove(blk->pb, blk->pe);
    if (failure) {
      g_ranges.erase(b);
      g_index_gen.fetch_add(1, std::memory_order_release);
      g_nranges.store(g_ranges.size(), std::memory_order_release);
      release_blocks(pb, pe, dead);  // (the fresh blocks never entered the index: this drops the references taken on the existing ones)
    } else {
      for (auto &blk : fresh) {
        blk->refs = 1;
        g_blocks.emplace(blk->pb, blk);
      }
      HostRange &r = g_ranges.find(b)->second;
      r.dev_delta = covering.front()->dev_delta;
      for (auto &blk : covering) usable = usable && blk->dev_delta == r.dev_delta;
      r.usable = usable;
      r.pending = false;
    }
  }
  writer.unlock();
  g_reg_cv.notify_all();
  (void)unmap_dead_blocks(dead);  // (blocks of a FAILED registration: nobody was handed their addresses)
  if (failure) std::rethrow_exception(failure);
  if (!usable) log_msg(1, "registered host range is covered by blocks with different device addresses: its chunks take the staged path");
}

bool unregister_host_memory(const void *base) {
  const uintptr_t b = reinterpret_cast<uintptr_t>(base);
  std::vector<std::shared_ptr<PageBlock>> dead;
  {
    std::lock_guard<std::mutex> writer(g_reg_mu);
    std::unique_lock<std::shared_mutex> lk(g_index_mu);
    auto rit = g_ranges.find(b);
    if (rit == g_ranges.end() || rit->second.pending) return false;  // (a range still being registered by another thread is not there yet)
    const uintptr_t pb = b & ~uintptr_t(4095), pe = (rit->second.end + 4095) & ~uintptr_t(4095);
    g_ranges.erase(rit);
    g_index_gen.fetch_add(1, std::memory_order_release);
    g_nranges.store(g_ranges.size(), std::memory_order_release);
    release_blocks(pb, pe, dead);
  }
  // the dead blocks are out of the index; they are unmapped once the calls that pinned them have finished -- with no lock held: other
  // registrations and unregistrations proceed unless they touch these very pages
  if (!unmap_dead_blocks(dead))
    throw InferaError::onnx("host memory range still in use: zero-copy calls were still reading it after " + std::to_string(kUnmapWaitSeconds) +
                            " s; the range is no longer served but its pages stay pinned -- keep the memory mapped");
  return true;
}

// Resolves n host runs to device-visible addresses; every run must lie inside ONE registered range.  The blocks under the runs are PINNED
// (reader count) until the returned guard dies -- hold it until the GPU has finished reading.  O(log n) per run, one shared lock.
ZeroCopyPins::~ZeroCopyPins() {
  // (the last access to each block: the unregistering thread that waits for readers == 0 holds the block object until then)
  for (size_t i = 0; i < count; i++) static_cast<PageBlock *>(blocks[i])->readers.fetch_sub(1, std::memory_order_release);
}

namespace {
// A caller thread's last resolution of run i of its chunks.  A table scan walks a column segment vector by vector: the next chunk's run i lies
// in the SAME registered range and block as this one's 31 times out of 32 (DuckDB: 32 vectors per 256 KiB block) -- with 20,000 blocks of a
// 10M x 128 table registered one by one, the two tree walks per run cost 13 us per chunk for a lone caller and 25 us at two or more (cache
// misses on tree nodes every caller walks; round 6, tools/r06_segments_ranges.sh).  Valid while g_index_gen has not moved (nothing left the
// index since) -- checked under the shared index lock, so nothing can leave while the hit is being pinned either.
struct RunHit {
  uintptr_t rb = 1, re = 0;  // the registered range
  intptr_t delta = 0;
  PageBlock *blk = nullptr;
  std::shared_ptr<PageBlock> keep;  // (keeps the block OBJECT alive for the raw pointer above; not its registration)
};
struct RunCache {
  uint64_t gen = ~uint64_t(0);
  RunHit hit[kern::kMaxZeroCopyCols];
};
thread_local RunCache t_runs;
}  // namespace

bool lookup_host_memory_many(size_t n, const void *const *ptrs, const size_t *bytes, const void **out, ZeroCopyPins &pins) {
  if (g_nranges.load(std::memory_order_acquire) == 0) return false;
  std::shared_lock<std::shared_mutex> lk(g_index_mu);
  RunCache &rc = t_runs;
  const uint64_t gen = g_index_gen.load(std::memory_order_acquire);
  if (rc.gen != gen) {
    for (auto &h : rc.hit) h = RunHit();
    rc.gen = gen;
  }
  uintptr_t hb = 0, he = 0;  // the range the previous run lay in (columns of one table usually share it)
  intptr_t hd = 0;
  PageBlock *last = nullptr;  // the block pinned last (consecutive runs usually share it too)
  for (size_t i = 0; i < n; i++) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptrs[i]), ae = a + bytes[i];
    RunHit *hit = i < size_t(kern::kMaxZeroCopyCols) ? &rc.hit[i] : nullptr;
    if (hit && hit->blk && a >= hit->rb && ae <= hit->re && a >= hit->blk->pb && ae <= hit->blk->pe) {  // where run i was last time
      out[i] = reinterpret_cast<const void *>(intptr_t(a) + hit->delta);
      if (hit->blk != last) {
        if (pins.count == ZeroCopyPins::kMax) return false;
        hit->blk->readers.fetch_add(1, std::memory_order_acquire);
        pins.blocks[pins.count++] = hit->blk;
        last = hit->blk;
      }
      continue;
    }
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
      pins.blocks[pins.count++] = bit->second.get();
      last = bit->second.get();
      if (hit && a >= last->pb && ae <= last->pe) {  // the whole run in one block: remember it for this thread's next chunk
        hit->rb = hb, hit->re = he, hit->delta = hd, hit->blk = last;
        hit->keep = bit->second;
      }
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

// 2-D copies a GPU has in flight (between a call's enqueue and its return).  The runtime executes them ONE AT A TIME however many streams
// issue them (tools/ubench/chunk_pull_probe.hip: 28.6 us per MiB from 1 to 16 streams), which is what makes two or three callers alternate
// nicely -- and what stops a GPU at 80 M rows/s.  The pulling kernel runs beside it on the shader cores.  So a chunk takes the 2-D copy while
// fewer than INFERA_ZERO_COPY_RECT_INFLIGHT (default 3) are in flight on its GPU and the pulling kernel otherwise.  Counted per PHYSICAL GPU, like
// SubmitGate (host_path.cpp gate_for_slot): two device slots on one GPU (INFERA_DEVICES=0,0) share the runtime's one-at-a-time 2-D copy path.
namespace {
std::atomic<int> g_rect_inflight[64];
}
int rect_copy_acquire() {
  const int limit = Config::get().zero_copy_rect_inflight;
  const int gpu = int(size_t(devices().ids[size_t(rt::home_slot())]) % 64);
  std::atomic<int> &n = g_rect_inflight[gpu];
  if (limit > 0 && n.fetch_add(1, std::memory_order_relaxed) >= limit) {
    n.fetch_sub(1, std::memory_order_relaxed);
    return -1;
  }
  return limit > 0 ? gpu : 64;  // (the ticket is the GPU's index; 64: no limit, nothing to give back)
}
void rect_copy_release(int ticket) {
  if (ticket >= 0 && ticket < 64) g_rect_inflight[size_t(ticket)].fetch_sub(1, std::memory_order_relaxed);
}

}  // namespace infera_hip
