// host_path.cpp -- the host ABI's data path (infera_predict / infera_predict_columns / infera_predict_from_blob[_batch]): the caller's rows are
// borrowed for the call only (SURVEY.md 8b "Ownership"), so a call leases a staging context of its thread's home GPU, stages (or lets the
// GPU fetch) its rows, is admitted, enqueues copy + kernels, naps until the device is done and copies the result out.  Replaces the copy-in /
// run / copy-out of engine.rs:139-154.  Four ways through, chosen per call:
//   run_chunks     calls up to one staging pass (a DataChunk): gather -> gate -> {H2D | read in place} -> kernels -> wait -> copy out
//   zero-copy      the same with `dfill`: the GPU pulls the caller's REGISTERED column runs itself (no CPU gather, no pinned staging)
//   hipGraph       INFERA_HIPGRAPH=1: {H2D, kernels[, D2H]} replayed from a per-(model, rows) graph (off by default: DESIGN.md 4)
//   run_pipelined  calls above 24 MB (image batches): two staging slots, the CPU copy of pass i + 1 beside the GPU's pass i
// and around them run_host_redealt: a device fault takes the slot out of service and the call runs again on another one.
#include <cstring>
#include <optional>
#include <thread>

#include "profile.hpp"
#include "runtime.hpp"

namespace infera_hip {
namespace rt {
namespace {

// One copy stream per device slot for the big-row pipeline's H2D copies (run_pipelined: why), created on first use; `mu` serialises
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
    return ++in_flight_;
  }
  void release(int limit) {
    if (limit <= 0) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      in_flight_--;
    }
    cv_.notify_one();
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int in_flight_ = 0;
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

// ---------------------------------------------------------------------------------------------------------------------------------------
// One host-ABI call on the staging context it leased.
struct HostCall {
  const LoadedModel &m;
  const FillFn &fill;          // stages rows [r0, r0 + nr) into pinned memory (row-major, or column-major with col_major) ...
  const DeviceFillFn *dfill;   // ... or (zero-copy) makes the GPU write them as one column-major chunk into HBM
  float *const h_out;
  const int64_t rows;
  const bool col_major;
  const int slot;
  ThreadCtx &ctx;
  const DeviceModel &dm;
  const size_t in_row, out_row, widest;  // bytes per row
  const bool use_graph = Config::get().use_hipgraph;

  HostCall(const LoadedModel &model, const FillFn &f, const DeviceFillFn *df, float *out, int64_t n, bool cm, int sl, ThreadCtx &c)
      : m(model), fill(f), dfill(df), h_out(out), rows(n), col_major(cm), slot(sl), ctx(c), dm(device_model(model, sl)),
        in_row(size_t(model.plan.in_per_row()) * 4), out_row(size_t(model.plan.out_per_row()) * 4), widest(std::max(in_row, out_row)) {}

  size_t h2d_bytes(int64_t nr) const { return size_t(nr) * in_row; }
  // what kind of wait this is, for the nap estimates: model and row count (a context that served a 30 ms image batch must not sleep 2 ms on
  // the 50 us table chunk that follows it)
  uint64_t wait_key(int64_t nr, bool pipelined = false) const { return m.uid * 0x9E3779B97F4A7C15ull ^ uint64_t(nr) ^ (pipelined ? uint64_t(1) << 63 : 0); }
  // calls longer than this run as a two-slot pipeline of passes
  bool is_big() const { return !use_graph && size_t(rows) * widest > kPipePassBytes + kPipePassBytes / 2 && rows > 1; }

  // H2D of one pass on the context's stream; a column-major pass lands in dev_cm first and is transposed into the row-major table on the GPU
  void upload_pass(const float *pin, float *din, int64_t nr) {
    if (col_major) {
      ctx.ensure_dev(ctx.dev_cm, ctx.dev_cm_cap, h2d_bytes(nr));
      HIP_TRY(hipMemcpyAsync(ctx.dev_cm, pin, h2d_bytes(nr), hipMemcpyHostToDevice, ctx.stream));
      kern::transpose_cm(ctx.stream, ctx.dev_cm, din, nr, int64_t(in_row / 4));
    } else {
      HIP_TRY(hipMemcpyAsync(din, pin, h2d_bytes(nr), hipMemcpyHostToDevice, ctx.stream));
    }
  }

  int64_t pipeline_pass_rows() const;
  void run_pipelined();
  void enqueue_chunk(int64_t r0, int64_t nr, bool single_pass, bool direct_out, int in_flight);
  bool graph_chunk(int64_t r0, int64_t nr, bool direct_out);
  void run_chunks(uint64_t lease_ns);
};

// Pass size of the big-row pipeline: 16 MB keeps the CPU copy and the H2D of table rows overlapped best; rows as big as images (602 KB) get
// many more of them per pass, because a 27-image pass leaves the conv kernels half empty (ResNet-18, 16 threads x 256-image calls: 18.5k
// img/s with 16 MB passes, 29.7k -- the resident rate -- with 64 MB).  Round 3: up to 256 such rows per pass -- 96-image passes left C5 at
// 0.90 of its resident rate end to end, 256-image passes reach 0.96 (31.97k -> 34.0k img/s at 16 callers).  The pinned staging this costs --
// two passes of inputs and two of results per context -- is bounded by the context's share of kPinnedBudgetPerSlot: 221 images of 602 KB.
// Equal passes, at least two when the call is worth cutting: the CPU copy of pass 2 must overlap the GPU's pass 1 also when ONE caller
// brings one batch (256 images as 128 + 128, not as a single pass with nothing to overlap).
int64_t HostCall::pipeline_pass_rows() const {
  const int64_t by_bytes = std::max<int64_t>(1, int64_t(kPipePassBytes / widest));
  const int64_t share_rows = int64_t(kPinnedBudgetPerSlot / size_t(std::max(1, Config::get().host_contexts)) / (2 * (in_row + out_row)));
  const int64_t by_rows = std::min<int64_t>(std::min<int64_t>(256, std::max<int64_t>(16, share_rows)), std::max<int64_t>(1, int64_t(4 * kHostPassBytes / widest)));
  const int64_t p0 = std::min<int64_t>(rows, std::max(by_bytes, by_rows));
  const int64_t npass = std::max<int64_t>(rows >= 64 ? 2 : 1, (rows + p0 - 1) / p0);
  return (rows + npass - 1) / npass;
}

// Larger host inputs (a whole BLOB batch, a big infera_predict call): two staging slots.  The CPU copy of pass i + 1 into pinned memory -- the
// slowest stage, the caller's buffer is only borrowed -- overlaps the H2D / kernels / D2H of pass i instead of following them.
// Row-major passes are copied to the device on ONE copy stream per GPU slot, shared by all its contexts, and a pass's kernels wait for the
// copy's event.  (1) On the context's own stream the copy of pass i + 1 queues behind the kernels of pass i.  (2) Copies of several callers on
// several streams SHARE the link: sixteen 133 MB copies issued together all arrive after 38 ms, where one at a time the first arrives after
// 2.4 ms and its kernels start -- a kernel + copy trace of the C5 scan at 16 callers showed the matrix cores idle 8 % of the time, always with
// copies running (profiles/r04_e2e_timeline.txt).  One stream is first-come-first-served at full link speed.  (A column-major pass lands in
// the single dev_cm buffer first: it stays on the context's stream.)
void HostCall::run_pipelined() {
  const int64_t P = pipeline_pass_rows();
  ctx.ensure_pinned(ctx.pin_in, ctx.pin_in_cap, 2 * size_t(P) * in_row);
  ctx.ensure_pinned(ctx.pin_out, ctx.pin_out_cap, 2 * size_t(P) * out_row);
  ctx.ensure_dev(ctx.dev_in, ctx.dev_in_cap, 2 * size_t(P) * in_row);
  ctx.ensure_dev(ctx.dev_out, ctx.dev_out_cap, 2 * size_t(P) * out_row);
  for (auto &e : ctx.pipe_ev)
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipStream_t copy_stream = nullptr;
  std::mutex *copy_mu = nullptr;
  if (!col_major) {
    copy_stream = big_copy_stream(ctx.slot, copy_mu);
    for (auto &e : ctx.h2d_ev)
      if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  int64_t pend_r0[2] = {0, 0}, pend_nr[2] = {0, 0};
  auto slot_ptr = [&](float *base, int k, size_t row_bytes) { return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + size_t(k) * size_t(P) * row_bytes); };
  auto drain = [&](int k) {  // waits for the pass that last used staging slot k and copies its results out
    if (!pend_nr[k]) return;
    prof::Range range("infera:drain");
    const uint64_t t0 = now_ns();
    ctx.wait_event(ctx.pipe_ev[k], ctx.pipe_est, wait_key(pend_nr[k], true));  // (naps: a BLOB batch is milliseconds of GPU time per pass)
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
      {
        prof::Range range("infera:fill");
        fill(pin, r0, nr);
      }
      const uint64_t t_f1 = now_ns();
      prof::Range range("infera:enqueue");
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
}

// {H2D | nothing: small inputs are read from pinned memory by the kernel itself | the GPU's own fetch} + the plan's kernels [+ D2H] of ONE chunk
// whose staged input is in ctx.pin_in, on ctx.stream; results land in ctx.pin_out.  Direct (non-graph) enqueue.
void HostCall::enqueue_chunk(int64_t r0, int64_t nr, bool single_pass, bool direct_out, int in_flight) {
  const float *pin = ctx.pin_in;
  float *din = ctx.dev_in;
  // column-major chunk straight into the model's first kernel when it can read one (no transpose launch)
  const bool cm_direct = col_major && m.in_colmajor_ok && nr <= m.in_colmajor_max_rows && single_pass;
  // Small inputs (a narrow table's chunk, a point query; INFERA_HOST_DIRECT_IN bytes, default 128 KB) are not copied to HBM first:
  // the kernel that reads them -- the plan's first kernel, or the transpose in front of it -- loads them from the pinned
  // (host-coherent) buffer over PCIe itself.  One submission less per call, and a DMA engine's start-up latency is as long as
  // such a transfer: 13-column table +6..17 % at 2..64 threads.  (1 MB chunks read this way reach 34 GB/s against the copy
  // engines' 50: C2 and every larger chunk keep the H2D copy.)  A quiet GPU reads larger chunks this way -- twice that size with at
  // most four calls in flight, four times with at most two: a few kernels pulling over PCIe do not yet compete with each other, and the
  // copy engine's latency is the larger part of such a call (30 -> 100 -> 2, 245 KB chunks: +17 / +8 / +6 / +4 % at 1 / 2 / 4 / 8 threads).
  const int quiet_mult = in_flight <= 2 ? 4 : in_flight <= 4 ? 2 : 1;
  const bool small_in = int64_t(nr) * int64_t(in_row) <= int64_t(Config::get().host_direct_in_bytes) * quiet_mult;
  const float *kin = din;
  if (dfill) {
    // zero-copy: the GPU pulls the caller's (registered) column runs itself -- straight into the chunk the first kernel reads when it
    // reads column-major chunks, else into the transpose's source
    if (cm_direct) {
      prof::Section sec(8, "enqueue: fetch launch");
      (*dfill)(ctx.stream, din, r0, nr);
    } else {
      ctx.ensure_dev(ctx.dev_cm, ctx.dev_cm_cap, h2d_bytes(nr));
      (*dfill)(ctx.stream, ctx.dev_cm, r0, nr);
      kern::transpose_cm(ctx.stream, ctx.dev_cm, din, nr, int64_t(in_row / 4));
    }
  } else if (cm_direct) {
    if (small_in) kin = pin;
    else {
      prof::Section sec(8, "enqueue: H2D copy");
      HIP_TRY(hipMemcpyAsync(din, pin, h2d_bytes(nr), hipMemcpyHostToDevice, ctx.stream));
    }
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
    prof::Section sec(9, "enqueue: model launch");
    exec_plan(m, dm, ctx, kin, ctx.pin_out, nr, cm_direct);
  } else {
    exec_plan(m, dm, ctx, kin, ctx.dev_out, nr, cm_direct);
    HIP_TRY(hipMemcpyAsync(ctx.pin_out, ctx.dev_out, size_t(nr) * out_row, hipMemcpyDeviceToHost, ctx.stream));
  }
}

// hipGraph mode: replays the chunk's graph (false: the caller waits for it as for a direct enqueue).  First chunk of this (model, rows) on this
// context: runs it directly -- one-time work such as hipFuncSetAttribute / code-object loading must not happen inside a capture -- copies the
// result out, then records the graph (capturing does not execute anything) for the chunks that follow (true: this chunk is done).
bool HostCall::graph_chunk(int64_t r0, int64_t nr, bool direct_out) {
  const bool cm_graph = col_major;  // (implies m.in_colmajor_ok: checked by run_chunks)
  const int64_t key = nr | (cm_graph ? int64_t(1) << 62 : 0);
  for (auto &g : ctx.graphs)
    if (g.uid == m.uid && g.rows == key) {
      g.last_use = ++ctx.graph_clock;
      HIP_TRY(hipGraphLaunch(g.exec, ctx.stream));
      return false;
    }
  float *result = direct_out ? ctx.pin_out : ctx.dev_out;
  auto enqueue_all = [&]() -> hipError_t {
    hipError_t e = hipMemcpyAsync(ctx.dev_in, ctx.pin_in, h2d_bytes(nr), hipMemcpyHostToDevice, ctx.stream);
    if (e == hipSuccess) exec_plan(m, dm, ctx, ctx.dev_in, result, nr, cm_graph);
    if (e == hipSuccess && !direct_out) e = hipMemcpyAsync(ctx.pin_out, ctx.dev_out, size_t(nr) * out_row, hipMemcpyDeviceToHost, ctx.stream);
    return e;
  };
  HIP_TRY(enqueue_all());
  HIP_TRY(hipStreamSynchronize(ctx.stream));
  std::memcpy(h_out + size_t(r0) * (out_row / 4), ctx.pin_out, size_t(nr) * out_row);
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::shared_lock<std::shared_mutex> capture_lock(g_capture_mu);
  HIP_TRY(hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeThreadLocal));
  hipError_t e;
  try {
    e = enqueue_all();
  } catch (...) {
    (void)hipStreamEndCapture(ctx.stream, &graph);
    if (graph) (void)hipGraphDestroy(graph);
    throw;
  }
  const hipError_t e2 = hipStreamEndCapture(ctx.stream, &graph);
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
  ctx.graphs.push_back({m.uid, key, exec, ++ctx.graph_clock});
  return true;
}

// Calls of up to one staging pass (64 MB) per device pass: a DataChunk, a point query, a medium infera_predict call.
// Measured and dropped, twice:
//   * one chunk as two halves on the call's stream, the gather of the second under the H2D + kernel of the first (round 3, and again in round 5
//     with rows/s at 1 and 2 callers and CPU per chunk as the yardsticks): a loss at every caller count -- 1 caller 20.5 -> 17.8 M rows/s, 2 callers
//     37.8 -> 32.6, CPU per chunk 55-66 -> 75-82 us.  The wait behind the second enqueue is as long as a whole chunk's (46-52 us): it is latency
//     (copy-engine start-up, launch, completion), not the 19 us of transfer (profiles/r05_half_chunk_ab.txt, r03_host_cpu_ab_split_pollq.txt);
//   * STREAMED chunks (round 5): the fused-MLP tile kernel launched BEFORE the gather, consuming the columns out of pinned staging 16 at a time
//     behind per-group flags, no H2D copy at all.  Bit-identical; a lone caller gains 13-23 % (93 -> 80 us per chunk) -- and two callers LOSE
//     whenever their streams share one of the runtime's hardware queues (a kernel that waits for its host blocks the queue behind it: 35 -> 26 M
//     rows/s), four and more lose 15-20 % (profiles/r05_stream_chunk_ab.txt).  Even with the host spinning on the event the tail AFTER the last
//     column group was there took 29 us: what is left of a chunk's wait is the device's completion path, which no pipelining of the input hides;
//   * waiting by READING the results (round 5): pinned result buffer filled with a sentinel before the enqueue, the words copied out as they
//     change -- one nap until shortly before they are due, then a short spin -- instead of an event behind the kernel: no gain at 1-2 callers
//     (wait 46.6-48 vs 45-47 us: the results are there when the event says so, the 7 us "tail" of a rocprofv3 timeline is the profiler's own),
//     a loss at 4+ (profiles/r05_result_poll_ab.txt).  What a lone caller waits for is the device's pipeline: ~5 us until the copy starts,
//     24 us of copy (18.4 at link speed), 8 us between the copy engine's completion and the kernel's start, 12 us of tile kernel
//     (profiles/r05_e2e_ranges.txt);
//   * staging contexts dealt so that concurrent callers never share a hardware queue (round 5; the runtime maps streams onto four queues in
//     creation order 1 2 3 4 | 4 3 2 1 | ..., and two callers on one queue run strictly one behind the other): no gain, staged 3-4 callers lose
//     4-7 % -- on one queue two callers ALTERNATE (one's copy under the other's kernel), on two they fall into lockstep and share the link
//     (profiles/r05_ctx_queue_groups_ab.txt, r05_stream_queue_probe.txt); likewise ONE high-priority fetch stream per GPU for all zero-copy
//     pulls with the kernels waiting for its events: the cross-queue dependency costs a lone caller 6 us per chunk and four callers collapse
//     to 36 M rows/s (profiles/r05_fetch_stream_ab.txt);
//   * a staged chunk PULLED out of pinned staging by a kernel instead of copied by a copy engine while few calls are in flight (round 5): a lone
//     caller +4-6 %, two and more lose 5-15 % and pay 5-15 us more CPU per chunk (profiles/r05_staged_pull_ab.txt).
void HostCall::run_chunks(uint64_t lease_ns) {
  // hipGraph mode captures {H2D, kernels[, D2H]}: a column-major chunk only when the model's first kernel reads it itself (the transposing
  // path allocates per pass, which a capture cannot contain)
  if (col_major && use_graph && !m.in_colmajor_ok) throw InferaError::onnx("internal: column-major staging is not captured in hipGraph mode for this plan");
  const int64_t rows_pass = std::min(rows, std::max<int64_t>(1, int64_t(kHostPassBytes / widest)));
  const bool direct_out = m.out_write_once && size_t(rows_pass) * out_row <= (1u << 20);
  if (!dfill) ctx.ensure_pinned(ctx.pin_in, ctx.pin_in_cap, size_t(rows_pass) * in_row);
  ctx.ensure_pinned(ctx.pin_out, ctx.pin_out_cap, size_t(rows_pass) * out_row);
  ctx.ensure_dev(ctx.dev_in, ctx.dev_in_cap, size_t(rows_pass) * in_row);
  ctx.ensure_dev(ctx.dev_out, ctx.dev_out_cap, size_t(rows_pass) * out_row);
  const bool profiling = prof::enabled();
  for (int64_t r0 = 0; r0 < rows; r0 += rows_pass) {
    const int64_t nr = std::min(rows_pass, rows - r0);
    // The caller's buffer is only borrowed for the call (SURVEY.md 8b "Ownership"): stage it.
    const uint64_t t_f0 = now_ns();
    if (!dfill) {
      prof::Range range("infera:gather");
      fill(ctx.pin_in, r0, nr);
    }
    const uint64_t t_f1 = now_ns();
    if (profiling) prof::push("infera:gate");
    GateHold admitted(gate_for_slot(slot), Config::get().max_inflight, Config::get().max_inflight_total);  // until this pass has been synchronised
    if (profiling) prof::pop();
    const uint64_t t_g = now_ns();
    // one device pass for the whole chunk?  (plans with activation scratch split long calls; may reallocate -- and drop graphs --
    // so before the lookup).  A column-major chunk is only handed to the first kernel as it lies when it is.
    std::optional<prof::Section> sec_scratch(std::in_place, 7, "prepare scratch");
    const bool single_pass = prepare_scratch(m, ctx, nr) == nr;
    sec_scratch.reset();
    bool done = false;
    {
      prof::Range range("infera:enqueue");
      if (use_graph) {
        if (col_major && !single_pass)
          throw InferaError::onnx("internal: column-major chunk of " + std::to_string(nr) + " rows needs several device passes; not captured in hipGraph mode");
        done = graph_chunk(r0, nr, direct_out);
      } else {
        enqueue_chunk(r0, nr, single_pass, direct_out, admitted.in_flight);
      }
    }
    if (done) continue;  // (first chunk of a graph: its result is already in h_out)
    const uint64_t t_e = now_ns();
    {
      prof::Range range("infera:wait");
      ctx.wait_stream(wait_key(nr));  // (spinning on hipStreamQuery instead measured slower: 41 vs 69 M rows/s at 16 threads)
    }
    const uint64_t t_w = now_ns();
    {
      prof::Range range("infera:copy_out");
      prof::Section sec(14, "copy out");
      std::memcpy(h_out + size_t(r0) * (out_row / 4), ctx.pin_out, size_t(nr) * out_row);
    }
    const uint64_t t_c = now_ns();
    prof::Section sec_counters(15, "phase counters");
    if (r0 == 0) g_phase_ns[kPhLease].fetch_add(lease_ns, std::memory_order_relaxed);
    g_phase_ns[kPhGather].fetch_add(t_f1 - t_f0, std::memory_order_relaxed);
    g_phase_ns[kPhGate].fetch_add(t_g - t_f1, std::memory_order_relaxed);
    g_phase_ns[kPhEnqueue].fetch_add(t_e - t_g, std::memory_order_relaxed);
    g_phase_ns[kPhWait].fetch_add(t_w - t_e, std::memory_order_relaxed);
    g_phase_ns[kPhCopyOut].fetch_add(t_c - t_w, std::memory_order_relaxed);
    g_phase_calls.fetch_add(1, std::memory_order_relaxed);
  }
}

// One host-ABI call on the calling thread's home slot (false: a zero-copy call this path does not take -- the caller stages it instead).
bool run_host_impl(const LoadedModel &m, const FillFn &fill, const DeviceFillFn *dfill, float *h_out, int64_t rows, bool col_major) {
  if (m.dev.empty()) throw InferaError::onnx("HIP backend unavailable: " + m.device_error);
  if (rows <= 0) return true;
  if (dfill) {  // zero-copy: one host pass, direct (non-graph) enqueue only -- anything else goes the staging way
    const size_t widest_row = std::max(size_t(m.plan.in_per_row()), size_t(m.plan.out_per_row())) * 4;
    if (Config::get().use_hipgraph || size_t(rows) * widest_row > kPipePassBytes + kPipePassBytes / 2) return false;
  }
  const int slot = home_slot();
  prof::Range range("infera:chunk");
  const uint64_t t_entry = now_ns();
  std::optional<prof::Range> lease_range(std::in_place, "infera:lease");
  std::optional<prof::Section> sec_lease(std::in_place, 5, "lease (+ hipSetDevice)");
  HostLease lease(slot);
  sec_lease.reset();
  lease_range.reset();
  const uint64_t t_leased = now_ns();
  if (fault_injected(slot)) hip_fail(hipErrorLaunchFailure, "injected fault (INFERA_FAULT_INJECT)");
  std::optional<prof::Section> sec_setup(std::in_place, 6, "call setup");
  HostCall call(m, fill, dfill, h_out, rows, col_major, slot, *lease.c);
  sec_setup.reset();
  g_slot_calls[size_t(slot) % 64].fetch_add(1, std::memory_order_relaxed);
  g_slot_rows[size_t(slot) % 64].fetch_add(uint64_t(rows), std::memory_order_relaxed);
  try {
    if (call.is_big()) call.run_pipelined();
    else call.run_chunks(t_leased - t_entry);
  } catch (...) {
    // Whatever threw (a launch error in exec_plan, a non-NotReady error from wait_stream, ...) may have left a copy, a 2-D copy or the pulling
    // kernel of this chunk in flight: nothing may still read the caller's registered pages (the ZeroCopyPins taken in capi.cpp are released while
    // unwinding -- an application may then unmap them) or this context's staging (the next lessee overwrites it) when the lease goes back
    // (run_pipelined's own handler also drains the shared copy stream).
    (void)hipStreamSynchronize(lease.c->stream);
    throw;
  }
  if (range.on || prof::sections_enabled()) prof::note_call(t_entry, now_ns(), uint64_t(rows));
  return true;
}

// Is the device behind `slot` answering?  Asked before a call that failed with a device fault on one slot is run again on the next: on ROCm a
// real execution fault is usually sticky for the whole process, and then every slot would be taken out of service one after another, one
// retry each, before the error surfaces.  A device that cannot even synchronise is not a candidate.
bool slot_responds(int slot) {
  UnsafeOpGuard guard;
  (void)hipGetLastError();
  const bool ok = hipSetDevice(devices().ids[size_t(slot)]) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  (void)hipGetLastError();
  return ok;
}

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
      // the slot this thread would be dealt next must answer, or the fault is the process's, not the device's: surface it, leave the others
      // in service (they are marked when their own callers fail)
      if (!slot_responds(home_slot())) throw;
    }
  }
}

// Zero-copy fetches a GPU has in flight right now, for INFERA_ZERO_COPY_MAX_INFLIGHT (default 0 = no limit): with a limit, the chunks beyond it
// take the staged path (false = "stage it", exactly as for a chunk outside the registered ranges).  Round 4 shipped a limit of four because
// in-place fetches stopped at 80 M rows/s per GPU whatever mechanism issued them and only the copy engines' linear copies filled the rest of
// the link -- at 46-63 us of CPU per chunk at 8-16 callers.  Round 5 found what the 80 were: a 2-D copy takes 24-28 us and the runtime runs
// them one at a time; the pulling kernel's launch shape cost it 10 %.  With at most three 2-D copies in flight per GPU and the (re-shaped)
// pulling kernel for every other chunk (zero_copy.cpp, rect_copy_acquire) every chunk is fetched in place: 89.5 / 97.8 / 101.8 M rows/s at
// 4 / 6 / 8 callers at 22-24 us of CPU per chunk (profiles/r05_zero_copy_ab.txt).
std::atomic<int> g_zc_fetches[64];

}  // namespace
}  // namespace rt
using namespace rt;

// What the host link really delivers on this box: `threads` threads, each keeping TWO hipMemcpyAsync(bytes) in flight from its own pinned
// buffers on its own stream (copy i + 1 is enqueued before copy i is waited for: with one copy per thread the link idled between a completion
// and the next enqueue, and the "ceiling" read lower than the pipeline it is meant to bound -- VERDICT r4) -- what the host path's
// "fraction of PCIe" is shown beside.  NOT an upper bound of the pipeline: bare 1 MiB copies from 8-24 threads reach 35-48 GB/s on these
// boxes and 8 MiB copies 53-55, while the host path moves its 1 MiB chunks at 56 (a kernel behind every copy keeps the queues busier than a
// copy alone) -- the fraction that matters is the one against the link's raw 64 GB/s.
double h2d_probe_gbs(int device_ordinal, size_t bytes, int iters, int threads) {
  if (threads < 1) threads = 1;
  if (iters < 2) iters = 2;
  std::atomic<int> failed{0}, ready{0};
  std::atomic<bool> go{false};
  std::vector<std::chrono::steady_clock::time_point> done;
  done.resize(size_t(threads));
  // (buffers, streams and events are set up BEFORE the clock starts and torn down after every thread's own end stamp: round 4's probe timed them too)
  auto worker = [&](int t) {
    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);  // (this thread's own: the 2 us sleeps below must not become 52)
    hipStream_t s = nullptr;
    char *pin[2] = {nullptr, nullptr}, *dev[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ok = hipSetDevice(device_ordinal) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; ok && k < 2; k++) {
      ok = hipHostMalloc(reinterpret_cast<void **>(&pin[k]), bytes, hipHostMallocDefault) == hipSuccess &&
           hipMalloc(reinterpret_cast<void **>(&dev[k]), bytes) == hipSuccess &&
           hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) == hipSuccess;
      if (ok) std::memset(pin[k], 1, bytes);
    }
    auto enqueue = [&](int k) { return hipMemcpyAsync(dev[k], pin[k], bytes, hipMemcpyHostToDevice, s) == hipSuccess && hipEventRecord(ev[k], s) == hipSuccess; };
    // waits like the host path does -- short sleeps between queries: hipEventSynchronize spins on ROCm 7.2, and sixteen spinning threads on a
    // 16-CPU quota are throttled by the cgroup, which is the link's "ceiling" no more
    auto wait = [&](int k) {
      for (;;) {
        const hipError_t e = hipEventQuery(ev[k]);
        if (e == hipSuccess) return true;
        if (e != hipErrorNotReady) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(2));
      }
    };
    if (ok) ok = enqueue(0) && wait(0);  // (one untimed copy: first-use work of the stream)
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
    if (ok) ok = enqueue(0);
    for (int i = 1; ok && i < iters; i++) ok = enqueue(i & 1) && wait((i - 1) & 1);
    if (ok) ok = wait((iters - 1) & 1);
    done[size_t(t)] = std::chrono::steady_clock::now();
    if (!ok) failed = 1;
    if (s) (void)hipStreamSynchronize(s);
    for (int k = 0; k < 2; k++) {
      if (ev[k]) (void)hipEventDestroy(ev[k]);
      if (dev[k]) (void)hipFree(dev[k]);
      if (pin[k]) (void)hipHostFree(pin[k]);
    }
    if (s) (void)hipStreamDestroy(s);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(worker, t);
  while (ready.load() < threads) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto &x : th) x.join();
  const double sec = std::chrono::duration<double>(*std::max_element(done.begin(), done.end()) - t0).count();
  (void)hipGetLastError();
  return failed ? -1.0 : double(bytes) * iters * threads / sec / 1e9;
}

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

void run_host_fill(const LoadedModel &m, const FillFn &fill, float *h_out, int64_t rows, bool col_major) {
  (void)run_host_redealt(m, fill, nullptr, h_out, rows, col_major);
}

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

void run_host(const LoadedModel &m, const float *h_in, float *h_out, int64_t rows) {
  const size_t in_per_row = size_t(m.plan.in_per_row());
  run_host_fill(m, [&](float *dst, int64_t r0, int64_t nr) { std::memcpy(dst, h_in + size_t(r0) * in_per_row, size_t(nr) * in_per_row * 4); },
                h_out, rows);
}

}  // namespace infera_hip
