// backend.hpp -- MI355X execution backend: device discovery, per-model HBM residency, per-thread
// streams / pinned staging, and the plan executor.
//
// This replaces the half of engine.rs that talks to Tract (engine.rs:49-55 load, :139-154 run).
// Threading model mirrors the reference's contract (SURVEY.md section 8b "Threading"): any number
// of caller threads run inferences concurrently; each thread owns a stream, pinned staging and
// scratch on "its" GPU (threads are dealt round-robin over the selected devices, which is how a
// DuckDB scan spreads its DataChunks over the 8 GPUs of a node with no collective).
#pragma once

#include <hip/hip_runtime_api.h>

#include <functional>
#include <memory>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../host/plan.hpp"
#include "kernels.hpp"

namespace infera_hip {

// Device discovery (cached).  count==0 -> `why` says what hipGetDeviceCount reported.
struct DeviceSet {
  std::vector<int> ids;  // HIP ordinals selected by INFERA_DEVICES (default: all)
  std::vector<int> cus;  // multiprocessor count per selected device
  std::vector<std::string> arch;
  std::vector<int> numa;  // NUMA node of the device's PCIe function (/sys/bus/pci/devices/<bdf>/numa_node), -1 = unknown
  std::string why;
};
// Which device slot a new caller thread gets ("NUMA-pin threads / pinned buffers to the socket that hosts the
// GPU"): round-robin over the slots on the thread's own NUMA node when there are any -- its gathers then write pinned
// staging that is local to both the thread and the GPU's root complex -- else round-robin over all slots.  Pure function
// of its arguments (per-node tickets are kept by the caller); thread_node < 0 or all-unknown topology = plain round-robin.
int choose_slot(const std::vector<int> &slot_numa, int thread_node, uint64_t ticket_on_node, uint64_t ticket_global);
// the policy home_slot() applies: NUMA-local first, bounded by load (slot_threads[i] = caller threads homed on slot i)
int choose_slot_balanced(const std::vector<int> &slot_numa, const std::vector<int> &slot_threads, int thread_node);
const DeviceSet &devices();
// false once a HIP error took the slot out of service (its callers were re-dealt to the other slots); *fault = the error text
bool slot_health(int slot, std::string *fault);
// host-ABI calls / table rows served so far by device slot `slot` (index into DeviceSet::ids)
void slot_counters(int slot, uint64_t *calls, uint64_t *rows);
uint64_t slot_pinned_bytes(int slot);  // pinned staging the slot's contexts hold right now
// {"passes":n,"lease_ns":..,"gather_ns":..,"gate_ns":..,"enqueue_ns":..,"wait_ns":..,"copy_out_ns":..}: where the wall
// time of the host-ABI calls went so far (single-pass path), summed over all caller threads
std::string host_phase_json();
// GB/s of plain pinned hipMemcpyAsync H2D on this box (threads x iters transfers of `bytes`); < 0 on failure
double h2d_probe_gbs(int device_ordinal, size_t bytes, int iters, int threads);

// Constants of one plan step resident in one GPU's HBM.
struct DeviceStep {
  float *W = nullptr, *bias = nullptr, *cst = nullptr, *scale = nullptr, *shift = nullptr;
};

// How the executor runs a step.
enum class ExecKind : int { Normal = 0, Skipped = 1, Mlp3Head = 2, DenseSoftmax = 3, ConvTiled = 4, ConvPatch = 5, ConvDepthwise = 6, DenseTiled = 7, DenseArgMax = 8, ChainHead = 9 };

struct DeviceModel {
  int device = -1;  // HIP ordinal
  int num_cus = 0;
  std::vector<DeviceStep> steps;
  float *mlp3_packed = nullptr;
  std::vector<float *> chain_packed;  // parameter block per LoadedModel::chains entry
  ~DeviceModel();
};

class LoadedModel {
 public:
  uint64_t uid = 0;  // unique per build_model call (keys the per-thread hipGraph cache)
  std::string name;
  Plan plan;
  // execution schedule (device independent)
  std::vector<ExecKind> exec;
  kern::Mlp3Shape mlp3_shape{};
  // Runs of small Dense layers (optionally behind a PadCols, optionally ending in Softmax / ArgMax) executed by one
  // load-time specialised kernel: steps [first, first + nsteps) of the plan; `pad` = 1 when the first one is a PadCols.
  struct ChainRun {
    int first = 0, nsteps = 0, pad = 0;
    kern::ChainShape shape;
  };
  std::vector<ChainRun> chains;
  const ChainRun *chain_at(size_t step) const {
    for (const auto &c : chains)
      if (size_t(c.first) == step) return &c;
    return nullptr;
  }
  // Convolutional plans keep every 4-D activation except the caller's input CHANNELS-LAST (NHWC) so
  // the implicit-GEMM gathers and stores are 16-byte vectors; decided per plan in schedule().
  bool cq_mode = false;
  // the served output is stored exactly once, by the plan's last kernel, and never read back: that kernel may write
  // straight into host-visible pinned memory (host path, small results)
  bool out_write_once = false;
  // the plan's first kernel is the only reader of the input table and has a variant that reads a column-major chunk
  // [cols][rows] directly (host path: no transpose kernel between the H2D copy and the model)
  bool in_colmajor_ok = false;
  int64_t in_colmajor_max_rows = 0;  // longest column-major chunk the first kernel reads itself (load-time specialised MLP chains: their tile kernel's range)
  bool in_single_reader = false;  // the input buffer is read by exactly one kernel of the plan (small host inputs: straight from pinned memory)
  // ... except the caller's input and what elementwise preprocessing makes of it (x/255, (x - mean) / std in the graph):
  // those few-channel tensors stay NCHW and the first convolution reads them with the patch kernel.
  std::vector<char> nchw_buf;
  // A ConvTiled step that absorbed the residual Add (+ activation) following it: per conv step, the
  // index of the fused BinaryAct step (-1: none) and which of its operands is the skip tensor.
  std::vector<int> conv_fused_pool;  // per step: the MaxPool 3x3/2 step a ConvPatch stem computes in its own kernel, or -1
  std::vector<int> conv_fused_add;
  std::vector<int> conv_residual_buf;
  std::vector<int> conv_fold;  // per ConvTiled step: the 1x1 projection-shortcut step computed inside it as extra K stages (conv_split.hip SecondInput), or -1
  std::vector<char> conv_split6;  // ConvTiled steps on conv2d_split6 (default; INFERA_PRECISION=fp32 leaves them on the exact-fp32 kernels)
  std::vector<char> stem_split6;  // ConvPatch + fused MaxPool steps that run conv2d_stem_split6 (same arithmetic)
  std::vector<int> slot_of_buf;        // scratch slot per activation buffer (-1: external in/out)
  std::vector<int64_t> slot_per_row;   // floats per row of each scratch slot
  int64_t scratch_per_row = 0;         // sum over slots
  // per selected device residency; empty when no GPU is visible (then `device_error` says why and
  // every predict fails loudly -- there is no CPU execution path in this library).
  std::vector<std::unique_ptr<DeviceModel>> dev;
  std::string device_error;

  std::string describe_json() const;
};

// Parse + lower + upload.  Throws InferaError.
std::shared_ptr<LoadedModel> build_model(const std::string &name, const std::string &path, const std::string &output_select = "");

// Host-memory inference (infera_predict / infera_predict_from_blob): `h_in` is rows x in_per_row
// f32 in pageable host memory; result written to `h_out` (rows x out_per_row).  Blocks until done.
void run_host(const LoadedModel &m, const float *h_in, float *h_out, int64_t rows);
// Same, but the input rows are PRODUCED straight into the pinned staging buffer by `fill(dst, row0,
// nrows)` (columnar gather, blob concatenation): no intermediate host copy.
using FillFn = std::function<void(float *dst, int64_t row0, int64_t nrows)>;
// col_major: `fill` writes the pass as [in_per_row][nrows] (column-major -- flat columns are copied as they are) and
// the transpose to the row-major table happens on the GPU.
void run_host_fill(const LoadedModel &m, const FillFn &fill, float *h_out, int64_t rows, bool col_major = false);
// Zero-copy host path (round 3): `dfill(stream, dst, row0, nrows)` makes the GPU itself write rows [row0, row0 + nrows) as one
// column-major chunk [in_per_row][nrows] into `dst` (HBM) on `stream` -- the caller's columns live in REGISTERED host memory and are
// read in place over PCIe; no CPU copy, no pinned staging, no hipMemcpyAsync.  Calls longer than one host pass fall back (return
// false: the caller then stages through the CPU as usual).
using DeviceFillFn = std::function<void(hipStream_t stream, float *dst, int64_t row0, int64_t nrows)>;
bool run_host_device_fill(const LoadedModel &m, const DeviceFillFn &dfill, float *h_out, int64_t rows);
// Registered host memory: [base, base + bytes) is pinned and mapped into every selected GPU (hipHostRegister on whole pages; pages a
// neighbouring range already pinned are shared, never re-registered).  Thread-safe; registering never waits for calls in flight, and
// unregistering waits only for the calls that are reading the pages it unmaps.
void register_host_memory(const void *base, size_t bytes);
bool unregister_host_memory(const void *base);
// Pins (reader counts) on the page blocks a zero-copy call reads: hold until the GPU has finished with them.
struct ZeroCopyPins {
  static constexpr size_t kMax = 520;  // (kMaxZeroCopyCols runs, a few of them straddling a block border)
  void *blocks[kMax];  // PageBlock * (raw: a pinned block's object outlives its last reader, zero_copy.cpp; nothing to construct per chunk)
  size_t count = 0;
  ZeroCopyPins() = default;
  ZeroCopyPins(const ZeroCopyPins &) = delete;
  ZeroCopyPins &operator=(const ZeroCopyPins &) = delete;
  ~ZeroCopyPins();
};
// device-visible addresses of n host runs, each inside ONE registered range (false otherwise), their blocks pinned in `pins`; O(log n) per run
bool lookup_host_memory_many(size_t n, const void *const *ptrs, const size_t *bytes, const void **out, ZeroCopyPins &pins);
const void *lookup_host_memory(const void *p, size_t bytes);  // (address only, no pin: tests / diagnostics)
size_t registered_host_ranges();
// `height` runs of `width` bytes, `src_pitch` bytes apart in REGISTERED host memory -> one column-major chunk at dst, as ONE 2-D copy on `stream`
void copy_rect_to_device(hipStream_t stream, float *dst, const void *src, size_t src_pitch, size_t width, size_t height);
bool zero_copy_rect_enabled();  // Config::zero_copy_rect
int rect_copy_acquire();             // a 2-D copy ticket of the calling thread's GPU (-1: enough of them in flight, use the pulling kernel)
void rect_copy_release(int ticket);
// whether a host call of `rows` rows can be handed to the plan's first kernel as column-major chunks (one device pass per host pass)
bool colmajor_direct_ok(const LoadedModel &m, int64_t rows);

// Device-resident inference: d_in / d_out live on HIP device `device_ordinal`.  Enqueues on the
// calling thread's stream for that device and returns without synchronising.
void run_device(const LoadedModel &m, int device_ordinal, const float *d_in, float *d_out, int64_t rows);
// Blocks until the calling thread's stream on that device is idle.
void sync_device(int device_ordinal);
hipStream_t thread_stream(int device_ordinal);

}  // namespace infera_hip
