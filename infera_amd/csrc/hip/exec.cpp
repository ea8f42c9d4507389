// exec.cpp -- the plan executor: one model, one context, rows already on the device.  Cuts the rows into device passes (activation scratch),
// long convolutional passes into two lanes, and launches what schedule.cpp decided, step by step.  Replaces `SimplePlan::run`
// (engine.rs:142-145).  Also the device-resident entry points (infera_hip_predict_device).
#include <cstdlib>

#include "runtime.hpp"

namespace infera_hip {
namespace rt {

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
// 17.94 ms (-2.6 %); three or four lanes: no gain (profiles/r04_conv_lanes_ab.txt).  With the lanes a 1024-image pass is within 0.2 % of the per-image
// time of a 1003-image one (whole rounds everywhere), and splitting the last round finer only costs (round 5, profiles/r05_tail_split_ab.txt).  Not under a stream capture (INFERA_HIPGRAPH=1), not
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

// Rows r0 .. r0 + nr - 1 of one device pass through every step of the plan, on one stream, with one set of scratch slots.
struct PassRunner {
  const LoadedModel &m;
  const DeviceModel &dm;
  ThreadCtx &ctx;
  const Plan &p;
  const std::vector<Step> &st;
  const float *d_in;
  float *d_out;
  const bool in_colmajor;
  hipStream_t stream = nullptr;
  int64_t r0 = 0, nr = 0;
  std::vector<int64_t> slot_base;  // floats into the scratch, per slot

  PassRunner(const LoadedModel &model, const DeviceModel &dmodel, ThreadCtx &c, const float *in, float *out, bool cm)
      : m(model), dm(dmodel), ctx(c), p(model.plan), st(model.plan.steps), d_in(in), d_out(out), in_colmajor(cm), slot_base(model.slot_per_row.size(), 0) {}

  float *buf(int b) const {
    if (b == 0) return const_cast<float *>(d_in) + r0 * p.in_per_row();
    if (b == p.out_buf) return d_out + r0 * p.out_per_row();
    return ctx.scratch + slot_base[size_t(m.slot_of_buf[size_t(b)])];
  }
  bool cq(int b) const { return m.cq_mode && !m.nchw_buf[size_t(b)]; }  // a channel-quad tensor (not the caller's NCHW one)
  // scratch slots sized for slot_rows rows, from scratch_off floats into the scratch
  void run(hipStream_t s, int64_t first_row, int64_t rows, int64_t slot_rows, int64_t scratch_off) {
    stream = s;
    r0 = first_row;
    nr = rows;
    int64_t off = scratch_off;
    for (size_t i = 0; i < m.slot_per_row.size(); i++) {
      slot_base[i] = off;
      off += m.slot_per_row[i] * slot_rows;
    }
    for (size_t i = 0; i < st.size(); i++) {
      size_t skip = 0;
      if (!launch_fused(i, &skip)) launch_plain(i);
      i += skip;
    }
    HIP_TRY(hipGetLastError());
  }
  bool launch_fused(size_t i, size_t *skip);  // steps schedule.cpp gave a fused / specialised kernel (true: handled)
  void launch_conv_tiled(size_t i);
  void launch_conv_patch(size_t i);
  void launch_plain(size_t i);                // one kernel per step, by step kind
};

void PassRunner::launch_conv_tiled(size_t i) {
  const Step &x = st[i];
  const DeviceStep &d = dm.steps[i];
  const int fj = m.conv_fused_add[i];
  const kern::ConvGeom gp = kern::conv2d_tiled_geom(conv_geom(x));
  const Step &last = fj >= 0 ? st[size_t(fj)] : x;  // whose activation and output the launch carries (a fused residual Add's)
  if (m.conv_split6[i] && m.conv_fold[i] >= 0) {
    const Step &q = st[size_t(m.conv_fold[i])];
    const kern::SecondInput x2{buf(q.in0), int(q.C), int(q.H), int(q.Wd), int(q.sh), int(q.sw)};
    kern::conv2d_split6(stream, buf(x.in0), d.W, d.bias, nullptr, buf(last.out), nr, gp, act_of(last), x2);
    return;
  }
  const float *residual = fj >= 0 ? buf(m.conv_residual_buf[i]) : nullptr;
  if (m.conv_split6[i]) kern::conv2d_split6(stream, buf(x.in0), d.W, d.bias, residual, buf(last.out), nr, gp, act_of(last));
  else kern::conv2d_tiled(stream, buf(x.in0), d.W, d.bias, residual, buf(last.out), nr, gp, act_of(last));
}

void PassRunner::launch_conv_patch(size_t i) {
  const Step &x = st[i];
  const DeviceStep &d = dm.steps[i];
  const kern::ConvGeom gp = kern::conv2d_patch_geom(conv_geom(x));
  const int fj = m.conv_fused_pool[i];
  if (fj < 0) {
    kern::conv2d_patch(stream, buf(x.in0), d.W, d.bias, buf(x.out), nr, gp, act_of(x), dm.num_cus);
    return;
  }
  const Step &q = st[size_t(fj)];
  const char *sse = getenv("INFERA_STEM_SPLIT");  // 0: the exact-fp32 stem kernels under a split plan (read per launch: tests, A/B)
  if (m.stem_split6[i] && d.cst && !(sse && atoi(sse) == 0))
    kern::conv2d_stem_split6(stream, buf(x.in0), d.cst, d.bias, buf(q.out), nr, gp, act_of(x), pool_tail(q), dm.num_cus);
  else
    kern::conv2d_patch_pool(stream, buf(x.in0), d.W, d.bias, buf(q.out), nr, gp, act_of(x), pool_tail(q), dm.num_cus);
}

bool PassRunner::launch_fused(size_t i, size_t *skip) {
  const Step &x = st[i];
  const DeviceStep &d = dm.steps[i];
  const bool cm = in_colmajor && x.in0 == 0;  // this step reads the caller's column-major chunk
  switch (m.exec[i]) {
    case ExecKind::Skipped: return true;
    case ExecKind::Mlp3Head: {
      std::string why;
      if (!kern::mlp3(stream, m.mlp3_shape, buf(x.in0), dm.mlp3_packed, buf(st[i + 2].out), nr, dm.num_cus, &why, cm))
        throw InferaError::onnx("fused MLP kernel launch failed: " + why);
      return true;
    }
    case ExecKind::ChainHead: {
      const LoadedModel::ChainRun &run = *m.chain_at(i);
      std::string why;
      if (!kern::chain(stream, run.shape, buf(x.in0), dm.chain_packed[size_t(&run - m.chains.data())], buf(st[i + size_t(run.nsteps) - 1].out), nr,
                       dm.num_cus, &why, cm))
        throw InferaError::onnx("fused chain kernel launch failed: " + why);
      return true;
    }
    case ExecKind::DenseArgMax:
      if (cm || kern::dense_can_fuse_argmax(buf(x.in0), int(x.K), int(x.M))) {  // (both column-major kernels have the epilogue)
        kern::dense(stream, buf(x.in0), d.W, d.bias, buf(st[i + 1].out), nr, int(x.K), int(x.M), act_of(x), 3, cm);
        *skip = 1;  // the ArgMax step is done
        return true;
      }
      return false;  // as two kernels
    case ExecKind::DenseSoftmax:
      kern::dense(stream, buf(x.in0), d.W, d.bias, buf(st[i + 1].out), nr, int(x.K), int(x.M), act_of(x), st[i + 1].log_softmax ? 2 : 1, cm);
      return true;
    case ExecKind::ConvTiled: launch_conv_tiled(i); return true;
    case ExecKind::DenseTiled: kern::conv2d_tiled(stream, buf(x.in0), d.W, d.bias, nullptr, buf(x.out), nr, dense_as_conv(x), act_of(x)); return true;
    case ExecKind::ConvDepthwise: kern::conv2d_depthwise(stream, buf(x.in0), d.W, d.bias, buf(x.out), nr, conv_geom(x), act_of(x)); return true;
    case ExecKind::ConvPatch: launch_conv_patch(i); return true;
    default: return false;
  }
}

void PassRunner::launch_plain(size_t i) {
  const Step &x = st[i];
  const DeviceStep &d = dm.steps[i];
  switch (x.kind) {
    case StepKind::Dense: kern::dense(stream, buf(x.in0), d.W, d.bias, buf(x.out), nr, int(x.K), int(x.M), act_of(x), 0, in_colmajor && x.in0 == 0); break;
    case StepKind::Unary: kern::unary(stream, buf(x.in0), buf(x.out), nr * p.buf_per_row[size_t(x.out)], act_of(x)); break;
    case StepKind::AffineChannel:
      kern::affine_channel(stream, buf(x.in0), d.scale, d.shift, buf(x.out), nr, x.C, x.S, act_of(x), cq(x.in0));
      break;
    case StepKind::BinaryConst:
      kern::binary_const(stream, buf(x.in0), d.cst, buf(x.out), nr, p.buf_per_row[size_t(x.out)], x.bop, x.const_left, act_of(x));
      break;
    case StepKind::BinaryAct:
      if (x.S > 1) kern::binary_gate(stream, buf(x.in0), buf(x.in1), buf(x.out), nr, x.C, x.S, x.bop, act_of(x), cq(x.in0));
      else kern::binary_act(stream, buf(x.in0), buf(x.in1), buf(x.out), nr * p.buf_per_row[size_t(x.out)], x.bop, act_of(x));
      break;
    case StepKind::Softmax: kern::softmax(stream, buf(x.in0), buf(x.out), nr, x.sm_outer, x.sm_len, x.sm_inner, x.sm_norm ? 1 + x.sm_norm : int(x.log_softmax)); break;
    case StepKind::Conv2d: {
      kern::conv2d(stream, buf(x.in0), d.W, d.bias, buf(x.out), nr, conv_geom(x), act_of(x), cq(x.in0),
                   cq(x.out));
      break;
    }
    case StepKind::Pool2d:
      kern::pool2d(stream, buf(x.in0), buf(x.out), nr, int(x.C), int(x.H), int(x.Wd), int(x.OH), int(x.OW), int(x.kh), int(x.kw),
                   int(x.sh), int(x.sw), int(x.pt), int(x.pl), int(x.dh), int(x.dw), x.is_max, x.count_pad,
                   cq(x.in0));
      break;
    case StepKind::GlobalAvgPool:
      kern::global_avgpool(stream, buf(x.in0), buf(x.out), nr, int(x.C), int(x.S), cq(x.in0), x.is_max);
      break;
    case StepKind::CopyCols:
      kern::copy_cols(stream, buf(x.in0), buf(x.out), nr, p.buf_per_row[size_t(x.in0)], p.buf_per_row[size_t(x.in0)], 0,
                      p.buf_per_row[size_t(x.out)], x.col_off);
      break;
    case StepKind::PadCols: kern::pad_cols(stream, buf(x.in0), buf(x.out), nr, x.K, x.M); break;
    case StepKind::LRN:
      kern::lrn(stream, buf(x.in0), buf(x.out), nr, int(x.C), int(x.S), int(x.lrn_size), x.lrn_alpha, x.lrn_beta, x.lrn_bias,
                cq(x.in0));
      break;
    case StepKind::ChannelShuffle:
      kern::channel_shuffle(stream, buf(x.in0), buf(x.out), nr, int(x.C), int(x.S), int(x.groups), cq(x.in0));
      break;
    case StepKind::SliceCols:
      kern::copy_cols(stream, buf(x.in0), buf(x.out), nr, x.K, p.buf_per_row[size_t(x.in0)], x.col_off, x.K, 0);
      break;
    case StepKind::ArgMax: kern::argmax_rows(stream, buf(x.in0), buf(x.out), nr, x.K); break;
  }
}

// in_colmajor: d_in is one column-major chunk [in_per_row][rows] (only with m.in_colmajor_ok, which implies a single pass)
void exec_plan(const LoadedModel &m, const DeviceModel &dm, ThreadCtx &ctx, const float *d_in, float *d_out, int64_t rows, bool in_colmajor) {
  const Plan &p = m.plan;
  if (rows <= 0) return;
  if (p.out_buf == 0) {  // pure alias / Identity graph
    HIP_TRY(hipMemcpyAsync(d_out, d_in, size_t(rows) * size_t(p.in_per_row()) * 4, hipMemcpyDeviceToDevice, ctx.stream));
    return;
  }
  const int64_t rows_pass = prepare_scratch(m, ctx, rows);
  // a column-major chunk [K][rows] cannot be cut into row passes (pass r0 would start at a row-major offset with stride nr):
  // callers check single_pass() first and stage such calls row-major instead; this guards every kernel family at once
  if (in_colmajor && rows_pass != rows) throw InferaError::onnx("internal: column-major input needs a single pass");
  PassRunner runner(m, dm, ctx, d_in, d_out, in_colmajor);
  for (int64_t r0 = 0; r0 < rows; r0 += rows_pass) {
    const int64_t nr = std::min(rows_pass, rows - r0);
    const int nl = in_colmajor ? 1 : lanes_of(m, nr);
    if (nl == 1) {
      runner.run(ctx.stream, r0, nr, rows_pass, 0);
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
        runner.run(ls, r0 + l0, ln, nlane, int64_t(m.scratch_per_row) * nlane * l);
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


}  // namespace rt
using namespace rt;

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

}  // namespace infera_hip
