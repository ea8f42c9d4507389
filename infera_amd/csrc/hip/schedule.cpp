// schedule.cpp -- plan -> kernels (device independent, done once per infera_load_model): which steps fuse into which kernel, the
// activation layout of convolutional plans, the arithmetic of the convolutions, where the served result and the input may live, and
// the scratch slots of what is left.  This is the half of engine.rs:49-55 (`into_optimized().into_runnable()`) that decides; model.cpp
// packs and uploads what is decided here, exec.cpp runs it.
#include <sstream>

#include "runtime.hpp"

namespace infera_hip {
namespace rt {

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

// One infera_load_model's scheduling state: the passes below run in this order, each reading what the earlier ones decided.
struct Scheduler {
  LoadedModel &m;
  const std::vector<Step> &st;
  const size_t n;
  std::vector<int> uses;  // readers per buffer (the served output counts as read)
  const Config &cfg = Config::get();

  explicit Scheduler(LoadedModel &model) : m(model), st(model.plan.steps), n(model.plan.steps.size()), uses(model.plan.buf_per_row.size(), 0) {
    m.exec.assign(n, ExecKind::Normal);
    for (const auto &s : st) {
      if (s.in0 >= 0) uses[size_t(s.in0)]++;
      if (s.in1 >= 0) uses[size_t(s.in1)]++;
    }
    uses[size_t(m.plan.out_buf)]++;
  }
  bool is4d(int b) const { return b >= 0 && m.plan.buf_shape[size_t(b)].size() == 4; }
  int64_t spatial(int b) const { return is4d(b) ? m.plan.buf_shape[size_t(b)][2] * m.plan.buf_shape[size_t(b)][3] : int64_t(1); }

  void fuse_tabular();               // Dense x3 -> fused MLP, runs of small layers -> chain kernel, Dense + Softmax / ArgMax epilogues, tiled Dense
  void decide_layout();              // channel-quad planes for convolutional plans (+ the weight / constant permutations that follow from it)
  void assign_conv_kernels();        // patch / tiled / depthwise kernels, the stem's fused MaxPool
  void fuse_residual_adds();         // a block's Add (+ activation) into the epilogue of its last producer
  void choose_conv_arithmetic();     // bf16 x three exact parts (default) or exact fp32
  void fold_projection_shortcuts();  // a block's 1x1 shortcut as extra K stages of its second convolution
  void classify_io(const std::vector<EffStep> &eff);   // may the result be stored into pinned memory, the input be read from it / column-major?
  void place_scratch(const std::vector<EffStep> &eff); // scratch slots by liveness
};

void Scheduler::fuse_tabular() {
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
}

void Scheduler::decide_layout() {
  // ---- layout decision for convolutional plans ----
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
}

void Scheduler::assign_conv_kernels() {
  m.conv_fused_pool.assign(n, -1);
  if (m.cq_mode)
    for (size_t i = 0; i < n; i++) {
      const Step &s = st[i];
      if (m.exec[i] != ExecKind::Normal || s.kind != StepKind::Conv2d) continue;
      kern::ConvGeom g = conv_geom(s);
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
}

void Scheduler::fuse_residual_adds() {
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

}

void Scheduler::choose_conv_arithmetic() {
  // the default arithmetic of the tiled convolutions and the 7x7 / stride-2 stem: bf16 x three exact parts (conv_split.hip)
  m.conv_split6.assign(n, 0);
  m.stem_split6.assign(n, 0);
  if (ScheduleKnobs::read().conv_bf16x6 && m.cq_mode) {
    for (size_t i = 0; i < n; i++) {
      if (m.exec[i] != ExecKind::ConvTiled) continue;
      const Step &c = st[i];
      const kern::ConvGeom g = conv_geom(c);
      if (kern::conv2d_split6_supported(kern::conv2d_tiled_geom(g)) && !m.nchw_buf[size_t(c.in0)]) m.conv_split6[i] = 1;
    }
    for (size_t i = 0; i < n; i++) {  // the 7x7 / stride-2 stem + max-pool in the same arithmetic
      if (m.exec[i] != ExecKind::ConvPatch || m.conv_fused_pool[i] < 0) continue;
      const Step &c = st[i], &q = st[size_t(m.conv_fused_pool[i])];
      const kern::ConvGeom g = conv_geom(c);
      const kern::ConvGeom gp = kern::conv2d_patch_geom(g);
      if (gp.mvalid == 0 && kern::conv2d_stem_split6_supported(gp, pool_tail(q))) m.stem_split6[i] = 1;
    }
  }
}

void Scheduler::fold_projection_shortcuts() {
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
      const kern::ConvGeom gc = conv_geom(c);
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

}

void Scheduler::classify_io(const std::vector<EffStep> &eff) {
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
}

// a slot is reused once its buffer has been read for the last time
void Scheduler::place_scratch(const std::vector<EffStep> &eff) {
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

void schedule(LoadedModel &m) {
  Scheduler s(m);
  s.fuse_tabular();
  s.decide_layout();
  s.assign_conv_kernels();
  s.fuse_residual_adds();
  s.choose_conv_arithmetic();
  s.fold_projection_shortcuts();
  const auto eff = effective_steps(m);
  s.classify_io(eff);
  s.place_scratch(eff);
}

}  // namespace rt
using namespace rt;

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
