// lowering.cpp -- ONNX graph -> Plan (shape inference with a symbolic row axis + peephole fusion).
//
// Fusions performed while walking the (topologically sorted) node list:
//   MatMul + Add(const [M])            -> Dense with bias
//   per-feature (x-mean)/std, x*s+t ... + MatMul/Gemm -> folded into the Dense weights and bias
//   Gemm(transB, alpha, beta)          -> Dense (constants folded into W / bias)
//   Dense|Conv|Binary|Affine + act     -> trailing activation fused into the producing step
//   Conv + BatchNormalization          -> BN folded into conv weights/bias
//   Identity/Dropout/Flatten/Reshape/Squeeze/Unsqueeze/Cast(f32) -> buffer alias (no kernel)
//   Shape/Gather/Concat/Unsqueeze/Squeeze/Slice/Cast on constants -> folded (the usual exporter pattern
//     Shape -> Gather -> Unsqueeze -> Concat -> Reshape; the symbolic row count is carried as 0 = "copy")
//   ReduceMean over the spatial axes -> GlobalAvgPool;  Concat(axis=1) -> one CopyCols step per input
// Dense chains are kept as consecutive Dense steps; the device executor decides whether a chain
// runs as one whole-chain fused kernel or layer by layer.
#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <set>
#include <numeric>
#include <sstream>

#include "common.hpp"
#include "plan.hpp"

namespace infera_hip {

namespace {

using onnx::NodeDef;
using onnx::TensorData;

struct Val {
  bool is_const = false;
  std::shared_ptr<TensorData> c;
  int buf = -1;
  std::vector<int64_t> shape;  // activations: dim0 = -1 (symbolic rows) or fixed batch
  // zero padding (top, left, bottom, right) a Pad node asked for and the consuming Conv still has to apply; `shape`
  // is the UNPADDED tensor that `buf` holds
  int64_t pend[4] = {0, 0, 0, 0};
  bool padded() const { return pend[0] || pend[1] || pend[2] || pend[3]; }
};

int64_t prod(const std::vector<int64_t> &v, size_t from = 0, size_t to = SIZE_MAX) {
  // Shapes come from a model file: a product that leaves int64 is an error, not a wrapped number.
  int64_t p = 1;
  for (size_t i = from; i < std::min(to, v.size()); i++)
    if (__builtin_mul_overflow(p, v[i], &p)) throw InferaError::onnx("tensor shape too large");
  return p;
}

// Largest activation row the planner accepts (elements): 2^31 floats = 8 GiB per table row is beyond anything the
// staging or scratch sizing could serve, and keeping per-row counts in 31 bits keeps rows*per_row inside int64.
constexpr int64_t kMaxPerRow = int64_t(1) << 31;

std::string shape_str(const std::vector<int64_t> &s) {
  std::string o = "[";
  for (size_t i = 0; i < s.size(); i++) o += (i ? "," : "") + std::to_string(s[i]);
  return o + "]";
}

[[noreturn]] void unsupported(const NodeDef &n, const std::string &why) {
  throw InferaError::onnx("node '" + (n.name.empty() ? n.op : n.name) + "' (" + n.op + "): " + why);
}

struct Lowerer {
  const onnx::Model &m;
  Plan plan;
  std::map<std::string, Val> vals;
  std::map<std::string, int> uses;
  std::map<int, int> producer;  // buffer id -> index of the step that wrote it
  std::map<int, std::vector<std::string>> buf_names;  // value names that denote each buffer
  std::map<int, int> alias_edges;  // nodes folded away whose input AND output denote the buffer
  std::set<int> int_bufs;  // buffers whose f32 values are whole numbers by construction (ArgMax, integer Cast, label arithmetic)

  Lowerer(const onnx::Model &model, const std::string &output_select_in) : m(model) {
    // Which graph output is served.  Default: the first (engine.rs:146-149).  A selector names another one, by output
    // name or by decimal index (infera_load_model(name, "model.onnx#probabilities"), SURVEY.md 8f-3 "named / multi outputs").
    if (m.outputs.empty()) throw InferaError::onnx("model has no outputs");
    // (a leading '?' makes the selector optional -- a URL fragment that names no output leaves the default, capi.cpp)
    const bool optional = !output_select_in.empty() && output_select_in[0] == '?';
    const std::string output_select = optional ? output_select_in.substr(1) : output_select_in;
    if (!output_select.empty()) {
      bool found = false;
      for (size_t i = 0; i < m.outputs.size() && !found; i++)
        if (m.outputs[i].name == output_select) out_index = i, found = true;
      if (!found && output_select.find_first_not_of("0123456789") == std::string::npos && output_select.size() < 6 &&
          size_t(std::atoi(output_select.c_str())) < m.outputs.size())
        out_index = size_t(std::atoi(output_select.c_str())), found = true;
      if (!found && !optional) {
        std::string names;
        for (const auto &o : m.outputs) names += (names.empty() ? "" : ", ") + o.name;
        throw InferaError::onnx("model has no output '" + output_select + "' (outputs: " + names + ")");
      }
    }
  }
  size_t out_index = 0;

  int new_buf(const std::vector<int64_t> &shape) {
    // [N,C,L] tensors (1-D convolutional nets) are laid out and scheduled as [N,C,1,L]; values keep their 3-D shape
    plan.buf_shape.push_back(shape.size() == 3 ? std::vector<int64_t>{shape[0], shape[1], 1, shape[2]} : shape);
    const int64_t per_row = prod(shape, 1);
    if (per_row < 0 || per_row > kMaxPerRow) throw InferaError::onnx("activation of " + std::to_string(per_row) + " elements per row is too large");
    plan.buf_per_row.push_back(per_row);
    return int(plan.buf_shape.size()) - 1;
  }

  const Val &get(const NodeDef &n, size_t i) {
    if (i >= n.inputs.size() || n.inputs[i].empty()) unsupported(n, "missing input " + std::to_string(i));
    auto it = vals.find(n.inputs[i]);
    if (it != vals.end()) return it->second;
    auto ci = m.initializers.find(n.inputs[i]);
    if (ci != m.initializers.end()) {
      Val v;
      v.is_const = true;
      v.c = ci->second;
      v.shape = ci->second->dims;
      return vals[n.inputs[i]] = v;
    }
    unsupported(n, "input '" + n.inputs[i] + "' is not produced by any earlier node");
  }
  bool has_input(const NodeDef &n, size_t i) const { return i < n.inputs.size() && !n.inputs[i].empty(); }

  const std::vector<float> &cf32(const NodeDef &n, const Val &v) {
    if (!v.is_const || v.c->dtype != onnx::kFloat) unsupported(n, "expected a constant f32 tensor");
    return v.c->f32;
  }

  // Binds the node's first output to `buf`.  `folded` = the node itself emitted no step (alias or
  // fused into its producer), i.e. it is an edge inside the buffer rather than a consumer of it.
  void set_act(const NodeDef &n, int buf, const std::vector<int64_t> &shape, bool folded = false) {
    Val v;
    v.buf = buf;
    v.shape = shape;
    vals[n.outputs[0]] = v;
    buf_names[buf].push_back(n.outputs[0]);
    if (folded) alias_edges[buf]++;
  }

  // Number of not-yet-folded consumers (incl. graph outputs) of a buffer across all its names.
  int live_uses(int buf) {
    int total = 0;
    for (const auto &nm : buf_names[buf]) total += uses[nm];
    return total - alias_edges[buf];
  }

  // The step that produced `v` if it can still absorb a trailing op: sole consumer is this node.
  Step *fusable_producer(const NodeDef &n, size_t in_idx) {
    const Val &v = get(n, in_idx);
    if (v.is_const || v.buf <= 0) return nullptr;
    if (live_uses(v.buf) != 1) return nullptr;
    auto it = producer.find(v.buf);
    if (it == producer.end()) return nullptr;
    return &plan.steps[size_t(it->second)];
  }

  Step &emit(Step s, const NodeDef &n, const std::vector<int64_t> &out_shape) {
    s.out = new_buf(out_shape);
    s.origin = n.op + (n.name.empty() ? "" : ":" + n.name);
    plan.steps.push_back(std::move(s));
    producer[plan.steps.back().out] = int(plan.steps.size()) - 1;
    set_act(n, plan.steps.back().out, out_shape);
    return plan.steps.back();
  }

  // ------------------------------------------------------------------------------------------
  void dense(const NodeDef &n, bool gemm) {
    const Val &a = get(n, 0);
    const Val &b = get(n, 1);
    if (a.is_const) unsupported(n, "constant left operand is not supported");
    if (!b.is_const) unsupported(n, "right operand must be a constant weight matrix");
    if (a.shape.size() != 2 || b.shape.size() != 2) unsupported(n, "only [rows,K] x [K,M] is supported, got " + shape_str(a.shape) + " x " + shape_str(b.shape));
    bool tA = gemm && n.attr_i("transA", 0) != 0, tB = gemm && n.attr_i("transB", 0) != 0;
    if (tA) unsupported(n, "transA=1 mixes table rows and is not supported");
    float alpha = gemm ? n.attr_f("alpha", 1.f) : 1.f, beta = gemm ? n.attr_f("beta", 1.f) : 1.f;
    const auto &w = cf32(n, b);
    int64_t K = tB ? b.shape[1] : b.shape[0], M = tB ? b.shape[0] : b.shape[1];
    if (a.shape[1] != K) unsupported(n, "inner dimensions differ: " + shape_str(a.shape) + " x " + shape_str(b.shape));
    int in_buf = a.buf;
    const std::vector<int64_t> a_shape = a.shape;
    // Per-feature affine preprocessing in front of the layer -- (x - mean) / std, x * scale + shift: the sklearn
    // StandardScaler / MinMaxScaler + linear model pipeline -- is folded into the weights:
    //   ((x (op) c) . W + b)  ==  x . (s W) + (b + t . W)   with x' = s x + t composed over the chain,
    // so the elementwise passes over the table disappear (exact algebra; rounding differs within the tolerance).
    // Only when the preprocessing value has no other consumer and its steps are the last ones emitted.
    std::vector<double> fs(size_t(K), 1.0), ft(size_t(K), 0.0);
    bool folded = false;
    std::string folded_origin;
    for (;;) {
      auto pit = producer.find(in_buf);
      if (in_buf <= 0 || pit == producer.end() || pit->second != int(plan.steps.size()) - 1 || live_uses(in_buf) != 1) break;
      const Step &p = plan.steps.back();
      if (p.kind == StepKind::AffineChannel && p.act == Act::None && p.S == 1 && p.C == K) {  // BatchNormalization of the features
        for (int64_t k = 0; k < K; k++) {
          ft[size_t(k)] += fs[size_t(k)] * double(p.shift[size_t(k)]);
          fs[size_t(k)] *= double(p.scale[size_t(k)]);
        }
        folded = true;
        folded_origin = p.origin + (folded_origin.empty() ? "" : "+" + folded_origin);
        in_buf = p.in0;
        producer.erase(pit);
        plan.steps.pop_back();
        continue;
      }
      if (p.kind != StepKind::BinaryConst || p.act != Act::None || int64_t(p.cst.size()) != K) break;
      if (p.bop != '+' && p.bop != '-' && p.bop != '*' && !(p.bop == '/' && !p.const_left)) break;
      for (int64_t k = 0; k < K; k++) {
        const double c = p.cst[size_t(k)], sk = fs[size_t(k)], tk = ft[size_t(k)];
        switch (p.bop) {
          case '+': ft[size_t(k)] = sk * c + tk; break;
          case '-':
            if (p.const_left) { fs[size_t(k)] = -sk; ft[size_t(k)] = sk * c + tk; }  // c - u
            else ft[size_t(k)] = tk - sk * c;
            break;
          case '*': fs[size_t(k)] = sk * c; break;
          default: fs[size_t(k)] = sk / c; break;
        }
      }
      folded = true;
      folded_origin = p.origin + (folded_origin.empty() ? "" : "+" + folded_origin);
      in_buf = p.in0;
      producer.erase(pit);
      plan.steps.pop_back();
    }
    // A wide layer over rows whose length is not a multiple of 4 floats (30 features, ...): copy the rows into a
    // zero-padded matrix first so the MFMA kernels can read them in 16-byte quads; the padded k carry zero weights.
    // (Narrow heads stream their input once and take any K.)
    // (up to 64 outputs over more than 128 columns the wide-table kernel reads the rows as they are)
    const int64_t Kp = (M > 32 && K % 4 != 0 && !(K > 128 && M <= 64)) ? (K + 3) / 4 * 4 : K;
    if (Kp != K) {
      Step p;
      p.kind = StepKind::PadCols;
      p.in0 = in_buf;
      p.K = K;
      p.M = Kp;
      p.out = new_buf({a_shape[0], Kp});
      p.origin = n.op + (n.name.empty() ? "" : ":" + n.name) + "[pad]";
      plan.steps.push_back(std::move(p));
      producer[plan.steps.back().out] = int(plan.steps.size()) - 1;
      in_buf = plan.steps.back().out;
    }
    Step s;
    s.kind = StepKind::Dense;
    s.in0 = in_buf;
    s.K = Kp;
    s.M = M;
    s.W.assign(size_t(Kp * M), 0.f);
    for (int64_t k = 0; k < K; k++)
      for (int64_t j = 0; j < M; j++) {
        float v = tB ? w[size_t(j * K + k)] : w[size_t(k * M + j)];
        s.W[size_t(k * M + j)] = alpha == 1.f ? v : alpha * v;
      }
    if (gemm && has_input(n, 2)) {
      const Val &c = get(n, 2);
      const auto &cv = cf32(n, c);
      if (int64_t(cv.size()) != M && cv.size() != 1) unsupported(n, "bias C must have M or 1 elements (row-independent)");
      s.bias.resize(size_t(M));
      for (int64_t j = 0; j < M; j++) {
        float v = cv.size() == 1 ? cv[0] : cv[size_t(j)];
        s.bias[size_t(j)] = beta == 1.f ? v : beta * v;
      }
    }
    if (folded) {
      std::vector<double> extra(size_t(M), 0.0);
      for (int64_t k = 0; k < K; k++)
        for (int64_t j = 0; j < M; j++) {
          const double wkj = s.W[size_t(k * M + j)];
          extra[size_t(j)] += ft[size_t(k)] * wkj;
          s.W[size_t(k * M + j)] = float(fs[size_t(k)] * wkj);
        }
      if (s.bias.empty()) s.bias.assign(size_t(M), 0.f);
      for (int64_t j = 0; j < M; j++) s.bias[size_t(j)] = float(double(s.bias[size_t(j)]) + extra[size_t(j)]);
    }
    Step &e = emit(std::move(s), n, {a_shape[0], M});
    if (folded) e.origin = folded_origin + "+" + e.origin;
  }

  // Broadcast a constant against an activation's per-row shape; returns per_row floats.
  std::vector<float> broadcast_const(const NodeDef &n, const Val &c, const std::vector<int64_t> &act_shape) {
    const auto &cv = cf32(n, c);
    std::vector<int64_t> cs = c.shape;
    if (cs.size() > act_shape.size()) unsupported(n, "constant operand has higher rank than the activation");
    while (cs.size() < act_shape.size()) cs.insert(cs.begin(), 1);
    if (cs[0] != 1) unsupported(n, "constant operand varies along the row axis");
    for (size_t i = 1; i < cs.size(); i++)
      if (cs[i] != 1 && cs[i] != act_shape[i]) unsupported(n, "constant operand " + shape_str(c.shape) + " does not broadcast to " + shape_str(act_shape));
    int64_t per_row = prod(act_shape, 1);
    std::vector<float> out(size_t(per_row), 0.f);
    size_t r = act_shape.size();
    for (int64_t flat = 0; flat < per_row; flat++) {
      int64_t rem = flat, idx = 0, stride = 1;
      for (size_t i = r; i-- > 1;) {
        int64_t coord = rem % act_shape[i];
        rem /= act_shape[i];
        if (cs[i] != 1) idx += coord * stride;
        stride *= cs[i];
      }
      out[size_t(flat)] = cv[size_t(idx)];
    }
    return out;
  }

  static float fold_bop(char op, float u, float v) {
    switch (op) {
      case '+': return u + v;
      case '-': return u - v;
      case '*': return u * v;
      case '/': return u / v;
      case 'm': return std::fmin(u, v);
      case 'M': return std::fmax(u, v);
      case '^': return std::pow(u, v);
      default: return u >= 0.f ? u : v * u;  // 'p'
    }
  }

  void binary(const NodeDef &n, char op) {
    if (n.inputs.size() != 2) unsupported(n, "exactly two inputs are supported");
    const Val &a = get(n, 0);
    const Val &b = get(n, 1);
    if (a.is_const && b.is_const && a.c->dtype == onnx::kInt64 && b.c->dtype == onnx::kInt64) return fold_int_binary(n, op, a, b);
    if (a.is_const && b.is_const) {  // fold
      const auto &x = cf32(n, a), &y = cf32(n, b);
      if (a.shape != b.shape && x.size() != 1 && y.size() != 1) unsupported(n, "constant folding needs equal shapes or a scalar");
      auto t = std::make_shared<TensorData>();
      t->dtype = onnx::kFloat;
      t->dims = x.size() >= y.size() ? a.shape : b.shape;
      size_t cnt = std::max(x.size(), y.size());
      t->f32.resize(cnt);
      for (size_t i = 0; i < cnt; i++) {
        float u = x[x.size() == 1 ? 0 : i], v = y[y.size() == 1 ? 0 : i];
        t->f32[i] = fold_bop(op, u, v);
      }
      Val v;
      v.is_const = true;
      v.c = t;
      v.shape = t->dims;
      vals[n.outputs[0]] = v;
      return;
    }
    if (!a.is_const && !b.is_const && op == '*' && a.shape == b.shape) {
      // x * Sigmoid(x) (Swish / SiLU as exporters spell it): the Sigmoid pass and the multiply become ONE activation on x,
      // which then folds into the epilogue of the convolution / layer that produced x when nothing else reads it
      for (int side = 0; side < 2; side++) {
        const Val &x = side ? b : a, &sg = side ? a : b;
        auto pit = producer.find(sg.buf);
        if (sg.buf <= 0 || pit == producer.end() || pit->second != int(plan.steps.size()) - 1 || live_uses(sg.buf) != 1) continue;
        const Step &ps = plan.steps.back();
        if (ps.kind != StepKind::Unary || ps.act != Act::Sigmoid || ps.in0 != x.buf) continue;
        producer.erase(pit);
        plan.steps.pop_back();
        alias_edges[x.buf]++;  // the Sigmoid node no longer consumes x
        apply_unary(n, side ? 1 : 0, Act::Swish, 0.f, 0.f, "Swish");
        return;
      }
    }
    if (!a.is_const && !b.is_const) {
      // per-channel gate: [N,C,H,W] (op) [N,C,1,1]  (squeeze-and-excitation blocks); either order for + * min max
      auto is_gate = [](const Val &big, const Val &small) {
        if (big.shape.size() < 3 || small.shape.size() != big.shape.size() || small.shape[0] != big.shape[0] || small.shape[1] != big.shape[1]) return false;
        for (size_t i = 2; i < small.shape.size(); i++)
          if (small.shape[i] != 1) return false;
        return prod(big.shape, 2) > 1;
      };
      const bool commutes = op == '+' || op == '*' || op == 'm' || op == 'M';
      if (a.shape != b.shape && (is_gate(a, b) || (commutes && is_gate(b, a)))) {
        const bool swap = !is_gate(a, b);
        const Val &big = swap ? b : a, &small = swap ? a : b;
        Step s;
        s.kind = StepKind::BinaryAct;
        s.in0 = big.buf;
        s.in1 = small.buf;
        s.bop = op;
        s.C = big.shape[1];
        s.S = prod(big.shape, 2);  // > 1 marks the broadcast form
        std::vector<int64_t> shape = big.shape;
        emit(std::move(s), n, shape);
        return;
      }
      if (a.shape != b.shape) unsupported(n, "activation operands must have equal shapes (or [N,C,H,W] with [N,C,1,1]), got " + shape_str(a.shape) + " and " + shape_str(b.shape));
      Step s;
      s.kind = StepKind::BinaryAct;
      s.in0 = a.buf;
      s.in1 = b.buf;
      s.bop = op;
      emit(std::move(s), n, a.shape);
      return;
    }
    const bool const_left = a.is_const;
    const Val &act = const_left ? b : a;
    const Val &cst = const_left ? a : b;
    std::vector<int64_t> act_shape = act.shape;
    if (op == 'p' && const_left) unsupported(n, "PRelu needs a constant slope");
    if ((op == 'm' || op == 'M') && const_left) {  // commutative: keep the activation on the left
      Step s;
      s.kind = StepKind::BinaryConst;
      s.in0 = act.buf;
      s.bop = op;
      s.cst = broadcast_const(n, cst, act_shape);
      emit(std::move(s), n, act_shape);
      return;
    }
    // MatMul + Add(const over M) -> Dense bias
    if (op == '+') {
      Step *p = fusable_producer(n, const_left ? 1 : 0);
      const auto &cv = cf32(n, cst);
      bool over_m = p && p->kind == StepKind::Dense && p->bias.empty() && p->act == Act::None && int64_t(cv.size()) == p->M &&
                    (cst.shape.size() == 1 || (cst.shape.size() == 2 && cst.shape[0] == 1));
      if (over_m) {
        p->bias = cv;
        p->origin += "+Add";
        set_act(n, act.buf, act_shape, true);
        return;
      }
    }
    // Per-feature affine arithmetic on a [rows, K] table (x - mean, / std, * scale, + shift ...) or per-channel on an
    // [N,C,H,W] tensor (a bias Add / scale Mul an exporter left behind a convolution, in-graph pixel normalisation) is
    // one multiply-add per element however many nodes spell it: consecutive ones compose into a single AffineChannel
    // pass, fold into the convolution that produced the tensor, and the Dense fold above still finds them.
    if (act_shape.size() >= 2 && (op == '+' || op == '-' || op == '*' || (op == '/' && !const_left))) {
      const std::vector<float> cv = broadcast_const(n, cst, act_shape);
      const size_t C = size_t(act_shape[1]), S = size_t(prod(act_shape, 2));
      bool per_channel = cv.size() == C * S;
      for (size_t c = 0; per_channel && c < C; c++)
        for (size_t q = 1; q < S; q++)
          if (cv[c * S + q] != cv[c * S]) { per_channel = false; break; }
      if (per_channel) {
        std::vector<double> sc(C, 1.0), sh(C, 0.0);
        for (size_t k = 0; k < C; k++) {
          const double c = cv[k * S];
          if (op == '+') sh[k] = c;
          else if (op == '-') { sc[k] = const_left ? -1.0 : 1.0; sh[k] = const_left ? c : -c; }
          else if (op == '*') sc[k] = c;
          else sc[k] = 1.0 / c;
        }
        const std::string label = n.op + (n.name.empty() ? "" : ":" + n.name);
        Step *p = fusable_producer(n, const_left ? 1 : 0);
        if (p && p->kind == StepKind::AffineChannel && p->act == Act::None && size_t(p->S) == S && size_t(p->C) == C) {
          for (size_t k = 0; k < C; k++) {
            p->shift[k] = float(sc[k] * double(p->shift[k]) + sh[k]);
            p->scale[k] = float(sc[k] * double(p->scale[k]));
          }
          p->origin += "+" + label;
          set_act(n, act.buf, act_shape, true);
          return;
        }
        if (p && p->kind == StepKind::Conv2d && p->act == Act::None && size_t(p->Mo) == C) {  // into the conv's weights and bias
          const size_t per_m = size_t(p->K);
          for (size_t mo = 0; mo < C; mo++)
            for (size_t k = 0; k < per_m; k++) p->W[mo * per_m + k] = float(sc[mo] * double(p->W[mo * per_m + k]));
          if (p->bias.empty()) p->bias.assign(C, 0.f);
          for (size_t mo = 0; mo < C; mo++) p->bias[mo] = float(sc[mo] * double(p->bias[mo]) + sh[mo]);
          p->origin += "+" + label;
          set_act(n, act.buf, act_shape, true);
          return;
        }
        Step a;
        a.kind = StepKind::AffineChannel;
        a.in0 = act.buf;
        a.C = int64_t(C);
        a.S = int64_t(S);
        a.scale.resize(C);
        a.shift.resize(C);
        for (size_t k = 0; k < C; k++) {
          a.scale[k] = float(sc[k]);
          a.shift[k] = float(sh[k]);
        }
        emit(std::move(a), n, act_shape);
        return;
      }
    }
    Step s;
    s.kind = StepKind::BinaryConst;
    s.in0 = act.buf;
    s.bop = op;
    s.const_left = const_left;
    s.cst = broadcast_const(n, cst, act_shape);
    const int src = act.buf;
    const Step &e = emit(std::move(s), n, act_shape);
    if (int_bufs.count(src) && (op == '+' || op == '-' || op == '*') &&
        std::all_of(e.cst.begin(), e.cst.end(), [](float c) { return c == std::nearbyint(c); }))
      int_bufs.insert(e.out);
  }

  // ---- constant folding of the integer (shape) sub-graphs exporters emit around Reshape ----
  void set_const_i64(const NodeDef &n, std::vector<int64_t> v, std::vector<int64_t> dims) {
    auto t = std::make_shared<TensorData>();
    t->dtype = onnx::kInt64;
    t->dims = std::move(dims);
    t->i64 = std::move(v);
    Val o;
    o.is_const = true;
    o.c = t;
    o.shape = t->dims;
    vals[n.outputs[0]] = o;
  }
  void fold_int_binary(const NodeDef &n, char op, const Val &a, const Val &b) {
    const auto &x = a.c->i64, &y = b.c->i64;
    if (x.size() != y.size() && x.size() != 1 && y.size() != 1) unsupported(n, "integer folding needs equal sizes or a scalar");
    std::vector<int64_t> o(std::max(x.size(), y.size()));
    for (size_t i = 0; i < o.size(); i++) {
      const int64_t u = x[x.size() == 1 ? 0 : i], v = y[y.size() == 1 ? 0 : i];
      if (op == '/' && v == 0) unsupported(n, "integer division by zero");
      o[i] = op == '+' ? u + v : op == '-' ? u - v : op == '*' ? u * v : op == '/' ? u / v : op == 'm' ? std::min(u, v) : std::max(u, v);
    }
    set_const_i64(n, o, x.size() >= y.size() ? a.shape : b.shape);
  }
  void shape_op(const NodeDef &n) {
    const Val &a = get(n, 0);
    std::vector<int64_t> d = a.shape;
    // the symbolic row count is carried as 0, which Reshape reads as "copy this dim from the input"
    if (!a.is_const && !d.empty() && d[0] < 0) d[0] = 0;
    int64_t r = int64_t(d.size()), st = n.attr_i("start", 0), en = n.attr_i("end", r);
    if (st < 0) st += r;
    if (en < 0) en += r;
    st = std::clamp<int64_t>(st, 0, r);
    en = std::clamp<int64_t>(en, st, r);
    std::vector<int64_t> o(d.begin() + st, d.begin() + en);
    const int64_t cnt = int64_t(o.size());
    set_const_i64(n, std::move(o), {cnt});
  }
  std::vector<int64_t> const_ints(const NodeDef &n, size_t i, const char *what) {
    const Val &v = get(n, i);
    if (!v.is_const || v.c->dtype != onnx::kInt64) unsupported(n, std::string(what) + " must be a constant integer tensor");
    return v.c->i64;
  }
  void gather(const NodeDef &n) {
    const Val &d = get(n, 0);
    const Val &ix = get(n, 1);
    // column pick out of a [rows, K] activation (the probability of one class, a feature subset): a contiguous range
    if (!d.is_const && d.shape.size() == 2 && ix.is_const && ix.c->dtype == onnx::kInt64 && !ix.c->i64.empty() && ix.shape.size() <= 1) {
      int64_t axis = n.attr_i("axis", 0);
      if (axis < 0) axis += 2;
      if (axis != 1) unsupported(n, "only axis 1 keeps rows independent");
      std::vector<int64_t> v = ix.c->i64;
      for (auto &i : v)
        if (i < 0) i += d.shape[1];
      for (size_t i = 1; i < v.size(); i++)
        if (v[i] != v[0] + int64_t(i)) unsupported(n, "only a contiguous column range");
      if (v[0] < 0 || v[0] + int64_t(v.size()) > d.shape[1]) unsupported(n, "column index out of range");
      const Val src = d;
      emit_slice_cols(n, src, v[0], v[0] + int64_t(v.size()), n.outputs[0]);
      if (ix.shape.empty()) {  // scalar index: the axis disappears ([rows] instead of [rows, 1])
        Val &o = vals[n.outputs[0]];
        o.shape = {src.shape[0]};
      }
      return;
    }
    if (!d.is_const || !ix.is_const || ix.c->dtype != onnx::kInt64) unsupported(n, "only constant data with constant indices is folded");
    if (d.shape.size() > 1 || n.attr_i("axis", 0) != 0) unsupported(n, "only 1-D data / axis 0");
    const int64_t len = d.c->dtype == onnx::kInt64 ? int64_t(d.c->i64.size()) : int64_t(d.c->f32.size());
    auto at = [&](int64_t i) {
      if (i < 0) i += len;
      if (i < 0 || i >= len) unsupported(n, "index out of range");
      return size_t(i);
    };
    if (d.c->dtype == onnx::kInt64) {
      std::vector<int64_t> o;
      for (auto i : ix.c->i64) o.push_back(d.c->i64[at(i)]);
      set_const_i64(n, std::move(o), ix.shape);
    } else {
      auto t = std::make_shared<TensorData>();
      t->dtype = onnx::kFloat;
      t->dims = ix.shape;
      for (auto i : ix.c->i64) t->f32.push_back(d.c->f32[at(i)]);
      Val o;
      o.is_const = true;
      o.c = t;
      o.shape = t->dims;
      vals[n.outputs[0]] = o;
    }
  }
  // feature-axis slice of a [rows, K] activation: out = in[:, b:e]
  // ... or a channel range of an [N,C,H,W] activation: channels [b, e) are one contiguous block of every sample, in
  // NCHW and (whole quads) in the channel-quad layout alike
  void emit_slice_cols(const NodeDef &n, const Val &a, int64_t b, int64_t e, const std::string &out_name) {
    const int64_t inner = prod(a.shape, 2);
    std::vector<int64_t> oshape = a.shape;
    oshape[1] = e - b;
    Step s;
    s.kind = StepKind::SliceCols;
    s.in0 = a.buf;
    s.col_off = b * inner;
    s.K = (e - b) * inner;
    s.out = new_buf(oshape);
    s.origin = n.op + (n.name.empty() ? "" : ":" + n.name);
    plan.steps.push_back(std::move(s));
    producer[plan.steps.back().out] = int(plan.steps.size()) - 1;
    Val v;
    v.buf = plan.steps.back().out;
    v.shape = oshape;
    vals[out_name] = v;
    buf_names[v.buf].push_back(out_name);
  }
  void split(const NodeDef &n) {
    const Val &a = get(n, 0);
    if (a.is_const || a.shape.size() < 2) unsupported(n, "only [rows, K] or [N,C,...] activations");
    int64_t axis = n.attr_i("axis", 0);
    if (axis < 0) axis += int64_t(a.shape.size());
    if (axis != 1) unsupported(n, "only axis 1 keeps rows independent");
    std::vector<int64_t> sizes;
    if (has_input(n, 1)) sizes = const_ints(n, 1, "split");
    else if (auto *p = n.attr_ints("split")) sizes = *p;
    const int64_t K = a.shape[1], nout = int64_t(n.outputs.size());
    if (sizes.empty()) {
      const int64_t parts = n.attr_i("num_outputs", nout), each = (K + parts - 1) / parts;
      for (int64_t i = 0; i < parts; i++) sizes.push_back(std::min(each, K - i * each));
    }
    if (int64_t(sizes.size()) != nout || std::accumulate(sizes.begin(), sizes.end(), int64_t(0)) != K) unsupported(n, "split sizes do not cover the axis");
    const Val src = a;  // `a` may dangle once vals grows
    int64_t off = 0;
    for (int64_t i = 0; i < nout; i++) {
      if (sizes[size_t(i)] <= 0) unsupported(n, "empty split piece");
      if (!n.outputs[size_t(i)].empty() && uses.count(n.outputs[size_t(i)])) emit_slice_cols(n, src, off, off + sizes[size_t(i)], n.outputs[size_t(i)]);
      off += sizes[size_t(i)];
    }
  }
  void slice(const NodeDef &n) {
    const Val &d = get(n, 0);
    if (!d.is_const && d.shape.size() >= 2) {  // activation: feature / channel axis slice with step 1
      std::vector<int64_t> st, en, ax, sp;
      if (has_input(n, 1)) {
        st = const_ints(n, 1, "starts");
        en = const_ints(n, 2, "ends");
        if (has_input(n, 3)) ax = const_ints(n, 3, "axes");
        if (has_input(n, 4)) sp = const_ints(n, 4, "steps");
      } else {
        if (auto *p = n.attr_ints("starts")) st = *p;
        if (auto *p = n.attr_ints("ends")) en = *p;
        if (auto *p = n.attr_ints("axes")) ax = *p;
      }
      if (st.size() != 1 || en.size() != 1 || ax.size() > 1 || (!sp.empty() && sp[0] != 1)) unsupported(n, "one axis, step 1");
      const int64_t axis = ax.empty() ? 0 : (ax[0] < 0 ? ax[0] + int64_t(d.shape.size()) : ax[0]);
      if (axis != 1) unsupported(n, "only axis 1 (features / channels)");
      const int64_t K = d.shape[1];
      int64_t b = st[0] < 0 ? st[0] + K : st[0], e = en[0] < 0 ? en[0] + K : en[0];
      b = std::clamp<int64_t>(b, 0, K);
      e = std::clamp<int64_t>(e, b, K);
      if (e == b) unsupported(n, "empty slice");
      const Val src = d;
      emit_slice_cols(n, src, b, e, n.outputs[0]);
      return;
    }
    if (!d.is_const || d.c->dtype != onnx::kInt64 || d.shape.size() != 1) unsupported(n, "only 1-D constant integer data is folded");
    std::vector<int64_t> st, en, ax, sp;
    if (has_input(n, 1)) {
      st = const_ints(n, 1, "starts");
      en = const_ints(n, 2, "ends");
      if (has_input(n, 3)) ax = const_ints(n, 3, "axes");
      if (has_input(n, 4)) sp = const_ints(n, 4, "steps");
    } else {
      if (auto *p = n.attr_ints("starts")) st = *p;
      if (auto *p = n.attr_ints("ends")) en = *p;
    }
    if (st.size() != 1 || en.size() != 1 || (!ax.empty() && ax[0] != 0 && ax[0] != -1)) unsupported(n, "one axis only");
    const int64_t len = int64_t(d.c->i64.size()), step = sp.empty() ? 1 : sp[0];
    if (step != 1) unsupported(n, "step must be 1");
    int64_t b = st[0] < 0 ? st[0] + len : st[0], e = en[0] < 0 ? en[0] + len : en[0];
    b = std::clamp<int64_t>(b, 0, len);
    e = std::clamp<int64_t>(e, b, len);
    std::vector<int64_t> o(d.c->i64.begin() + b, d.c->i64.begin() + e);
    const int64_t cnt = int64_t(o.size());
    set_const_i64(n, std::move(o), {cnt});
  }
  void cast(const NodeDef &n) {
    const Val &a = get(n, 0);
    const int64_t to = n.attr_i("to", onnx::kFloat);
    const bool to_int = to == onnx::kInt64 || to == onnx::kInt32, to_f = to == onnx::kFloat || to == onnx::kDouble;
    if (!to_int && !to_f) unsupported(n, "only casts to f32/f64/int32/int64");
    if (a.is_const) {
      if ((a.c->dtype == onnx::kInt64) == to_int) { vals[n.outputs[0]] = a; return; }
      if (to_int) {
        std::vector<int64_t> o;
        for (float f : a.c->f32) o.push_back(int64_t(f));
        set_const_i64(n, std::move(o), a.shape);
      } else {
        auto t = std::make_shared<TensorData>();
        t->dtype = onnx::kFloat;
        t->dims = a.shape;
        for (int64_t i : a.c->i64) t->f32.push_back(float(i));
        Val o;
        o.is_const = true;
        o.c = t;
        o.shape = t->dims;
        vals[n.outputs[0]] = o;
      }
      return;
    }
    // activations are always f32 here: a float cast is an alias, an integer cast truncates toward zero and
    // the values stay in f32 storage (the C ABI returns f32, rust.h:28-49)
    if (to_f || int_bufs.count(a.buf)) { alias(n, a.shape); return; }  // (already whole numbers: ArgMax labels and the like)
    Step s;
    s.kind = StepKind::Unary;
    s.in0 = a.buf;
    s.act = Act::Trunc;
    std::vector<int64_t> shape = a.shape;
    int_bufs.insert(emit(std::move(s), n, shape).out);
  }
  void concat(const NodeDef &n) {
    if (n.inputs.empty()) unsupported(n, "no inputs");
    bool all_const = true;
    for (size_t i = 0; i < n.inputs.size(); i++) all_const = all_const && get(n, i).is_const;
    if (all_const) {
      std::vector<int64_t> o;
      for (size_t i = 0; i < n.inputs.size(); i++) {
        const Val &v = get(n, i);
        if (v.c->dtype != onnx::kInt64 || v.shape.size() > 1) unsupported(n, "only 1-D integer constants are folded");
        o.insert(o.end(), v.c->i64.begin(), v.c->i64.end());
      }
      const int64_t cnt = int64_t(o.size());
      set_const_i64(n, std::move(o), {cnt});
      return;
    }
    const Val &first = get(n, 0);
    if (first.is_const) unsupported(n, "mixing constants and activations");
    const int64_t rank = int64_t(first.shape.size());
    int64_t axis = n.attr_i("axis", 1);
    if (axis < 0) axis += rank;
    if (axis != 1) unsupported(n, "only axis 1 (features / channels) is supported");
    std::vector<int64_t> out_shape = first.shape;
    out_shape[1] = 0;
    for (size_t i = 0; i < n.inputs.size(); i++) {
      const Val &v = get(n, i);
      if (v.is_const) unsupported(n, "mixing constants and activations");
      if (v.shape.size() != first.shape.size()) unsupported(n, "rank mismatch");
      for (size_t d = 0; d < v.shape.size(); d++)
        if (d != 1 && v.shape[d] != first.shape[d]) unsupported(n, "shape mismatch " + shape_str(v.shape) + " vs " + shape_str(first.shape));
      out_shape[1] += v.shape[1];
    }
    const int out = new_buf(out_shape);
    int64_t off = 0;
    for (size_t i = 0; i < n.inputs.size(); i++) {
      const Val &v = get(n, i);
      Step s;
      s.kind = StepKind::CopyCols;
      s.in0 = v.buf;
      s.out = out;
      s.col_off = off;
      s.origin = n.op + (n.name.empty() ? "" : ":" + n.name) + "[" + std::to_string(i) + "]";
      off += prod(v.shape, 1);
      plan.steps.push_back(std::move(s));
    }
    producer[out] = int(plan.steps.size()) - 1;
    set_act(n, out, out_shape);
  }
  void reduce_mean(const NodeDef &n) {
    const Val &a = get(n, 0);
    if (a.is_const || a.shape.size() < 3) unsupported(n, "only spatial means of [N,C,...] activations");
    std::vector<int64_t> axes;
    if (has_input(n, 1)) axes = const_ints(n, 1, "axes");
    else if (auto *p = n.attr_ints("axes")) axes = *p;
    const int64_t rank = int64_t(a.shape.size());
    std::vector<bool> red(size_t(rank), false);
    for (auto ax : axes) {
      const int64_t na = ax < 0 ? ax + rank : ax;
      if (na < 0 || na >= rank) unsupported(n, "axis " + std::to_string(ax) + " is out of range for rank " + std::to_string(rank));
      red[size_t(na)] = true;
    }
    for (int64_t i = 0; i < rank; i++)
      if (red[size_t(i)] != (i >= 2)) unsupported(n, "axes must be exactly the spatial axes");
    Step s;
    s.kind = StepKind::GlobalAvgPool;
    s.in0 = a.buf;
    s.C = a.shape[1];
    s.S = prod(a.shape, 2);
    std::vector<int64_t> shape = {a.shape[0], a.shape[1]};
    if (n.attr_i("keepdims", 1) != 0)
      for (int64_t i = 2; i < rank; i++) shape.push_back(1);
    emit(std::move(s), n, shape);
  }
  void argmax(const NodeDef &n) {
    const Val &a = get(n, 0);
    if (a.is_const || a.shape.size() != 2) unsupported(n, "only [rows, classes] activations");
    int64_t axis = n.attr_i("axis", 0);
    if (axis < 0) axis += 2;
    if (axis != 1) unsupported(n, "only axis 1 keeps rows independent");
    if (n.attr_i("select_last_index", 0) != 0) unsupported(n, "select_last_index=1");
    Step s;
    s.kind = StepKind::ArgMax;
    s.in0 = a.buf;
    s.K = a.shape[1];
    std::vector<int64_t> shape = {a.shape[0]};
    if (n.attr_i("keepdims", 1) != 0) shape.push_back(1);
    int_bufs.insert(emit(std::move(s), n, shape).out);
  }

  void unary(const NodeDef &n) {
    Act act;
    float pa = 0.f, pb = 0.f;
    static const std::map<std::string, Act> simple = {
        {"Relu", Act::Relu}, {"Sigmoid", Act::Sigmoid}, {"Tanh", Act::Tanh}, {"Exp", Act::Exp}, {"Log", Act::Log},
        {"Sqrt", Act::Sqrt}, {"Neg", Act::Neg}, {"Abs", Act::Abs}, {"Softplus", Act::Softplus}, {"HardSwish", Act::HardSwish},
        {"Erf", Act::Erf}, {"Reciprocal", Act::Reciprocal}, {"Floor", Act::Floor}, {"Ceil", Act::Ceil},
        {"Softsign", Act::Softsign}, {"Round", Act::Round}};
    auto si = simple.find(n.op);
    if (si != simple.end()) act = si->second;
    else if (n.op == "LeakyRelu") { act = Act::LeakyRelu; pa = n.attr_f("alpha", 0.01f); }
    else if (n.op == "Elu") { act = Act::Elu; pa = n.attr_f("alpha", 1.0f); }
    else if (n.op == "Selu") { act = Act::Selu; pa = n.attr_f("alpha", 1.67326319217681884765625f); pb = n.attr_f("gamma", 1.05070102214813232421875f); }
    else if (n.op == "HardSigmoid") { act = Act::HardSigmoid; pa = n.attr_f("alpha", 0.2f); pb = n.attr_f("beta", 0.5f); }
    else if (n.op == "Gelu") {
      if (n.attr_s("approximate", "none") != "none") unsupported(n, "only the exact (erf) form");
      act = Act::Gelu;
    }
    else {  // Clip
      act = Act::Clip;
      pa = -INFINITY;
      pb = INFINITY;
      if (auto *a = n.attr("min")) pa = a->f;
      if (auto *a = n.attr("max")) pb = a->f;
      if (has_input(n, 1)) { const Val &v = get(n, 1); if (cf32(n, v).size() != 1) unsupported(n, "min must be a scalar constant"); pa = v.c->f32[0]; }
      if (has_input(n, 2)) { const Val &v = get(n, 2); if (cf32(n, v).size() != 1) unsupported(n, "max must be a scalar constant"); pb = v.c->f32[0]; }
    }
    apply_unary(n, 0, act, pa, pb, n.op);
  }
  // activation `act` on input `idx` of node n: into the epilogue of the step that produced it when that step takes one
  // and nothing else reads the value, else an elementwise pass
  void apply_unary(const NodeDef &n, size_t idx, Act act, float pa, float pb, const std::string &label) {
    const Val &a = get(n, idx);
    if (a.is_const) unsupported(n, "activation of a constant");
    std::vector<int64_t> shape = a.shape;
    if (Step *p = fusable_producer(n, idx)) {
      const bool mfma_step = p->kind == StepKind::Dense || p->kind == StepKind::Conv2d;
      const bool takes_act = p->kind == StepKind::Dense || p->kind == StepKind::Conv2d || p->kind == StepKind::AffineChannel ||
                             p->kind == StepKind::BinaryConst || p->kind == StepKind::BinaryAct;
      if (p->act == Act::None && takes_act && (!mfma_step || mfma_fusable(act))) {
        p->act = act;
        p->act_a = pa;
        p->act_b = pb;
        p->origin += "+" + label;
        set_act(n, a.buf, shape, true);
        return;
      }
    }
    Step s;
    s.kind = StepKind::Unary;
    s.in0 = a.buf;
    s.act = act;
    s.act_a = pa;
    s.act_b = pb;
    emit(std::move(s), n, shape);
  }

  void alias(const NodeDef &n, const std::vector<int64_t> &new_shape) {
    const Val &a = get(n, 0);
    if (prod(new_shape, 1) != prod(a.shape, 1) || new_shape.empty() || new_shape[0] != a.shape[0])
      unsupported(n, "reshape " + shape_str(a.shape) + " -> " + shape_str(new_shape) + " does not preserve the row axis");
    int buf = a.buf;
    set_act(n, buf, new_shape, true);
  }

  void reshape_like(const NodeDef &n) {
    const Val &a = get(n, 0);
    if (a.is_const) {
      if (n.op == "Identity") { vals[n.outputs[0]] = a; return; }
      if ((n.op == "Unsqueeze" || n.op == "Squeeze") && a.shape.size() <= 1) {  // scalar <-> [1] in shape sub-graphs
        Val v = a;
        auto t = std::make_shared<TensorData>(*a.c);
        t->dims = n.op == "Unsqueeze" ? std::vector<int64_t>{1} : std::vector<int64_t>{};
        if (n.op == "Unsqueeze" && a.shape.size() == 1) unsupported(n, "only scalar constants are unsqueezed");
        v.c = t;
        v.shape = t->dims;
        vals[n.outputs[0]] = v;
        return;
      }
      unsupported(n, "reshaping constants is not supported");
    }
    std::vector<int64_t> out;
    const int64_t rank = int64_t(a.shape.size());
    if (n.op == "Identity" || n.op == "Dropout") {
      out = a.shape;
    } else if (n.op == "Flatten") {
      int64_t axis = n.attr_i("axis", 1);
      if (axis < 0) axis += rank;
      if (axis != 1) unsupported(n, "only axis=1 keeps the row axis");
      out = {a.shape[0], prod(a.shape, 1)};
    } else if (n.op == "Reshape") {
      std::vector<int64_t> tgt;
      if (has_input(n, 1)) {
        const Val &s = get(n, 1);
        if (!s.is_const || s.c->dtype != onnx::kInt64) unsupported(n, "shape must be a constant int64 tensor");
        tgt = s.c->i64;
      } else if (auto *p = n.attr_ints("shape")) tgt = *p;
      else unsupported(n, "missing shape");
      if (tgt.empty()) unsupported(n, "empty target shape");
      int64_t per_row = prod(a.shape, 1);
      // leading entry must denote the row axis: 0 (copy), the fixed batch, or -1 with the rest complete
      int64_t rest = 1;
      int neg = -1;
      for (size_t i = 1; i < tgt.size(); i++) {
        int64_t d = tgt[i];
        if (d == 0) { if (i >= a.shape.size()) unsupported(n, "0 entry out of range"); d = a.shape[i]; tgt[i] = d; }
        if (d == -1) { if (neg >= 0) unsupported(n, "more than one -1"); neg = int(i); continue; }
        rest *= d;
      }
      bool lead_ok = tgt[0] == 0 || (tgt[0] == a.shape[0] && a.shape[0] > 0) || (tgt[0] == -1 && neg < 0 && rest == per_row);
      if (!lead_ok) unsupported(n, "target shape " + shape_str(tgt) + " does not keep the row axis of " + shape_str(a.shape));
      if (neg >= 0) {
        if (rest == 0 || per_row % rest) unsupported(n, "cannot infer -1");
        tgt[size_t(neg)] = per_row / rest;
      }
      tgt[0] = a.shape[0];
      out = tgt;
    } else {  // Squeeze / Unsqueeze
      std::vector<int64_t> axes;
      if (has_input(n, 1)) {
        const Val &s = get(n, 1);
        if (!s.is_const || s.c->dtype != onnx::kInt64) unsupported(n, "axes must be constant");
        axes = s.c->i64;
      } else if (auto *p = n.attr_ints("axes")) axes = *p;
      if (n.op == "Squeeze") {
        for (int64_t i = 0; i < rank; i++) {
          bool drop = axes.empty() ? (a.shape[size_t(i)] == 1 && i != 0) : false;
          for (auto ax : axes) if ((ax < 0 ? ax + rank : ax) == i) drop = true;
          if (drop && i == 0) unsupported(n, "cannot squeeze the row axis");
          if (!drop) out.push_back(a.shape[size_t(i)]);
        }
      } else {
        int64_t nr = rank + int64_t(axes.size());
        size_t src = 0;
        for (int64_t i = 0; i < nr; i++) {
          bool ins = false;
          for (auto ax : axes) if ((ax < 0 ? ax + nr : ax) == i) ins = true;
          if (ins && i == 0) unsupported(n, "cannot insert an axis before the row axis");
          if (!ins && src >= a.shape.size()) unsupported(n, "axes out of range");
          out.push_back(ins ? 1 : a.shape[src++]);
        }
        if (src != a.shape.size()) unsupported(n, "axes out of range");
      }
    }
    alias(n, out);
  }

  void softmax(const NodeDef &n, bool logsm) {
    const Val &a = get(n, 0);
    if (a.is_const) unsupported(n, "softmax of a constant");
    const int64_t rank = int64_t(a.shape.size());
    int64_t axis = n.attr_i("axis", m.opset >= 13 ? -1 : 1);
    if (axis < 0) axis += rank;
    if (axis < 1 || axis >= rank) unsupported(n, "axis must address a non-row axis");
    Step s;
    s.kind = StepKind::Softmax;
    s.in0 = a.buf;
    s.log_softmax = logsm;
    s.sm_outer = prod(a.shape, 1, size_t(axis));
    if (m.opset >= 13) {
      s.sm_len = a.shape[size_t(axis)];
      s.sm_inner = prod(a.shape, size_t(axis) + 1);
    } else {
      s.sm_len = prod(a.shape, size_t(axis));
      s.sm_inner = 1;
    }
    std::vector<int64_t> shape = a.shape;
    emit(std::move(s), n, shape);
  }

  void spatial(const NodeDef &n, Step &s, int64_t H, int64_t W, const int64_t *extra_pad = nullptr) {
    s.sh = s.sw = s.dh = s.dw = 1;
    s.pt = s.pl = s.pb = s.pr = 0;
    // (1-D operators run as [N,C,1,L]: one stride / dilation, two pads, all on the W axis)
    if (auto *p = n.attr_ints("strides")) { if (p->size() == 1) s.sw = (*p)[0]; else if (p->size() != 2) unsupported(n, "only 1-D / 2-D"); else { s.sh = (*p)[0]; s.sw = (*p)[1]; } }
    if (auto *p = n.attr_ints("dilations")) { if (p->size() == 1) s.dw = (*p)[0]; else if (p->size() != 2) unsupported(n, "only 1-D / 2-D"); else { s.dh = (*p)[0]; s.dw = (*p)[1]; } }
    if (auto *p = n.attr_ints("pads")) {
      if (p->size() == 2) { s.pl = (*p)[0]; s.pr = (*p)[1]; }
      else if (p->size() != 4) unsupported(n, "only 1-D / 2-D");
      else { s.pt = (*p)[0]; s.pl = (*p)[1]; s.pb = (*p)[2]; s.pr = (*p)[3]; }
    }
    std::string ap = n.attr_s("auto_pad", "NOTSET");
    if (ap == "VALID") s.pt = s.pl = s.pb = s.pr = 0;
    else if (ap == "SAME_UPPER" || ap == "SAME_LOWER") {
      int64_t oh = (H + s.sh - 1) / s.sh, ow = (W + s.sw - 1) / s.sw;
      int64_t ph = std::max<int64_t>(0, (oh - 1) * s.sh + (s.kh - 1) * s.dh + 1 - H);
      int64_t pw = std::max<int64_t>(0, (ow - 1) * s.sw + (s.kw - 1) * s.dw + 1 - W);
      bool up = ap == "SAME_UPPER";
      s.pt = up ? ph / 2 : ph - ph / 2; s.pb = ph - s.pt;
      s.pl = up ? pw / 2 : pw - pw / 2; s.pr = pw - s.pl;
    } else if (ap != "NOTSET") unsupported(n, "auto_pad " + ap);
    if (extra_pad && (extra_pad[0] || extra_pad[1] || extra_pad[2] || extra_pad[3])) {  // a Pad node in front (TF exporters)
      if (ap == "SAME_UPPER" || ap == "SAME_LOWER") unsupported(n, "auto_pad SAME behind an explicit Pad");
      s.pt += extra_pad[0]; s.pl += extra_pad[1]; s.pb += extra_pad[2]; s.pr += extra_pad[3];
    }
    // A model file is untrusted input: attributes that would divide by zero or index backwards are rejected here
    // (the reference's parser returns an error for them; it must never take the host process down).
    if (s.sh < 1 || s.sw < 1) unsupported(n, "strides must be >= 1");
    if (s.dh < 1 || s.dw < 1) unsupported(n, "dilations must be >= 1");
    if (s.kh < 1 || s.kw < 1) unsupported(n, "kernel extents must be >= 1");
    if (s.pt < 0 || s.pl < 0 || s.pb < 0 || s.pr < 0) unsupported(n, "negative pads");
    const int64_t lim = int64_t(1) << 20;  // keeps every extent product below far inside int64
    if (s.sh > lim || s.sw > lim || s.dh > lim || s.dw > lim || s.kh > lim || s.kw > lim || s.pt > lim || s.pl > lim || s.pb > lim || s.pr > lim)
      unsupported(n, "spatial attribute out of range");
    // pooling only: ceil_mode=1 rounds the extent up, and a last window that would start beyond the input plus
    // its leading pad is dropped (ONNX MaxPool / AveragePool); the kernels already ignore out-of-image taps
    const bool ceil_mode = n.attr_i("ceil_mode", 0) != 0;
    auto extent = [&](int64_t in, int64_t p0, int64_t p1, int64_t k, int64_t d, int64_t st) {
      const int64_t num = in + p0 + p1 - (d * (k - 1) + 1);
      int64_t o = (ceil_mode ? (num + st - 1) / st : num / st) + 1;
      if (ceil_mode && (o - 1) * st >= in + p0) o--;
      return o;
    };
    s.OH = extent(H, s.pt, s.pb, s.kh, s.dh, s.sh);
    s.OW = extent(W, s.pl, s.pr, s.kw, s.dw, s.sw);
    if (s.OH <= 0 || s.OW <= 0) unsupported(n, "empty spatial output");
  }

  void conv(const NodeDef &n) {
    const Val &a = get(n, 0);
    const Val &w = get(n, 1);
    const bool one_d = a.shape.size() == 3 && w.shape.size() == 3;  // Conv1d: [N,C,L] as [N,C,1,L], kernel [M,C/g,k] as [M,C/g,1,k]
    if (a.is_const || (a.shape.size() != 4 && !one_d)) unsupported(n, "only [N,C,L] / [N,C,H,W] activations");
    if (!w.is_const || (w.shape.size() != 4 && !one_d)) unsupported(n, "weights must be a constant [M,C/g,kh,kw] (or [M,C/g,k])");
    Step s;
    s.kind = StepKind::Conv2d;
    s.in0 = a.buf;
    s.C = a.shape[1]; s.H = one_d ? 1 : a.shape[2]; s.Wd = one_d ? a.shape[2] : a.shape[3];
    s.Mo = w.shape[0]; s.kh = one_d ? 1 : w.shape[2]; s.kw = one_d ? w.shape[2] : w.shape[3];
    s.groups = n.attr_i("group", 1);
    if (s.groups < 1 || s.C != w.shape[1] * s.groups || s.Mo % s.groups) unsupported(n, "channel/group mismatch");
    spatial(n, s, s.H, s.Wd, a.pend);
    s.W = cf32(n, w);
    if (has_input(n, 2)) {
      s.bias = cf32(n, get(n, 2));
      if (int64_t(s.bias.size()) != s.Mo) unsupported(n, "bias size mismatch");
    }
    s.K = (s.C / s.groups) * s.kh * s.kw;
    s.M = s.Mo;
    std::vector<int64_t> shape = {a.shape[0], s.Mo, s.OH, s.OW};
    if (one_d) shape = {a.shape[0], s.Mo, s.OW};
    emit(std::move(s), n, shape);
  }

  void batchnorm(const NodeDef &n) {
    const Val &a = get(n, 0);
    if (a.is_const || a.shape.size() < 2) unsupported(n, "bad input");
    const auto &sc = cf32(n, get(n, 1)), &bi = cf32(n, get(n, 2)), &mu = cf32(n, get(n, 3)), &var = cf32(n, get(n, 4));
    const int64_t C = a.shape[1];
    if (int64_t(sc.size()) != C || int64_t(bi.size()) != C || int64_t(mu.size()) != C || int64_t(var.size()) != C)
      unsupported(n, "parameter size mismatch");
    float eps = n.attr_f("epsilon", 1e-5f);
    std::vector<float> scale((size_t)C), shift((size_t)C);
    for (int64_t c = 0; c < C; c++) {
      float inv = 1.0f / std::sqrt(var[size_t(c)] + eps);
      scale[size_t(c)] = sc[size_t(c)] * inv;
      shift[size_t(c)] = bi[size_t(c)] - mu[size_t(c)] * scale[size_t(c)];
    }
    std::vector<int64_t> shape = a.shape;
    if (Step *p = fusable_producer(n, 0)) {
      if (p->kind == StepKind::Conv2d && p->act == Act::None) {  // fold into conv
        const int64_t per_m = p->K;
        for (int64_t mo = 0; mo < p->Mo; mo++)
          for (int64_t k = 0; k < per_m; k++) p->W[size_t(mo * per_m + k)] *= scale[size_t(mo)];
        if (p->bias.empty()) p->bias.assign(size_t(p->Mo), 0.f);
        for (int64_t mo = 0; mo < p->Mo; mo++) p->bias[size_t(mo)] = p->bias[size_t(mo)] * scale[size_t(mo)] + shift[size_t(mo)];
        p->origin += "+BatchNormalization";
        set_act(n, a.buf, shape, true);
        return;
      }
      if (p->kind == StepKind::Dense && p->act == Act::None && a.shape.size() == 2 && p->M == C) {  // Dense -> BN (Keras MLPs)
        for (int64_t k = 0; k < p->K; k++)
          for (int64_t j = 0; j < C; j++) p->W[size_t(k * C + j)] *= scale[size_t(j)];
        if (p->bias.empty()) p->bias.assign(size_t(C), 0.f);
        for (int64_t j = 0; j < C; j++) p->bias[size_t(j)] = p->bias[size_t(j)] * scale[size_t(j)] + shift[size_t(j)];
        p->origin += "+BatchNormalization";
        set_act(n, a.buf, shape, true);
        return;
      }
    }
    Step s;
    s.kind = StepKind::AffineChannel;
    s.in0 = a.buf;
    s.C = C;
    s.S = prod(a.shape, 2);
    s.scale = std::move(scale);
    s.shift = std::move(shift);
    emit(std::move(s), n, shape);
  }

  void pool(const NodeDef &n, bool is_max) {
    const Val &a = get(n, 0);
    const bool one_d = a.shape.size() == 3;
    if (a.is_const || (a.shape.size() != 4 && !one_d)) unsupported(n, "only [N,C,L] / [N,C,H,W] activations");
    auto *ks = n.attr_ints("kernel_shape");
    if (!ks || ks->size() != (one_d ? 1u : 2u)) unsupported(n, "kernel_shape must have one entry per spatial axis");
    Step s;
    s.kind = StepKind::Pool2d;
    s.in0 = a.buf;
    s.is_max = is_max;
    s.count_pad = n.attr_i("count_include_pad", 0) != 0;
    if (!is_max && s.count_pad && n.attr_i("ceil_mode", 0) != 0) unsupported(n, "ceil_mode=1 with count_include_pad=1");
    s.C = a.shape[1]; s.H = one_d ? 1 : a.shape[2]; s.Wd = one_d ? a.shape[2] : a.shape[3];
    s.kh = one_d ? 1 : (*ks)[0]; s.kw = one_d ? (*ks)[0] : (*ks)[1];
    spatial(n, s, s.H, s.Wd);
    std::vector<int64_t> shape = {a.shape[0], s.C, s.OH, s.OW};
    if (one_d) shape = {a.shape[0], s.C, s.OW};
    emit(std::move(s), n, shape);
  }

  // Pad with zeros on the two spatial axes of an [N,C,H,W] activation: no kernel -- the value keeps its buffer and
  // the Conv that consumes it widens its own padding (the form TF / Keras exporters write for 'same' convolutions).
  void pad(const NodeDef &n) {
    const Val &a = get(n, 0);
    if (a.is_const || a.shape.size() != 4) unsupported(n, "only [N,C,H,W] activations");
    if (n.attr_s("mode", "constant") != "constant") unsupported(n, "only constant (zero) padding");
    std::vector<int64_t> pads;
    if (has_input(n, 1)) pads = const_ints(n, 1, "pads");
    else if (auto *p = n.attr_ints("pads")) pads = *p;
    if (pads.size() != 8) unsupported(n, "pads must hold 8 entries for a 4-D tensor");
    float value = n.attr_f("value", 0.f);
    if (has_input(n, 2)) {
      const auto &cv = cf32(n, get(n, 2));
      if (cv.size() != 1) unsupported(n, "constant_value must be a scalar");
      value = cv[0];
    }
    if (value != 0.f) unsupported(n, "only zero padding folds into a convolution");
    if (has_input(n, 3)) unsupported(n, "axes input");
    if (pads[0] || pads[1] || pads[4] || pads[5]) unsupported(n, "padding of the batch / channel axes");
    for (auto v : pads)
      if (v < 0) unsupported(n, "negative pads (cropping)");
    Val v = a;
    v.pend[0] += pads[2]; v.pend[1] += pads[3]; v.pend[2] += pads[6]; v.pend[3] += pads[7];
    vals[n.outputs[0]] = v;
    buf_names[v.buf].push_back(n.outputs[0]);
    alias_edges[v.buf]++;
  }
  void lrn(const NodeDef &n) {
    const Val &a = get(n, 0);
    if (a.is_const || a.shape.size() < 3) unsupported(n, "only [N,C,...] activations");
    Step s;
    s.kind = StepKind::LRN;
    s.in0 = a.buf;
    s.C = a.shape[1];
    s.S = prod(a.shape, 2);
    s.lrn_size = n.attr_i("size", 0);
    if (s.lrn_size < 1) unsupported(n, "size attribute required");
    if (s.lrn_size > (int64_t(1) << 20)) unsupported(n, "size out of range");  // the kernels carry it as int
    s.lrn_alpha = n.attr_f("alpha", 1e-4f);
    s.lrn_beta = n.attr_f("beta", 0.75f);
    s.lrn_bias = n.attr_f("bias", 1.f);
    std::vector<int64_t> shape = a.shape;
    emit(std::move(s), n, shape);
  }
  // Transpose: constants are folded (2-D weight matrices); on activations only the channel shuffle of ShuffleNet-style
  // blocks -- Reshape [N,C,H,W] -> [N,g,C/g,H,W], Transpose(0,2,1,3,4), Reshape back -- which is a channel permutation.
  void transpose(const NodeDef &n) {
    const Val &a = get(n, 0);
    const int64_t rank = int64_t(a.shape.size());
    std::vector<int64_t> perm;
    if (auto *p = n.attr_ints("perm")) perm = *p;
    else for (int64_t i = rank; i-- > 0;) perm.push_back(i);
    if (int64_t(perm.size()) != rank) unsupported(n, "perm length");
    if (a.is_const) {
      if (rank != 2 || a.c->dtype != onnx::kFloat) unsupported(n, "only 2-D f32 constants are folded");
      if (perm[0] == 0 && perm[1] == 1) { vals[n.outputs[0]] = a; return; }
      const int64_t R = a.shape[0], Cc = a.shape[1];
      std::vector<float> t(size_t(R * Cc));
      for (int64_t r = 0; r < R; r++)
        for (int64_t c = 0; c < Cc; c++) t[size_t(c * R + r)] = a.c->f32[size_t(r * Cc + c)];
      vals[n.outputs[0]] = const_f32(std::move(t), {Cc, R});
      return;
    }
    const bool shuffle = rank >= 4 && perm[0] == 0 && perm[1] == 2 && perm[2] == 1 && [&] {
      for (int64_t i = 3; i < rank; i++)
        if (perm[size_t(i)] != i) return false;
      return true;
    }();
    if (!shuffle) unsupported(n, "on activations only the channel shuffle (0,2,1,3,...) keeps rows independent and is supported");
    Step s;
    s.kind = StepKind::ChannelShuffle;
    s.in0 = a.buf;
    s.groups = a.shape[1];
    s.C = a.shape[1] * a.shape[2];
    s.S = prod(a.shape, 3);
    // the buffer is registered as the [N,C,spatial...] tensor it is for the layout rules; the value carries the 5-D shape
    std::vector<int64_t> bshape = {a.shape[0], s.C};
    for (int64_t i = 3; i < rank; i++) bshape.push_back(a.shape[size_t(i)]);
    std::vector<int64_t> vshape = a.shape;
    std::swap(vshape[1], vshape[2]);
    s.out = new_buf(bshape);
    s.origin = n.op + (n.name.empty() ? "" : ":" + n.name);
    plan.steps.push_back(std::move(s));
    producer[plan.steps.back().out] = int(plan.steps.size()) - 1;
    set_act(n, plan.steps.back().out, vshape);
  }
  // Sum of any number of equal-shaped activations: a chain of residual adds
  void sum(const NodeDef &n) {
    if (n.inputs.empty()) unsupported(n, "no inputs");
    if (n.inputs.size() == 1) { lower_node(std_node(n, "Identity", {n.inputs[0]}, n.outputs[0])); return; }
    std::string cur = n.inputs[0];
    for (size_t i = 1; i < n.inputs.size(); i++) {
      const bool last = i + 1 == n.inputs.size();
      const std::string out = last ? n.outputs[0] : n.outputs[0] + "\x01sum" + std::to_string(i);
      if (!last) uses[out] = 1;
      lower_node(std_node(n, "Add", {cur, n.inputs[i]}, out));
      cur = out;
    }
  }
  void global_avgpool(const NodeDef &n, bool is_max) {
    const Val &a = get(n, 0);
    if (a.is_const || a.shape.size() < 3) unsupported(n, "bad input");
    Step s;
    s.kind = StepKind::GlobalAvgPool;
    s.is_max = is_max;
    s.in0 = a.buf;
    s.C = a.shape[1];
    s.S = prod(a.shape, 2);
    std::vector<int64_t> shape = a.shape;
    for (size_t i = 2; i < shape.size(); i++) shape[i] = 1;
    emit(std::move(s), n, shape);
  }

  // ------------------------------------------------------------------------------------------
  void lower_node(const NodeDef &n) {
    const std::string &op = n.op;
    if (op != "Conv")
      for (const auto &in_name : n.inputs) {
        auto it = vals.find(in_name);
        if (it != vals.end() && it->second.padded()) unsupported(n, "the output of a Pad node can only feed a Conv (its padding is folded into the convolution)");
      }
    if (op == "MatMul") dense(n, false);
    else if (op == "Gemm") dense(n, true);
    else if (op == "Add") binary(n, '+');
    else if (op == "Sub") binary(n, '-');
    else if (op == "Mul") binary(n, '*');
    else if (op == "Div") binary(n, '/');
    else if (op == "Min") binary(n, 'm');
    else if (op == "Max") binary(n, 'M');
    else if (op == "Pow") binary(n, '^');
    else if (op == "PRelu") binary(n, 'p');
    else if (op == "Relu" || op == "Sigmoid" || op == "Tanh" || op == "LeakyRelu" || op == "Clip" || op == "Exp" || op == "Log" ||
             op == "Sqrt" || op == "Neg" || op == "Abs" || op == "Elu" || op == "Selu" || op == "Softplus" || op == "HardSigmoid" ||
             op == "HardSwish" || op == "Erf" || op == "Gelu" || op == "Reciprocal" || op == "Floor" || op == "Ceil" ||
             op == "Softsign" || op == "Round")
      unary(n);
    else if (op == "Shape") shape_op(n);
    else if (op == "Gather") gather(n);
    else if (op == "Slice") slice(n);
    else if (op == "Split") split(n);
    else if (op == "Cast") cast(n);
    else if (op == "Concat") concat(n);
    else if (op == "ReduceMean") reduce_mean(n);
    else if (op == "ArgMax") argmax(n);
    else if (op == "Identity" || op == "Dropout" || op == "Flatten" || op == "Reshape" || op == "Squeeze" || op == "Unsqueeze") reshape_like(n);
    else if (op == "Softmax") softmax(n, false);
    else if (op == "LogSoftmax") softmax(n, true);
    else if (op == "Conv") conv(n);
    else if (op == "BatchNormalization") batchnorm(n);
    else if (op == "MaxPool") pool(n, true);
    else if (op == "AveragePool") pool(n, false);
    else if (op == "GlobalAveragePool") global_avgpool(n, false);
    else if (op == "GlobalMaxPool") global_avgpool(n, true);
    else if (op == "Pad") pad(n);
    else if (op == "Sum") sum(n);
    else if (op == "LRN") lrn(n);
    else if (op == "Transpose") transpose(n);
    else if (op == "Constant") {
      Val v;
      v.is_const = true;
      if (auto *a = n.attr("value"); a && a->t) {
        v.c = a->t;
      } else {  // scalar / 1-D attribute forms (opset 12+)
        auto t = std::make_shared<TensorData>();
        if (auto *f = n.attr("value_float")) { t->dtype = onnx::kFloat; t->f32 = {f->f}; }
        else if (auto *i = n.attr("value_int")) { t->dtype = onnx::kInt64; t->i64 = {i->i}; }
        else if (auto *is = n.attr_ints("value_ints")) { t->dtype = onnx::kInt64; t->i64 = *is; t->dims = {int64_t(is->size())}; }
        else unsupported(n, "only the value / value_float / value_int / value_ints forms are supported");
        v.c = t;
      }
      v.shape = v.c->dims;
      vals[n.outputs[0]] = v;
    } else {
      unsupported(n, "unsupported operator");
    }
  }

  // ---- ai.onnx.ml: the classical-ML nodes sklearn exporters write (tract-onnx 0.22 ops/ml is what serves them for
  // the reference, engine.rs:49-56).  Each is rewritten into the standard operators above, so a Scaler folds into the
  // linear model behind it and a LinearClassifier's scores go through the Dense (+Softmax) kernels; semantics follow
  // the ONNX-ML operator specification (tract's sources are not in /root/reference).
  Val const_f32(std::vector<float> v, std::vector<int64_t> dims) {
    auto t = std::make_shared<TensorData>();
    t->dtype = onnx::kFloat;
    t->dims = std::move(dims);
    t->f32 = std::move(v);
    Val o;
    o.is_const = true;
    o.c = t;
    o.shape = t->dims;
    return o;
  }
  static NodeDef std_node(const NodeDef &from, const char *op, std::vector<std::string> in, std::string out) {
    NodeDef d;
    d.op = op;
    d.name = from.name.empty() ? from.op : from.op + ":" + from.name;
    d.inputs = std::move(in);
    d.outputs = {std::move(out)};
    return d;
  }
  static void set_i(NodeDef &d, const char *k, int64_t v) {
    onnx::Attribute a;
    a.name = k;
    a.type = 2;
    a.i = v;
    d.attrs[k] = a;
  }
  bool wanted(const NodeDef &n, size_t i) const {
    if (i >= n.outputs.size() || n.outputs[i].empty()) return false;
    auto it = uses.find(n.outputs[i]);
    return it != uses.end() && it->second > 0;
  }
  std::vector<float> ml_floats(const NodeDef &n, const char *k, int64_t want, float dflt, bool allow_scalar) {
    const onnx::Attribute *a = n.attr(k);
    std::vector<float> v = a ? a->floats : std::vector<float>{};
    if (v.empty()) return std::vector<float>(size_t(want), dflt);
    if (allow_scalar && v.size() == 1) return std::vector<float>(size_t(want), v[0]);
    if (int64_t(v.size()) != want) unsupported(n, std::string(k) + " holds " + std::to_string(v.size()) + " values, expected " + std::to_string(want));
    return v;
  }
  // post_transform over raw scores `raw` -> value `out`
  void ml_post_transform(const NodeDef &n, const std::string &raw, const std::string &out) {
    const std::string pt = n.attr_s("post_transform", "NONE");
    if (pt == "NONE") lower_node(std_node(n, "Identity", {raw}, out));
    else if (pt == "LOGISTIC") lower_node(std_node(n, "Sigmoid", {raw}, out));
    else if (pt == "SOFTMAX") {
      NodeDef d = std_node(n, "Softmax", {raw}, out);
      set_i(d, "axis", 1);
      lower_node(d);
    } else unsupported(n, "post_transform " + pt);
  }
  // ArrayFeatureExtractor: Y = X[..., indices].  Two forms occur in exported pipelines: a constant class table indexed
  // by the ArgMax of the scores (label lookup; the table must be evenly spaced, then it is one multiply-add on the
  // index), and a contiguous column range picked out of the feature matrix.
  void array_feature_extractor(const NodeDef &n) {
    const Val &x = get(n, 0);
    const Val &ix = get(n, 1);
    const std::string tmp = n.outputs[0] + "\x01";
    if (x.is_const && !ix.is_const) {
      if (!int_bufs.count(ix.buf)) unsupported(n, "indices must be whole numbers (an ArgMax output)");
      if (x.shape.size() != 1) unsupported(n, "only a 1-D class table");
      std::vector<double> tab;
      if (x.c->dtype == onnx::kInt64) tab.assign(x.c->i64.begin(), x.c->i64.end());
      else tab.assign(x.c->f32.begin(), x.c->f32.end());
      if (tab.size() < 2) unsupported(n, "class table needs two entries");
      const double a0 = tab[0], step = tab[1] - tab[0];
      for (size_t i = 0; i < tab.size(); i++)
        if (tab[i] != a0 + step * double(i)) unsupported(n, "class table must be evenly spaced (label = a + b * index)");
      std::string cur = n.inputs[1];
      if (step == 1.0 && a0 == 0.0) { lower_node(std_node(n, "Identity", {cur}, n.outputs[0])); return; }
      if (step != 1.0) {
        vals[tmp + "step"] = const_f32({float(step)}, {});
        const std::string nxt = a0 == 0.0 ? n.outputs[0] : tmp + "scaled";
        if (nxt != n.outputs[0]) uses[nxt] = 1;
        lower_node(std_node(n, "Mul", {cur, tmp + "step"}, nxt));
        cur = nxt;
      }
      if (a0 != 0.0) {
        vals[tmp + "first"] = const_f32({float(a0)}, {});
        lower_node(std_node(n, "Add", {cur, tmp + "first"}, n.outputs[0]));
      }
      return;
    }
    if (!x.is_const && ix.is_const && x.shape.size() == 2 && ix.c->dtype == onnx::kInt64 && !ix.c->i64.empty()) {
      const auto &v = ix.c->i64;
      for (size_t i = 1; i < v.size(); i++)
        if (v[i] != v[0] + int64_t(i)) unsupported(n, "only a contiguous column range");
      if (v[0] < 0 || v[0] + int64_t(v.size()) > x.shape[1]) unsupported(n, "column index out of range");
      const Val src = x;
      emit_slice_cols(n, src, v[0], v[0] + int64_t(v.size()), n.outputs[0]);
      return;
    }
    unsupported(n, "only (constant table, index activation) or (activation, constant contiguous indices)");
  }
  void ml_node(const NodeDef &n) {
    if (n.op == "ArrayFeatureExtractor") return array_feature_extractor(n);
    const Val &x = get(n, 0);
    if (x.is_const || x.shape.size() != 2) unsupported(n, "only [rows, features] activations");
    const int64_t F = x.shape[1];
    const std::string tmp = n.outputs[0] + "\x01";
    if (n.op == "Scaler") {
      vals[tmp + "offset"] = const_f32(ml_floats(n, "offset", F, 0.f, true), {F});
      vals[tmp + "scale"] = const_f32(ml_floats(n, "scale", F, 1.f, true), {F});
      uses[tmp + "centered"] = 1;
      lower_node(std_node(n, "Sub", {n.inputs[0], tmp + "offset"}, tmp + "centered"));
      lower_node(std_node(n, "Mul", {tmp + "centered", tmp + "scale"}, n.outputs[0]));
    } else if (n.op == "LinearRegressor" || n.op == "LinearClassifier") {
      const bool cls = n.op == "LinearClassifier";
      std::vector<int64_t> labels;
      if (cls) {
        if (n.attr("classlabels_strings")) unsupported(n, "string class labels cannot be returned as numbers");
        if (auto *p = n.attr_ints("classlabels_ints")) labels = *p;
        if (labels.size() < 2) unsupported(n, "needs at least two classlabels_ints");
      }
      const int64_t E = cls ? int64_t(labels.size()) : n.attr_i("targets", 1);
      if (E < 1) unsupported(n, "targets must be positive");
      vals[tmp + "coef"] = const_f32(ml_floats(n, "coefficients", E * F, 0.f, false), {E, F});
      vals[tmp + "icpt"] = const_f32(ml_floats(n, "intercepts", E, 0.f, false), {E});
      const bool want_label = cls && wanted(n, 0), want_scores = cls ? wanted(n, 1) : true;
      const std::string raw = tmp + "raw";
      uses[raw] = int(want_label) + int(want_scores);
      NodeDef g = std_node(n, "Gemm", {n.inputs[0], tmp + "coef", tmp + "icpt"}, raw);
      set_i(g, "transB", 1);
      lower_node(g);
      if (want_label) {
        // label = classlabels_ints[argmax(raw scores)]; the class table must be an arithmetic progression
        // (0..E-1, 1..E, {-1, 1}, ...), which is then one multiply-add on the index
        const int64_t a0 = labels[0], step = labels[1] - labels[0];
        for (size_t i = 0; i < labels.size(); i++)
          if (labels[i] != a0 + step * int64_t(i)) unsupported(n, "classlabels_ints must be evenly spaced (label = a + b * index)");
        std::string cur = (step == 1 && a0 == 0) ? n.outputs[0] : tmp + "index";
        if (cur != n.outputs[0]) uses[cur] = 1;
        NodeDef am = std_node(n, "ArgMax", {raw}, cur);
        set_i(am, "axis", 1);
        set_i(am, "keepdims", 0);
        lower_node(am);
        if (step != 1) {
          vals[tmp + "step"] = const_f32({float(step)}, {});
          const std::string nxt = a0 == 0 ? n.outputs[0] : tmp + "scaled";
          if (nxt != n.outputs[0]) uses[nxt] = 1;
          lower_node(std_node(n, "Mul", {cur, tmp + "step"}, nxt));
          cur = nxt;
        }
        if (a0 != 0) {
          vals[tmp + "first"] = const_f32({float(a0)}, {});
          lower_node(std_node(n, "Add", {cur, tmp + "first"}, n.outputs[0]));
        }
      }
      if (want_scores) ml_post_transform(n, raw, cls ? n.outputs[1] : n.outputs[0]);
    } else if (n.op == "Normalizer") {
      const std::string norm = n.attr_s("norm", "MAX");
      Step s;
      s.kind = StepKind::Softmax;
      s.in0 = x.buf;
      s.sm_norm = norm == "MAX" ? 1 : norm == "L1" ? 2 : norm == "L2" ? 3 : 0;
      if (!s.sm_norm) unsupported(n, "norm " + norm);
      s.sm_len = F;
      std::vector<int64_t> shape = x.shape;
      emit(std::move(s), n, shape);
    } else {
      unsupported(n, "unsupported operator");
    }
  }

  Plan run() {
    plan.opset = m.opset;
    const onnx::ValueDef &in = m.inputs[0];
    if (!in.has_shape) throw InferaError::onnx("input '" + in.name + "' has no declared shape");
    if (in.elem_type != 0 && in.elem_type != onnx::kFloat) throw InferaError::onnx("input '" + in.name + "' is not f32");
    if (in.dims.size() < 2) throw InferaError::onnx("input rank " + std::to_string(in.dims.size()) + " has no row axis + feature axis; rank >= 2 is required");
    for (size_t i = 1; i < in.dims.size(); i++)
      if (in.dims[i] <= 0) throw InferaError::onnx("only the leading (row/batch) dimension of the input may be symbolic, got " + shape_str(in.dims));
    plan.input_shape = in.dims;
    if (in.dims[0] > 0) plan.fixed_batch = in.dims[0];
    else plan.input_shape[0] = -1;
    if (m.inputs.size() > 1) {
      // Several runtime inputs (the reference feeds input 0 only, engine.rs:139-145, so such a model cannot run
      // there at all): the feature columns of the call are split across the inputs in declaration order.  Every
      // input must be an f32 [rows, k_i] matrix with the same leading dimension.
      int64_t total = 0;
      for (const auto &v : m.inputs) {
        if (!v.has_shape || v.dims.size() != 2 || v.dims[1] <= 0 || (v.elem_type != 0 && v.elem_type != onnx::kFloat) ||
            (v.dims[0] > 0 ? v.dims[0] : -1) != plan.input_shape[0])
          throw InferaError::onnx("multi-input models need f32 [rows, k] inputs with one common leading dimension; input '" + v.name + "' is " +
                                  (v.has_shape ? shape_str(v.dims) : std::string("unshaped")));
        total += v.dims[1];
      }
      plan.input_shape[1] = total;
    }
    {
      Val v;
      v.buf = new_buf(plan.input_shape);
      v.shape = plan.input_shape;
      if (m.inputs.size() == 1) {
        vals[in.name] = v;
        buf_names[v.buf].push_back(in.name);
      }
    }
    // Only the first output is served (engine.rs:146-149): nodes that do not feed it are dead -- a second
    // output (e.g. the probabilities next to a label) must neither cost kernels nor block loading.
    std::vector<char> live(m.nodes.size(), 0);
    {
      std::map<std::string, size_t> producer_of;
      for (size_t i = 0; i < m.nodes.size(); i++)
        for (const auto &o : m.nodes[i].outputs) producer_of[o] = i;
      std::vector<std::string> work{m.outputs[out_index].name};
      while (!work.empty()) {
        const std::string v = work.back();
        work.pop_back();
        auto it = producer_of.find(v);
        if (it == producer_of.end() || live[it->second]) continue;
        live[it->second] = 1;
        for (const auto &i : m.nodes[it->second].inputs) work.push_back(i);
      }
    }
    for (size_t i = 0; i < m.nodes.size(); i++)
      if (live[i])
        for (const auto &in_name : m.nodes[i].inputs) uses[in_name]++;
    uses[m.outputs[out_index].name]++;

    if (m.inputs.size() > 1) {
      int64_t off = 0;
      for (const auto &v : m.inputs) {
        if (uses.count(v.name) && uses[v.name] > 0) {
          Step s;
          s.kind = StepKind::SliceCols;
          s.in0 = 0;
          s.col_off = off;
          s.K = v.dims[1];
          s.out = new_buf({plan.input_shape[0], v.dims[1]});
          s.origin = "input:" + v.name;
          plan.steps.push_back(std::move(s));
          producer[plan.steps.back().out] = int(plan.steps.size()) - 1;
          Val a;
          a.buf = plan.steps.back().out;
          a.shape = {plan.input_shape[0], v.dims[1]};
          vals[v.name] = a;
          buf_names[a.buf].push_back(v.name);
        }
        off += v.dims[1];
      }
    }
    for (size_t ni = 0; ni < m.nodes.size(); ni++) {
      if (!live[ni]) continue;
      const auto &n = m.nodes[ni];
      if (n.outputs.empty()) throw InferaError::onnx("node " + n.op + " has no outputs");
      if (n.domain == "ai.onnx.ml") ml_node(n);
      else if (!n.domain.empty() && n.domain != "ai.onnx") unsupported(n, "operator domain '" + n.domain + "'");
      else lower_node(n);
    }
    // one output is served: the first (engine.rs:146-149) unless the load call selected another
    const onnx::ValueDef &out = m.outputs[out_index];
    auto it = vals.find(out.name);
    if (it == vals.end()) throw InferaError::onnx("output '" + out.name + "' is never produced");
    if (it->second.is_const) throw InferaError::onnx("output '" + out.name + "' is a constant; nothing to run");
    if (it->second.padded()) throw InferaError::onnx("output '" + out.name + "' is a Pad result; padding is only folded into a following Conv");
    if (out.elem_type != 0 && out.elem_type != onnx::kFloat) {
      // integer outputs (ArgMax labels, Cast to int) are returned as f32 VALUES: the C ABI carries f32 only
      // (rust.h:28-49; the reference itself rejects non-f32 outputs at engine.rs:150-152)
      const bool int_valued = int_bufs.count(it->second.buf) > 0;
      if (!(int_valued && (out.elem_type == onnx::kInt64 || out.elem_type == onnx::kInt32)))
        throw InferaError::onnx("output '" + out.name + "' is not f32");
      plan.output_declared_type = out.elem_type == onnx::kInt64 ? "int64" : "int32";
    }
    plan.output_name = out.name;
    plan.out_buf = it->second.buf;
    plan.output_shape = it->second.shape;
    if (plan.fixed_batch < 0) plan.output_shape[0] = -1;
    // declared output dims must agree where both are known
    if (out.has_shape) {
      if (out.dims.size() != plan.output_shape.size())
        throw InferaError::onnx("declared output rank " + std::to_string(out.dims.size()) + " differs from inferred " + shape_str(plan.output_shape));
      for (size_t i = 0; i < out.dims.size(); i++)
        if (out.dims[i] > 0 && plan.output_shape[i] > 0 && out.dims[i] != plan.output_shape[i])
          throw InferaError::onnx("declared output shape " + shape_str(out.dims) + " conflicts with inferred " + shape_str(plan.output_shape));
    }
    return std::move(plan);
  }
};

}  // namespace

Plan lower_model(const onnx::Model &m, const std::string &output_select) { return Lowerer(m, output_select).run(); }

double Plan::flops_per_row() const {
  double f = 0;
  for (const auto &s : steps) {
    if (s.kind == StepKind::Dense) f += 2.0 * double(s.K) * double(s.M);
    else if (s.kind == StepKind::Conv2d) f += 2.0 * double(s.K) * double(s.Mo) * double(s.OH) * double(s.OW);
  }
  return f;
}

std::string Plan::describe_json() const {
  static const char *kinds[] = {"Dense", "Unary", "AffineChannel", "BinaryConst", "BinaryAct", "Softmax", "Conv2d", "Pool2d", "GlobalAvgPool", "CopyCols", "ArgMax", "SliceCols", "PadCols", "LRN", "ChannelShuffle"};
  static const char *acts[] = {"", "Relu", "Sigmoid", "Tanh", "LeakyRelu", "Clip", "Exp", "Log", "Sqrt", "Neg", "Abs", "Elu", "Selu", "Softplus",
                               "HardSigmoid", "HardSwish", "Erf", "Gelu", "Reciprocal", "Floor", "Ceil", "Softsign", "Trunc", "Round", "Swish"};
  std::ostringstream o;
  o << "{\"input_shape\":" << json_int_array(input_shape) << ",\"output_shape\":" << json_int_array(output_shape)
    << ",\"flops_per_row\":" << (long long)flops_per_row() << ",\"steps\":[";
  for (size_t i = 0; i < steps.size(); i++) {
    const Step &s = steps[i];
    if (i) o << ",";
    o << "{\"kind\":\"" << kinds[int(s.kind)] << "\",\"in\":" << s.in0 << ",\"out\":" << s.out;
    if (s.in1 >= 0) o << ",\"in1\":" << s.in1;
    if (s.kind == StepKind::Dense) o << ",\"K\":" << s.K << ",\"M\":" << s.M << ",\"bias\":" << (s.bias.empty() ? "false" : "true");
    if (s.kind == StepKind::Conv2d) o << ",\"C\":" << s.C << ",\"M\":" << s.Mo << ",\"k\":[" << s.kh << "," << s.kw << "],\"out_hw\":[" << s.OH << "," << s.OW << "]";
    if (s.act != Act::None) o << ",\"act\":\"" << acts[int(s.act)] << "\"";
    o << ",\"origin\":" << json_str(s.origin) << "}";
  }
  o << "]}";
  return o.str();
}

}  // namespace infera_hip
