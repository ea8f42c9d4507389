// model.cpp -- infera_load_model's device half: every scheduled step's constants packed in the order its kernel walks them and uploaded to
// each selected GPU, once.  (engine.rs:49-55: the reference builds a Tract plan here; nothing model-dependent is left per chunk.)
#include "runtime.hpp"
#include "../host/onnx_model.hpp"

namespace infera_hip {
namespace rt {
namespace {

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

}  // namespace

namespace {

// `v` padded with zeros to n floats (channel / feature counts rounded up to what a kernel's tiles need)
std::vector<float> zero_padded(const std::vector<float> &v, size_t n) {
  std::vector<float> out(n, 0.f);
  std::copy(v.begin(), v.end(), out.begin());
  return out;
}

void upload_mlp3(const LoadedModel &m, DeviceModel &dm, size_t i, hipStream_t us) {
  const auto &st = m.plan.steps;
  std::vector<float> packed(kern::mlp3_packed_floats(m.mlp3_shape));
  const Step &s1 = st[i], &s2 = st[i + 1], &s3 = st[i + 2];
  kern::mlp3_pack(m.mlp3_shape, s1.W.data(), s1.bias.empty() ? nullptr : s1.bias.data(), s2.W.data(), s2.bias.empty() ? nullptr : s2.bias.data(),
                  s3.W.data(), s3.bias.empty() ? nullptr : s3.bias.data(), packed.data());
  dm.mlp3_packed = upload(packed, us);
}

void upload_chain(const LoadedModel &m, DeviceModel &dm, size_t i, hipStream_t us) {
  const auto &st = m.plan.steps;
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
}

// returns true when the step's bias is uploaded here too (padded, or summed with a folded shortcut's)
bool upload_conv_tiled(const LoadedModel &m, DeviceStep &d, size_t i, hipStream_t us) {
  const auto &st = m.plan.steps;
  const Step &s = st[i];
  const kern::ConvGeom g = conv_geom(s), gp = kern::conv2d_tiled_geom(g);
  std::vector<float> packed(kern::conv2d_tiled_packed_floats(gp));
  if (gp.padc) {  // channel counts padded to 32: zero weights and zero bias beyond the real ones
    const size_t taps = size_t(g.kh) * g.kw;
    std::vector<float> wt(size_t(gp.M) * gp.C * taps, 0.f);
    for (int mo = 0; mo < g.M; mo++) std::copy_n(s.W.begin() + size_t(mo) * g.C * taps, size_t(g.C) * taps, wt.begin() + size_t(mo) * gp.C * taps);
    kern::conv2d_tiled_pack(gp, wt.data(), packed.data());
    d.W = upload(packed, us);
    if (!s.bias.empty()) d.bias = upload(zero_padded(s.bias, size_t(gp.M)), us);
    return true;
  }
  if (!m.conv_split6[i]) {
    kern::conv2d_tiled_pack(g, s.W.data(), packed.data());
    d.W = upload(packed, us);
    return false;
  }
  const size_t main_floats = kern::conv2d_split6_packed_floats(g);
  packed.resize(main_floats);
  kern::conv2d_split6_pack(g, s.W.data(), packed.data());
  const int fl = m.conv_fold[i];
  if (fl < 0) {
    d.W = upload(packed, us);
    return false;
  }
  // the folded 1x1 shortcut: its chunks behind the main filter's, its bias added to this layer's
  const Step &q = st[size_t(fl)];
  const kern::ConvGeom gq{int(q.C), int(q.H), int(q.Wd), int(q.Mo), int(q.OH), int(q.OW), 1, 1, int(q.sh), int(q.sw), 0, 0, 1, 1, 1};
  packed.resize(main_floats + kern::conv2d_split6_packed_floats(gq));
  kern::conv2d_split6_pack(gq, q.W.data(), packed.data() + main_floats);
  d.W = upload(packed, us);
  std::vector<float> b(s.bias);
  for (size_t k = 0; k < b.size() && k < q.bias.size(); k++) b[k] += q.bias[k];
  d.bias = upload(b, us);
  return true;
}

void upload_dense_tiled(const Step &s, DeviceStep &d, hipStream_t us) {
  const kern::ConvGeom g = dense_as_conv(s);
  std::vector<float> wt(size_t(g.C) * g.M, 0.f), packed(kern::conv2d_tiled_packed_floats(g));
  for (int64_t k = 0; k < s.K; k++)
    for (int64_t j = 0; j < s.M; j++) wt[size_t(j) * g.C + size_t(k)] = s.W[size_t(k * s.M + j)];  // [K][M] -> conv's [Mp][Cp]
  kern::conv2d_tiled_pack(g, wt.data(), packed.data());
  d.W = upload(packed, us);
  if (!s.bias.empty()) d.bias = upload(zero_padded(s.bias, size_t(g.M)), us);
}

// returns true when the step's bias is uploaded here too (padded)
bool upload_conv_patch(const LoadedModel &m, DeviceStep &d, size_t i, hipStream_t us) {
  const auto &st = m.plan.steps;
  const Step &s = st[i];
  const kern::ConvGeom g = conv_geom(s), gp = kern::conv2d_patch_geom(g);
  std::vector<float> packed(kern::conv2d_patch_packed_floats(gp));
  if (gp.mvalid > 0) {  // output features padded to whole tiles: zero weights and bias beyond the real ones
    kern::conv2d_patch_pack(gp, zero_padded(s.W, size_t(gp.M) * g.C * g.kh * g.kw).data(), packed.data());
    d.W = upload(packed, us);
    if (!s.bias.empty()) d.bias = upload(zero_padded(s.bias, size_t(gp.M)), us);
    return true;
  }
  if (const int fj = m.conv_fused_pool[i]; fj >= 0) {
    const kern::PoolTail tail = pool_tail(st[size_t(fj)]);
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
  return false;
}

}  // namespace

void upload_to_device(const LoadedModel &m, DeviceModel &dm) {
  UnsafeOpGuard guard;
  hipStream_t us = ctx_for_slot(slot_of_ordinal(dm.device)).stream;  // also does hipSetDevice
  const auto &st = m.plan.steps;
  dm.steps.resize(st.size());
  for (size_t i = 0; i < st.size(); i++) {
    const Step &s = st[i];
    DeviceStep &d = dm.steps[i];
    bool bias_done = false;
    switch (m.exec[i]) {
      case ExecKind::Skipped: continue;
      case ExecKind::Mlp3Head: upload_mlp3(m, dm, i, us); continue;
      case ExecKind::ChainHead: upload_chain(m, dm, i, us); continue;
      case ExecKind::ConvTiled: bias_done = upload_conv_tiled(m, d, i, us); break;
      case ExecKind::DenseTiled:
        upload_dense_tiled(s, d, us);
        bias_done = true;
        break;
      case ExecKind::ConvPatch: bias_done = upload_conv_patch(m, d, i, us); break;
      case ExecKind::ConvDepthwise: {
        std::vector<float> packed(s.W.size());
        kern::conv2d_depthwise_pack(conv_geom(s), s.W.data(), packed.data());
        d.W = upload(packed, us);
        break;
      }
      default:
        if (s.kind == StepKind::Conv2d) {
          const kern::ConvGeom g = conv_geom(s);
          if (!kern::conv2d_generic_supported(g))
            throw InferaError::onnx("Conv with (C/group)*kh*kw = " + std::to_string(s.K) + " > 8192 is not supported by the generic kernel");
          std::vector<float> packed(s.W.size());
          kern::conv2d_generic_pack(g, s.W.data(), packed.data());
          d.W = upload(packed, us);
        } else {
          d.W = upload(s.W, us);
        }
    }
    if (!bias_done) d.bias = upload(s.bias, us);
    if (!d.cst) d.cst = upload(s.cst, us);  // (a split stem keeps its bf16 blob there: convolutions have no constants)
    d.scale = upload(s.scale, us);
    d.shift = upload(s.shift, us);
  }
}

}  // namespace rt
using namespace rt;

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

}  // namespace infera_hip
