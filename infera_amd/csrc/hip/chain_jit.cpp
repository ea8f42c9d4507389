// chain_jit.cpp -- host side of the fused small-MLP chain (chain_device.inc): shape limits, the parameter block's
// layout, load-time specialisation with hipRTC and the launch.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "kernels.hpp"
#include "mlp_jit.hpp"

// generated at build time from hip/chain_device.inc (csrc/Makefile)
#include "chain_device_src.inc"

namespace infera_hip::kern {

namespace {

constexpr int kMaxWidth = 128;           // table columns and layer widths
constexpr size_t kLdsBudget = 160 * 1024;

int groups(int d) { return (d + 15) / 16; }
int width(const ChainShape &s, int l) { return l == 0 ? s.k0 : s.dims[size_t(l - 1)]; }
// the last layer of a chain of >= 2 with at most 4 outputs runs on the VALU (chain_device.inc, VH)
bool vhead(const ChainShape &s) { return s.dims.size() >= 2 && s.dims.back() <= 4; }
size_t layer_floats(const ChainShape &s, int l) {
  if (vhead(s) && l == int(s.dims.size())) return size_t(groups(width(s, l - 1))) * 16 * size_t(width(s, l)) + 4;
  return size_t(groups(width(s, l - 1))) * groups(width(s, l)) * 256 + 16 * size_t(groups(width(s, l)));
}
// 32-row groups per trip: about 6 KB of table per wave in flight, as long as the tiles leave room for four workgroups
// per CU beside the parameters
int row_groups(const ChainShape &s) {
  int rt = std::clamp(6144 / (128 * s.k0), 1, 8);
  const size_t par = chain_packed_floats(s) * sizeof(float);
  while (rt > 1 && par + size_t(4) * 32 * rt * (16 * groups(s.k0) + 4) * sizeof(float) > 40 * 1024) rt--;
  return rt;
}
size_t lds_bytes_with(const ChainShape &s, int waves) {
  return (chain_packed_floats(s) + size_t(waves) * 32 * row_groups(s) * (16 * groups(s.k0) + 4)) * sizeof(float);
}
// Waves per workgroup: 4, or 8 when the parameter block is so big that only one 4-wave workgroup fits a CU (the eight
// waves then share one copy of it and the CU still runs two waves per SIMD)
int waves_of(const ChainShape &s) {
  static const int forced = getenv("INFERA_CHAIN_WAVES") ? atoi(getenv("INFERA_CHAIN_WAVES")) : 0;
  if (forced == 4 || forced == 8) return lds_bytes_with(s, forced) <= kLdsBudget ? forced : 4;
  return (2 * lds_bytes_with(s, 4) > kLdsBudget && lds_bytes_with(s, 8) <= kLdsBudget) ? 8 : 4;
}
size_t lds_bytes(const ChainShape &s) { return lds_bytes_with(s, waves_of(s)); }

int bits_of(float f) {
  int b;
  std::memcpy(&b, &f, 4);
  return b;
}

std::string ints(const std::vector<int> &v) {
  std::string o = "infera_hip::kern::chaindev::Ints<";
  for (size_t i = 0; i < v.size(); i++) o += (i ? "," : "") + std::to_string(v[i]);
  return o + ">";
}

std::string expr_of(const ChainShape &s) {
  std::vector<int> pa, pb;
  for (float f : s.pa) pa.push_back(bits_of(f));
  for (float f : s.pb) pb.push_back(bits_of(f));
  return "infera_hip::kern::chaindev::chain_kernel<infera_hip::kern::chaindev::Cfg<" + std::to_string(s.k0) + "," + ints(s.dims) + "," +
         ints(s.acts) + "," + ints(pa) + "," + ints(pb) + "," + std::to_string(s.sm) + "," + std::to_string(row_groups(s)) + "," + std::to_string(waves_of(s)) + "," + (vhead(s) ? "1" : "0") + ">>";
}

struct Compiled {
  std::mutex mu;  // compile + per-device module load of THIS shape; other shapes' launches never wait on it
  bool ok = false;
  std::string why, expr, lowered;
  std::vector<char> code;
  std::map<int, hipFunction_t> fn_by_device;
};
std::mutex g_mu;  // the map only
std::map<std::string, std::shared_ptr<Compiled>> g_cache;  // keyed by the name expression (it spells out the whole shape)

bool shape_ok(const ChainShape &s, std::string &why) {
  const size_t L = s.dims.size();
  if (L < 1 || s.acts.size() != L || s.pa.size() != L || s.pb.size() != L) {
    why = "malformed chain";
    return false;
  }
  if (s.k0 < 1 || s.k0 > kMaxWidth) {
    why = "table rows of 1.." + std::to_string(kMaxWidth) + " columns";
    return false;
  }
  for (size_t l = 0; l < L; l++) {
    if (s.dims[l] < 1 || s.dims[l] > kMaxWidth) {
      why = "layer widths of 1.." + std::to_string(kMaxWidth);
      return false;
    }
    if (s.acts[l] < 0 || s.acts[l] > 5) {
      why = "only None/Relu/Sigmoid/Tanh/LeakyRelu/Clip can be fused";
      return false;
    }
  }
  if (s.sm < 0 || s.sm > 3) {
    why = "unknown epilogue";
    return false;
  }
  if (lds_bytes(s) > kLdsBudget) {
    why = "parameters + row tiles exceed the 160 KiB LDS";
    return false;
  }
  return true;
}

// entry of the shape, compiled (or failed) on return; the caller holds c.mu
Compiled &compile_locked(const ChainShape &s, std::unique_lock<std::mutex> &held) {
  const std::string expr = expr_of(s);
  std::shared_ptr<Compiled> entry;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &slot = g_cache[expr];
    if (!slot) slot = std::make_shared<Compiled>();
    entry = slot;  // entries are never erased: the reference stays valid
  }
  Compiled &c = *entry;
  held = std::unique_lock<std::mutex>(c.mu);
  if (c.ok || !c.why.empty()) return c;
  c.expr = expr;
  if (!shape_ok(s, c.why)) return c;
  c.ok = jit_compile(kChainDeviceSrc, "infera_chain_jit.hip", c.expr, c.code, c.lowered, c.why);
  return c;
}

}  // namespace

size_t chain_packed_floats(const ChainShape &s) {
  size_t n = 0;
  for (int l = 1; l <= int(s.dims.size()); l++) n += layer_floats(s, l);
  return n;
}

// Layer l: fragment (g, mt), lane, j  <->  W[k = 16g + 4(lane>>4) + j][m = 16mt + (lane&15)], zero outside [Kin, Mout].
void chain_pack(const ChainShape &s, const std::vector<const float *> &W, const std::vector<const float *> &bias, float *out) {
  for (int l = 1; l <= int(s.dims.size()); l++) {
    const int kin = width(s, l - 1), mout = width(s, l), G = groups(kin), MT = groups(mout);
    const float *w = W[size_t(l - 1)], *b = bias[size_t(l - 1)];
    if (vhead(s) && l == int(s.dims.size())) {  // [g][q][j][m], zero past the real input width; then 4 bias slots
      for (int g = 0; g < G; g++)
        for (int q = 0; q < 4; q++)
          for (int j = 0; j < 4; j++)
            for (int m = 0; m < mout; m++) {
              const int k = 16 * g + 4 * q + j;
              *out++ = k < kin ? w[size_t(k) * mout + m] : 0.f;
            }
      for (int m = 0; m < 4; m++) *out++ = (b != nullptr && m < mout) ? b[m] : 0.f;
      continue;
    }
    for (int g = 0; g < G; g++)
      for (int mt = 0; mt < MT; mt++)
        for (int lane = 0; lane < 64; lane++)
          for (int j = 0; j < 4; j++) {
            const int k = 16 * g + 4 * (lane >> 4) + j, m = 16 * mt + (lane & 15);
            *out++ = (k < kin && m < mout) ? w[size_t(k) * mout + m] : 0.f;
          }
    for (int m = 0; m < 16 * MT; m++) *out++ = (b != nullptr && m < mout) ? b[m] : 0.f;
  }
}

bool chain_supported(const ChainShape &s, std::string *why) {
  std::unique_lock<std::mutex> lk;
  Compiled &c = compile_locked(s, lk);
  if (!c.ok && why) *why = c.why;
  return c.ok;
}

bool chain(hipStream_t st, const ChainShape &s, const float *X, const float *packed, float *Y, int64_t rows, int num_cus,
           std::string *why, bool x_colmajor) {
  hipFunction_t fn = nullptr;
  const int lds = int(lds_bytes(s));
  {
    std::unique_lock<std::mutex> lk;
    Compiled &c = compile_locked(s, lk);
    if (!c.ok) {
      if (why) *why = c.why;
      return false;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto it = c.fn_by_device.find(dev);
    if (it == c.fn_by_device.end()) {
      hipModule_t mod = nullptr;
      hipError_t e = hipModuleLoadData(&mod, c.code.data());
      if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, c.lowered.c_str());
      if (e == hipSuccess && lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) {
        if (why) *why = std::string("hipModuleLoadData/GetFunction: ") + hipGetErrorString(e);
        return false;
      }
      c.fn_by_device[dev] = fn;
    } else {
      fn = it->second;
    }
  }
  if (rows <= 0) return true;
  const int64_t tile_rows = 32 * row_groups(s), ntiles = (rows + tile_rows - 1) / tile_rows;
  const int per_cu = int(std::min<size_t>(8, std::max<size_t>(1, kLdsBudget / size_t(lds))));
  const int waves = waves_of(s);
  int64_t blocks = std::min<int64_t>((ntiles + waves - 1) / waves, int64_t(num_cus) * per_cu);
  int aligned = ((reinterpret_cast<uintptr_t>(X) & 15) == 0 ? 1 : 0) | (x_colmajor ? 2 : 0);  // (chain_device.inc: x_aligned16 bits)
  void *args[] = {(void *)&X, (void *)&packed, (void *)&Y, (void *)&rows, (void *)&aligned};
  hipError_t e = hipModuleLaunchKernel(fn, unsigned(blocks), 1, 1, unsigned(waves) * 64, 1, 1, unsigned(lds), st, args, nullptr);
  if (e != hipSuccess) {
    if (why) *why = std::string("hipModuleLaunchKernel: ") + hipGetErrorString(e);
    return false;
  }
  return true;
}

std::string chain_kernel_name(const ChainShape &s) {
  std::string o = "chain_kernel<" + std::to_string(s.k0);
  for (int d : s.dims) o += "x" + std::to_string(d);
  static const char *sm[] = {"", "+softmax", "+logsoftmax", "+argmax"};
  return o + sm[s.sm] + "> [hipRTC]";
}

}  // namespace infera_hip::kern
