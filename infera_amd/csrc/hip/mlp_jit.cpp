#include "mlp_jit.hpp"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>

#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "../host/common.hpp"
#include "../host/remote.hpp"
#include "mlp_layout.hpp"

// generated at build time from hip/mlp_device.inc (csrc/Makefile): `static const char kMlpDeviceSrc[] = R"..."`
#include "mlp_device_src.inc"

namespace infera_hip::kern {

namespace {

// ---- minimal hipRTC binding, resolved with dlopen so libinfera.so has no link-time dependency on it ----
using hiprtcProgram = struct _hiprtcProgram *;
struct Rtc {
  void *lib = nullptr;
  int (*CreateProgram)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*AddNameExpression)(hiprtcProgram, const char *) = nullptr;
  int (*CompileProgram)(hiprtcProgram, int, const char **) = nullptr;
  int (*GetLoweredName)(hiprtcProgram, const char *, const char **) = nullptr;
  int (*GetCodeSize)(hiprtcProgram, size_t *) = nullptr;
  int (*GetCode)(hiprtcProgram, char *) = nullptr;
  int (*GetProgramLogSize)(hiprtcProgram, size_t *) = nullptr;
  int (*GetProgramLog)(hiprtcProgram, char *) = nullptr;
  int (*DestroyProgram)(hiprtcProgram *) = nullptr;
  int (*Version)(int *, int *) = nullptr;  // optional: part of the disk-cache key
  std::string why;
  bool ok = false;
};

const Rtc &rtc() {
  static const Rtc r = [] {
    Rtc x;
    for (const char *name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
      x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (x.lib) break;
    }
    if (!x.lib) {
      x.why = std::string("hipRTC not available: ") + dlerror();
      return x;
    }
#define SYM(field, sym)                                                            \
  x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.lib, sym));               \
  if (!x.field) {                                                                  \
    x.why = std::string("hipRTC symbol missing: ") + sym;                         \
    return x;                                                                      \
  }
    SYM(CreateProgram, "hiprtcCreateProgram")
    SYM(AddNameExpression, "hiprtcAddNameExpression")
    SYM(CompileProgram, "hiprtcCompileProgram")
    SYM(GetLoweredName, "hiprtcGetLoweredName")
    SYM(GetCodeSize, "hiprtcGetCodeSize")
    SYM(GetCode, "hiprtcGetCode")
    SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
    SYM(GetProgramLog, "hiprtcGetProgramLog")
    SYM(DestroyProgram, "hiprtcDestroyProgram")
#undef SYM
    x.Version = reinterpret_cast<decltype(x.Version)>(dlsym(x.lib, "hiprtcVersion"));
    x.ok = true;
    return x;
  }();
  return r;
}


}  // namespace

// Code objects are kept on disk (INFERA_JIT_CACHE_DIR, default $TMPDIR/infera_jit_cache; "off" disables): the key is
// sha256 over the device source, the name expression, the GPU architecture and the hipRTC version, so a second process
// loading the same model shape skips the compile (0.5-3 s).  Files are written whole and renamed into place.
namespace {
std::string jit_cache_dir() {
  const char *e = getenv("INFERA_JIT_CACHE_DIR");
  if (e && (std::string(e) == "off" || std::string(e) == "0")) return "";
  if (e && *e) return e;
  const char *t = getenv("TMPDIR");
  return std::string(t && *t ? t : "/tmp") + "/infera_jit_cache";
}
bool jit_cache_read(const std::string &path, std::vector<char> &code, std::string &lowered) {
  std::ifstream f(path, std::ios::binary);
  std::string magic, low;
  size_t n = 0;
  if (!f || !std::getline(f, magic) || magic != "INFERAJIT1" || !std::getline(f, low) || !(f >> n) || f.get() != '\n' || n == 0 ||
      n > (size_t(1) << 30))
    return false;
  std::vector<char> buf(n);
  if (!f.read(buf.data(), std::streamsize(n)) || f.peek() != EOF) return false;
  code = std::move(buf);
  lowered = low;
  return true;
}
void jit_cache_write(const std::string &dir, const std::string &path, const std::vector<char> &code, const std::string &lowered) {
  (void)::mkdir(dir.c_str(), 0755);
  const std::string tmp = path + "." + std::to_string(long(::getpid())) + ".part";
  {
    std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
    if (!f) return;
    f << "INFERAJIT1\n" << lowered << "\n" << code.size() << "\n";
    f.write(code.data(), std::streamsize(code.size()));
    if (!f.good()) {
      f.close();
      (void)::unlink(tmp.c_str());
      return;
    }
  }
  if (::rename(tmp.c_str(), path.c_str()) != 0) (void)::unlink(tmp.c_str());
}
}  // namespace

bool jit_compile(const char *src, const char *file, const std::string &expr, std::vector<char> &code, std::string &lowered,
                 std::string &why) {
  const Rtc &r = rtc();
  if (!r.ok) {
    why = r.why;
    return false;
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    why = "no HIP device to compile for";
    return false;
  }
  int vmaj = 0, vmin = 0;
  if (r.Version) (void)r.Version(&vmaj, &vmin);
  const std::string dir = jit_cache_dir();
  const std::string cached =
      dir.empty() ? "" : dir + "/" + remote::sha256_hex(std::string(src) + '\n' + expr + '\n' + prop.gcnArchName + '\n' + std::to_string(vmaj) + "." + std::to_string(vmin)) + ".hsaco";
  if (!cached.empty() && jit_cache_read(cached, code, lowered)) return true;
  // hipRTC pre-includes its own runtime header (threadIdx, __global__, device math); no #include needed
  hiprtcProgram prog = nullptr;
  if (r.CreateProgram(&prog, src, file, 0, nullptr, nullptr) != 0) {
    why = "hiprtcCreateProgram failed";
    return false;
  }
  r.AddNameExpression(prog, expr.c_str());
  const std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
  const char *opts[] = {arch.c_str(), "-O3", "-std=c++17"};
  const int rc = r.CompileProgram(prog, 3, opts);
  if (rc != 0) {
    size_t n = 0;
    r.GetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) r.GetProgramLog(prog, log.data());
    why = "hipRTC compile failed: " + log.substr(0, 600);
    r.DestroyProgram(&prog);
    return false;
  }
  const char *low = nullptr;
  size_t n = 0;
  if (r.GetLoweredName(prog, expr.c_str(), &low) != 0 || !low || r.GetCodeSize(prog, &n) != 0 || n == 0) {
    why = "hipRTC produced no code object";
    r.DestroyProgram(&prog);
    return false;
  }
  lowered = low;
  code.resize(n);
  r.GetCode(prog, code.data());
  r.DestroyProgram(&prog);
  if (!cached.empty()) jit_cache_write(dir, cached, code, lowered);
  return true;
}

namespace {

using Key = std::tuple<int, int, int, int, int, int, int>;
Key key_of(const Mlp3Shape &s) { return {s.d0, s.d1, s.d2, s.d3, s.act1, s.act2, s.act3}; }

// The latency-shaped kernel for short launches (mlp3_tile_kernel, mlp_device.inc: one workgroup per 32-row tile, bit-identical to the
// persistent kernels) in its row-major and column-major (XCM) forms: what the host path's 2048-row chunks run on.
struct Variant {
  bool ok = false;
  std::string expr, lowered, why;
  std::vector<char> code;
  std::map<int, hipFunction_t> fn_by_device;
};

struct Compiled {
  std::mutex mu;  // compile + per-device module load of THIS shape
  bool ok = false;
  std::string why, expr, lowered, cfg;
  std::vector<char> code;
  int threads = 256, lds_bytes = 0;
  std::map<int, hipFunction_t> fn_by_device;
  Variant tile, tile_xcm;
  Variant tile16, tile16_xcm;  // mlp3_tile16_kernel (chunks up to kTile16MaxRows rows: 16-row tiles on the 16x16x4 instruction)
};

std::mutex g_mu;  // the map only
std::map<Key, std::shared_ptr<Compiled>> g_cache;

// Picks the kernel template and its tuning parameters for a shape; returns the name expression.
bool plan_kernel(const Mlp3Shape &s, const Mlp3Layout &L, std::string &expr, std::string &cfg, int &threads, std::string &why) {
  if (s.d0 % 8 || s.d1 % 32 || s.d2 % 32 || s.d0 < 8 || s.d1 < 32 || s.d2 < 32 || s.d3 < 1 || s.d3 > 32) {
    why = "chain dims must satisfy d0%8==0, d1%32==0, d2%32==0, 1<=d3<=32";
    return false;
  }
  for (int a : {s.act1, s.act2, s.act3})
    if (a < 0 || a > 3) {
      why = "only None/Relu/Sigmoid/Tanh can be fused";
      return false;
    }
  if (!L.fits_lds) {
    why = "layer-1 fragments exceed the 160 KiB LDS";
    return false;
  }
  const std::string cfg_head = "infera_hip::kern::mlpdev::Cfg<" + std::to_string(s.d0) + "," + std::to_string(s.d1) + "," +
                               std::to_string(s.d2) + "," + std::to_string(s.d3) + "," + std::to_string(s.act1) + "," +
                               std::to_string(s.act2) + "," + std::to_string(s.act3) + ",4,";
  const int U2 = L.G1 * L.MT2;
  // two waves per SIMD (<= 256 registers): x = G0*4, two layer-1 tiles, acc2 = MT2*16, rings
  if (L.l3v && L.MT1 >= 2) {
    const int mth = (L.MT1 % 2 == 0) ? 2 : 1, split = L.MT1 / mth;
    const int u2h = (L.G1 / split) * L.MT2, p2s = u2h >= 8 ? 8 : 4;
    const int regs = L.G0 * 4 + mth * 16 + L.MT2 * 16 + 12 + p2s * 4 + 40;
    const int p1 = L.G0 * mth < 3 ? L.G0 * mth : 3;
    if (regs <= 250 && u2h >= p2s && L.G1 % split == 0) {
      cfg = cfg_head + std::to_string(p1) + ",16>";
      expr = "infera_hip::kern::mlpdev::mlp3_split_kernel<" + cfg + "," + std::to_string(split) + "," + std::to_string(p2s) + ",8>";
      threads = 512;
      return true;
    }
  }
  // one wave per SIMD (<= 512 registers)
  int p2 = 16;
  while (p2 > U2) p2 >>= 1;
  const int regs = L.G0 * 4 + L.MT1 * 16 + L.MT2 * 16 * (L.l3v ? 2 : 1) + (L.l3v ? 0 : 16) + 12 + p2 * 4 + 48;
  const int nq = L.MT2 * 4;
  const int u1 = L.G0 * L.MT1, p1 = u1 < 3 ? u1 : 3;
  if (regs <= 480 && p2 >= 1 && (!L.l3v || u1 > nq) && (L.l3v || L.G2 >= 2)) {
    cfg = cfg_head + std::to_string(p1) + "," + std::to_string(p2) + ">";
    expr = "infera_hip::kern::mlpdev::mlp3_kernel<" + cfg + ">";
    threads = 256;
    return true;
  }
  why = "chain too wide for the register budget of the fused kernels (est. " + std::to_string(regs) + " registers)";
  return false;
}

// entry of the shape, compiled (or failed) on return; the caller holds c.mu (other shapes' launches never wait on it)
Compiled &compile_locked(const Mlp3Shape &s, std::unique_lock<std::mutex> &held) {
  std::shared_ptr<Compiled> entry;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &slot = g_cache[key_of(s)];
    if (!slot) slot = std::make_shared<Compiled>();
    entry = slot;  // entries are never erased
  }
  Compiled &c = *entry;
  held = std::unique_lock<std::mutex>(c.mu);
  if (c.ok || !c.why.empty()) return c;
  const Mlp3Layout L = mlp3_layout(s.d0, s.d1, s.d2, s.d3);
  if (!plan_kernel(s, L, c.expr, c.cfg, c.threads, c.why)) return c;
  c.lds_bytes = L.N_LDS * 4;
  c.ok = jit_compile(kMlpDeviceSrc, "infera_mlp_jit.hip", c.expr, c.code, c.lowered, c.why);
  // the tile kernel's preconditions (mlp_device.inc: VALU head, at most four layer-2 tiles, at most six layer-1 tiles per wave, its
  // 32 x (D1+4) activation tile in static LDS); a failed compile of a variant only means short launches keep the persistent kernel
  if (c.ok && L.l3v && L.MT2 <= 4 && 32 * (s.d1 + 4) * 4 <= 60 * 1024 && (L.MT1 % 4 == 0 ? L.MT1 / 4 : L.MT1 % 2 == 0 ? L.MT1 / 2 : L.MT1) <= 6) {
    for (Variant *v : {&c.tile, &c.tile_xcm}) {
      v->expr = "infera_hip::kern::mlpdev::mlp3_tile_kernel<" + c.cfg + "," + (v == &c.tile_xcm ? "true" : "false") + ">";
      v->ok = jit_compile(kMlpDeviceSrc, "infera_mlp_jit.hip", v->expr, v->code, v->lowered, v->why);
      if (!v->ok) log_msg(1, "fused MLP " + c.expr + ": no tile kernel (" + v->why.substr(0, 300) + ")");
    }
  }
  // ... and the 16-row form for the shortest launches (mlp_device.inc, mlp3_tile16_kernel: VALU head, D1 in 64s, D2 <= 128, its
  // 16 x (D1+4) activation tile in static LDS)
  if (c.ok && c.tile.ok && c.tile_xcm.ok && L.l3v && s.d1 % 64 == 0 && s.d2 <= 128 && 16 * (s.d1 + 4 + s.d2 + 4) * 4 <= 60 * 1024) {
    for (Variant *v : {&c.tile16, &c.tile16_xcm}) {
      v->expr = "infera_hip::kern::mlpdev::mlp3_tile16_kernel<" + c.cfg + "," + (v == &c.tile16_xcm ? "true" : "false") + ">";
      v->ok = jit_compile(kMlpDeviceSrc, "infera_mlp_jit.hip", v->expr, v->code, v->lowered, v->why);
      if (!v->ok) log_msg(1, "fused MLP " + c.expr + ": no 16-row tile kernel (" + v->why.substr(0, 300) + ")");
    }
  }
  return c;
}

// the function of a compiled code object on the current device (loaded on first use); the caller holds the shape's mutex
hipFunction_t function_on_device(std::vector<char> &code, const std::string &lowered, std::map<int, hipFunction_t> &cache, int lds_bytes,
                                 std::string *why) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto it = cache.find(dev);
  if (it != cache.end()) return it->second;
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  hipError_t e = hipModuleLoadData(&mod, code.data());
  if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, lowered.c_str());
  if (e == hipSuccess && lds_bytes > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) {
    if (why) *why = std::string("hipModuleLoadData/GetFunction: ") + hipGetErrorString(e);
    return nullptr;
  }
  cache[dev] = fn;
  return fn;
}

}  // namespace

bool mlp3_jit_prepare(const Mlp3Shape &sh, std::string *why) {
  std::unique_lock<std::mutex> lk;
  Compiled &c = compile_locked(sh, lk);
  if (!c.ok && why) *why = c.why;
  return c.ok;
}

constexpr int64_t kTileKernelMaxRows = 32768;  // (as for the ahead-of-time configurations, mlp_fused.hip)
constexpr int64_t kTile16MaxRows = 4096;
int64_t mlp3_tile16_max_rows() {
  static const int64_t v = [] {
    const char *e = std::getenv("INFERA_MLP_TILE16_MAX_ROWS");
    return e ? int64_t(std::atoll(e)) : kTile16MaxRows;
  }();
  return v;
}

int64_t mlp3_jit_colmajor_max_rows(const Mlp3Shape &sh) {
  std::unique_lock<std::mutex> lk;
  Compiled &c = compile_locked(sh, lk);
  return c.ok && c.tile_xcm.ok ? kTileKernelMaxRows : 0;
}

bool mlp3_jit_launch(hipStream_t s, const Mlp3Shape &sh, const float *X, const float *packed, float *Y, int64_t rows,
                     int num_cus, std::string *why, bool x_colmajor) {
  hipFunction_t fn = nullptr;
  int threads = 256, lds = 0;
  bool tile = false, tile16 = false;
  {
    std::unique_lock<std::mutex> lk;
    Compiled &c = compile_locked(sh, lk);
    if (!c.ok) {
      if (why) *why = c.why;
      return false;
    }
    Variant &v = x_colmajor ? c.tile_xcm : c.tile;
    Variant &v16 = x_colmajor ? c.tile16_xcm : c.tile16;
    if (rows <= mlp3_tile16_max_rows() && v16.ok) {
      fn = function_on_device(v16.code, v16.lowered, v16.fn_by_device, 0, why);
      tile = tile16 = fn != nullptr;  // (a 16-row module that does not load on this device: the 32-row tile variant below serves the chunk)
    }
    if (fn) {
    } else if (rows <= kTileKernelMaxRows && v.ok) {
      tile = true;
      fn = function_on_device(v.code, v.lowered, v.fn_by_device, 0, why);
    } else if (x_colmajor) {
      if (why) *why = "no column-major kernel of this chain for " + std::to_string(rows) + " rows";
      return false;
    } else {
      fn = function_on_device(c.code, c.lowered, c.fn_by_device, c.lds_bytes, why);
      threads = c.threads;
      lds = c.lds_bytes;
    }
    if (!fn) return false;
  }
  const int64_t ntiles = (rows + 31) / 32;
  int64_t blocks = tile16 ? (rows + 15) / 16 : ntiles;  // tile kernels: one workgroup per 32-row (16-row) tile
  if (!tile) {
    const int waves = threads / 64;
    blocks = (ntiles + waves - 1) / waves;
    if (blocks > num_cus) blocks = num_cus;
    if (blocks < 1) blocks = 1;
  }
  void *args[] = {(void *)&X, (void *)&packed, (void *)&Y, (void *)&rows};
  hipError_t e = hipModuleLaunchKernel(fn, unsigned(blocks), 1, 1, unsigned(threads), 1, 1, unsigned(lds), s, args, nullptr);
  if (e != hipSuccess) {
    if (why) *why = std::string("hipModuleLaunchKernel: ") + hipGetErrorString(e);
    return false;
  }
  return true;
}

std::string mlp3_jit_kernel_name(const Mlp3Shape &sh) {
  std::shared_ptr<Compiled> entry;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cache.find(key_of(sh));
    if (it == g_cache.end()) return std::string();
    entry = it->second;
  }
  std::lock_guard<std::mutex> el(entry->mu);
  return entry->ok ? entry->expr + " [hipRTC]" : std::string();
}

}  // namespace infera_hip::kern
