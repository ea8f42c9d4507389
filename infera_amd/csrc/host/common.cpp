#include "common.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace infera_hip {

namespace {
thread_local std::string g_last_error;
thread_local bool g_has_error = false;
}  // namespace

void set_last_error(const std::string &text) {
  // CString::new fails on interior NULs (error.rs:79) -- in that case the slot keeps its old value.
  if (text.find('\0') != std::string::npos) return;
  g_last_error = text;
  // Error texts quote names out of model files; a corrupted file can put arbitrary bytes there.  The reference's
  // strings are Rust `String`s (always UTF-8) and DuckDB expects UTF-8: bytes of an invalid sequence become '?'.
  if (!is_valid_utf8(g_last_error.c_str()))
    for (auto &ch : g_last_error)
      if (static_cast<unsigned char>(ch) >= 0x80) ch = '?';
  g_has_error = true;
}

const char *last_error_cstr() { return g_has_error ? g_last_error.c_str() : nullptr; }

bool is_valid_utf8(const char *s) {
  const unsigned char *p = reinterpret_cast<const unsigned char *>(s);
  while (*p) {
    unsigned char c = *p;
    int n;
    uint32_t cp;
    if (c < 0x80) { p++; continue; }
    else if ((c & 0xE0) == 0xC0) { n = 1; cp = c & 0x1F; if (c < 0xC2) return false; }
    else if ((c & 0xF0) == 0xE0) { n = 2; cp = c & 0x0F; }
    else if ((c & 0xF8) == 0xF0) { n = 3; cp = c & 0x07; if (c > 0xF4) return false; }
    else return false;
    for (int i = 1; i <= n; i++) {
      if ((p[i] & 0xC0) != 0x80) return false;
      cp = (cp << 6) | (p[i] & 0x3F);
    }
    if (n == 2 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) return false;
    if (n == 3 && (cp < 0x10000 || cp > 0x10FFFF)) return false;
    p += n + 1;
  }
  return true;
}

static std::string env_or(const char *k, const std::string &d) {
  const char *v = std::getenv(k);
  return v ? std::string(v) : d;
}
static bool env_flag(const char *k, bool d) {
  const char *v = std::getenv(k);
  if (!v) return d;
  return !(v[0] == '0' || v[0] == 'f' || v[0] == 'F' || v[0] == 'n' || v[0] == 'N' || v[0] == 0);
}
static uint64_t env_u64(const char *k, uint64_t d) {
  const char *v = std::getenv(k);
  if (!v || !*v) return d;
  char *end = nullptr;
  unsigned long long x = std::strtoull(v, &end, 10);
  return (end && *end == 0) ? (uint64_t)x : d;  // invalid -> default (config.rs:129-133)
}

const Config &Config::get() {
  static const Config cfg = [] {
    Config c;
    std::string tmp = env_or("TMPDIR", "/tmp");
    c.cache_dir = env_or("INFERA_CACHE_DIR", tmp + "/infera_cache");
    c.cache_size_limit = env_u64("INFERA_CACHE_SIZE_LIMIT", 1024ull * 1024 * 1024);
    c.http_timeout_secs = env_u64("INFERA_HTTP_TIMEOUT", 30);
    c.http_retry_attempts = uint32_t(env_u64("INFERA_HTTP_RETRY_ATTEMPTS", 3));
    c.http_retry_delay_ms = env_u64("INFERA_HTTP_RETRY_DELAY", 1000);
    std::string lvl = env_or("INFERA_LOG_LEVEL", "WARN");
    for (auto &ch : lvl) ch = (char)std::toupper((unsigned char)ch);
    c.log_level = lvl == "ERROR" ? 0 : (lvl == "INFO" ? 2 : (lvl == "DEBUG" ? 3 : 1));
    std::string devs = env_or("INFERA_DEVICES", "");
    size_t pos = 0;
    while (pos < devs.size()) {
      size_t e = devs.find(',', pos);
      if (e == std::string::npos) e = devs.size();
      std::string tok = devs.substr(pos, e - pos);
      if (!tok.empty() && tok.find_first_not_of("0123456789") == std::string::npos) c.devices.push_back(std::atoi(tok.c_str()));
      pos = e + 1;
    }
    c.use_hipgraph = env_flag("INFERA_HIPGRAPH", false);
    c.max_inflight = int(env_u64("INFERA_MAX_INFLIGHT", 12));
    c.host_contexts = std::max(1, int(env_u64("INFERA_HOST_CONTEXTS", 24)));
    c.max_inflight_total = int(env_u64("INFERA_MAX_INFLIGHT_TOTAL", 0));
    c.host_zero_copy = env_flag("INFERA_HOST_ZERO_COPY", true);
    c.zero_copy_rect = env_flag("INFERA_ZERO_COPY_RECT", true);
    c.zero_copy_rect_inflight = int(env_u64("INFERA_ZERO_COPY_RECT_INFLIGHT", 3));
    c.zero_copy_max_inflight = int(env_u64("INFERA_ZERO_COPY_MAX_INFLIGHT", 0));
    c.numa_slots = env_flag("INFERA_NUMA_SLOTS", true);
    c.host_direct_in_bytes = (long long)env_u64("INFERA_HOST_DIRECT_IN", 128 * 1024);
    c.fused_mlp = env_flag("INFERA_FUSED_MLP", true);
    c.max_rows_per_pass = env_u64("INFERA_MAX_ROWS_PER_PASS", 1ull << 18);
    c.batch_split = env_flag("INFERA_BATCH_SPLIT", false);
    return c;
  }();
  return cfg;
}

ScheduleKnobs ScheduleKnobs::read() {
  // INFERA_PRECISION: unset / "bf16x6" = the default convolution arithmetic, "fp32" = the exact-fp32 matrix instruction; anything else is a
  // typo (or a mode of an earlier round) and must not silently select one or the other: warn and keep the default
  const std::string prec = env_or("INFERA_PRECISION", "bf16x6");
  if (prec != "bf16x6" && prec != "fp32") log_msg(1, "INFERA_PRECISION='" + prec + "' is not one of bf16x6 | fp32: using the default (bf16x6)");
  return ScheduleKnobs{env_flag("INFERA_STEM_POOL", true), env_flag("INFERA_CHAIN_XCM", true), env_flag("INFERA_DENSE_XCM", true), prec != "fp32",
                       env_flag("INFERA_CONV_FOLD_SHORTCUT", true)};
}

void log_msg(int level, const std::string &msg) {
  static const char *names[] = {"ERROR", "WARN", "INFO", "DEBUG"};
  if (level <= Config::get().log_level) std::fprintf(stderr, "[%s] %s\n", names[level & 3], msg.c_str());
}

std::string json_escape(const std::string &s) {
  std::string o;
  o.reserve(s.size() + 2);
  for (unsigned char c : s) {
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      case '\b': o += "\\b"; break;
      case '\f': o += "\\f"; break;
      default:
        if (c < 0x20) {
          char buf[8];
          std::snprintf(buf, sizeof buf, "\\u%04x", c);
          o += buf;
        } else {
          o += (char)c;
        }
    }
  }
  return o;
}

std::string json_str_array(const std::vector<std::string> &v) {
  std::string o = "[";
  for (size_t i = 0; i < v.size(); i++) {
    if (i) o += ",";
    o += json_str(v[i]);
  }
  return o + "]";
}

}  // namespace infera_hip
