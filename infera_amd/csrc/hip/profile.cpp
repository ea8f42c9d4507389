// profile.cpp -- see profile.hpp.
#include "profile.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>

#include "runtime.hpp"

namespace infera_hip {
namespace prof {
namespace {

using PushFn = int (*)(const char *);
using PopFn = int (*)();
PushFn g_push = nullptr;
PopFn g_pop = nullptr;
std::atomic<uint64_t> g_first_ns{0}, g_last_ns{0}, g_rows{0}, g_calls{0};

void report_at_exit() {
  using namespace rt;
  static const char *names[kPhCount] = {"lease", "gather", "gate", "enqueue", "wait", "copy_out"};
  const uint64_t passes = g_phase_calls.load(std::memory_order_relaxed);
  const uint64_t calls = g_calls.load(), rows = g_rows.load();
  const double span_s = double(g_last_ns.load() - g_first_ns.load()) / 1e9;
  std::fprintf(stderr, "[infera profile] host-ABI calls %llu, device passes %llu, rows %llu, first call -> last return %.3f s",
               (unsigned long long)calls, (unsigned long long)passes, (unsigned long long)rows, span_s);
  if (span_s > 0) std::fprintf(stderr, " = %.4g rows/s", double(rows) / span_s);
  std::fprintf(stderr, "\n");
  if (!passes) return;
  uint64_t total = 0;
  for (int i = 0; i < kPhCount; i++) total += g_phase_ns[i].load(std::memory_order_relaxed);
  for (int i = 0; i < kPhCount; i++) {
    const uint64_t ns = g_phase_ns[i].load(std::memory_order_relaxed);
    std::fprintf(stderr, "[infera profile]   %-8s %12llu ns total  %9.1f ns per pass  %5.1f %%\n", names[i], (unsigned long long)ns, double(ns) / double(passes),
                 total ? 100.0 * double(ns) / double(total) : 0.0);
  }
}

constexpr int kMaxSections = 48;
std::atomic<uint64_t> g_sec_ns[kMaxSections], g_sec_n[kMaxSections];
std::atomic<const char *> g_sec_name[kMaxSections];
void report_sections_at_exit() {
  const uint64_t calls = g_calls.load();
  if (!calls) return;
  std::fprintf(stderr, "[infera profile] sections, ns per host-ABI call (%llu calls; a section entered several times per call counts every time):\n", (unsigned long long)calls);
  for (int i = 0; i < kMaxSections; i++)
    if (const char *n = g_sec_name[i].load())
      std::fprintf(stderr, "[infera profile]   %-28s %9.1f ns per call  (%.2f entries per call)\n", n, double(g_sec_ns[i].load()) / double(calls),
                   double(g_sec_n[i].load()) / double(calls));
}

bool init() {
  const char *e = std::getenv("INFERA_PROFILE");
  if (e && std::atoi(e) == 2) {  // sections only: no roctx ranges (the marker library serialises the callers that push them)
    std::atexit(report_sections_at_exit);
    std::atexit(report_at_exit);
    return false;
  }
  if (!e || std::atoi(e) != 1) return false;
  for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
    if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
      g_push = reinterpret_cast<PushFn>(dlsym(h, "roctxRangePushA"));
      g_pop = reinterpret_cast<PopFn>(dlsym(h, "roctxRangePop"));
      if (g_push && g_pop) break;
      g_push = nullptr;
      g_pop = nullptr;
    }
  }
  if (!g_push) log_msg(1, "INFERA_PROFILE=1: no roctx marker library found (librocprofiler-sdk-roctx.so / libroctx64.so): the exit report only");
  std::atexit(report_at_exit);
  return true;
}

}  // namespace

bool enabled() {
  static const bool on = init();
  return on;
}
bool sections_enabled() {
  static const bool on = [] {
    (void)enabled();  // (registers the exit reports)
    const char *e = std::getenv("INFERA_PROFILE");
    return e && std::atoi(e) == 2;
  }();
  return on;
}
void section_add(int id, const char *name, uint64_t ns) {
  if (id < 0 || id >= kMaxSections) return;
  g_sec_name[id].store(name, std::memory_order_relaxed);
  g_sec_ns[id].fetch_add(ns, std::memory_order_relaxed);
  g_sec_n[id].fetch_add(1, std::memory_order_relaxed);
}
Section::Section(int i, const char *n) : id(i), name(n) {
  if (sections_enabled()) t0 = rt::now_ns();
}
Section::~Section() {
  if (t0) section_add(id, name, rt::now_ns() - t0);
}
void push(const char *name) {
  if (g_push) (void)g_push(name);
}
void pop() {
  if (g_pop) (void)g_pop();
}
void note_call(uint64_t t_begin_ns, uint64_t t_end_ns, uint64_t rows) {
  uint64_t zero = 0;
  (void)g_first_ns.compare_exchange_strong(zero, t_begin_ns);
  uint64_t last = g_last_ns.load(std::memory_order_relaxed);
  while (last < t_end_ns && !g_last_ns.compare_exchange_weak(last, t_end_ns)) {
  }
  g_rows.fetch_add(rows, std::memory_order_relaxed);
  g_calls.fetch_add(1, std::memory_order_relaxed);
}

}  // namespace prof
}  // namespace infera_hip
