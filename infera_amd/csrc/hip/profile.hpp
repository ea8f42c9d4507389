// profile.hpp -- timeline markers and the per-stage report of the host path (SURVEY.md 5 "Tracing / profiling"; the reference has no
// spans, timers or counters: config.rs:200-207 is its only logging).
//
//   INFERA_PROFILE=1   (read once)  every host-ABI call pushes roctx ranges -- "infera:chunk" around the call, "infera:lease", ":gather", ":gate",
//                      ":enqueue", ":wait", ":copy_out" inside it (big-row calls: ":fill", ":enqueue", ":drain" per pipeline pass) -- which
//                      `rocprofv3 --marker-trace` records on the caller threads' rows of the timeline beside the kernels and copies
//                      (tools/e2e_timeline.py reads them); and at process exit one line per stage goes to stderr: calls, ns per chunk, and the
//                      rows/s of the whole interval between the first and the last host-ABI call.
//   unset / 0          one relaxed load per range: nothing is resolved, nothing is pushed (the library has no link-time dependency on the
//                      marker library: it is dlopen'ed on first use -- librocprofiler-sdk-roctx.so, else libroctx64.so).
#pragma once

#include <atomic>
#include <cstdint>

namespace infera_hip {
namespace prof {

bool enabled();                 // INFERA_PROFILE=1
void push(const char *name);    // roctxRangePushA (no-op when the marker library is absent)
void pop();                     // roctxRangePop
void note_call(uint64_t t_begin_ns, uint64_t t_end_ns, uint64_t rows);  // for the exit report's rows/s

// INFERA_PROFILE=2 (round 6): additionally, fine-grained SECTIONS of a host-ABI call's CPU work (registry read, validation, run lookup, lease,
// the two launches, event record, naps, queries, result copy ...) are clocked and reported at exit as ns per call: what the 18-20 us of CPU per
// chunk of the registered path are made of.  Off: one relaxed load per section.
bool sections_enabled();
void section_add(int id, const char *name, uint64_t ns);
struct Section {
  const int id;
  const char *const name;
  uint64_t t0 = 0;
  Section(int i, const char *n);
  ~Section();
  Section(const Section &) = delete;
  Section &operator=(const Section &) = delete;
};

struct Range {
  const bool on;
  explicit Range(const char *name) : on(enabled()) {
    if (on) push(name);
  }
  ~Range() {
    if (on) pop();
  }
  Range(const Range &) = delete;
  Range &operator=(const Range &) = delete;
};

}  // namespace prof
}  // namespace infera_hip
