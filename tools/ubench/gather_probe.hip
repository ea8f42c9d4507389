// gather_probe.hip -- what the host side of the scan can deliver with the link taken out: T threads copy 2048-row x 128-col
// chunks (128 column runs of 8 KiB each, DuckDB's row-group layout) out of a multi-GB host table into their own pinned staging
// buffer, with different copy loops.  Reports wall GB/s AND process CPU time per chunk (getrusage: user + sys), because under a
// cgroup CPU quota it is CPU time per chunk, not wall time per thread, that bounds an 8-GPU scan.
// build: hipcc --offload-arch=gfx950 -O3 -mavx2 -mavx512f -o tools/ubench/gather_probe tools/ubench/gather_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); std::exit(1); } } while (0)

static double cpu_seconds() {
  rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  return ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6 + ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
}

// ---- copy loops for one 8 KiB run (n floats, 64-byte aligned destination) ----
static void copy_memcpy(float *d, const float *s, size_t n) { std::memcpy(d, s, n * 4); }
__attribute__((target("avx2"))) static void copy_nt256(float *d, const float *s, size_t n) {
  for (size_t i = 0; i < n; i += 32) {
    __m256i a = _mm256_loadu_si256((const __m256i *)(s + i)), b = _mm256_loadu_si256((const __m256i *)(s + i + 8));
    __m256i c = _mm256_loadu_si256((const __m256i *)(s + i + 16)), e = _mm256_loadu_si256((const __m256i *)(s + i + 24));
    _mm256_stream_si256((__m256i *)(d + i), a);
    _mm256_stream_si256((__m256i *)(d + i + 8), b);
    _mm256_stream_si256((__m256i *)(d + i + 16), c);
    _mm256_stream_si256((__m256i *)(d + i + 24), e);
  }
}
__attribute__((target("avx512f"))) static void copy_nt512(float *d, const float *s, size_t n) {
  for (size_t i = 0; i < n; i += 64) {
    __m512i a = _mm512_loadu_si512(s + i), b = _mm512_loadu_si512(s + i + 16), c = _mm512_loadu_si512(s + i + 32), e = _mm512_loadu_si512(s + i + 48);
    _mm512_stream_si512((__m512i *)(d + i), a);
    _mm512_stream_si512((__m512i *)(d + i + 16), b);
    _mm512_stream_si512((__m512i *)(d + i + 32), c);
    _mm512_stream_si512((__m512i *)(d + i + 48), e);
  }
}
__attribute__((target("avx512f"))) static void copy_st512(float *d, const float *s, size_t n) {  // regular (cached) 64-byte stores
  for (size_t i = 0; i < n; i += 64) {
    __m512i a = _mm512_loadu_si512(s + i), b = _mm512_loadu_si512(s + i + 16), c = _mm512_loadu_si512(s + i + 32), e = _mm512_loadu_si512(s + i + 48);
    _mm512_store_si512((__m512i *)(d + i), a);
    _mm512_store_si512((__m512i *)(d + i + 16), b);
    _mm512_store_si512((__m512i *)(d + i + 32), c);
    _mm512_store_si512((__m512i *)(d + i + 48), e);
  }
}
static void copy_movsb(float *d, const float *s, size_t n) {
  size_t bytes = n * 4;
  asm volatile("rep movsb" : "+D"(d), "+S"(s), "+c"(bytes) : : "memory");
}
static inline void prefetch_run(const float *s, size_t lines) {
  for (size_t l = 0; l < lines; l++) _mm_prefetch((const char *)s + 64 * l, _MM_HINT_T0);
}

struct Variant {
  const char *name;
  void (*copy)(float *, const float *, size_t);
  int prefetch_lines;  // lines of the NEXT column run requested before copying this one
  bool fence;
};

int main(int argc, char **argv) {
  const size_t rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 6000000;
  const char *only = argc > 2 ? argv[2] : "";
  const int ncols = 128, CH = 2048;
  const size_t RG = 122880;
  const size_t total = rows * ncols;
  // table: anonymous mapping, optionally hugepage-advised, first-touched by the worker threads of the first run
  const bool huge = getenv("PROBE_HUGE") && getenv("PROBE_HUGE")[0] == '1';
  float *table = (float *)mmap(nullptr, total * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (table == MAP_FAILED) { perror("mmap"); return 1; }
  if (huge) madvise(table, total * 4, MADV_HUGEPAGE);
  {
    std::vector<std::thread> th;
    const int T = 16;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t] {
        const size_t per = (total + T - 1) / T, a = per * t, b = std::min(total, a + per);
        for (size_t i = a; i < b; i++) table[i] = float(i & 1023) * 0.001f;
      });
    for (auto &x : th) x.join();
  }
  const size_t nchunks = rows / CH;
  const Variant variants[] = {{"memcpy", copy_memcpy, 0, false},      {"nt256", copy_nt256, 0, true},       {"nt512", copy_nt512, 0, true},
                              {"st512", copy_st512, 0, false},        {"movsb", copy_movsb, 0, false},      {"nt512+pf4", copy_nt512, 4, true},
                              {"nt512+pf16", copy_nt512, 16, true},   {"memcpy+pf8", copy_memcpy, 8, false}, {"st512+pf8", copy_st512, 8, false}};
  const int nbuf = getenv("PROBE_NBUF") ? atoi(getenv("PROBE_NBUF")) : 1;  // staging buffers a thread cycles through
  const int passes = getenv("PROBE_PASSES") ? atoi(getenv("PROBE_PASSES")) : 1;  // scans of the table per measurement
  // PROBE_PIN: none | spread (thread t on its own CCD: physical core 8t, alternating sockets) | pack (thread t on core t)
  const std::string pin_mode = getenv("PROBE_PIN") ? getenv("PROBE_PIN") : "none";
  const int ncpu = int(sysconf(_SC_NPROCESSORS_ONLN)), nphys = ncpu / 2;
  std::printf("rows=%zu chunks=%zu huge=%d nbuf=%d passes=%d pin=%s affinity_cpus=%d\n", rows, nchunks, int(huge), nbuf, passes, pin_mode.c_str(), [] { cpu_set_t s; sched_getaffinity(0, sizeof s, &s); return CPU_COUNT(&s); }());
  for (const Variant &v : variants) {
    if (only[0] && !std::strstr(only, v.name)) continue;
    for (int T : {1, 2, 4, 8, 12, 16, 24, 32}) {
      std::atomic<size_t> next{0};
      std::atomic<int> ready{0};
      std::atomic<bool> go{false};
      std::atomic<int> tid{0};
      auto worker = [&] {
        const int me = tid.fetch_add(1);
        if (pin_mode != "none") {
          // EPYC 9575F: 2 sockets x 8 CCDs x 8 cores; logical CPU c < 128 is a physical core, c + 128 its SMT sibling; socket = c / 64
          int cpu = pin_mode == "pack" ? me % nphys : ((me % 2) * (nphys / 2) + (me / 2) * 8 % (nphys / 2) + (me / 2) * 8 / (nphys / 2)) % nphys;
          cpu_set_t set;
          CPU_ZERO(&set);
          CPU_SET(cpu, &set);
          sched_setaffinity(0, sizeof set, &set);
        }
        std::vector<float *> pins(nbuf);
        for (auto &p : pins) {
          CK(hipHostMalloc((void **)&p, size_t(CH) * ncols * 4, hipHostMallocDefault));
          std::memset(p, 0, size_t(CH) * ncols * 4);
        }
        ready++;
        while (!go.load()) std::this_thread::yield();
        size_t k = 0;
        for (;;) {
          size_t c = next.fetch_add(1);
          if (c >= nchunks * size_t(passes)) break;
          c %= nchunks;
          const size_t row0 = c * CH, g0 = row0 / RG * RG, gr = std::min(RG, rows - g0);
          const float *base = table + g0 * ncols + (row0 - g0);
          float *pin = pins[k++ % size_t(nbuf)];
          if (v.prefetch_lines) prefetch_run(base, size_t(v.prefetch_lines));
          for (int j = 0; j < ncols; j++) {
            if (v.prefetch_lines && j + 1 < ncols) prefetch_run(base + size_t(j + 1) * gr, size_t(v.prefetch_lines));
            v.copy(pin + size_t(j) * CH, base + size_t(j) * gr, CH);
          }
          if (v.fence) _mm_sfence();
        }
        for (auto &p : pins) CK(hipHostFree(p));
      };
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++) th.emplace_back(worker);
      while (ready.load() < T) std::this_thread::yield();
      const double c0 = cpu_seconds();
      const auto t0 = std::chrono::steady_clock::now();
      go = true;
      for (auto &x : th) x.join();
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const double cpu = cpu_seconds() - c0;
      const double nch = double(nchunks) * passes;
      std::printf("%-12s threads=%2d  %6.1f GB/s  %7.1f M rows/s  wall/chunk/thread %6.1f us  cpu/chunk %6.1f us  (cpu %.2f s / wall %.2f s = %.1f cpus)\n", v.name, T,
                  nch * CH * ncols * 4 / sec / 1e9, nch * CH / sec / 1e6, sec * T / nch * 1e6, cpu / nch * 1e6, cpu, sec, cpu / sec);
      std::fflush(stdout);
    }
  }
  return 0;
}
