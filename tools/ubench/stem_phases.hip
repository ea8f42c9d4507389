// Cycle sums per phase of conv2d_stem_split6_kernel's tile loop (first workgroup pair, wave 0 of each half-channel workgroup).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DINFERA_STEM_TIMING -I infera_amd/csrc/hip -I include -o tools/ubench/stem_phases tools/ubench/stem_phases.hip
#include "../../infera_amd/csrc/hip/conv.hip"

#include <cstdio>
#include <random>

int main() {
  using namespace infera_hip::kern;
  ConvGeom g{};
  g.C = 3; g.H = g.W = 224; g.M = 64; g.OH = g.OW = 112; g.kh = g.kw = 7; g.sh = g.sw = 2; g.pt = g.pl = 3; g.dh = g.dw = 1; g.groups = 1;
  PoolTail pool{56, 56, 1, 1};
  if (!conv2d_stem_split6_supported(g, pool)) return printf("unsupported\n"), 1;
  const int64_t rows = 1024;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> w(size_t(g.M) * g.C * 49), x(size_t(rows) * 3 * 224 * 224), packed(conv2d_stem_split6_packed_floats()), bias(64);
  for (auto &v : w) v = U(rng) * 0.08f;
  for (auto &v : x) v = U(rng);
  for (auto &v : bias) v = U(rng) * 0.1f;
  conv2d_stem_split6_pack(g, w.data(), packed.data());
  float *dx, *dp, *db, *dy;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dp, packed.size() * 4); hipMalloc(&db, 256); hipMalloc(&dy, size_t(rows) * 64 * 56 * 56 * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dp, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, bias.data(), 256, hipMemcpyHostToDevice);
  ActParam act; act.kind = 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; it++) conv2d_stem_split6(nullptr, dx, dp, db, dy, rows, g, act, pool, 256);
  hipEventRecord(e0);
  for (int it = 0; it < 5; it++) conv2d_stem_split6(nullptr, dx, dp, db, dy, rows, g, act, pool, 256);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("kernel %.1f us per launch\n", ms * 200);
  unsigned long long ph[2][10];
  hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_stem_phase), sizeof(ph));
  const char *names[10] = {"k loop", "barrier (out of k loop)", "exchange write", "barrier", "pool + store", "barrier", "park", "barrier", "tile head (origin)", "first fetch"};
  for (int h = 0; h < 2; h++) {
    unsigned long long tot = 0;
    for (int i = 0; i < 10; i++) tot += ph[h][i];
    printf("half %d: total %llu ticks over 224 tiles = %.0f per tile\n", h, tot, tot / 224.0);
    for (int i = 0; i < 10; i++) printf("   %-26s %8.0f per tile  %5.1f %%\n", names[i], ph[h][i] / 224.0, 100.0 * ph[h][i] / tot);
  }
  return 0;
}
