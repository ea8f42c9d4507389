// Cycle sums per phase of conv2d_split6_kernel's stage loop (wave 0 of every 64th workgroup): matrix/VALU section, the wait for the stage's
// loads (weights slab + next gather), the barrier.  One ResNet-18 layer geometry per line.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DINFERA_SPLIT_TIMING -I infera_amd/csrc/hip -I include -o tools/ubench/split6_phases tools/ubench/split6_phases.hip
#include "../../infera_amd/csrc/hip/conv_split.hip"

#include <cstdio>
#include <random>

namespace infera_hip::kern {
// (conv_split.hip asks conv.hip for these; the probe links neither)
bool conv2d_tiled_supported(const ConvGeom &) { return true; }
}  // namespace infera_hip::kern

int main(int argc, char **argv) {
  using namespace infera_hip::kern;
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 1024;
  struct L { int C, HW, M; } layers[] = {{64, 56, 64}, {128, 28, 128}, {256, 14, 256}, {512, 7, 512}};
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (const L &l : layers) {
    ConvGeom g{};
    g.C = l.C; g.H = g.W = l.HW; g.M = l.M; g.OH = g.OW = l.HW; g.kh = g.kw = 3; g.sh = g.sw = 1; g.pt = g.pl = 1; g.dh = g.dw = 1; g.groups = 1;
    std::vector<float> w(size_t(g.M) * g.C * 9), x(size_t(rows) * g.C * g.H * g.W), packed(conv2d_split6_packed_floats(g));
    for (auto &v : w) v = U(rng) * 0.05f;
    for (auto &v : x) v = U(rng);
    conv2d_split6_pack(g, w.data(), packed.data());
    float *dx, *dp, *dy;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dp, packed.size() * 4); hipMalloc(&dy, size_t(rows) * g.M * g.OH * g.OW * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dp, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
    ActParam act; act.kind = 1;
    SecondInput none{};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) conv2d_split6(nullptr, dx, dp, nullptr, nullptr, dy, rows, g, act, none);
    unsigned long long zero[4] = {0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_split_phase), zero, sizeof(zero));
    hipEventRecord(e0);
    const int reps = 5;
    for (int it = 0; it < reps; it++) conv2d_split6(nullptr, dx, dp, nullptr, nullptr, dy, rows, g, act, none);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ph[4];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_split_phase), sizeof(ph));
    const double tot = double(ph[0] + ph[1] + ph[2]);
    const int nst = g.M % 128 ? (g.C / 16) * 3 : 9 * g.C / 32;
    printf("C %3d %2dx%2d M %3d: %7.1f us per launch | per workgroup %8.0f ticks, per stage %6.0f | matrix+VALU %5.1f %%  load wait %5.1f %%  barrier %5.1f %%\n", g.C, g.H,
           g.W, g.M, ms * 1000 / reps, tot / ph[3], tot / ph[3] / nst, 100 * ph[0] / tot, 100 * ph[1] / tot, 100 * ph[2] / tot);
    hipFree(dx); hipFree(dp); hipFree(dy);
  }
  return 0;
}
