// mlp_layout.hpp -- run-time mirror of mlpdev::Cfg's packed-blob / LDS-image arithmetic (mlp_device.inc),
// so that chains compiled at load time by hipRTC are packed by the same host code as the ahead-of-time
// instantiations.  mlp_fused.hip asserts the two agree for every AOT config.
#pragma once

#include <cstddef>

namespace infera_hip::kern {

struct Mlp3Layout {
  int d0, d1, d2, d3;
  bool l3v;
  int G0, G1, G2, MT1, MT2, GRP2;
  int OFF_W1, N_W1, OFF_W2, N_W2, OFF_SMALL;
  int S_B1, N_B1, S_B2, N_B2, S_W3, N_W3, S_B3, N_B3, N_SMALL, N_TOTAL;
  int FIT2, NL2, L_SMALL, N_LDS;
  bool fits_lds;
};

inline Mlp3Layout mlp3_layout(int d0, int d1, int d2, int d3) {
  Mlp3Layout L{};
  L.d0 = d0; L.d1 = d1; L.d2 = d2; L.d3 = d3;
  L.l3v = d3 <= 4;
  L.G0 = d0 / 8; L.G1 = d1 / 8; L.G2 = d2 / 8;
  L.MT1 = d1 / 32; L.MT2 = d2 / 32;
  L.GRP2 = L.MT2 * 256;
  L.OFF_W1 = 0; L.N_W1 = L.G0 * L.MT1 * 256;
  L.OFF_W2 = L.N_W1; L.N_W2 = L.G1 * L.GRP2;
  L.OFF_SMALL = L.OFF_W2 + L.N_W2;
  L.S_B1 = 0; L.N_B1 = L.MT1 * 32;
  L.S_B2 = L.S_B1 + L.N_B1; L.N_B2 = L.MT2 * 32;
  L.S_W3 = L.S_B2 + L.N_B2; L.N_W3 = L.l3v ? L.MT2 * 32 * d3 : L.G2 * 256;
  L.S_B3 = L.S_W3 + L.N_W3; L.N_B3 = L.l3v ? 4 : 32;
  L.N_SMALL = L.S_B3 + L.N_B3;
  L.N_TOTAL = L.OFF_SMALL + L.N_SMALL;
  const int lds_floats = 160 * 1024 / 4;
  const int room = lds_floats - L.N_W1 - L.N_SMALL;
  L.fits_lds = room >= 0 && L.GRP2 > 0;
  L.FIT2 = L.fits_lds ? room / L.GRP2 : 0;
  L.NL2 = L.FIT2 > L.G1 ? L.G1 : L.FIT2;
  L.L_SMALL = L.N_W1 + L.NL2 * L.GRP2;
  L.N_LDS = L.L_SMALL + L.N_SMALL;
  return L;
}

}  // namespace infera_hip::kern
