// abi_minimal.cpp -- the C ABI (include/rcm.h) from a plain C++ program, nothing else: the calls a
// maintainer of the reference would put into main.cpp in place of its time loop (route B).
//
//   g++ -O2 -std=c++17 -Iinclude examples/abi_minimal.cpp -Lour_first_climate_model_amd -lrcm_hip \
//       -Wl,-rpath,$PWD/our_first_climate_model_amd -o abi_minimal        (or: make examples)
//   ./abi_minimal tests/golden/test.atm tests/golden/Reduced10Forcing.nc 100
//
// Prints the surface temperature and the outgoing longwave radiation after every 10 iterations.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rcm.h"

#define CK(call)                                                          \
  do {                                                                    \
    if ((call) != RCM_OK) {                                               \
      std::fprintf(stderr, "%s: %s\n", #call, rcm_last_error());          \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s ATM_FILE TABLE(.nc|.rtab) [STEPS]\n", argv[0]);
    return 2;
  }
  const int nlayer = 20, nlevel = 21, nangle = 30;  // main.cpp:78-80
  const int steps = argc > 3 ? std::atoi(argv[3]) : 100;

  // main.cpp:396-485: atmosphere file -> layer quantities of the one column
  std::vector<double> atm(9 * (size_t)nlevel), plevel(nlevel), player(nlayer), T(nlayer), conv(nlayer),
      vmr(9 * (size_t)nlayer), rel(nlayer);
  int nlev = 0, ncols = 0;
  CK(rcm_atm_load(argv[1], nlevel, atm.data(), &nlev, &ncols));
  CK(rcm_column_init(nlevel, atm.data(), /*co2_factor*/ 1.0, /*rh_scale*/ 1.0, plevel.data(), player.data(), T.data(),
                     conv.data(), vmr.data(), rel.data()));

  // main.cpp:494-504: the repwvl table goes to the device once
  rcm_ensemble *ens = nullptr;
  CK(rcm_create(/*ncol*/ 1, nlayer, nangle, /*device*/ 0, /*stream*/ nullptr, &ens));
  CK(rcm_set_grid(ens, plevel.data()));
  long long d[4];
  double *blob = nullptr;
  CK(rcm_rtab_load(argv[2], d, &blob));
  const double *p_grid = blob, *t_ref = p_grid + d[3], *t_pert = t_ref + d[3], *vmrs_ref = t_pert + d[0],
               *wvl = vmrs_ref + d[1] * d[3], *weight = wvl + d[2], *xsec = weight + d[2];
  CK(rcm_set_repwvl_tables(ens, (int)d[0], (int)d[1], (int)d[2], (int)d[3], p_grid, t_ref, t_pert, wvl, weight, xsec));
  rcm_free(blob);

  // main.cpp:512-519: solar set-up, cloud in layer 17 with tau_s / 2
  double solar[7];
  CK(rcm_solar_setup(2.0, std::cos(60 * M_PI / 180.0), 20, 0.85, 0.12, 0.5, 1361.0, solar));
  CK(rcm_set_forcing(ens, solar[6], 17, 1.0));
  const double tsfc0 = 288.2;  // main.cpp:357
  CK(rcm_set_columns(ens, T.data(), &tsfc0, vmr.data(), rel.data()));

  // main.cpp:531-583 on the GPU
  std::vector<double> Ed(nlevel), Eu(nlevel), dE(nlayer);
  for (int i = 0; i < steps; i += 10) {
    CK(rcm_step(ens, std::min(10, steps - i)));
    double tsfc, dt;
    float time_h;
    CK(rcm_get_state(ens, nullptr, &tsfc, &time_h, &dt));
    CK(rcm_get_fluxes(ens, Ed.data(), Eu.data(), dE.data()));
    std::printf("iteration %4d  time %9.2f h  dt %8.1f s  T_surface %.6f K  OLR %.6f W/m2\n", std::min(i + 10, steps),
                time_h, dt, tsfc, Eu[0]);
  }
  CK(rcm_destroy(ens));
  return 0;
}
