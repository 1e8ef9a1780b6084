// rcm_main.cpp -- drop-in for the reference's main() (main.cpp:353-596) on top of the C ABI.
//
// Same sequence: read the atmosphere file, build the layer quantities, load the optical-depth
// tables, solar set-up, banner, then the time loop i = 0..n_steps INCLUSIVE with an output_conv
// block every output_steps iterations BEFORE that iteration's radiative transfer, all appended to
// output.txt in the reference's formats.  What the reference compiles in (Consts, main.cpp:66-92;
// file paths main.cpp:397,500; CO2 factor main.cpp:452) is a command-line option here, with the
// reference's value as the default.  The loop body runs on the GPU (rcm_step); with --ncol N the
// column is replicated into a CO2 sweep and column 0.. are integrated together.
//
//   rcm_main [--atm F] [--table F.nc|F.rtab | --lbl-dir D [--lbl-cache F]] [--steps N] [--output-steps K]
//            [--co2-factor X] [--ncol N] [--tsfc T] [--tau-s X] [--output F] [--device I | --devices a,b,..]
//            [--print-cols a,b,..] [--equilibrium TOL_K] [--restart F] [--checkpoint F] [--gather-out F]
// --devices: the ensemble is cut into contiguous blocks of ceil(ncol/n) columns, one handle per listed
//            GPU, all driven from this one process (rcm_step only enqueues); at the end ONE RCCL
//            all-gather over the GPUs collects (Tlayer[20], T_surface) of every column.
//            A device listed twice gets two handles and the gather goes through the host instead.
// --gather-out: the gathered state, one line per column: col,T_surface,Tlayer[0..19].
// --lbl-cache: binary cache of the five ASCII tables (read if present, else written after parsing).
// --print-cols: columns whose profile block is written at every output time (default 0).
// --equilibrium: instead of a fixed count, step until the adjusted profile changes by <= TOL_K per
//                step in every column (rcm_run_to_equilibrium), at most --steps iterations.
// --restart / --checkpoint: resume from / write the binary restart file (rcm_checkpoint_*); with several
//            handles each has its own file, F.0, F.1, ..
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rcm.h"

#ifdef RCM_WITH_RCCL
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#endif

namespace {

struct Options {
  std::string atm = "./repwvl_V2.01_cpp/test.atm";                // main.cpp:397
  std::string table = "./repwvl_V2.01_cpp/Reduced100Forcing.nc";   // main.cpp:500 (or a flattened .rtab)
  std::string lbl_dir, lbl_cache, restart, checkpoint, gather_out;
  std::vector<int> devices;  // empty: the single --device, no gather
  std::vector<int> print_cols = {0};
  double equilibrium = -1;  // < 0: fixed step count
  std::string output = "output.txt";                               // main.cpp:105,512
  int steps = 1;          // Consts::n_steps, main.cpp:83
  int output_steps = 10;  // Consts::output_steps, main.cpp:84
  double co2_factor = 1;  // main.cpp:452
  double tsfc = 288.2;    // main.cpp:357
  double tau_s = 2.0;     // main.cpp:86
  int ncol = 1, device = 0;
};

[[noreturn]] void die(const std::string &what) {
  std::fprintf(stderr, "rcm_main: %s: %s\n", what.c_str(), rcm_last_error());
  std::exit(1);
}

#define CK(call)                 \
  do {                           \
    if ((call) != RCM_OK) die(#call); \
  } while (0)

Options parse(int argc, char **argv) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char * {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "rcm_main: %s needs a value\n", a.c_str());
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "--help" || a == "-h") {
      std::printf(
          "rcm_main: the reference's main() (main.cpp:353-596) on the MI355X library.\n"
          "  --atm F            atmosphere file (default ./repwvl_V2.01_cpp/test.atm)\n"
          "  --table F          repwvl table, Reduced*Forcing.nc or its .rtab blob (default ./repwvl_V2.01_cpp/Reduced100Forcing.nc)\n"
          "  --lbl-dir D        line-by-line path: D/lbl.{co2,h2o,o3,ch4,n2o}.asc (lbl.arts/README)\n"
          "  --lbl-cache F      binary cache of the five tables: read if present, else written\n"
          "  --steps N          Consts::n_steps (1): iterations i = 0..N inclusive\n"
          "  --output-steps K   Consts::output_steps (10)\n"
          "  --co2-factor X     CO2 multiplier (main.cpp:452)\n"
          "  --tsfc T           initial surface temperature (288.2)\n"
          "  --tau-s X          optical thickness of the cloud in the solar set-up (2.0)\n"
          "  --ncol N           ensemble: CO2 factor x linspace(0.5, 4) over N columns\n"
          "  --print-cols a,b   columns whose profile block is written at every output time (0)\n"
          "  --equilibrium TOL  stop when the adjusted profile changes by <= TOL K per step in every column\n"
          "  --restart F / --checkpoint F   binary restart file\n"
          "  --device I | --devices a,b,..  one GPU, or one handle per listed GPU and one RCCL all-gather\n"
          "  --gather-out F     the gathered state: col,T_surface,Tlayer[0..19] per line\n"
          "  --output F         output.txt (appended, reference formats)\n");
      std::exit(0);
    }
    else if (a == "--atm") o.atm = val();
    else if (a == "--table") o.table = val();
    else if (a == "--lbl-dir") o.lbl_dir = val();
    else if (a == "--output") o.output = val();
    else if (a == "--steps") o.steps = std::atoi(val());
    else if (a == "--output-steps") o.output_steps = std::atoi(val());
    else if (a == "--co2-factor") o.co2_factor = std::atof(val());
    else if (a == "--tsfc") o.tsfc = std::atof(val());
    else if (a == "--tau-s") o.tau_s = std::atof(val());
    else if (a == "--ncol") o.ncol = std::atoi(val());
    else if (a == "--device") o.device = std::atoi(val());
    else if (a == "--lbl-cache") o.lbl_cache = val();
    else if (a == "--restart") o.restart = val();
    else if (a == "--checkpoint") o.checkpoint = val();
    else if (a == "--equilibrium") o.equilibrium = std::atof(val());
    else if (a == "--gather-out") o.gather_out = val();
    else if (a == "--devices") {
      for (const char *p = val(); *p;) {
        char *e = nullptr;
        o.devices.push_back((int)std::strtol(p, &e, 10));
        if (e == p) break;
        p = (*e == ',') ? e + 1 : e;
      }
    }
    else if (a == "--print-cols" || a == "--print-col") {
      o.print_cols.clear();
      for (const char *p = val(); *p;) {
        char *e = nullptr;
        o.print_cols.push_back((int)std::strtol(p, &e, 10));
        if (e == p) break;
        p = (*e == ',') ? e + 1 : e;
      }
    }
    else {
      std::fprintf(stderr, "rcm_main: unknown option %s\n", a.c_str());
      std::exit(2);
    }
  }
  bool ok = o.ncol >= 1 && o.steps >= 0 && o.output_steps >= 1 && !o.print_cols.empty();
  for (int c : o.print_cols) ok = ok && c >= 0 && c < o.ncol;
  for (int dv : o.devices) ok = ok && dv >= 0;
  if (!ok) {
    std::fprintf(stderr, "rcm_main: bad option value\n");
    std::exit(2);
  }
  return o;
}

// lbl.{co2,h2o,o3,ch4,n2o}.asc (lbl.arts/README:5-9) -> wavelength nodes + 5 tables [n][20].
void load_lbl(const std::string &dir, int nlayer, std::vector<double> &wvl, std::vector<std::vector<double>> &tabs) {
  const char *names[RCM_NSPEC_LBL] = {"co2", "h2o", "o3", "ch4", "n2o"};
  tabs.assign(RCM_NSPEC_LBL, {});
  for (int s = 0; s < RCM_NSPEC_LBL; ++s) {
    const std::string path = dir + "/lbl." + names[s] + ".asc";
    int nx = 0, ny = 0;
    double *x = nullptr, *y = nullptr;
    const int st = rcm_ascii_file2xy2D(path.c_str(), &nx, &ny, &x, &y);
    if (st != 0) {  // testlblarts.cpp:28-31
      std::fprintf(stderr, "Error %d reading %s\n", st, path.c_str());
      std::exit(1);
    }
    std::fprintf(stderr, " ... read %d wavelengths and %d layers from %s\n", nx, ny, path.c_str());  // testlblarts.cpp:33
    if (ny != nlayer || nx < 2) {
      std::fprintf(stderr, "rcm_main: %s: expected %d layer columns and >= 2 rows\n", path.c_str(), nlayer);
      std::exit(1);
    }
    const bool descending = x[0] > x[nx - 1];  // ARTS writes frequency-ordered grids; cplkavg needs lambda ascending
    std::vector<double> w(nx), t((size_t)nx * ny);
    for (int r = 0; r < nx; ++r) {
      const int src = descending ? nx - 1 - r : r;
      w[r] = x[src];
      std::memcpy(&t[(size_t)r * ny], &y[(size_t)src * ny], ny * sizeof(double));
    }
    rcm_free(x);
    rcm_free(y);
    if (s == 0) wvl = w;
    else if (w != wvl) {
      std::fprintf(stderr, "rcm_main: %s: wavelength grid differs from lbl.co2.asc\n", path.c_str());
      std::exit(1);
    }
    tabs[s] = std::move(t);
  }
}


struct Shard {
  rcm_ensemble *h = nullptr;
  int device = 0, lo = 0, hi = 0;  // columns [lo, hi) of the ensemble
};

// (Tlayer[0..nlayer-1], T_surface) of every column, [ncol][nlev], from all handles.
// Distinct GPUs: rcm_pack_state on each, then ONE ncclAllGather of equal (zero-padded) blocks over
// xGMI -- every GPU ends with the whole ensemble, GPU 0's copy comes back.  Otherwise: through the host.
const char *gather(std::vector<Shard> &shards, int ncol, int nlev, std::vector<double> &full) {
  full.assign((size_t)ncol * nlev, 0.0);
  const int n = (int)shards.size();
  bool distinct = true;
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b) distinct = distinct && shards[a].device != shards[b].device;
#ifdef RCM_WITH_RCCL
  if (distinct) {
    auto hipck = [](hipError_t e, const char *what) {
      if (e != hipSuccess) {
        std::fprintf(stderr, "rcm_main: %s: %s\n", what, hipGetErrorString(e));
        std::exit(1);
      }
    };
    auto ncclck = [](ncclResult_t e, const char *what) {
      if (e != ncclSuccess) {
        std::fprintf(stderr, "rcm_main: %s: %s\n", what, ncclGetErrorString(e));
        std::exit(1);
      }
    };
    int per = 0;
    for (const Shard &sh : shards) per = std::max(per, sh.hi - sh.lo);
    const size_t blk = (size_t)per * nlev;
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n);
    std::vector<double *> recv(n);
    for (int r = 0; r < n; ++r) devs[r] = shards[r].device;
    ncclck(ncclCommInitAll(comms.data(), n, devs.data()), "ncclCommInitAll");
    for (int r = 0; r < n; ++r) {
      hipck(hipSetDevice(devs[r]), "hipSetDevice");
      hipck(hipMalloc((void **)&recv[r], blk * n * sizeof(double)), "hipMalloc");
    }
    // rcm_gather = pack + zero padding + ncclAllGather on the handle's own stream (include/rcm.h)
    ncclck(ncclGroupStart(), "ncclGroupStart");
    for (int r = 0; r < n; ++r) CK(rcm_gather(shards[r].h, comms[r], recv[r], per));
    ncclck(ncclGroupEnd(), "ncclGroupEnd");
    for (int r = 0; r < n; ++r) CK(rcm_sync(shards[r].h));
    std::vector<double> all(blk * n);
    hipck(hipSetDevice(devs[0]), "hipSetDevice");
    hipck(hipMemcpy(all.data(), recv[0], all.size() * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
    for (int r = 0; r < n; ++r) {
      std::copy(all.begin() + r * blk, all.begin() + r * blk + (size_t)(shards[r].hi - shards[r].lo) * nlev,
                full.begin() + (size_t)shards[r].lo * nlev);
      hipck(hipSetDevice(devs[r]), "hipSetDevice");
      (void)hipFree(recv[r]);
      ncclCommDestroy(comms[r]);
    }
    return "RCCL (one all-gather)";
  }
#endif
  (void)distinct;
  const int nl = nlev - 1;
  std::vector<double> T, Ts;
  for (Shard &sh : shards) {
    const int m = sh.hi - sh.lo;
    T.resize((size_t)m * nl), Ts.resize(m);
    CK(rcm_get_state(sh.h, T.data(), Ts.data(), nullptr, nullptr));
    for (int c = 0; c < m; ++c) {
      double *row = &full[(size_t)(sh.lo + c) * nlev];
      std::copy(T.begin() + (size_t)c * nl, T.begin() + (size_t)(c + 1) * nl, row);
      row[nl] = Ts[c];
    }
  }
  return "the host (no RCCL in this build, or a GPU listed twice)";
}

}  // namespace

int main(int argc, char **argv) {
  const Options o = parse(argc, argv);
  const int nlayer = 20, nlevel = 21, nangle = 30;  // main.cpp:78-80

  // ---- profiles: main.cpp:396-485 ------------------------------------------------------------
  std::vector<double> atm(9 * (size_t)nlevel);
  int nlev_read = 0, ncols_found = 0;
  CK(rcm_atm_load(o.atm.c_str(), nlevel, atm.data(), &nlev_read, &ncols_found));
  if (nlev_read != nlevel) {
    std::fprintf(stderr, "rcm_main: %s has %d levels, need %d\n", o.atm.c_str(), nlev_read, nlevel);
    return 1;
  }
  const bool lbl = !o.lbl_dir.empty() || !o.lbl_cache.empty();
  const int ngas = lbl ? RCM_NSPEC_LBL : RCM_NSPEC_REPWVL;
  std::vector<double> plevel(nlevel), player(nlayer), Tlayer(nlayer), conv(nlayer), vmr9(9 * (size_t)nlayer),
      rel(nlayer), h2o_ref(nlayer);
  CK(rcm_column_init(nlevel, atm.data(), 1.0, 1.0, plevel.data(), player.data(), Tlayer.data(), conv.data(),
                     vmr9.data(), rel.data()));
  std::copy(vmr9.begin(), vmr9.begin() + nlayer, h2o_ref.begin());

  std::vector<double> T0((size_t)o.ncol * nlayer), Ts0(o.ncol, o.tsfc), gas((size_t)o.ncol * ngas * nlayer),
      relhum((size_t)o.ncol * nlayer);
  for (int c = 0; c < o.ncol; ++c) {
    // one column: the reference's factor; an ensemble: factor * linspace(0.5, 4.0)
    const double f = o.ncol == 1 ? o.co2_factor : o.co2_factor * (0.5 + 3.5 * c / (double)(o.ncol - 1));
    CK(rcm_column_init(nlevel, atm.data(), f, 1.0, plevel.data(), player.data(), Tlayer.data(), conv.data(),
                       vmr9.data(), rel.data()));
    std::copy(Tlayer.begin(), Tlayer.end(), T0.begin() + (size_t)c * nlayer);
    std::copy(rel.begin(), rel.end(), relhum.begin() + (size_t)c * nlayer);
    double *g = &gas[(size_t)c * ngas * nlayer];
    if (lbl) {
      for (int i = 0; i < ngas * nlayer; ++i) g[i] = 1.0;
      for (int l = 0; l < nlayer; ++l) g[l] = f;  // CO2 table multiplier
    } else {
      std::copy(vmr9.begin(), vmr9.end(), g);
    }
  }

  // ---- ensemble + optical-depth tables: main.cpp:494-504 -------------------------------------
  // one handle per device entry, contiguous blocks of ceil(ncol/n) columns (empty blocks are dropped)
  std::vector<Shard> shards;
  {
    const std::vector<int> devs = o.devices.empty() ? std::vector<int>{o.device} : o.devices;
    const int per = (o.ncol + (int)devs.size() - 1) / (int)devs.size();
    for (size_t r = 0; r < devs.size(); ++r) {
      Shard sh;
      sh.device = devs[r];
      sh.lo = std::min((int)r * per, o.ncol);
      sh.hi = std::min(sh.lo + per, o.ncol);
      if (sh.hi > sh.lo) shards.push_back(sh);
    }
  }
  for (Shard &sh : shards) {
    CK(rcm_create(sh.hi - sh.lo, nlayer, nangle, sh.device, nullptr, &sh.h));
    CK(rcm_set_grid(sh.h, plevel.data()));
  }
  auto shard_of = [&](int c) -> Shard & {
    for (Shard &sh : shards)
      if (c >= sh.lo && c < sh.hi) return sh;
    return shards[0];
  };
  if (lbl) {
    std::vector<double> wvl;
    std::vector<std::vector<double>> tabs;
    const double *tp[RCM_NSPEC_LBL];
    int n_nodes = 0, nl_file = 0;
    double *cw = nullptr, *ct = nullptr;
    FILE *probe = o.lbl_cache.empty() ? nullptr : std::fopen(o.lbl_cache.c_str(), "rb");
    if (probe) {  // parse the ASCII spectra once, then reuse the binary cache
      std::fclose(probe);
      CK(rcm_lbl_cache_read(o.lbl_cache.c_str(), &n_nodes, &nl_file, &cw, &ct));
      if (nl_file != nlayer) {
        std::fprintf(stderr, "rcm_main: %s has %d layers, need %d\n", o.lbl_cache.c_str(), nl_file, nlayer);
        return 1;
      }
      for (int s = 0; s < RCM_NSPEC_LBL; ++s) tp[s] = ct + (size_t)s * n_nodes * nlayer;
      for (Shard &sh : shards) CK(rcm_set_lbl_tables(sh.h, n_nodes, cw, tp, h2o_ref.data()));
      rcm_free(cw);
      rcm_free(ct);
    } else {
      if (o.lbl_dir.empty()) {
        std::fprintf(stderr, "rcm_main: %s does not exist and no --lbl-dir was given to build it from\n", o.lbl_cache.c_str());
        return 1;
      }
      load_lbl(o.lbl_dir, nlayer, wvl, tabs);
      for (int s = 0; s < RCM_NSPEC_LBL; ++s) tp[s] = tabs[s].data();
      if (!o.lbl_cache.empty()) CK(rcm_lbl_cache_write(o.lbl_cache.c_str(), (int)wvl.size(), nlayer, wvl.data(), tp));
      for (Shard &sh : shards) CK(rcm_set_lbl_tables(sh.h, (int)wvl.size(), wvl.data(), tp, h2o_ref.data()));
    }
  } else {
    long long dims[4];
    double *blob = nullptr;
    CK(rcm_rtab_load(o.table.c_str(), dims, &blob));
    const size_t nt = dims[0], ns = dims[1], nw = dims[2], np = dims[3];
    const double *p_grid = blob, *t_ref = p_grid + np, *t_pert = t_ref + np, *vmrs_ref = t_pert + nt,
                 *wvl = vmrs_ref + ns * np, *weight = wvl + nw, *xsec = weight + nw;
    for (Shard &sh : shards)
      CK(rcm_set_repwvl_tables(sh.h, (int)nt, (int)ns, (int)nw, (int)np, p_grid, t_ref, t_pert, wvl, weight, xsec));
    rcm_free(blob);
  }

  // ---- solar set-up + banner: main.cpp:512-519, 528 -------------------------------------------
  double solar[7];
  const double mu_s = std::cos(60 * M_PI / 180.0);  // main.cpp:87
  CK(rcm_solar_setup(o.tau_s, mu_s, 20, 0.85, 0.12, (double)0.5f, 1361.0, solar));
  auto shard_file = [&](const std::string &f, size_t r) { return shards.size() == 1 ? f : f + "." + std::to_string(r); };
  for (size_t r = 0; r < shards.size(); ++r) {
    Shard &sh = shards[r];
    CK(rcm_set_forcing(sh.h, solar[6], 17, o.tau_s / 2.0));  // main.cpp:92, 267
    CK(rcm_set_columns(sh.h, &T0[(size_t)sh.lo * nlayer], &Ts0[sh.lo], &gas[(size_t)sh.lo * ngas * nlayer],
                       &relhum[(size_t)sh.lo * nlayer]));
    if (!o.restart.empty()) CK(rcm_checkpoint_load(sh.h, shard_file(o.restart, r).c_str()));
  }
  FILE *out = std::fopen(o.output.c_str(), "a");  // main.cpp:512 (append)
  if (!out) {
    std::fprintf(stderr, "rcm_main: cannot open %s\n", o.output.c_str());
    return 1;
  }
  CK(rcm_write_banner(out, o.tau_s, mu_s, solar));

  // ---- model run: main.cpp:531-583 ------------------------------------------------------------
  std::vector<double> Tadj(nlayer), theta(nlayer);
  float time_h = 0.f;
  const auto t0 = std::chrono::steady_clock::now();
  auto write_blocks = [&]() {  // main.cpp:543-546 for every selected column
    for (int c : o.print_cols) {
      Shard &sh = shard_of(c);
      CK(rcm_get_profile(sh.h, c - sh.lo, Tadj.data(), theta.data(), &time_h));
      CK(rcm_write_profile(out, nlayer, player.data(), Tadj.data(), theta.data(), time_h));
    }
  };
  int i = 0;
  double resid = -1;
  while (i <= o.steps) {
    if (i % o.output_steps == 0) write_blocks();  // before this iteration's radiative transfer
    const int next_out = (i / o.output_steps + 1) * o.output_steps;
    const int run = std::min(next_out, o.steps + 1) - i;  // iterations until the next output or the end
    for (Shard &sh : shards) CK(rcm_step(sh.h, run));  // only enqueues: the GPUs run side by side
    i += run;
    if (o.equilibrium >= 0) {
      // the criterion of rcm_run_to_equilibrium on the WHOLE ensemble: worst column of all handles
      resid = 0;
      std::vector<double> r;
      for (Shard &sh : shards) {
        r.resize(sh.hi - sh.lo);
        CK(rcm_get_residual(sh.h, r.data()));
        for (double v : r) resid = (v > resid || v != v) ? v : resid;  // a NaN column is never "converged"
      }
      if (resid <= o.equilibrium) {
        write_blocks();  // the equilibrium profile
        break;
      }
    }
  }
  const int iterations = std::min(i, o.steps + 1);
  for (Shard &sh : shards) CK(rcm_sync(sh.h));
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::fclose(out);

  for (size_t r = 0; r < shards.size(); ++r)
    if (!o.checkpoint.empty()) CK(rcm_checkpoint_save(shards[r].h, shard_file(o.checkpoint, r).c_str()));
  std::vector<double> Tsfc(o.ncol);
  std::vector<float> tm(o.ncol);
  for (Shard &sh : shards) CK(rcm_get_state(sh.h, nullptr, &Tsfc[sh.lo], &tm[sh.lo], nullptr));
  std::fprintf(stderr, "rcm_main: %d column(s) x %d iterations (%d spectral points) on %zu handle(s) in %.3f s = %.1f column-timesteps/s\n",
               o.ncol, iterations, rcm_nspectral(shards[0].h), shards.size(), secs, o.ncol * (double)iterations / secs);
  if (o.equilibrium >= 0)
    std::fprintf(stderr, "rcm_main: equilibrium residual %.3e K/step (tolerance %.3e)\n", resid, o.equilibrium);
  for (int c : o.print_cols)
    std::fprintf(stderr, "rcm_main: column %d: T_surface %.10f K, time %.6f h\n", c, Tsfc[c], tm[c]);
  if (!o.devices.empty() || !o.gather_out.empty()) {
    std::vector<double> full;
    const char *how = gather(shards, o.ncol, nlayer + 1, full);
    std::fprintf(stderr, "rcm_main: gathered %d x %d state values over %s\n", o.ncol, nlayer + 1, how);
    for (int c : o.print_cols)
      if (full[(size_t)c * (nlayer + 1) + nlayer] != Tsfc[c]) {
        std::fprintf(stderr, "rcm_main: gathered T_surface of column %d differs from the handle's\n", c);
        return 1;
      }
    if (!o.gather_out.empty()) {
      FILE *g = std::fopen(o.gather_out.c_str(), "w");
      if (!g) {
        std::fprintf(stderr, "rcm_main: cannot open %s\n", o.gather_out.c_str());
        return 1;
      }
      for (int c = 0; c < o.ncol; ++c) {
        const double *row = &full[(size_t)c * (nlayer + 1)];
        std::fprintf(g, "%d,%.17g", c, row[nlayer]);
        for (int l = 0; l < nlayer; ++l) std::fprintf(g, ",%.17g", row[l]);
        std::fputc('\n', g);
      }
      std::fclose(g);
    }
  }
  for (Shard &sh : shards) CK(rcm_destroy(sh.h));
  return 0;
}
