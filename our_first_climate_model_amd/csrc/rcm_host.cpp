// rcm_host.cpp -- host-side pieces of the reference driver (no device code here).
//
// Mirrors, with the reference's own semantics and error behaviour:
//   * the atmosphere loader             main.cpp:396-430
//   * the per-column initialisation     main.cpp:439-485
//   * the solar set-up                  main.cpp:214-264
//   * the lbl.arts ASCII matrix loader  lbl.arts/ascii.cpp:225-296,1296-1337,1391-1422,1631-1691
//   * the output.txt formats            main.cpp:102-114, 513-519, 528
// plus the reader of the flattened repwvl table blob (tools/flatten_repwvl.py).
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "rcm.h"
#include "rcm_internal.hpp"

namespace rcm {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

}  // namespace rcm

using rcm::fail;

extern "C" {

int rcm_abi_version(void) { return RCM_ABI_VERSION; }

const char *rcm_last_error(void) { return rcm::g_last_error.c_str(); }

void rcm_free(void *p) { std::free(p); }

// ---------------------------------------------------------------------------------------------
// atmosphere file: main.cpp:396-430.  `ss >> value` semantics: a row contributes its leading
// run of parseable numbers, one per column; rows with none are skipped by construction.
// ---------------------------------------------------------------------------------------------
int rcm_atm_load(const char *path, int nlevel_max, double *cols, int *nlevel_out,
                 int *ncols_found) try {
  if (!path || !cols || nlevel_max < 2) return fail(RCM_ERR_ARG, "rcm_atm_load: bad argument");
  std::ifstream in(path);
  if (!in) return fail(RCM_ERR_IO, std::string("rcm_atm_load: cannot open ") + path);
  std::string line;
  for (int i = 0; i < 4; ++i) std::getline(in, line);  // main.cpp:402-404
  int nrow = 0, maxc = 0;
  while (std::getline(in, line)) {  // main.cpp:410-417
    std::stringstream ss(line);
    double value;
    int c = 0;
    while (ss >> value) {
      if (c < 9) {
        if (nrow >= nlevel_max) return fail(RCM_ERR_FORMAT, "rcm_atm_load: more levels than nlevel_max");
        cols[(size_t)c * nlevel_max + nrow] = value;
      }
      ++c;
    }
    if (c > 0) {
      if (c > maxc) maxc = c;
      ++nrow;
    }
  }
  if (nrow < 2) return fail(RCM_ERR_FORMAT, std::string("rcm_atm_load: no profile rows in ") + path);
  if (maxc < 6) return fail(RCM_ERR_FORMAT, "rcm_atm_load: need at least z p T air H2O O3");
  if (maxc < 9) {  // lbl.arts/fpda.lbl.atm: well-mixed gases from lbl.arts/README:14-17
    const double dflt[3] = {400.0, 1.7, 0.315};
    for (int c = maxc; c < 9; ++c)
      for (int r = 0; r < nrow; ++r) cols[(size_t)c * nlevel_max + r] = dflt[c - 6];
  }
  if (nlevel_out) *nlevel_out = nrow;
  if (ncols_found) *ncols_found = maxc;
  return RCM_OK;
} RCM_ABI_CATCH

// ---------------------------------------------------------------------------------------------
// Column initialisation: main.cpp:439-485.
// ---------------------------------------------------------------------------------------------
static double magnus_hPa(double T) {  // main.cpp:277-279
  return 6.1094 * std::exp(17.625 * (T - 273.15) / (T - 273.15 + 243.04));
}

int rcm_column_init(int nlevel, const double *a, double co2_factor, double rh_scale,
                    double *plevel, double *player, double *Tlayer, double *conv, double *vmr9,
                    double *rel_hum) try {
  if (nlevel < 2 || !a || !plevel || !player || !Tlayer || !conv || !vmr9 || !rel_hum)
    return fail(RCM_ERR_ARG, "rcm_column_init: bad argument");
  const int nl = nlevel - 1;
  const double kappa = 2.0 / 7.0;  // main.cpp:66
  const double *p_lev = a + 1 * (size_t)nlevel, *T_lev = a + 2 * (size_t)nlevel;
  // file column -> read_tau slot: H2O(4)->0, O3(5)->2, CO2(6)->1, CH4(7)->5, N2O(8)->3
  const int file_col[5] = {4, 5, 6, 7, 8};
  const int slot[5] = {0, 2, 1, 5, 3};
  for (int s = 0; s < RCM_NSPEC_REPWVL; ++s)
    for (int i = 0; i < nl; ++i) vmr9[(size_t)s * nl + i] = 0.0;  // main.cpp:459: CO, O2, HNO3, N2 = 0
  for (int k = 0; k < 5; ++k) {
    const double *lev = a + (size_t)file_col[k] * nlevel;
    double *lay = vmr9 + (size_t)slot[k] * nl;
    const double to_abs = (slot[k] == 1) ? co2_factor * 1E-6 : 1E-6;  // main.cpp:456
    for (int i = 0; i < nl; ++i) {
      lay[i] = (lev[i] + lev[i + 1]) / 2.0;  // main.cpp:144
      lay[i] *= to_abs;
    }
  }
  for (int i = 0; i < nl; ++i) vmr9[i] *= rh_scale;  // ensemble perturbation; 1 = reference
  for (int i = 0; i < nlevel; ++i) plevel[i] = p_lev[i];
  for (int i = 0; i < nl; ++i) {
    // main.cpp:467-468 pairs the LEVEL pressure and temperature with the LAYER mixing ratio
    rel_hum[i] = vmr9[i] * p_lev[i] / magnus_hPa(T_lev[i]);
    player[i] = (p_lev[i] + p_lev[i + 1]) / 2.0;    // main.cpp:472
    Tlayer[i] = (T_lev[i] + T_lev[i + 1]) / 2.0;    // main.cpp:473
    conv[i] = std::pow(1000.0 / player[i], kappa);  // main.cpp:474
  }
  return RCM_OK;
} RCM_ABI_CATCH

// ---------------------------------------------------------------------------------------------
// Solar set-up: main.cpp:214-264 (runs once; O(doublings) flops; stays on the host).
// ---------------------------------------------------------------------------------------------
int rcm_solar_setup(double tau_s, double mu_s, int doublings, double g_asym, double albedo,
                    double daytime, double E_0, double out[7]) try {
  if (!out || doublings < 0 || !(mu_s > 0)) return fail(RCM_ERR_ARG, "rcm_solar_setup: bad argument");
  // thin starting layer, main.cpp:216-227
  double dtau = (1 - g_asym) * tau_s / std::pow(2, doublings);
  double r = 0.5 * dtau / mu_s, t = 1.0 - r;
  double r_dir = dtau / mu_s * 0.5, s_dir = r_dir, t_dir = 1 - dtau / mu_s;
  for (int i = 0; i < doublings; ++i) {  // main.cpp:232-250
    const double d = 1 - r * r;
    const double r2 = r + (r * t * t) / d;
    const double t2 = (t * t) / d;
    const double td2 = std::pow(t_dir, 2);
    const double sd2 = (t * s_dir + t_dir * r_dir * r * t) / d + t_dir * s_dir;
    const double rd2 = (t * s_dir * r + t * t_dir * r) / d + r_dir;
    r = r2, t = t2, t_dir = td2, s_dir = sd2, r_dir = rd2;
  }
  const double r_total = r_dir + (t_dir + s_dir) / (1 - albedo * r) * t * albedo;  // main.cpp:258
  out[0] = r_dir, out[1] = s_dir, out[2] = t_dir, out[3] = r, out[4] = t, out[5] = r_total;
  out[6] = daytime * E_0 * mu_s * (1 - r_total);  // main.cpp:260
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_default_solar(double out[7]) try {  // main.cpp:69, 86-91
  return rcm_solar_setup(2.0, std::cos(60 * M_PI / 180.0), 20, 0.85, 0.12, (double)0.5f, 1361.0, out);
} RCM_ABI_CATCH

// ---------------------------------------------------------------------------------------------
// ASCII matrix loader with ASCII_file2xy2D semantics (lbl.arts/ascii.cpp:1631-1691):
//  * tokens are separated by blank, tab, newline (strtok " \t\n", ascii.cpp:257,268);
//  * a line whose first token starts with '%' or '#' is skipped; a later token starting with
//    one ends the row (ascii.cpp:259,274, 2686-2692); empty lines are skipped;
//  * every row must have the same token count, else NOT_A_RECTANGULAR_MATRIX = -5
//    (ascii.cpp:1650-1652; an empty file lands here too: min INT_MAX != max 0);
//  * tokens go through strtod: a non-number reads as 0 (ascii.cpp:1405-1417).
// One pass over the file instead of the reference's count/allocate-strings/re-read.
// ---------------------------------------------------------------------------------------------
int rcm_ascii_file2xy2D(const char *path, int *nx, int *ny, double **x, double **y) try {
  if (!path || !nx || !ny || !x || !y) return fail(RCM_ERR_ARG, "rcm_ascii_file2xy2D: bad argument");
  FILE *f = std::fopen(path, "r");
  if (!f) {
    std::fprintf(stderr, "Error, file '%s' not found!\n", path);  // ascii.cpp:247
    return fail(-1, std::string("file not found: ") + path);      // ASCIIFILE_NOT_FOUND
  }
  struct Owner {  // closes the file and frees the line buffer on every way out, an exception included
    FILE *f;
    char *line = nullptr;
    ~Owner() {
      std::free(line);
      if (f) std::fclose(f);
    }
  } own{f};
  std::vector<double> vals;
  // the reference reads with fgets(line, MAX_LENGTH_OF_LINE, f) (ascii.cpp:155,233,255): a physical line of
  // 1 048 575 characters or more is cut there and continues as the next "row" (a number across the cut splits
  // in two); same buffer size and same call here so that such a file fails or parses exactly as it does there
  constexpr int kMaxLine = 1048576;
  char *line = own.line = (char *)std::malloc((size_t)kMaxLine + 1);
  if (!line) return fail(-2, "no memory");  // ASCII_NO_MEMORY
  int rows = 0, min_col = INT_MAX, max_col = 0;
  while (std::fgets(line, kMaxLine, f) != nullptr) {
    int ncol = 0;
    char *save = nullptr;
    for (char *tok = strtok_r(line, " \t\n", &save); tok; tok = strtok_r(nullptr, " \t\n", &save)) {
      if (tok[0] == '%' || tok[0] == '#') break;
      char *end = nullptr;
      vals.push_back(std::strtod(tok, &end));
      ++ncol;
    }
    if (ncol > 0) {
      ++rows;
      if (ncol < min_col) min_col = ncol;
      if (ncol > max_col) max_col = ncol;
    }
  }
  if (min_col != max_col) return fail(-5, "not a rectangular matrix");  // NOT_A_RECTANGULAR_MATRIX
  const int ncy = max_col - 1;
  double *xs = (double *)std::calloc((size_t)rows, sizeof(double));
  double *ys = (double *)std::calloc((size_t)rows * (ncy > 0 ? ncy : 1), sizeof(double));
  if (!xs || !ys) {
    std::free(xs);
    std::free(ys);
    return fail(-2, "no memory");  // ASCII_NO_MEMORY
  }
  for (int r = 0; r < rows; ++r) {
    xs[r] = vals[(size_t)r * max_col];
    for (int c = 0; c < ncy; ++c) ys[(size_t)r * ncy + c] = vals[(size_t)r * max_col + 1 + c];
  }
  *nx = rows;
  *ny = ncy;
  *x = xs;
  *y = ys;
  return 0;
} catch (const std::bad_alloc &) {
  return fail(-2, "no memory");  // ASCII_NO_MEMORY, the reference's own code (no exception may cross the C ABI)
} catch (const std::exception &) {
  return fail(-2, "no memory");  // std::length_error of the value vector: the same condition under another name
}

// Bytes left in f behind the current position (the readers below size their allocations from header fields of an
// untrusted file: a header that promises more than the file holds is refused BEFORE anything is allocated).
static long long bytes_left(FILE *f) {
  const long here = std::ftell(f);
  if (here < 0 || std::fseek(f, 0, SEEK_END) != 0) return -1;
  const long end = std::ftell(f);
  if (end < 0 || std::fseek(f, here, SEEK_SET) != 0) return -1;
  return (long long)end - here;
}

// ---------------------------------------------------------------------------------------------
// repwvl table: the .rtab blob (tools/flatten_repwvl.py) or the .nc file itself (rcm_nc.cpp).
// ---------------------------------------------------------------------------------------------
int rcm_rtab_load(const char *path, long long dims[4], double **blob) try {
  if (!path || !dims || !blob) return fail(RCM_ERR_ARG, "rcm_rtab_load: bad argument");
  rcm::File f(std::fopen(path, "rb"));  // closed on every way out, a throwing error message included
  if (!f) return fail(RCM_ERR_IO, std::string("rcm_rtab_load: cannot open ") + path);
  char magic[8];
  long long d[4];
  if (std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "\x89HDF\r\n\x1a\n", 8) == 0) {
    f.close();  // a Reduced*Forcing.nc itself (NetCDF-4 = HDF5), as the reference opens it
    return rcm::nc_table_load(path, dims, blob);
  }
  std::rewind(f);
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "RWVLTAB1", 8) != 0 ||
      std::fread(d, sizeof(long long), 4, f) != 4 || d[0] < 2 || d[1] < 1 || d[2] < 1 || d[3] < 2 ||
      d[0] > 4096 || d[1] > 4096 || d[2] > (1 << 24) || d[3] > 4096)
    return fail(RCM_ERR_FORMAT, std::string("rcm_rtab_load: not an RWVLTAB1 blob: ") + path);
  const size_t nt = d[0], ns = d[1], nw = d[2], np = d[3];
  const size_t n = np + np + nt + ns * np + nw + nw + nt * ns * nw * np;  // (< 2^61 with the limits above: no wrap)
  const long long have = bytes_left(f);
  if (have < 0 || (unsigned long long)have / sizeof(double) < n)
    return fail(RCM_ERR_FORMAT, std::string("rcm_rtab_load: truncated blob: ") + path);
  std::unique_ptr<double, decltype(&std::free)> buf((double *)std::malloc(n * sizeof(double)), &std::free);
  if (!buf) return fail(RCM_ERR_NOMEM, "rcm_rtab_load: out of memory");
  if (std::fread(buf.get(), sizeof(double), n, f) != n)
    return fail(RCM_ERR_FORMAT, std::string("rcm_rtab_load: truncated blob: ") + path);
  double *b = buf.release();
  for (int i = 0; i < 4; ++i) dims[i] = d[i];
  *blob = b;
  return RCM_OK;
} RCM_ABI_CATCH

// ---------------------------------------------------------------------------------------------
// Binary cache of the lbl tables: "LBLTAB1\0", int64 n_nodes, int64 nlayer, wvl[n_nodes],
// tab[5][nlayer][n_nodes] (node fastest).
// ---------------------------------------------------------------------------------------------
int rcm_lbl_cache_write(const char *path, int n_nodes, int nlayer, const double *wvl,
                        const double *const tabs[RCM_NSPEC_LBL]) try {
  if (!path || !wvl || !tabs || n_nodes < 2 || nlayer < 1) return fail(RCM_ERR_ARG, "rcm_lbl_cache_write: bad argument");
  for (int s = 0; s < RCM_NSPEC_LBL; ++s)
    if (!tabs[s]) return fail(RCM_ERR_ARG, "rcm_lbl_cache_write: NULL table");
  rcm::File f(std::fopen(path, "wb"));
  if (!f) return fail(RCM_ERR_IO, std::string("rcm_lbl_cache_write: cannot open ") + path);
  const long long dims[2] = {n_nodes, nlayer};
  bool ok = std::fwrite("LBLTAB1", 1, 8, f) == 8 && std::fwrite(dims, sizeof(long long), 2, f) == 2 &&
            std::fwrite(wvl, sizeof(double), (size_t)n_nodes, f) == (size_t)n_nodes;
  std::vector<double> row((size_t)n_nodes);
  for (int s = 0; ok && s < RCM_NSPEC_LBL; ++s)
    for (int l = 0; ok && l < nlayer; ++l) {
      for (int b = 0; b < n_nodes; ++b) row[b] = tabs[s][(size_t)b * nlayer + l];
      ok = std::fwrite(row.data(), sizeof(double), (size_t)n_nodes, f) == (size_t)n_nodes;
    }
  ok = (f.close() == 0) && ok;
  return ok ? RCM_OK : fail(RCM_ERR_IO, std::string("rcm_lbl_cache_write: write failed: ") + path);
} RCM_ABI_CATCH

int rcm_lbl_cache_read(const char *path, int *n_nodes, int *nlayer, double **wvl, double **tabs) try {
  if (!path || !n_nodes || !nlayer || !wvl || !tabs) return fail(RCM_ERR_ARG, "rcm_lbl_cache_read: bad argument");
  rcm::File f(std::fopen(path, "rb"));
  if (!f) return fail(RCM_ERR_IO, std::string("rcm_lbl_cache_read: cannot open ") + path);
  char magic[8];
  long long dims[2];
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "LBLTAB1", 8) != 0 ||
      std::fread(dims, sizeof(long long), 2, f) != 2 || dims[0] < 2 || dims[0] > (1LL << 28) || dims[1] < 1 ||
      dims[1] > 4096)
    return fail(RCM_ERR_FORMAT, std::string("rcm_lbl_cache_read: not an LBLTAB1 file: ") + path);
  const size_t n = (size_t)dims[0], nl = (size_t)dims[1];
  const long long have = bytes_left(f);
  if (have < 0 || (unsigned long long)have / sizeof(double) < n + (size_t)RCM_NSPEC_LBL * n * nl)
    return fail(RCM_ERR_FORMAT, std::string("rcm_lbl_cache_read: truncated file: ") + path);
  using Buf = std::unique_ptr<double, decltype(&std::free)>;  // freed on every way out; released to the caller at the end
  Buf w((double *)std::malloc(n * sizeof(double)), &std::free);
  Buf t((double *)std::malloc((size_t)RCM_NSPEC_LBL * n * nl * sizeof(double)), &std::free);
  // one species' block [nlayer][n] at a time: nlayer sequential read streams, ONE sequential write stream (a row at a
  // time scattered 8 bytes every nlayer*8 and took 3.2 s for 131 072 nodes; this takes 0.2 s)
  Buf blk((double *)std::malloc(n * nl * sizeof(double)), &std::free);
  if (!w || !t || !blk) return fail(RCM_ERR_NOMEM, std::string("rcm_lbl_cache_read: out of memory: ") + path);
  bool ok = std::fread(w.get(), sizeof(double), n, f) == n;
  for (size_t s = 0; ok && s < RCM_NSPEC_LBL; ++s) {
    ok = std::fread(blk.get(), sizeof(double), n * nl, f) == n * nl;
    if (!ok) break;
    const double *src = blk.get();
    double *dst = t.get() + s * n * nl;
    for (size_t b = 0; b < n; ++b)
      for (size_t l = 0; l < nl; ++l) dst[b * nl + l] = src[l * n + b];
  }
  if (!ok) return fail(RCM_ERR_FORMAT, std::string("rcm_lbl_cache_read: truncated file: ") + path);
  *n_nodes = (int)n;
  *nlayer = (int)nl;
  *wvl = w.release();
  *tabs = t.release();
  return RCM_OK;
} RCM_ABI_CATCH

// ---------------------------------------------------------------------------------------------
// output.txt formats: main.cpp:513-519, 528 (banner) and :109-110 (rows).
// ---------------------------------------------------------------------------------------------
int rcm_write_banner(void *fp, double tau_s, double mu_s, const double solar[7]) try {
  FILE *f = (FILE *)fp;
  if (!f || !solar) return fail(RCM_ERR_ARG, "rcm_write_banner: bad argument");
  std::fprintf(f, "\n ===================new run===================== \n");
  std::fprintf(f, "Planetary albedo for an optical thickness of %2.3f and an cos(SZA) of %2.2f: %f \n",
               tau_s, mu_s, solar[5]);
  std::fprintf(f, "rdir %f, sdir %f, and tdir %f \n", solar[0], solar[1], solar[2]);
  std::fprintf(f, "sum: %f \n", solar[0] + solar[2] + solar[1]);
  std::fprintf(f, "solar irradiance: %f \n", solar[6]);
  std::fprintf(f, "layer,player,Tlayer,theta,time\n");
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_write_profile(void *fp, int nlayer, const double *player, const double *Tlayer,
                      const double *theta, float time_h) try {
  FILE *f = (FILE *)fp;
  if (!f || !player || !Tlayer || !theta) return fail(RCM_ERR_ARG, "rcm_write_profile: bad argument");
  for (int i = 0; i < nlayer; ++i)
    std::fprintf(f, "%d,%f,%f,%f,%f\n", i, player[i], Tlayer[i], theta[i], time_h);
  return RCM_OK;
} RCM_ABI_CATCH

}  // extern "C"
