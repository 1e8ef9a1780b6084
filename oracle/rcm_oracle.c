/*
 * rcm_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see rcm_oracle.h).
 *
 * Every function restates one reference function and cites it.  Expression
 * order follows the reference token by token so that, built with
 * `gcc -O2 -ffp-contract=off` on x86-64, results are bit-identical to the
 * reference built with `g++ -O2` (no FMA contraction on baseline x86-64).
 *
 * The reference is /root/reference (pabloconrat/our_first_climate_model).
 *
 * PARITY: pinned bit for bit by the reference's own code compiled here (oracle/build_ref.sh ->
 * oracle/_ref/: main.cpp's body, read_tau's netCDF-free statements, LowerPos, cplkavg.cpp, ascii.cpp)
 * and by committed outputs of it (tests/golden/ref_main_vectors.npz) for every function that has a
 * reference counterpart; UNPINNED by construction: the lbl composition (orc_lbl_tau and the band
 * convention), for which the reference has no driver and no spectra.  Details: rcm_oracle.h,
 * oracle/README.md, DESIGN.md section 2.
 */
#define _GNU_SOURCE
#include "rcm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* constants: main.cpp:66-92                                           */
/* ------------------------------------------------------------------ */
static const double C_KAPPA = 2.0 / 7.0;
static const double C_CAIR = 1004;
static const double C_G = 9.80665;
static const double C_E0 = 1361.0;
static const double C_H = 6.62607e-34;
static const double C_C = 299792458;
static const double C_KB = 1.380649e-23;
static const double C_TKELVIN = 273.15;
static const float C_MAX_DT = 5;

/* ------------------------------------------------------------------ */
/* solar: main.cpp:214-264                                             */
/* ------------------------------------------------------------------ */
void orc_solar_setup(double tau_s, double mu_s, int doublings, double g_asym, double albedo,
                     double daytime, double E_0, double out[7]) {
  /* doubling_adding, main.cpp:214-253 */
  double tau = (1 - g_asym) * tau_s;
  double dtau = tau / pow(2, doublings);
  double r_new, one_minus_rsq, t_new, r_dir_new, s_dir_new, t_dir_new;
  double r = 0.5 * dtau / mu_s;
  double t = 1.0 - r;
  double r_dir = dtau / mu_s * 0.5;
  double s_dir = r_dir;
  double t_dir = 1 - dtau / mu_s;
  for (int i = 0; i < doublings; ++i) {
    one_minus_rsq = (1 - r * r);
    r_new = r + (r * t * t) / one_minus_rsq;
    t_new = (t * t) / one_minus_rsq;
    t_dir_new = pow(t_dir, 2);
    s_dir_new = (t * s_dir + t_dir * r_dir * r * t) / one_minus_rsq + t_dir * s_dir;
    r_dir_new = (t * s_dir * r + t * t_dir * r) / one_minus_rsq + r_dir;
    r = r_new;
    t = t_new;
    t_dir = t_dir_new;
    s_dir = s_dir_new;
    r_dir = r_dir_new;
    dtau = 2 * dtau;
  }
  /* solar_radiative_transfer_setup, main.cpp:255-264 */
  double r_total = r_dir + (t_dir + s_dir) / (1 - albedo * r) * t * albedo;
  double solar_irr = daytime * E_0 * mu_s * (1 - r_total);
  out[0] = r_dir;
  out[1] = s_dir;
  out[2] = t_dir;
  out[3] = r;
  out[4] = t;
  out[5] = r_total;
  out[6] = solar_irr;
}

void orc_default_solar(double out[7]) {
  /* main.cpp:86-91: tau_s 2.0, mu_s cos(60 deg), 20 doublings, g 0.85, albedo 0.12, daytime 0.5f */
  orc_solar_setup(2.0, cos(60 * M_PI / 180.0), 20, 0.85, 0.12, (double)0.5f, C_E0, out);
}

/* ------------------------------------------------------------------ */
/* Planck, monochromatic * weight: main.cpp:186-204                    */
/* ------------------------------------------------------------------ */
double orc_planck_mono(double wvl, double weight, double T) {
  wvl *= 1e-9;
  double radiance =
      weight * 2 * C_H * pow(C_C, 2) / (pow(wvl, 5) * (exp(C_H * C_C / (wvl * C_KB * T)) - 1)) / 1e9;
  return radiance;
}

/* ------------------------------------------------------------------ */
/* Band-integrated Planck: cplkavg.cpp:124-243                         */
/* ------------------------------------------------------------------ */
static int g_cplkavg_errors = 0;
int orc_cplkavg_errors(void) { return g_cplkavg_errors; }

static inline double plkf(double x) { return x * x * x / (exp(x) - 1.); } /* cplkavg.cpp:109-113 */

double orc_cplkavg(double wvllo, double wvlhi, double t) {
  const double A1 = (1. / 3.), A2 = (-1. / 8.), A3 = (1. / 60.), A4 = (-1. / 5040.),
               A5 = (1. / 272160.), A6 = (-1. / 13305600.);
  const double C2 = 1.438786, SIGMA = 5.67032E-8, VCUT = 1.5;
  const double vcp[7] = {10.25, 5.7, 3.9, 2.9, 2.3, 1.9, 0.0};
  int i, k, m, mmax, n, smallv, converged;
  double del, ex, exm, hh, mv, oldval, val, val0, vsq, d[2] = {0, 0}, p[2] = {0, 0}, v[2], ans;
  const double vmax = log(DBL_MAX);           /* :133 */
  const double sigdpi = SIGMA / M_PI;         /* :134 */
  const double conc = 15. / pow(M_PI, 4.);    /* :135 */

  double wnumhi = 1.0E7 / wvllo; /* :141 */
  double wnumlo = 1.0E7 / wvlhi; /* :142 */

  if (t < 0. || wnumhi <= wnumlo || wnumlo < 0.) { /* :144-146, reference exits */
    ++g_cplkavg_errors;
    return NAN;
  }
  if (t < 1.e-4) return 0.; /* :148 */

  v[0] = C2 * wnumlo / t;
  v[1] = C2 * wnumhi / t;

  if (v[0] > DBL_EPSILON && v[1] < vmax && (wnumhi - wnumlo) / wnumhi < 1.e-2) { /* :155 */
    hh = v[1] - v[0];
    oldval = 0.;
    val0 = plkf(v[0]) + plkf(v[1]);
    converged = 0;
    val = 0.;
    for (n = 1; n <= 10; n++) {
      del = hh / (2 * n);
      val = val0;
      for (k = 1; k <= 2 * n - 1; k++) {
        val += (double)(2 * (1 + k % 2)) * plkf(v[0] + (double)k * del);
      }
      val *= del * A1;
      if (fabs((val - oldval) / val) <= 1.e-6) {
        converged = 1;
        break;
      }
      oldval = val;
    }
    (void)converged; /* reference only warns (:177-179) */
    return sigdpi * pow(t, 4.0) * conc * val; /* :181 */
  }

  smallv = 0; /* :187 */
  for (i = 1; i <= 2; i++) {
    if (v[i - 1] < VCUT) {
      smallv++;
      vsq = (v[i - 1]) * (v[i - 1]);
      p[i - 1] = conc * vsq * v[i - 1] *
                 (A1 + v[i - 1] * (A2 + v[i - 1] * (A3 + vsq * (A4 + vsq * (A5 + vsq * A6)))));
    } else {
      mmax = 1;
      while (v[i - 1] < vcp[mmax - 1]) mmax++;
      ex = exp(-v[i - 1]);
      exm = 1.;
      d[i - 1] = 0.;
      for (m = 1; m <= mmax; m++) {
        mv = (double)m * v[i - 1];
        exm = ex * exm;
        d[i - 1] += exm * (6. + mv * (6. + mv * (3. + mv))) / (m * m * m * m);
      }
      d[i - 1] *= conc;
    }
  }
  if (smallv == 2)
    ans = p[1] - p[0];
  else if (smallv == 1)
    ans = 1. - p[0] - d[1];
  else
    ans = d[0] - d[1];
  ans *= sigdpi * pow(t, 4.0);
  return ans;
}

/* ------------------------------------------------------------------ */
/* Magnus: main.cpp:277-279                                            */
/* ------------------------------------------------------------------ */
double orc_magnus(double T) {
  return 6.1094 * exp(17.625 * (T - C_TKELVIN) / (T - C_TKELVIN + 243.04));
}

/* ------------------------------------------------------------------ */
/* repwvl: repwvl_thermal.cpp                                          */
/* ------------------------------------------------------------------ */
static inline int sgn_d(double v) { return (0.0 < v) - (v < 0.0); } /* :13-15 */

int orc_lower_pos(const double *grid, int n, double x) { /* :19-45 */
  int lowerPos = 0;
  int sign = sgn_d(grid[0] - x);
  for (int p = 1; p != n; ++p) {
    double delta = grid[p] - x;
    if (sign != sgn_d(delta)) {
      lowerPos = p - 1;
      return lowerPos;
    } else if (p == n - 1) {
      lowerPos = p - 1;
      return lowerPos;
    }
    sign = sgn_d(delta);
  }
  return lowerPos;
}

void orc_read_tau(int n_tpert, int n_species, int n_wvl, int n_p, const double *p_grid,
                  const double *t_ref, const double *t_pert, const double *xsec, int nlevel,
                  const double *plevel_hPa, const double *Tlayer, const double *vmr, double *tau) {
  const int nlyr = nlevel - 1;
  const double avog = 6.02214076e23, molMassAir = 0.0289647, earthAccel = 9.80665; /* :124-126 */
  double *cp = (double *)malloc(sizeof(double) * nlevel);
  double *ct = (double *)malloc(sizeof(double) * nlyr);
  double *numDens = (double *)malloc(sizeof(double) * nlyr);
  double *tol = (double *)malloc(sizeof(double) * n_tpert);
  /* hPa -> Pa (:70-73) and reversal to surface-first (:89-99) */
  for (int k = 0; k < nlevel; ++k) cp[k] = plevel_hPa[nlevel - 1 - k] * 100.0;
  for (int k = 0; k < nlyr; ++k) ct[k] = Tlayer[nlyr - 1 - k];
  for (int lyr = 0; lyr != nlyr; ++lyr) /* :201-202 */
    numDens[lyr] = (cp[lyr] - cp[lyr + 1]) * avog / molMassAir / earthAccel;

#define XS(t, s, r, p) xsec[(((size_t)(t) * n_species + (s)) * n_wvl + (r)) * n_p + (p)]
  for (int w = 0; w != n_wvl; ++w) { /* :205 */
    const int r = n_wvl - w - 1;     /* :206 read row; :244 store column -- the same index */
    for (int lyr = 0; lyr < nlyr; ++lyr) {
      double acc = 0.0; /* tauMat zeroed :197-199 */
      for (int spec = 0; spec != n_species; ++spec) {
        double midP = (cp[lyr + 1] + cp[lyr]) / 2;                /* :214 */
        double midT = ct[lyr];                                    /* :217 prop_at_Lev==0 */
        double midVMR = vmr[(size_t)spec * nlyr + (nlyr - 1 - lyr)]; /* :222 on reversed VMRS */
        int lp = orc_lower_pos(p_grid, n_p, midP);                /* :226 */
        for (int k = 0; k != n_tpert; ++k) tol[k] = t_ref[lp] + t_pert[k]; /* :229-231 */
        int lt = orc_lower_pos(tol, n_tpert, midT);               /* :232 */
        double c0 = XS(lt, spec, r, lp);                          /* :235 */
        double cT = XS(lt + 1, spec, r, lp) - c0;
        double cP = XS(lt, spec, r, lp + 1) - c0;
        double cPT = XS(lt + 1, spec, r, lp + 1) - cP - cT - c0;
        double delT = (midT - tol[lt]) / (tol[lt + 1] - tol[lt]);
        double delP = (midP - p_grid[lp]) / (p_grid[lp + 1] - p_grid[lp]);
        double tempXsec = c0 + cT * delT + cP * delP + cPT * delT * delP; /* :241 */
        acc += tempXsec * midVMR;                                 /* :244 */
      }
      acc *= numDens[lyr];                                        /* :246 */
      tau[(size_t)r * nlyr + (nlyr - 1 - lyr)] = acc;             /* :244,:257-259 */
    }
  }
#undef XS
  free(cp);
  free(ct);
  free(numDens);
  free(tol);
}

/* ------------------------------------------------------------------ */
/* lbl composition (no reference function; format lbl.arts/README:5-17) */
/* ------------------------------------------------------------------ */
void orc_lbl_tau(int n_nodes, int nlayer, const double *const tabs[ORC_NSPEC_LBL],
                 const double *scale, double *tau) {
  for (int b = 0; b < n_nodes - 1; ++b)
    for (int l = 0; l < nlayer; ++l) {
      double acc = 0.0;
      for (int s = 0; s < ORC_NSPEC_LBL; ++s)
        acc += scale[s * nlayer + l] * tabs[s][(size_t)b * nlayer + l];
      tau[(size_t)b * nlayer + l] = acc;
    }
}

void orc_cloud_into_tau(double *tau, int nwvl, int nlayer, int cloud_layer, double cloud_tau) {
  for (int i = 0; i < nwvl; ++i) tau[(size_t)i * nlayer + cloud_layer] += cloud_tau; /* main.cpp:269-271 */
}

/* ------------------------------------------------------------------ */
/* thermal RT: main.cpp:291-344                                        */
/* ------------------------------------------------------------------ */
void orc_radiative_transfer(int nspec, int nlayer, int nangle, const double *tau,
                            const double *wvl, const double *weight, int planck_mode,
                            const double *Tlayer, double T_surface, double solar_irr,
                            double *E_down, double *E_up, double *dE) {
  const int nlevel = nlayer + 1;
  const double dmu = 1.0 / (double)nangle; /* main.cpp:356 */
  double *B = (double *)malloc(sizeof(double) * nlayer);
  double *alpha = (double *)malloc(sizeof(double) * nlayer);
  double *mu = (double *)malloc(sizeof(double) * nangle);
  for (int i = 0; i < nangle; ++i) mu[i] = dmu / 2.0 + dmu * (double)i; /* main.cpp:482 */
  for (int i = 0; i < nlevel; ++i) E_down[i] = E_up[i] = 0.0;            /* :326-327 */

  for (int i_rad = 0; i_rad < nspec; ++i_rad) { /* :329 */
    const double *t = tau + (size_t)i_rad * nlayer;
    double B_sfc;
    if (planck_mode == 0) {
      for (int i = 0; i < nlayer; ++i) B[i] = orc_planck_mono(wvl[i_rad], weight[i_rad], Tlayer[i]);
      B_sfc = orc_planck_mono(wvl[i_rad], weight[i_rad], T_surface); /* :301, same value every angle */
    } else {
      for (int i = 0; i < nlayer; ++i) B[i] = orc_cplkavg(wvl[i_rad], wvl[i_rad + 1], Tlayer[i]);
      B_sfc = orc_cplkavg(wvl[i_rad], wvl[i_rad + 1], T_surface);
    }
    for (int imu = 0; imu < nangle; ++imu) { /* :297 */
      double L_down = 0.0;
      double L_up = B_sfc;
      E_up[nlevel - 1] += 2 * M_PI * L_up * mu[imu] * dmu;                     /* :302 */
      for (int i = 0; i < nlayer; ++i) alpha[i] = 1.0 - exp(-t[i] / mu[imu]);   /* :207-212, 20 used */
      for (int ilev = 1; ilev < nlevel; ++ilev) {                               /* :306-309 */
        L_down = (1 - alpha[ilev - 1]) * L_down + alpha[ilev - 1] * B[ilev - 1];
        E_down[ilev] += 2 * M_PI * L_down * mu[imu] * dmu;
      }
      for (int ilev = nlevel - 2; ilev >= 0; --ilev) {                          /* :311-314 */
        L_up = (1 - alpha[ilev]) * L_up + alpha[ilev] * B[ilev];
        E_up[ilev] += 2 * M_PI * L_up * mu[imu] * dmu;
      }
    }
  }
  for (int i = 0; i < nlayer; ++i) dE[i] = E_down[i] - E_down[i + 1] + E_up[i + 1] - E_up[i]; /* :337-339 */
  dE[nlayer - 1] += solar_irr + E_down[nlevel - 1] - E_up[nlevel - 1];                         /* :341 */
  free(B);
  free(alpha);
  free(mu);
}

/* ------------------------------------------------------------------ */
/* time stepping: main.cpp:123-176, 277-289, 536-540                   */
/* ------------------------------------------------------------------ */
double orc_calculate_timestep(double dp, const double *dE, int nlayer) { /* main.cpp:156-162 */
  double mx = dE[0];
  for (int i = 1; i < nlayer; ++i)
    if (mx < dE[i]) mx = dE[i]; /* std::max_element: first largest, signed */
  double timestep = C_MAX_DT / mx * (C_CAIR * dp * 100.0) / C_G;
  if (timestep > 3600 * 12) timestep = 3600 * 12;
  return timestep;
}

void orc_thermodynamics(double *Tlayer, int nlayer, double dp, const double *dE, double timestep,
                        double *T_surface, const double *conv) { /* main.cpp:164-176 */
  for (int i = 0; i < nlayer; ++i) Tlayer[i] += dE[i] * timestep * C_G / (C_CAIR * dp * 100.0);
  *T_surface = Tlayer[nlayer - 1] * conv[nlayer - 1];
}

static int cmp_desc(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x < y) - (x > y);
}

void orc_convective_adjustment(double *Tlayer, double *theta, const double *conv, int nlayer) {
  for (int i = 0; i < nlayer; ++i) theta[i] = Tlayer[i] * conv[i];  /* main.cpp:123-127, 536 */
  qsort(theta, nlayer, sizeof(double), cmp_desc);                   /* :539 sort(greater) */
  for (int i = 0; i < nlayer; ++i) Tlayer[i] = theta[i] / conv[i];  /* :135-139, 540 */
}

void orc_water_vapor_feedback(const double *Tlayer, const double *rel_hum, const double *player,
                              double *h2o_vmr, int nlayer) { /* main.cpp:281-289 */
  for (int i = 0; i < nlayer; ++i) {
    double e_sat = orc_magnus(Tlayer[i]);
    h2o_vmr[i] = rel_hum[i] * e_sat / player[i];
  }
}

/* ------------------------------------------------------------------ */
/* atmosphere file and column init: main.cpp:396-485                   */
/* ------------------------------------------------------------------ */
int orc_atm_load(const char *path, int nlevel_max, double *cols, int *ncols_found) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  char *line = NULL;
  size_t cap = 0;
  for (int i = 0; i < 4; ++i) /* main.cpp:402-404 */
    if (getline(&line, &cap, f) < 0) break;
  int nrow = 0, maxc = 0;
  while (getline(&line, &cap, f) >= 0) { /* :410-417 */
    char *s = line;
    int c = 0;
    for (;;) {
      char *e;
      double v = strtod(s, &e);
      if (e == s) break;
      if (c < 9 && nrow < nlevel_max) cols[c * nlevel_max + nrow] = v;
      ++c;
      s = e;
    }
    if (c > 0) {
      if (c > maxc) maxc = c;
      ++nrow;
    }
  }
  free(line);
  fclose(f);
  if (nrow > nlevel_max) return -2;
  if (maxc < 9) { /* fpda.lbl.atm has 6 columns: README:14-17 defaults */
    const double dflt[3] = {400.0, 1.7, 0.315};
    for (int c = (maxc < 6 ? 6 : maxc); c < 9; ++c)
      for (int r = 0; r < nrow; ++r) cols[c * nlevel_max + r] = dflt[c - 6];
  }
  if (ncols_found) *ncols_found = maxc;
  return nrow;
}

void orc_column_init(int nlevel, const double *a, double co2_factor, double rh_scale,
                     double *plevel, double *player, double *Tlayer, double *conv, double *vmr9,
                     double *rel_hum) {
  const int nlayer = nlevel - 1;
  const double *pl = a + 1 * nlevel, *Tl = a + 2 * nlevel;
  const double *h2o = a + 4 * nlevel, *o3 = a + 5 * nlevel, *co2 = a + 6 * nlevel,
               *ch4 = a + 7 * nlevel, *n2o = a + 8 * nlevel;
  double *H2O = vmr9 + 0 * nlayer, *CO2 = vmr9 + 1 * nlayer, *O3 = vmr9 + 2 * nlayer,
         *N2O = vmr9 + 3 * nlayer, *CO = vmr9 + 4 * nlayer, *CH4 = vmr9 + 5 * nlayer,
         *O2 = vmr9 + 6 * nlayer, *HNO3 = vmr9 + 7 * nlayer, *N2 = vmr9 + 8 * nlayer;
  for (int i = 0; i < nlevel; ++i) plevel[i] = pl[i];
  for (int i = 0; i < nlayer; ++i) { /* VMR_level_to_layer, main.cpp:141-147 */
    H2O[i] = (h2o[i] + h2o[i + 1]) / 2.0;
    O3[i] = (o3[i] + o3[i + 1]) / 2.0;
    CO2[i] = (co2[i] + co2[i + 1]) / 2.0;
    CH4[i] = (ch4[i] + ch4[i + 1]) / 2.0;
    N2O[i] = (n2o[i] + n2o[i + 1]) / 2.0;
  }
  for (int i = 0; i < nlayer; ++i) { /* main.cpp:454-460 (the reference's i==nlayer write is out of bounds) */
    H2O[i] *= 1E-6;
    O3[i] *= 1E-6;
    CO2[i] *= co2_factor * 1E-6;
    CH4[i] *= 1E-6;
    N2O[i] *= 1E-6;
    CO[i] = 0.0;
    O2[i] = 0.0;
    HNO3[i] = 0.0;
    N2[i] = 0.0;
    H2O[i] *= rh_scale; /* ensemble perturbation; 1.0 = reference */
  }
  for (int i = 0; i < nlayer; ++i) { /* main.cpp:467-468: LEVEL p and T with LAYER vmr */
    double e_sat = orc_magnus(Tl[i]);
    rel_hum[i] = H2O[i] * pl[i] / e_sat;
  }
  for (int i = 0; i < nlayer; ++i) { /* main.cpp:471-476 */
    player[i] = (pl[i] + pl[i + 1]) / 2.0;
    Tlayer[i] = (Tl[i] + Tl[i + 1]) / 2.0;
    conv[i] = pow(1000.0 / player[i], C_KAPPA);
  }
}

/* ------------------------------------------------------------------ */
/* table blob                                                          */
/* ------------------------------------------------------------------ */
int orc_rtab_load(const char *path, long long dims[4], double **blob) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  char magic[8];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "RWVLTAB1", 8) != 0) {
    fclose(f);
    return -2;
  }
  if (fread(dims, sizeof(long long), 4, f) != 4) {
    fclose(f);
    return -2;
  }
  size_t nt = dims[0], ns = dims[1], nw = dims[2], np = dims[3];
  size_t n = np + np + nt + ns * np + nw + nw + nt * ns * nw * np;
  double *b = (double *)malloc(n * sizeof(double));
  if (!b) {
    fclose(f);
    return -3;
  }
  if (fread(b, sizeof(double), n, f) != n) {
    free(b);
    fclose(f);
    return -2;
  }
  fclose(f);
  *blob = b;
  return 0;
}

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ */
/* ensemble: main.cpp:531-583 for many independent columns             */
/* ------------------------------------------------------------------ */
struct orc_ensemble {
  int ncol, nlayer, nlevel, nangle, mode, nthreads, threads_used;
  long long step_index; /* the reference's loop variable i (main.cpp:531) */
  double dp;
  double *plevel, *player, *conv;
  double *solar_irr, *cloud_tau, *resid, *Tprev; /* per column */
  double *Ttau; /* [ncol][nlayer]: the profile the optical depths were last formed from */
  int *cloud_layer;
  /* repwvl */
  int n_tpert, n_species, n_wvl, n_p;
  double *p_grid, *t_ref, *t_pert, *wvl, *weight, *xsec;
  /* lbl */
  int n_nodes;
  double *lbl_wvl, *lbl_tab[ORC_NSPEC_LBL], *h2o_ref;
  /* columns */
  int ngas;
  double *T, *Tsfc, *gas, *rel_hum, *timestep, *E_down, *E_up, *dE;
  float *time_h;
};

static double *dup_d(const double *src, size_t n) {
  double *p = (double *)malloc(n * sizeof(double));
  if (p && src) memcpy(p, src, n * sizeof(double));
  return p;
}

int orc_create(int ncol, int nlayer, int nangle, orc_ensemble **out) {
  if (ncol < 1 || nlayer < 1 || nangle < 1 || !out) return -1;
  orc_ensemble *h = (orc_ensemble *)calloc(1, sizeof(*h));
  if (!h) return -5;
  h->ncol = ncol;
  h->nlayer = nlayer;
  h->nlevel = nlayer + 1;
  h->nangle = nangle;
  h->mode = -1;
  h->nthreads = 1;
  h->dp = 1000.0 / (double)nlayer; /* main.cpp:355 */
  h->solar_irr = (double *)calloc(ncol, sizeof(double));
  h->cloud_tau = (double *)calloc(ncol, sizeof(double));
  h->cloud_layer = (int *)calloc(ncol, sizeof(int));
  h->resid = (double *)calloc(ncol, sizeof(double));
  h->Tprev = (double *)calloc((size_t)ncol * nlayer, sizeof(double));
  h->Ttau = (double *)calloc((size_t)ncol * nlayer, sizeof(double));
  for (int c = 0; c < ncol; ++c) {
    h->cloud_layer[c] = 17;      /* main.cpp:92 */
    h->cloud_tau[c] = 2.0 / 2.0; /* main.cpp:267 */
  }
  h->T = (double *)calloc((size_t)ncol * nlayer, sizeof(double));
  h->Tsfc = (double *)calloc(ncol, sizeof(double));
  h->time_h = (float *)calloc(ncol, sizeof(float));
  h->timestep = (double *)calloc(ncol, sizeof(double));
  h->rel_hum = (double *)calloc((size_t)ncol * nlayer, sizeof(double));
  h->E_down = (double *)calloc((size_t)ncol * h->nlevel, sizeof(double));
  h->E_up = (double *)calloc((size_t)ncol * h->nlevel, sizeof(double));
  h->dE = (double *)calloc((size_t)ncol * nlayer, sizeof(double));
  *out = h;
  return 0;
}

int orc_destroy(orc_ensemble *h) {
  if (!h) return -1;
  free(h->plevel); free(h->player); free(h->conv);
  free(h->solar_irr); free(h->cloud_tau); free(h->cloud_layer); free(h->resid); free(h->Tprev); free(h->Ttau);
  free(h->p_grid); free(h->t_ref); free(h->t_pert); free(h->wvl); free(h->weight); free(h->xsec);
  free(h->lbl_wvl); free(h->h2o_ref);
  for (int s = 0; s < ORC_NSPEC_LBL; ++s) free(h->lbl_tab[s]);
  free(h->T); free(h->Tsfc); free(h->gas); free(h->rel_hum); free(h->timestep);
  free(h->E_down); free(h->E_up); free(h->dE); free(h->time_h);
  free(h);
  return 0;
}

int orc_set_grid(orc_ensemble *h, const double *plevel_hPa) {
  if (!h || !plevel_hPa) return -1;
  free(h->plevel); free(h->player); free(h->conv);
  h->plevel = dup_d(plevel_hPa, h->nlevel);
  h->player = dup_d(NULL, h->nlayer);
  h->conv = dup_d(NULL, h->nlayer);
  for (int i = 0; i < h->nlayer; ++i) {
    h->player[i] = (plevel_hPa[i] + plevel_hPa[i + 1]) / 2.0; /* main.cpp:472 */
    h->conv[i] = pow(1000.0 / h->player[i], C_KAPPA);          /* main.cpp:474 */
  }
  return 0;
}

int orc_set_forcing(orc_ensemble *h, double solar_irr, int cloud_layer, double cloud_tau) {
  if (!h || cloud_layer < -1 || cloud_layer >= h->nlayer) return -1;
  for (int c = 0; c < h->ncol; ++c) {
    h->solar_irr[c] = solar_irr;
    h->cloud_layer[c] = cloud_layer;
    h->cloud_tau[c] = cloud_tau;
  }
  return 0;
}

int orc_set_column_forcing(orc_ensemble *h, const double *solar_irr, const int *cloud_layer,
                           const double *cloud_tau) {
  if (!h) return -1;
  for (int c = 0; c < h->ncol; ++c) {
    if (cloud_layer && (cloud_layer[c] < -1 || cloud_layer[c] >= h->nlayer)) return -1;
    if (solar_irr) h->solar_irr[c] = solar_irr[c];
    if (cloud_layer) h->cloud_layer[c] = cloud_layer[c];
    if (cloud_tau) h->cloud_tau[c] = cloud_tau[c];
  }
  return 0;
}

int orc_get_residual(orc_ensemble *h, double *resid) {
  if (!h || !resid) return -1;
  memcpy(resid, h->resid, h->ncol * sizeof(double));
  return 0;
}

int orc_set_repwvl_tables(orc_ensemble *h, int n_tpert, int n_species, int n_wvl, int n_p,
                          const double *p_grid, const double *t_ref, const double *t_pert,
                          const double *wvl, const double *weight, const double *xsec) {
  if (!h || n_species != ORC_NSPEC_REPWVL || n_tpert < 2 || n_p < 2 || n_wvl < 1) return -1;
  h->mode = ORC_MODE_REPWVL;
  h->n_tpert = n_tpert; h->n_species = n_species; h->n_wvl = n_wvl; h->n_p = n_p;
  free(h->p_grid); free(h->t_ref); free(h->t_pert); free(h->wvl); free(h->weight); free(h->xsec);
  h->p_grid = dup_d(p_grid, n_p);
  h->t_ref = dup_d(t_ref, n_p);
  h->t_pert = dup_d(t_pert, n_tpert);
  h->wvl = dup_d(wvl, n_wvl);
  h->weight = dup_d(weight, n_wvl);
  h->xsec = dup_d(xsec, (size_t)n_tpert * n_species * n_wvl * n_p);
  h->ngas = ORC_NSPEC_REPWVL;
  return 0;
}

int orc_set_lbl_tables(orc_ensemble *h, int n_nodes, const double *wvl,
                       const double *const tabs[ORC_NSPEC_LBL], const double *h2o_ref) {
  if (!h || n_nodes < 2 || !wvl || !tabs || !h2o_ref) return -1;
  for (int i = 0; i + 1 < n_nodes; ++i)
    if (!(wvl[i] < wvl[i + 1]) || !(wvl[i] > 0)) return -1; /* cplkavg.cpp:144 needs wnumhi > wnumlo */
  h->mode = ORC_MODE_LBL;
  h->n_nodes = n_nodes;
  free(h->lbl_wvl); free(h->h2o_ref);
  h->lbl_wvl = dup_d(wvl, n_nodes);
  h->h2o_ref = dup_d(h2o_ref, h->nlayer);
  for (int s = 0; s < ORC_NSPEC_LBL; ++s) {
    free(h->lbl_tab[s]);
    h->lbl_tab[s] = dup_d(tabs[s], (size_t)n_nodes * h->nlayer);
  }
  h->ngas = ORC_NSPEC_LBL;
  return 0;
}

int orc_set_columns(orc_ensemble *h, const double *Tlayer0, const double *T_surface0,
                    const double *gas, const double *rel_hum) {
  if (!h || h->mode < 0 || !Tlayer0 || !T_surface0 || !gas || !rel_hum) return -1;
  size_t n = (size_t)h->ncol * h->nlayer;
  memcpy(h->T, Tlayer0, n * sizeof(double));
  memcpy(h->Tsfc, T_surface0, h->ncol * sizeof(double));
  memcpy(h->rel_hum, rel_hum, n * sizeof(double));
  free(h->gas);
  h->gas = dup_d(gas, n * h->ngas);
  memset(h->time_h, 0, h->ncol * sizeof(float));
  memset(h->timestep, 0, h->ncol * sizeof(double));
  h->step_index = 0;
  return 0;
}

int orc_set_threads(orc_ensemble *h, int nthreads) {
  if (!h || nthreads < 1) return -1;
  h->nthreads = nthreads;
  return 0;
}

int orc_nspectral(orc_ensemble *h) {
  if (!h || h->mode < 0) return -1;
  return h->mode == ORC_MODE_REPWVL ? h->n_wvl : h->n_nodes - 1;
}

int orc_threads_used(orc_ensemble *h) { return h ? h->threads_used : -1; }

/* optical depths for one column at its current state: main.cpp:500-504 / 564-568 */
static void column_tau(const orc_ensemble *h, int c, double *tau) {
  const int nl = h->nlayer;
  memcpy(h->Ttau + (size_t)c * nl, h->T + (size_t)c * nl, nl * sizeof(double));
  const double *gas = h->gas + (size_t)c * h->ngas * nl;
  int nspec;
  if (h->mode == ORC_MODE_REPWVL) {
    orc_read_tau(h->n_tpert, h->n_species, h->n_wvl, h->n_p, h->p_grid, h->t_ref, h->t_pert,
                 h->xsec, h->nlevel, h->plevel, h->T + (size_t)c * nl, gas, tau);
    nspec = h->n_wvl;
  } else {
    const double *tabs[ORC_NSPEC_LBL];
    for (int s = 0; s < ORC_NSPEC_LBL; ++s) tabs[s] = h->lbl_tab[s];
    orc_lbl_tau(h->n_nodes, nl, tabs, gas, tau);
    nspec = h->n_nodes - 1;
  }
  if (h->cloud_layer[c] >= 0) orc_cloud_into_tau(tau, nspec, nl, h->cloud_layer[c], h->cloud_tau[c]);
}

/* H2O row of the gas block from relative humidity: main.cpp:562 */
static void column_feedback(const orc_ensemble *h, int c) {
  const int nl = h->nlayer;
  double *gas = h->gas + (size_t)c * h->ngas * nl;
  double vmr[256];
  orc_water_vapor_feedback(h->T + (size_t)c * nl, h->rel_hum + (size_t)c * nl, h->player, vmr, nl);
  if (h->mode == ORC_MODE_REPWVL) {
    for (int l = 0; l < nl; ++l) gas[0 * nl + l] = vmr[l];
  } else { /* tau_H2O is linear in VMR: scale = vmr / (vmr the table was computed at) */
    for (int l = 0; l < nl; ++l) gas[1 * nl + l] = vmr[l] / h->h2o_ref[l];
  }
}

int orc_step(orc_ensemble *h, int nsteps) {
  if (!h || h->mode < 0 || !h->gas || !h->plevel || nsteps < 0 || h->nlayer > 256) return -1;
  const int nl = h->nlayer, nlev = h->nlevel;
  const int nspec = orc_nspectral(h);
  int used = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(h->nthreads)
#endif
  {
#ifdef _OPENMP
#pragma omp single
    used = omp_get_num_threads();
#endif
    double *tau = (double *)malloc((size_t)nspec * nl * sizeof(double));
    double theta[256];
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int c = 0; c < h->ncol; ++c) {
      double *T = h->T + (size_t)c * nl;
      for (long long i = h->step_index; i < h->step_index + nsteps; ++i) { /* main.cpp:531 */
        if (i == 0) column_tau(h, c, tau); /* :500-504: first tau from the PRE-adjustment profile */
        orc_convective_adjustment(T, theta, h->conv, nl);                  /* :536-540 */
        { /* equilibrium residual of the build (no reference counterpart): change of the adjusted profile */
          double r = 0.0, *prev = h->Tprev + (size_t)c * nl;
          for (int l = 0; l < nl; ++l) { /* a NaN layer makes the residual NaN (as k_pre / post_column do): a */
            const double dl = fabs(T[l] - prev[l]); /* column gone NaN must never pass for converged */
            if (dl > r || dl != dl) r = dl;
            prev[l] = T[l];
          }
          h->resid[c] = (i == 0) ? 1e300 : r;
        }
        if (i != 0) {                                                      /* :549 */
          column_feedback(h, c);                                           /* :562 */
          column_tau(h, c, tau);                                           /* :564-568 */
        }
        orc_radiative_transfer(nspec, nl, h->nangle, tau,
                               h->mode == ORC_MODE_REPWVL ? h->wvl : h->lbl_wvl, h->weight,
                               h->mode == ORC_MODE_REPWVL ? 0 : 1, T, h->Tsfc[c], h->solar_irr[c],
                               h->E_down + (size_t)c * nlev, h->E_up + (size_t)c * nlev,
                               h->dE + (size_t)c * nl);                    /* :574 */
        h->timestep[c] = orc_calculate_timestep(h->dp, h->dE + (size_t)c * nl, nl); /* :576 */

        orc_thermodynamics(T, nl, h->dp, h->dE + (size_t)c * nl, h->timestep[c], &h->Tsfc[c],
                           h->conv);                                       /* :578 */
        h->time_h[c] += (float)h->timestep[c] / 3600;                      /* :581 */
      }
    }
    free(tau);
  }
  h->threads_used = used;
  h->step_index += nsteps;
  return 0;
}

int orc_get_state(orc_ensemble *h, double *Tlayer, double *T_surface, float *time_h,
                  double *timestep) {
  if (!h) return -1;
  if (Tlayer) memcpy(Tlayer, h->T, (size_t)h->ncol * h->nlayer * sizeof(double));
  if (T_surface) memcpy(T_surface, h->Tsfc, h->ncol * sizeof(double));
  if (time_h) memcpy(time_h, h->time_h, h->ncol * sizeof(float));
  if (timestep) memcpy(timestep, h->timestep, h->ncol * sizeof(double));
  return 0;
}

int orc_get_fluxes(orc_ensemble *h, double *E_down, double *E_up, double *dE) {
  if (!h) return -1;
  if (E_down) memcpy(E_down, h->E_down, (size_t)h->ncol * h->nlevel * sizeof(double));
  if (E_up) memcpy(E_up, h->E_up, (size_t)h->ncol * h->nlevel * sizeof(double));
  if (dE) memcpy(dE, h->dE, (size_t)h->ncol * h->nlayer * sizeof(double));
  return 0;
}

/* what output_conv would print now (main.cpp:536-546), without touching the state */
int orc_get_profile(orc_ensemble *h, int col, double *Tlayer_adj, double *theta, float *time_h) {
  if (!h || col < 0 || col >= h->ncol || !h->conv) return -1;
  double T[256], th[256];
  memcpy(T, h->T + (size_t)col * h->nlayer, h->nlayer * sizeof(double));
  orc_convective_adjustment(T, th, h->conv, h->nlayer);
  if (Tlayer_adj) memcpy(Tlayer_adj, T, h->nlayer * sizeof(double));
  if (theta) memcpy(theta, th, h->nlayer * sizeof(double));
  if (time_h) *time_h = h->time_h[col];
  return 0;
}

/* The optical depths of the most recent radiative transfer (before the first step: those it will
 * use), i.e. formed from the profile column_tau last saw, with the current gas block. */
int orc_get_tau(orc_ensemble *h, int col, double *tau) {
  if (!h || h->mode < 0 || col < 0 || col >= h->ncol || !tau) return -1;
  const int nl = h->nlayer;
  double keep[256], *T = h->T + (size_t)col * nl;
  if (nl > 256) return -1;
  if (h->step_index > 0) { /* evaluate at the saved profile, leave the state as it was */
    memcpy(keep, T, nl * sizeof(double));
    memcpy(T, h->Ttau + (size_t)col * nl, nl * sizeof(double));
    column_tau(h, col, tau);
    memcpy(T, keep, nl * sizeof(double));
  } else {
    column_tau(h, col, tau);
  }
  return 0;
}

int orc_get_gas(orc_ensemble *h, double *gas) {
  if (!h || !h->gas || !gas) return -1;
  memcpy(gas, h->gas, (size_t)h->ncol * h->ngas * h->nlayer * sizeof(double));
  return 0;
}
