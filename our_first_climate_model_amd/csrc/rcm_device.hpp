// rcm_device.hpp -- device-side physics for the gfx950 kernels.
//
// Every function names the reference function it restates.  fp64 throughout; no fast-math.
#pragma once
#include <hip/hip_runtime.h>

#include "rcm_exp_coef.hpp"

namespace rcm {

// ---------------------------------------------------------------------------------------------
// exp(x) for the transmission t = exp(-tau/mu) (main.cpp:209).
//
// EXPV 0 (reference variants): the ROCm device library's exp (~1 ulp).
// EXPV 1 (production): k = rint(y), y = x*log2(e), by the 1.5*2^52 trick (the integer falls out of the low
//         mantissa word for free), f = y - k in one FMA, near-minimax polynomial for 2^f with constant
//         term exactly 1, 2^k applied by v_ldexp_f64.
// ---------------------------------------------------------------------------------------------
// Transmission t = exp(-tau * s) with the per-angle constant a1 = -s*log2(e).
// tau must already be clamped to [kTauMin, kTauMax] (once per band-layer, clamp_tau) so that
// |k| < 2^31; v_ldexp_f64 takes care of k < -1022 (denormal, then 0).
constexpr double kTauMin = -10.0;  // keeps |k| < 2^31; exp(+10/mu) overflows to +inf above 1/mu = 71 (nangle >= 36), as exp() would
constexpr double kTauMax = 1.0e7;  // exp(-1e7) == 0 for every angle

// Clamp that keeps a NaN a NaN: v_max_f64 / v_min_f64 return the non-NaN operand, which would turn a NaN optical
// depth (NaN in a mixing ratio, a relative humidity or a table) into tau = -10 and finite garbage downstream,
// where the reference (and the oracle) propagate the NaN.  One compare and one select of the high word put it
// back (the low word of a NaN does not matter while the high word keeps the quiet bit every arithmetic NaN has).
__device__ __forceinline__ double clamp_tau(double tau) {
  const double c = fmin(fmax(tau, kTauMin), kTauMax);
  const int hi = (tau != tau) ? __double2hiint(tau) : __double2hiint(c);
  return __hiloint2double(hi, __double2loint(c));
}

#ifndef RCM_EXP_DEG
#define RCM_EXP_DEG 7  // degree of the polynomial of the transmissions (c0 = 1 exactly), and
#endif
#ifndef RCM_EXP_HALF
#define RCM_EXP_HALF 1  // 1: the polynomial is 2^(f/2) and the result is squared; 0: 2^f on the full range.
#endif                  // VALU instructions and max rel err of t:
                        //   half, degree 7 (default) 12, 7.4e-13;  half 8: 13, 1.5e-14;  half 6: 11, 6.8e-11
                        //   full, degree 8: 12, 2.1e-12;  9: 13, 3.7e-14;  10: 14, 5.8e-16
                        // Flux error of the kernel against a long-double evaluation of the reference's sweep
                        // (tools/accuracy_vs_longdouble.py): profiles/README.md; the contract is 1e-6 W/m^2.
__device__ __forceinline__ double exp_trans(double tau, double a1) {
  constexpr double kMagic = 0x1.8p+52;
  const double z = fma(tau, a1, kMagic);  // low word of z = k = rint(-tau*s*log2e), two's complement
  const double kf = z - kMagic;
  // f = y - k with y = tau*a1 = -tau*s*log2(e): one FMA, exact up to the rounding of a1
  // (relative 1.1e-16 of y: <= 4e-15 of the result wherever t >= 1e-15)
  const double f = fma(tau, a1, -kf);
#if RCM_EXP_HALF  // 2^(f/2), the halving folded into the coefficients, then one squaring
#if RCM_EXP_DEG == 6
  constexpr const double(&c)[7] = kExp2Half6;
#elif RCM_EXP_DEG == 7
  constexpr const double(&c)[8] = kExp2Half7;
#else
  constexpr const double(&c)[9] = kExp2Half8;
#endif
  constexpr int deg = sizeof(c) / sizeof(double) - 1;
  double q = c[deg];
#pragma unroll
  for (int i = deg - 1; i >= 0; --i) q = fma(q, f, c[i]);
  q *= q;
#elif RCM_EXP_DEG == 8
  double q = kExp2Poly8[8];
#pragma unroll
  for (int i = 7; i >= 0; --i) q = fma(q, f, kExp2Poly8[i]);
#elif RCM_EXP_DEG == 9
  double q = kExp2Poly9[9];
#pragma unroll
  for (int i = 8; i >= 0; --i) q = fma(q, f, kExp2Poly9[i]);
#else
  double q = kExp2Poly10[10];
#pragma unroll
  for (int i = 9; i >= 0; --i) q = fma(q, f, kExp2Poly10[i]);
#endif
  // v_ldexp_f64: one instruction, denormal/zero for k < -1022 (no integer clamp needed)
  return __builtin_amdgcn_ldexp(q, __double2loint(z));
}

// ---------------------------------------------------------------------------------------------
// Monochromatic Planck radiance times spectral weight: main.cpp:186-204.
//   weight*2*h*c^2 / (lambda^5 * (exp(h*c/(lambda*kB*T)) - 1)) / 1e9
// pnum = ((weight*2)*h)*pow(c,2) and lam5 = pow(lambda,5), lkb = lambda*kB are formed on the
// host once per wavelength in the reference's operation order (lambda in metres).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double planck_mono(double pnum, double lam5, double lkb, double T) {
  constexpr double kHC = 6.62607e-34 * 299792458.0;  // h*c, main.cpp:70-71 (one rounding, as h*c there)
  return pnum / (lam5 * (exp(kHC / (lkb * T)) - 1)) / 1e9;
}

// The same quantity for the lane-pair kernel, three IEEE divisions and a library exp cheaper:
// per wavelength (once per pass) xc = h*c/(lambda*kB) and sc = pnum/lam5/1e9; per temperature
// x = xc/T (correctly rounded through rT = 1/T), B = sc / (exp(x) - 1) with exp_pos and a
// Newton-refined reciprocal.  Differs from planck_mono by a few 1e-16 relative.
__device__ __forceinline__ double planck_mono_fast(double xc, double sc, double T, double rT);

// ---------------------------------------------------------------------------------------------
// Band-integrated Planck function: cplkavg.cpp:124-243 (libRadtran c_planck_func1).
// PROVENANCE: cplkavg_dev below restates, branch for branch and with the reference's identifier names, the
// algorithm of c_planck_func1 from libRadtran's cdisort.c (GPL; cplkavg.cpp:1-4 says where it was taken from).
// Parity with the reference needs the same branches, series lengths and constants; it is the COLD path
// (cplkavg_fast handles the narrow bands of the hot loop with arithmetic of its own).
// Takes the wavenumber bounds (cplkavg.cpp:141-142 done by the caller, once per band) and
// tpre = sigdpi * pow(t,4) (formed once per column-layer).  Same branches as the reference:
// Simpson iteration to 1e-6 when the bounds are close (:155-182), else power / exponential
// series (:187-237).  The reference's exit(1) on bad input (:144-146) cannot happen on the
// device: the host rejects non-increasing wavelengths and negative temperatures give NaN.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double plkf(double x) { return x * x * x / (exp(x) - 1.); }  // :109-113

constexpr double kConc = 0x1.3b5ef8356c19dp-3;  // 15/pi^4 = 15./pow(M_PI,4.) bit for bit, cplkavg.cpp:135

__device__ inline double cplkavg_dev(double wnumlo, double wnumhi, double t, double tpre) {
  constexpr double A1 = 1. / 3., A2 = -1. / 8., A3 = 1. / 60., A4 = -1. / 5040., A5 = 1. / 272160.,
                   A6 = -1. / 13305600.;
  constexpr double C2 = 1.438786, VCUT = 1.5;
  constexpr double kVmax = 0x1.62e42fefa39efp+9;                  // log(DBL_MAX), :133
  constexpr double kEps = 2.2204460492503131e-16;                 // DBL_EPSILON
  if (t < 1.e-4) return 0.;                                       // :148
  const double v0 = C2 * wnumlo / t, v1 = C2 * wnumhi / t;        // :152-153
  if (v0 > kEps && v1 < kVmax && (wnumhi - wnumlo) / wnumhi < 1.e-2) {  // :155
    const double hh = v1 - v0;
    double oldval = 0., val = 0.;
    const double val0 = plkf(v0) + plkf(v1);
    for (int n = 1; n <= 10; n++) {                               // :163
      const double del = hh / (2 * n);
      val = val0;
      for (int k = 1; k <= 2 * n - 1; k++)
        val += (double)(2 * (1 + k % 2)) * plkf(v0 + (double)k * del);
      val *= del * A1;
      if (fabs((val - oldval) / val) <= 1.e-6) break;             // :170
      oldval = val;
    }
    return tpre * kConc * val;                                    // :181
  }
  double d[2] = {0., 0.}, p[2] = {0., 0.};
  int smallv = 0;
  const double v[2] = {v0, v1};
#pragma unroll
  for (int i = 0; i < 2; i++) {                                   // :188
    const double vi = v[i];
    if (vi < VCUT) {
      smallv++;
      const double vsq = vi * vi;
      p[i] = kConc * vsq * vi * (A1 + vi * (A2 + vi * (A3 + vsq * (A4 + vsq * (A5 + vsq * A6)))));
    } else {
      // :203  `while (vi < vcp[mmax-1]) mmax++` over vcp = {10.25, 5.7, 3.9, 2.9, 2.3, 1.9, 0.0}: the table
      // decreases, so the loop stops at the first entry not above vi = the number of entries above vi
      // (vi >= VCUT here, the closing 0.0 never counts).  Compares instead of an indexed table: no scratch.
      const int mmax = 1 + (vi < 10.25) + (vi < 5.7) + (vi < 3.9) + (vi < 2.9) + (vi < 2.3) + (vi < 1.9);
      const double ex = exp(-vi);
      double exm = 1., di = 0.;
      for (int m = 1; m <= mmax; m++) {
        const double mv = (double)m * vi;
        exm = ex * exm;
        di += exm * (6. + mv * (6. + mv * (3. + mv))) / (double)(m * m * m * m);
      }
      d[i] = di * kConc;
    }
  }
  double ans;
  if (smallv == 2)
    ans = p[1] - p[0];
  else if (smallv == 1)
    ans = 1. - p[0] - d[1];
  else
    ans = d[0] - d[1];
  return ans * tpre;                                              // :237
}

// Out-of-line copy for callers that unroll the fast path: the wide-band branch is rare on
// line-by-line grids and its code is large.
__device__ __attribute__((noinline)) double cplkavg_slow(double wnumlo, double wnumhi, double t, double tpre) {
  return cplkavg_dev(wnumlo, wnumhi, t, tpre);
}

// Narrow-band fast path of cplkavg.  When the band is so narrow that hh = v1 - v0 < kHhFast, the
// reference's Simpson iteration (cplkavg.cpp:159-181) converges at n = 2 with a truncation error
// of hh^4/2880 * (f4/f) <= 6e-15 (f4 = fourth derivative), i.e. it returns the exact integral to
// rounding; the midpoint rule with the curvature term, hh * (f(xm) + hh^2/24 * f2(xm)) with
// f = x^3/(e^x - 1) and f2 its second derivative, has error hh^4/1920 * (f4/f) and is the same
// number to <= 4e-15 relative (checked against the reference's compiled cplkavg.cpp in tests/) for
// one exp and one reciprocal instead of six of each.  v0, v1 are formed as in cplkavg.cpp:152-153
// (C2*wnum/t): the product with rT = 1/t plus one residual FMA step is the correctly rounded
// quotient, so hh carries the reference's own rounding.  Returns false when the fast path does not
// apply (the caller then runs cplkavg_dev).
constexpr double kHhFast = 2.0e-3;

__device__ __forceinline__ double div_by_t(double c, double t, double rT) {
  const double q = c * rT;
  return fma(fma(-q, t, c), rT, q);
}

// exp(x) for the Planck argument, 0 < x < 700: same construction as exp_trans (base-2 reduction in
// one FMA, degree-10 polynomial 5.8e-16, v_ldexp_f64), no clamps needed in that range.
#ifndef RCM_EXPPOS_HALF
#define RCM_EXPPOS_HALF 1  // 1: degree 8 for 2^(f/2), squared (7.1e-15; one instruction fewer than 0: degree 10, 5.8e-16)
#endif
#ifndef RCM_RCP_NEWTON
#define RCM_RCP_NEWTON 1   // Newton steps after v_rcp_f64 (~2^-26 relative) in the Planck functions: one reaches ~2^-52
#endif
__device__ __forceinline__ double exp_pos(double x) {
  constexpr double kLog2e = 0x1.71547652b82fep+0;
  constexpr double kMagic = 0x1.8p+52;
  const double z = fma(x, kLog2e, kMagic);
  const double kf = z - kMagic;
  const double f = fma(x, kLog2e, -kf);
#if RCM_EXPPOS_HALF
  double q = kExp2Half8[8];
#pragma unroll
  for (int i = 7; i >= 0; --i) q = fma(q, f, kExp2Half8[i]);
  q *= q;
#else
  double q = kExp2Poly10[10];
#pragma unroll
  for (int i = 9; i >= 0; --i) q = fma(q, f, kExp2Poly10[i]);
#endif
  return __builtin_amdgcn_ldexp(q, __double2loint(z));
}

// 1/d from v_rcp_f64 (~2^-26 relative) and RCM_RCP_NEWTON Newton steps (each squares the error)
__device__ __forceinline__ double rcp_newton(double d) {
  double g = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int i = 0; i < RCM_RCP_NEWTON; ++i) g = fma(fma(-d, g, 1.0), g, g);
  return g;
}

// The fast formula itself, preconditions established by the caller: xm = argument at the band centre,
// hh = its width.
__device__ __forceinline__ double cplkavg_fast_mid(double xm, double hh, double pc) {
  const double d = exp_pos(xm) - 1.0;
  const double g = rcp_newton(d);
  const double g1 = fma(-g, g, -g);          // dg/dx = -g(1+g)
  const double g2 = -g1 * fma(2.0, g, 1.0);  // d2g/dx2
  const double x2 = xm * xm, x3 = x2 * xm;
  const double f = x3 * g;
  // f2 = x^3 g2 + 6 x^2 g1 + 6 x g = x * (x^2 g2 + 6 (x g1 + g))
  const double f2 = xm * fma(x2, g2, 6.0 * fma(xm, g1, g));
  return pc * (hh * fma(hh * hh * (1.0 / 24.0), f2, f));
}

__device__ __forceinline__ double cplkavg_fast_eval(double v0, double v1, double pc) {
  return cplkavg_fast_mid(0.5 * (v0 + v1), v1 - v0, pc);
}

__device__ __forceinline__ bool cplkavg_fast(double c2lo, double c2hi, double t, double rT,
                                             double pc, double *out) {
  const double v0 = div_by_t(c2lo, t, rT), v1 = div_by_t(c2hi, t, rT);
  const double hh = v1 - v0;
  if (!(hh < kHhFast && v0 > 2.2204460492503131e-16 && v1 < 700.0 && t >= 1.e-4)) return false;
  *out = cplkavg_fast_eval(v0, v1, pc);
  return true;
}

__device__ __forceinline__ double planck_mono_fast(double xc, double sc, double T, double rT) {
  const double d = exp_pos(div_by_t(xc, T, rT)) - 1.0;
  return sc * rcp_newton(d);
}

// Magnus saturation vapour pressure [hPa]: main.cpp:277-279
__device__ __forceinline__ double magnus_dev(double T) {
  return 6.1094 * exp(17.625 * (T - 273.15) / (T - 273.15 + 243.04));
}

// LowerPos on a small grid: repwvl_thermal.cpp:19-45 (first sign change of grid[i]-x, clamped).
__device__ __forceinline__ int sgn_dev(double v) { return (0.0 < v) - (v < 0.0); }

template <typename F>
__device__ __forceinline__ int lower_pos_dev(F grid, int n, double x) {
  int sign = sgn_dev(grid(0) - x);
  for (int p = 1; p != n; ++p) {
    const double delta = grid(p) - x;
    if (sign != sgn_dev(delta)) return p - 1;
    if (p == n - 1) return p - 1;
    sign = sgn_dev(delta);
  }
  return 0;
}

// 64-lane butterfly sum; every lane ends with the total (fixed order: deterministic).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace rcm
