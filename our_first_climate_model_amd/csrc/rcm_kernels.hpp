// rcm_kernels.hpp -- gfx950 (CDNA4) kernels of the radiative-convective column ensemble.
// Included once, by rcm_capi.hip, which holds the handle and the C ABI (include/rcm.h).
//
// Mapping: one wavefront (64 lanes) per (column, spectral chunk); spectral points
// on lanes; the column's level grid (temperatures, gas scalings, interpolation brackets) in
// LDS, read as broadcasts; 41 flux accumulators per lane in VGPRs, reduced across the wave by
// a butterfly at the end of the chunk; chunk partials summed in fixed order by the time-step
// kernel.  fp64 everywhere.  No MFMA: there is no dense contraction on this path.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rcm.h"
#include "rcm_device.hpp"
#include "rcm_internal.hpp"

namespace rcm {

constexpr int NL = 20;       // layers   (main.cpp:78); kernels are specialised for it
constexpr int NLEV = NL + 1; // levels   (main.cpp:79)
constexpr int PSTRIDE = 2 * NLEV;  // one partial record: E_down[21], E_up[21]
constexpr int WPB = 4;       // waves per block = columns per block (one per SIMD)
// Tuning knobs (compile time; tools/build_variants.py A/Bs them on the GPU).
#ifndef RCM_LB
#define RCM_LB 1  // waves per SIMD the lbl RT kernel is register-allocated for (1: spills go to AGPRs, not scratch)
#endif
#ifndef RCM_PAIR_LB
#define RCM_PAIR_LB 3  // waves per SIMD the lane-pair kernels are register-allocated for (<= 168 VGPRs)
#endif
#ifndef RCM_PLK_HOIST
#define RCM_PLK_HOIST 1  // decide the Planck regime once per band (wave vote) and run the loop branch-free
#endif
#ifndef RCM_PAIR_PLK_UNROLL
#define RCM_PAIR_PLK_UNROLL 3  // band-Planck evaluations in flight in the branch-free loop of the lane-pair lbl kernel
#endif
#ifndef RCM_PLK_XM_FAST
#define RCM_PLK_XM_FAST 1  // band-centre Planck argument as c2mid * (1/T) (two roundings: <= 1.5e-14 relative in B at x = 60,
#endif                     // 1e-15 at the 10 um peak) instead of the correctly rounded quotient: 2 FMAs per value.
                           // Measured (tools/ab_planck.py): with one Newton step and the half-range exp_pos -1.9 % kernel
                           // time, narrow-band slice and full-range deviations from the oracle unchanged (3.1e-13, 2.1e-11)
#ifndef RCM_PLK_UNROLL
#define RCM_PLK_UNROLL 1  // independent band-Planck evaluations in flight
#endif
#ifndef RCM_RW_TAU_UNROLL
#define RCM_RW_TAU_UNROLL 10  // layers of the repwvl optical-depth stage in flight (10 table loads each); with the
                              // wave priorities 10 beats 1 by 1.2 % (it made no difference before them)
#endif
#ifndef RCM_RW_PLK_UNROLL
#define RCM_RW_PLK_UNROLL 11  // monochromatic Planck evaluations in flight in the repwvl lane-pair kernels (all: another 0.7 %)
#endif
// Wave priority: the phases around the angle loop (column images, optical depths, Planck, reduction) are short and
// latency-bound; at a higher priority than the angle loops of the other waves of the SIMD they finish sooner.
#ifndef RCM_PRIO
#define RCM_PRIO 1
#endif
#if RCM_PRIO
// (Not the builtin and not `asm volatile`: either counts as a store to anything, and the kernel's scalar loads then
// become vector loads with a full wait each.  A plain asm tied to a value the next phase needs stays where it is put.)
#ifndef RCM_PRIO_A
#define RCM_PRIO_A 3  // before the angle loop
#define RCM_PRIO_B 0  // in it
#define RCM_PRIO_C 3  // after it
#endif
#define RCM_STR2(x) #x
#define RCM_STR(x) RCM_STR2(x)
#define RCM_PRIO_START(x) asm("s_setprio " RCM_STR(RCM_PRIO_A) : "+v"(x))
#define RCM_PRIO_LO(x) asm("s_setprio " RCM_STR(RCM_PRIO_B) : "+v"(x))
#define RCM_PRIO_HI(x) asm("s_setprio " RCM_STR(RCM_PRIO_C) : "+v"(x))
#else
#define RCM_PRIO_START(x) ((void)0)
#define RCM_PRIO_HI(x) ((void)0)
#define RCM_PRIO_LO(x) ((void)0)
#endif
#ifndef RCM_SWEEP_DIFF
#define RCM_SWEEP_DIFF 1  // recurrences of the lane-pair kernels in difference form (2 FMAs per crossing and angle);
#endif                    // 0: the emission form L = t*L + (1-t)*B (2.5), kept for A/B runs

enum { MODE_NONE = -1, MODE_REPWVL = 0, MODE_LBL = 1 };

#ifdef RCM_DBG_STAMPS  // profiling builds only (tools/phase_stamps.py): s_memtime at phase boundaries of sampled waves
__device__ unsigned long long *g_stamps = nullptr;  // set by rcm_dbg_stamps
__device__ __forceinline__ void stamp(bool sampled, int sample, int i) {
  if (sampled && g_stamps) g_stamps[(size_t)sample * 64 + i] = __builtin_amdgcn_s_memtime();
}
#define RCM_PT_STAMP(i) stamp((threadIdx.x & 63) == 0 && (blockIdx.x % 61) == 0 && (threadIdx.x >> 6) == 0, blockIdx.x / 61, (i))
#else
#define RCM_PT_STAMP(i) \
  do {                 \
  } while (0)
#endif

// Everything a kernel needs, passed by value.
struct Dev {
  int ncol, nangle, mode, ngas;
  int nspec;      // spectral points: n_wvl (repwvl) or bands = n_nodes-1 (lbl)
  int nchunk;     // spectral chunks per column
  int chunk_len;  // points per chunk (multiple of 32)
  double dp, sigdpi;
  // per-column forcing (rcm_set_forcing fills them uniformly): solar heating of the lowest layer
  // (main.cpp:341), grey cloud layer index (-1: none) and optical depth (main.cpp:266-274)
  const double *solar_irr, *cloud_tau;
  const int *cloud_layer;
  unsigned long long *dbg;  // phase time stamps of sampled waves (builds with -DRCM_DBG_STAMPS only; else null)
  double *resid;  // equilibrium residual per column: max_l |change of the ADJUSTED profile| between the
                  // last two iterations [K] (1e300 until two iterations have run)
  double *Tprev;  // adjusted profile of the previous iteration
  const double *player, *conv, *ang;  // ang = [inv_mu[nangle], cw[nangle], sum of cw, schedule {a1, code}[nsched]]
  int nsched;                         // entries of the root schedule (pair_all_angles)
  int flip;  // lbl: 1 = a wave takes its passes LAST FIRST (odd iterations of few-column runs: the passes a wave ended
             // the previous launch with are the ones still in its XCD's L2 -- the tables, 106 MB, exceed the 32 MiB)
  double *T, *Ttau, *Tsfc, *dt, *gas, *E_down, *E_up, *dE, *partial;
  const double *rel_hum;
  float *time_h;
  // lbl: tab[s][l][ld] (band fastest), wvl[n_nodes], h2o_ref[NL]
  const double *lbl_tab, *lbl_wvl, *h2o_ref;
  const double *lbl_wnum;  // 1.0E7 / lbl_wvl[i] (cplkavg.cpp:141-142), formed once on the host
  long lbl_ld;
  // repwvl: xsec[t][s][p][w] (wavelength fastest)
  const double *xsec, *t_ref, *t_pert, *delP, *numDens, *pnum, *lam5, *lkb;
  // xsec interpolated to each layer's mid pressure (host, with the grid), as PAIRS of neighbouring temperature rows:
  // xp[layer][t][s][w] = {x(t), x(t+1)}, t = 0 .. n_tpert-2: the two corners of a bracket come in ONE 16-byte load
  // (the optical-depth stage is bound by the number of load instructions, not by their bytes: profiles/README.md)
  const double *xp;
  const double *tref_l;  // [layer]: t_ref[ip[layer]], the reference temperature of the layer's pressure bracket (host)
  // repwvl, per column, written at the start of every iteration (k_pre / post_column, repwvl_image_store): the
  // complete column image of the flattened kernel -- temperature brackets, interpolation coefficients with the mixing
  // ratios folded in, T, 1/T, cloud -- which that kernel (2.3 blocks touch a column) only copies
  void *img;          // [ncol] RepwvlPairLds: the column image the flattened kernel copies into LDS
  const double *rw_xc, *rw_sc;  // per wavelength: h*c/(lambda*kB) and pnum/lam5/1e9 (planck_mono_fast), host-formed
  const int *ip;
  int n_tpert, n_species, n_p, n_wvl;
  unsigned species_mask;  // species with a non-zero mixing ratio somewhere (the reference-form getter and kernels)
  // Production kernels: species whose mixing-ratio profile is THE SAME IN EVERY COLUMN (O3, N2O, CH4 in a CO2 sweep) are
  // summed into one table row set on the host, xp[l][t][merged_slot][w] = sum_s vmr_s[l] * xp[l][t][s][w]
  // (derive_repwvl_layers): the kernels then load the bracket's corner pair for H2O, for every species that does vary,
  // and for the merged one (coefficient: the interpolation weight alone): 3 x 20 pairs per point instead of 5 x 20.
  unsigned prod_mask;     // species the production optical-depth stage visits ...
  int nact;               // ... their number and their list, ascending, padded to 9 entries with species outside the
  int act[RCM_NSPEC_REPWVL];  // mask (their image coefficients are 0): the stage runs WITHOUT a branch per species
  unsigned long long act_nibbles;  // the same list, 4 bits per entry (a per-lane lookup without an indexed array)
  int merged_slot;        // -1: nothing merged
};


// ---------------------------------------------------------------------------------------------
// lbl optical depth of band b (DESIGN.md "lbl composition"; format lbl.arts/README:5-17):
// tau_l = sum_s scale[s][l] * tab_s[b][l] (+ grey cloud, main.cpp:266-274).
// sc = the column's [5][NL] multipliers in LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double lbl_tau_layer(const Dev &d, const double *sc, long b, int l, int cloud_layer,
                                                double cloud_tau) {
  double acc = 0.0;
#pragma unroll
  for (int s = 0; s < RCM_NSPEC_LBL; ++s)
    acc = fma(sc[s * NL + l], d.lbl_tab[(size_t)(s * NL + l) * d.lbl_ld + b], acc);
  if (l == cloud_layer) acc += cloud_tau;
  return acc;
}

// ---------------------------------------------------------------------------------------------
// repwvl optical depth at wavelength w: read_tau, repwvl_thermal.cpp:197-248.
// Per column-layer (wave prologue, LDS): it = LowerPos(t_ref[ip]+t_pert, T) and delT (:229-239).
// Per layer (host, once): ip = LowerPos(p_grid, midP), delP, numDens (:201-202, 226, 240).
// ---------------------------------------------------------------------------------------------
struct RepwvlLds {
  double T[NL + 1];          // Tlayer, T_surface
  double delT[NL];
  double vmr[RCM_NSPEC_REPWVL * NL];
  int it[NL];
};

__device__ __forceinline__ void repwvl_prologue(const Dev &d, int col, int lane, RepwvlLds &s) {
  if (lane < NL) {
    s.T[lane] = d.T[(size_t)col * NL + lane];
    const double midT = d.Ttau[(size_t)col * NL + lane];  // :217 (prop_at_Lev == 0)
    const int ip = d.ip[lane];
    const double tr = d.t_ref[ip];
    const double *tp = d.t_pert;
    const int it = lower_pos_dev([&](int k) { return tr + tp[k]; }, d.n_tpert, midT);  // :229-232
    const double t0 = tr + tp[it], t1 = tr + tp[it + 1];
    s.it[lane] = it;
    s.delT[lane] = (midT - t0) / (t1 - t0);  // :239
  }
  if (lane == NL) s.T[NL] = d.Tsfc[col];
  for (int i = lane; i < d.ngas * NL; i += 64) s.vmr[i] = d.gas[(size_t)col * d.ngas * NL + i];
}

// One layer of read_tau's interpolation for the lane's wavelength (:210-246).
__device__ __forceinline__ double repwvl_tau_layer(const Dev &d, const RepwvlLds &s, int w, int l, int cloud_layer,
                                                   double cloud_tau) {
  const size_t ld = d.n_wvl;
  const int it = s.it[l], ip = d.ip[l];
  const double delT = s.delT[l], delP = d.delP[l];
  double acc = 0.0;
#pragma unroll
  for (int sp = 0; sp < RCM_NSPEC_REPWVL; ++sp) {
    if (!((d.species_mask >> sp) & 1u)) continue;  // vmr == 0 everywhere: the term is +0
    const double *x0 = d.xsec + (((size_t)it * d.n_species + sp) * d.n_p + ip) * ld + w;
    const double *x1 = d.xsec + (((size_t)(it + 1) * d.n_species + sp) * d.n_p + ip) * ld + w;
    const double c0 = x0[0];                      // :235
    const double cT = x1[0] - c0;                 // :236
    const double cP = x0[ld] - c0;                // :237
    const double cPT = x1[ld] - cP - cT - c0;     // :238
    const double xs = c0 + cT * delT + cP * delP + cPT * delT * delP;  // :241
    acc += xs * s.vmr[sp * NL + l];               // :244
  }
  acc *= d.numDens[l];                            // :246
  if (l == cloud_layer) acc += cloud_tau;         // main.cpp:266-274
  return acc;
}


// ---------------------------------------------------------------------------------------------
// Thermal RT, lbl spectra, LANE-PAIR layout (the production kernel).
//
// Two adjacent lanes share one band and split the COLUMN: the even lane owns layers 0..9, the odd
// lane layers 19..10 (position k of a lane = layer k, resp. 19-k).  Each lane forms the optical
// depths, Planck values and transmissions of its own ten layers only -- nothing transcendental is
// computed twice and nothing but one radiance crosses lanes:
//   phase 1 (k = 0..9):  even lane runs the reference's DOWN recurrence through layers 0..9
//                        (main.cpp:306-309, levels 1..10); odd lane runs the UP recurrence through
//                        layers 19..10 (main.cpp:311-314, levels 19..10), starting from the surface.
//   hand-over:           the two radiances at level 10 swap lanes (one DPP quad-permute per half).
//   phase 2 (k = 9..0):  even lane continues the UP recurrence through layers 9..0 (levels 9..0);
//                        odd lane continues the DOWN recurrence through layers 10..19 (levels 11..20).
// Both lanes execute one instruction stream:  L = B[k] + t[k]*(L - B[k]);  acc += w*L.
// Per lane: 21 accumulators, 10 transmissions, 10 Planck values, 10 optical depths.
// ---------------------------------------------------------------------------------------------
// Which lanes form a pair: a policy of the lane-pair kernels.
// Pair<0> pairs ADJACENT lanes (2i, 2i+1): the hand-over of the radiance at level 10 is a DPP quad-permute (two VALU
//   instructions, no latency to speak of), but a wave's 8-byte table load alternates between two table rows from
//   lane to lane, which the texture path handles at about half speed (tools/ubench/tcp_pattern.hip, L2-resident
//   table, 12 waves per CU: 34 cycles per wave-load against 17.5 when 16 consecutive lanes read one row).
// Pair<1> alternates ROWS OF 16 LANES: lanes 0..15 and 32..47 are the "even" halves (layers 0..9) of the spectral
//   points 0..15 and 16..31 of a pass, lanes 16..31 and 48..63 the "odd" halves (layers 19..10) of the same points;
//   partners are 16 lanes apart and the hand-over is a ds_swizzle (LDS crossbar, no LDS memory, no VALU
//   instruction, but ~100 cycles of latency inside the dependent chain of an angle).
// Pair<2> alternates HALF-ROWS OF 8 LANES (lanes 0..7 "even", 8..15 "odd" halves of the points 0..7, and so on): 8
//   consecutive lanes read 64 contiguous bytes of one row -- the texture path takes that as well as rows of 16
//   (17.2 cycles) -- and partners are 8 lanes apart inside one DPP row: the hand-over is row_ror:8, a DPP move again.
// Measured on one box (tools/ab_libs.py), lbl kernel per 2 048-column launch / flattened repwvl kernel per step:
//   Pair<0> 67.35 ms / 345.1 us,  Pair<1> 69.08 ms / 339.0 us (the swizzle's latency sits in the angle chain),
//   Pair<2> 66.76 ms / 335.9 us: both kernels use Pair<2> (RCM_LBL_LAYOUT / RCM_RW_LAYOUT).
#ifndef RCM_LBL_LAYOUT
#define RCM_LBL_LAYOUT 2
#endif
#ifndef RCM_RW_LAYOUT
#define RCM_RW_LAYOUT 2
#endif

// v of the lane OFF lanes away (xor): DPP inside a row of 16 lanes, ds_swizzle / ds_bpermute across rows.
template <int OFF>
__device__ __forceinline__ double lane_xor(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (OFF == 1) {  // quad_perm [1,0,3,2]
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, true);
  } else if constexpr (OFF == 2) {  // quad_perm [2,3,0,1]
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, true);
  } else if constexpr (OFF == 16) {  // ds_swizzle, bit-mask mode: lane' = lane ^ 0x10 within each group of 32
    lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
    hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
  } else {
    return __shfl_xor(v, OFF, 64);
  }
  return __hiloint2double(hi, lo);
}

template <int LAYOUT>
struct Pair;

template <>
struct Pair<0> {  // adjacent lanes
  static __device__ __forceinline__ int parity(int lane) { return lane & 1; }
  static __device__ __forceinline__ int index(int lane) { return lane >> 1; }  // 0..31
  static __device__ __forceinline__ int lane_of(int index, int parity) { return 2 * index + parity; }
  static __device__ __forceinline__ double swap(double v) { return lane_xor<1>(v); }
  // Sum of v over the lane pairs whose index differs in the low BITS bits only (same parity): a butterfly.
  template <int BITS>
  static __device__ __forceinline__ double sum(double v) {
#pragma unroll
    for (int b = BITS - 1; b >= 0; --b) v += __shfl_xor(v, 2 << b, 64);
    return v;
  }
};

template <>
struct Pair<1> {  // rows of 16 lanes
  static __device__ __forceinline__ int parity(int lane) { return (lane >> 4) & 1; }
  static __device__ __forceinline__ int index(int lane) { return (lane & 15) | ((lane >> 5) << 4); }
  static __device__ __forceinline__ int lane_of(int index, int parity) { return (index & 15) | (parity << 4) | ((index >> 4) << 5); }
  static __device__ __forceinline__ double swap(double v) { return lane_xor<16>(v); }
  // Within a row of 16 lanes the levels 4 and 8 are row rotations (only the lanes that store read the result:
  // the order of the sum is fixed).
  template <int BITS>
  static __device__ __forceinline__ double sum(double v) {
    if constexpr (BITS >= 1) v += lane_xor<1>(v);
    if constexpr (BITS >= 2) v += lane_xor<2>(v);
    if constexpr (BITS == 3) v += __shfl_xor(v, 4, 64);  // half a row: a rotation would reach into the other half
    if constexpr (BITS >= 4) {  // row_ror:4 -- after the quad levels every lane holds its quad's sum
      int lo = __double2loint(v), hi = __double2hiint(v);
      lo = __builtin_amdgcn_update_dpp(lo, lo, 0x124, 0xF, 0xF, true);
      hi = __builtin_amdgcn_update_dpp(hi, hi, 0x124, 0xF, 0xF, true);
      v += __hiloint2double(hi, lo);
      lo = __double2loint(v), hi = __double2hiint(v);  // row_ror:8
      lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xF, 0xF, true);
      hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xF, 0xF, true);
      v += __hiloint2double(hi, lo);
    }
    if constexpr (BITS >= 5) v += __shfl_xor(v, 32, 64);
    return v;
  }
};

template <>
struct Pair<2> {  // half-rows of 8 lanes alternate in parity; partners are 8 lanes apart IN THE SAME ROW of 16
  static __device__ __forceinline__ int parity(int lane) { return (lane >> 3) & 1; }
  static __device__ __forceinline__ int index(int lane) { return (lane & 7) | ((lane >> 4) << 3); }
  static __device__ __forceinline__ int lane_of(int index, int parity) { return (index & 7) | (parity << 3) | ((index >> 3) << 4); }
  static __device__ __forceinline__ double swap(double v) {  // row_ror:8 = xor 8 inside a row: one DPP move per half
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  }
  // index bits 0..2 = lane bits 0..2 (two quad-permutes and a half-row mirror: after the quad levels every lane of a
  // quad holds the quad's sum, so the mirror image 7-i is as good as i^4), bits 3, 4 = lane bits 4, 5
  template <int BITS>
  static __device__ __forceinline__ double sum(double v) {
    if constexpr (BITS >= 1) v += lane_xor<1>(v);
    if constexpr (BITS >= 2) v += lane_xor<2>(v);
    if constexpr (BITS >= 3) {  // row_half_mirror
      int lo = __double2loint(v), hi = __double2hiint(v);
      lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, true);
      hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, true);
      v += __hiloint2double(hi, lo);
    }
    if constexpr (BITS >= 4) v += lane_xor<16>(v);
    if constexpr (BITS >= 5) v += __shfl_xor(v, 32, 64);
    return v;
  }
};

using PairLbl = Pair<RCM_LBL_LAYOUT>;
using PairRw = Pair<RCM_RW_LAYOUT>;

constexpr int NH = NL / 2;  // layers per lane

// The recurrences in DIFFERENCE form.  The reference's L = (1-alpha)*L + alpha*B (main.cpp:307,312) with
// alpha = 1-t is L' - B = t*(L - B): carrying X = (radiance at a level) - (Planck value of the NEXT layer on the
// path) turns one crossing into ONE FMA whose addend does not depend on the angle,
//     X <- t[k]*X + dB[k],   dB[k] = B[k] - B[next layer on the path],
// and the flux sums take cw*X; the missing sum_i cw_i * B[next] is the same for every angle and is added
// once per spectral point (pair_flux_offsets).  2 FMAs per layer crossing and angle instead of 2.5 (the
// emission (1-t)*B per layer and angle is gone).  Positions: k = 0..9 along the lane's phase-1 direction,
// B[10] = the partner lane's B[9] (the layer across level 10), B[-1] = 0 (nothing beyond level 0 / level 20).
struct PairSrc {
  double dB[NH];  // phase 1 (k = 0..9) adds dB[k] = B[k] - B[k+1]; phase 2 (k = 9..1) adds B[k] - B[k-1] = -dB[k-1]
  double B0;      // phase 2, k = 0: B[0] - B[-1] = B[0]
  // X0 = Linit - B[0], the radiance entering the lane's first layer (main.cpp:300-301) minus its Planck value,
  // waits in the lane's LDS staging column (row NH) and is read once per angle: LDS reads are free beside a
  // saturated VALU, two VGPRs are not.
};

template <class P>
__device__ __forceinline__ void pair_source(const double (&B)[NH], double Linit, double wsum, PairSrc &src,
                                            double (*st)[64], int lane, double (&acc1)[NH + 1], double (&acc2)[NH]) {
  acc1[0] = fma(wsum, Linit, acc1[0]);  // the surface term of main.cpp:302, all angles at once
#if RCM_SWEEP_DIFF
  const double Bp = P::swap(B[NH - 1]);
#pragma unroll
  for (int k = 0; k < NH - 1; ++k) src.dB[k] = B[k] - B[k + 1];
  src.dB[NH - 1] = B[NH - 1] - Bp;
  src.B0 = B[0];
  st[NH][lane] = Linit - B[0];
  // what the difference form leaves out of the flux sums, for all angles at once (wsum = sum_i 2*pi*mu_i*dmu):
  // acc1[k+1] lacks wsum*B[k+1], acc2[k] lacks wsum*B[k-1]
#pragma unroll
  for (int k = 1; k < NH; ++k) acc1[k] = fma(wsum, B[k], acc1[k]);
  acc1[NH] = fma(wsum, Bp, acc1[NH]);
#pragma unroll
  for (int k = 1; k < NH; ++k) acc2[k] = fma(wsum, B[k - 1], acc2[k]);
#else
#pragma unroll
  for (int k = 0; k < NH; ++k) src.dB[k] = B[k];  // the emission form keeps the Planck values themselves
  src.B0 = 0.0;
  st[NH][lane] = Linit;
#endif
}

// One angle (odd angle number m) on the lane pair, then -- while m is divisible by 3 -- the angle m/3
// with the cubed transmissions (t_{m/3} = t_m^3).  Nested straight-line code (DEPTH levels; 3^4 = 81
// covers nangle <= 64) instead of a loop: a loop makes the compiler carry t[] through register copies.
template <class P, int DEPTH>
__device__ __forceinline__ void pair_chain(const Dev &d, const double (&t)[NH], const PairSrc &src,
                                           const double *x0, int m, double (&acc1)[NH + 1], double (&acc2)[NH]) {
  const double cw = d.ang[d.nangle + ((m - 1) >> 1)];
  double X = *x0;
#if RCM_SWEEP_DIFF
#pragma unroll
  for (int k = 0; k < NH; ++k) {  // main.cpp:306-309 (even lane) / :311-314 (odd lane)
    X = fma(t[k], X, src.dB[k]);
    acc1[k + 1] = fma(cw, X, acc1[k + 1]);
  }
  X = P::swap(X);  // level 10: the down radiance moves to the odd lane, the up radiance to the even lane
#pragma unroll
  for (int k = NH - 1; k >= 1; --k) {
    X = fma(t[k], X, -src.dB[k - 1]);
    acc2[k] = fma(cw, X, acc2[k]);
  }
  X = fma(t[0], X, src.B0);
  acc2[0] = fma(cw, X, acc2[0]);
#else
  // emission form: the layer's own emission (1-t)*B = alpha*B of main.cpp:307,312 is formed once per layer and
  // angle and both crossings are L = t*L + e
  double e[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) e[k] = fma(-t[k], src.dB[k], src.dB[k]);
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    X = fma(t[k], X, e[k]);
    acc1[k + 1] = fma(cw, X, acc1[k + 1]);
  }
  X = P::swap(X);
#pragma unroll
  for (int k = NH - 1; k >= 0; --k) {
    X = fma(t[k], X, e[k]);
    acc2[k] = fma(cw, X, acc2[k]);
  }
#endif
  if constexpr (DEPTH > 0) {
    if (m % 3 == 0) {
      double t3[NH];
#pragma unroll
      for (int k = 0; k < NH; ++k) t3[k] = t[k] * t[k] * t[k];
      pair_chain<P, DEPTH - 1>(d, t3, src, x0, m / 3, acc1, acc2);
    }
  }
}

// All angles of one spectral point on a lane pair (main.cpp:297).  mu_i = (2i+1)/(2*nangle): with odd
// angle numbers n = 2i+1 <= N = 2*nangle-1 the transmission is t_n = exp(-tau*2*nangle/n), so
//   * t_{n/3} = t_n^3: with CUBE only the "roots" n > N/3 need their own evaluation (pair_chain does the rest);
//   * two roots 3m and 5m can share ONE exponential u = exp(-tau*2*nangle/(15m)) at a fictitious angle:
//     t_5m = u^3, t_3m = u^5 (30 angles: 21/35, 27/45, 33/55).
// The host lays this out as a schedule behind the angle table (rcm_create): entries {a1, code} with
// a1 = -log2(e)/mu of the exponential to evaluate and code = 4*n + mode, mode 0: plain root n, 1: a1 is the
// generator of (n = 3m, 5m), u^3 is parked in the lane's LDS column st (free during the angle loop),
// 2: n = 5m, transmissions come back from st.
template <class P, int CUBE>
__device__ __forceinline__ void pair_all_angles(const Dev &d, const double (&tau)[NH], const double (&B)[NH],
                                                double Linit, double (*st)[64], int lane, double (&acc1)[NH + 1],
                                                double (&acc2)[NH]) {
  const int na = d.nangle;
  PairSrc src;
  pair_source<P>(B, Linit, d.ang[2 * na], src, st, lane, acc1, acc2);
  if constexpr (!CUBE) {
    for (int n = 2 * na - 1; n >= 1; n -= 2) {
      const double a1 = -d.ang[(n - 1) >> 1] * 0x1.71547652b82fep+0;
      double t[NH];
#pragma unroll
      for (int k = 0; k < NH; ++k) t[k] = exp_trans(tau[k], a1);  // main.cpp:209
      pair_chain<P, 0>(d, t, src, &st[NH][lane], n, acc1, acc2);
    }
  } else {
    const double *sched = d.ang + 2 * na + 1;
    for (int s = 0; s < d.nsched; ++s) {
      const double a1 = sched[2 * s];
      const int code = (int)__double_as_longlong(sched[2 * s + 1]);
      double t[NH];
      if ((code & 3) == 2) {
#pragma unroll
        for (int k = 0; k < NH; ++k) t[k] = st[k][lane];
      } else {
#pragma unroll
        for (int k = 0; k < NH; ++k) t[k] = exp_trans(tau[k], a1);  // main.cpp:209
        if (code & 1) {
#pragma unroll
          for (int k = 0; k < NH; ++k) {
            const double u2 = t[k] * t[k], u3 = u2 * t[k];
            st[k][lane] = u3;
            t[k] = u3 * u2;
          }
        }
      }
      pair_chain<P, 4>(d, t, src, &st[NH][lane], code >> 2, acc1, acc2);
    }
  }
}

// Reduce the accumulators over the 32 lanes of each parity and store the record.
// even lane: acc1[j] = E_down[j] (j = 0..10), acc2[k] = E_up[k] (k = 0..9)
// odd lane:  acc1[j] = E_up[20-j],           acc2[k] = E_down[20-k]
template <class P>
__device__ __forceinline__ void store_pair_partial(const Dev &d, int col, int chunk, int lane, bool valid,
                                                   double (&acc1)[NH + 1], double (&acc2)[NH]) {
  double *oEd = d.partial + ((size_t)col * d.nchunk + chunk) * PSTRIDE, *oEu = oEd + NLEV;
  const bool w0 = valid && lane == P::lane_of(0, 0), w1 = valid && lane == P::lane_of(0, 1);
#pragma unroll
  for (int j = 0; j <= NH; ++j) {
    const double v = P::template sum<5>(acc1[j]);
    if (w0) oEd[j] = v;
    if (w1) oEu[NL - j] = v;
  }
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    const double v = P::template sum<5>(acc2[k]);
    if (w0) oEu[k] = v;
    if (w1) oEd[NL - k] = v;
  }
}

// LDS image of one column, once per lane parity p (0: layers 0..9, 1: layers 19..10)
struct PairLds {
  double T[2][NH + 1];    // T[p][k] = Tlayer[layer(p,k)], k = NH: T_surface
  double tp[2][NH + 1];   // sigdpi * T^4            (cplkavg.cpp:237)
  double pc[2][NH + 1];   // sigdpi * T^4 * conc     (cplkavg.cpp:181)
  double rT[2][NH + 1];   // 1/T
  double gas[2][RCM_NSPEC_LBL][NH];
  double cloud[2][NH];    // grey cloud optical depth of the layer (main.cpp:266-274), 0 elsewhere
};

// Fills the LDS image of column `col` for the lane-pair lbl kernels.  Called by a whole wave.
__device__ __forceinline__ void lbl_pair_image(const Dev &d, int col, int lane, PairLds &s) {
  // lanes 0..21 fill (parity, position) = (lane / 11, lane % 11)
  if (lane < 2 * (NH + 1)) {
    const int q = lane / (NH + 1), k = lane - q * (NH + 1);
    const int layer = q ? NL - 1 - k : k;
    const double t = (k == NH) ? d.Tsfc[col] : d.T[(size_t)col * NL + layer];
    const double tp = d.sigdpi * pow(t, 4.0);
    s.T[q][k] = t;
    s.tp[q][k] = tp;
    s.pc[q][k] = tp * kConc;
    s.rT[q][k] = 1.0 / t;
  }
  for (int i = lane; i < 2 * RCM_NSPEC_LBL * NH; i += 64) {
    const int q = i / (RCM_NSPEC_LBL * NH), r = i - q * (RCM_NSPEC_LBL * NH);
    const int sp = r / NH, k = r - sp * NH;
    const int layer = q ? NL - 1 - k : k;
    s.gas[q][sp][k] = d.gas[((size_t)col * RCM_NSPEC_LBL + sp) * NL + layer];
  }
  if (lane < 2 * NH) {
    const int q = lane / NH, k = lane - q * NH;
    s.cloud[q][k] = ((q ? NL - 1 - k : k) == d.cloud_layer[col]) ? d.cloud_tau[col] : 0.0;
  }
}

// Table addressing of the lane-pair lbl kernels: one scalar base per species + a 32-bit byte offset per lane
// (the five tables are < 4 GiB together, rcm_set_lbl_tables checks), so a load costs no vector address
// arithmetic.  row0 = the lane's first row, rstep = the row step: layers 0,1,.. (even lane) or 19,18,.. (odd).
struct LblAddr {
  const char *sbase[RCM_NSPEC_LBL];
  unsigned row0, rstep;
};

__device__ __forceinline__ LblAddr lbl_pair_addr(const Dev &d, int p) {
  LblAddr a;
#pragma unroll
  for (int sp = 0; sp < RCM_NSPEC_LBL; ++sp) a.sbase[sp] = (const char *)(d.lbl_tab + (size_t)sp * NL * d.lbl_ld);
  a.row0 = (unsigned)((p ? NL - 1 : 0) * d.lbl_ld) * 8u;
  a.rstep = (unsigned)d.lbl_ld * 8u * (p ? -1 : 1);
  return a;
}

// Optical depths of the lane's ten layers for the band at byte offset `off` (lbl composition, DESIGN.md 4:
// tau = sum_s gas[s][l] * tab_s[b][l], + the grey cloud of main.cpp:266-274), clamped once, into the lane's
// staging column st[0..9][lane].  THE optical-depth arithmetic of the production kernel; k_tau_pair dumps it.
__device__ __forceinline__ void lbl_pair_tau(const PairLds &s, int p, const char *const (&sbase)[RCM_NSPEC_LBL],
                                             unsigned off, unsigned rstep, double (*st)[64], int lane) {
#pragma unroll 2
  for (int k = 0; k < NH; ++k, off += rstep) {  // rolled: bounds the loads (and VGPRs) in flight
    double a = 0.0;
#pragma unroll
    for (int sp = 0; sp < RCM_NSPEC_LBL; ++sp) a = fma(s.gas[p][sp][k], *(const double *)(sbase[sp] + off), a);
    st[k][lane] = clamp_tau(a + s.cloud[p][k]);  // + 0.0 leaves the other layers as they are
  }
}

// Reciprocals of the column's coldest and warmest temperature (for the once-per-band Planck regime test).
__device__ __forceinline__ void lbl_pair_trange(const PairLds &s, double &rTmin, double &rTmax) {
  rTmin = 0.0, rTmax = 1e300;
  for (int k = 0; k <= NH; ++k) {
    rTmin = fmax(rTmin, fmax(s.rT[0][k], s.rT[1][k]));
    rTmax = fmin(rTmax, fmin(s.rT[0][k], s.rT[1][k]));
  }
  if (!(rTmax > 0.0)) rTmax = 0.0;  // T < 1e-4 or NaN somewhere: never the fast path
}

// Band Planck source of the lane's ten layers and of the surface (cplkavg.cpp:124-243) for band b, into the
// lane's staging column st[0..NH][lane].
__device__ __forceinline__ void lbl_pair_planck(const Dev &d, const PairLds &s, int p, long b, double rTmin, double rTmax,
                                                double (*st)[64], int lane) {
  const double wnumhi = d.lbl_wnum[b];      // cplkavg.cpp:141
  const double wnumlo = d.lbl_wnum[b + 1];  // cplkavg.cpp:142
  const double c2lo = 1.438786 * wnumlo, c2hi = 1.438786 * wnumhi;  // cplkavg.cpp:152-153
  // gate of the fast path only (the slow path repeats cplkavg.cpp:155 itself): a shade inside
  // (wnumhi - wnumlo) / wnumhi < 1.e-2, without the division
  const bool narrow = wnumhi - wnumlo < 0.999e-2 * wnumhi;
  // Is every one of the column's 21 temperatures in the narrow-band regime for this lane's band
  // (hh = (c2hi-c2lo)/T < kHhFast, eps < v < 700)?  Decided once per band from Tmin/Tmax with a 0.1 %
  // margin; when the whole wave agrees the Planck loop runs branch-free.
  const bool lane_fast = narrow && (c2hi - c2lo) * rTmin < 0.999 * kHhFast && c2hi * rTmin < 699.0 &&
                         c2lo * rTmax > 1e-15;
  if (RCM_PLK_HOIST && __all(lane_fast)) {
    const double c2mid = 0.5 * (c2lo + c2hi), dc2 = c2hi - c2lo;
#pragma unroll RCM_PAIR_PLK_UNROLL
    for (int k = 0; k <= NH; ++k) {
      const double t = s.T[p][k], rT = s.rT[p][k];
#if RCM_PLK_XM_FAST  // x at the band centre as c2mid * (1/T): two roundings instead of the correctly rounded quotient
      (void)t;
      st[k][lane] = cplkavg_fast_mid(c2mid * rT, dc2 * rT, s.pc[p][k]);
#else
      st[k][lane] = cplkavg_fast_mid(div_by_t(c2mid, t, rT), dc2 * rT, s.pc[p][k]);
#endif
    }
  } else {
#pragma unroll 1
    for (int k = 0; k <= NH; ++k) {
      double Bk;
      if (!(narrow && cplkavg_fast(c2lo, c2hi, s.T[p][k], s.rT[p][k], s.pc[p][k], &Bk)))
        Bk = cplkavg_slow(wnumlo, wnumhi, s.T[p][k], s.tp[p][k]);
      st[k][lane] = Bk;
    }
  }
}

// BR = 1 (few columns x many chunks, where a column has hundreds of records: BASELINE configs[1]): a block owns WPB
// CONSECUTIVE chunks of ONE column and writes ONE record -- its waves' sums meet in LDS (three DPP levels inside the
// half-rows of 8 lanes, then the half-rows and the waves in a fixed order) -- so that a column has nchunk / WPB records
// and k_post sums them directly: the k_reduce_partials launch between the two is gone (one column x 131 071 bands:
// 2 048 -> 512 records, 4.8 us of a 73 us step) and the wave's tail is 21 x 3 DPP steps instead of 21 x 5 butterfly
// levels, two of which cross rows.
// BR = 2 (small ensembles, launches of <= 65 536 waves): the plain item mapping and records of BR = 0; like BR = 1 the
// phases around the angle loop run at wave priority 3 and the loop at 0 (-4 % at 2 columns ... -0.8 % at 64; the
// 10 000-column launch, whose waves live for 150 passes, does not care and keeps BR = 0).
template <int CUBE, int BR = 0>
__global__ __launch_bounds__(64 * WPB, RCM_PAIR_LB) void k_rt_lbl_pair(Dev d) {
  __shared__ PairLds s_col[WPB];
  __shared__ double s_stage[WPB][NH + 1][64];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // work item = (chunk, column), column fastest: the waves of a block and neighbouring blocks share
  // the chunk (its table tile stays in L2); a small ensemble still fills every wave slot
  const long nitem = (long)d.ncol * d.nchunk;
  constexpr bool REC = BR == 1;  // block records (+ flipped pass order); BR >= 1: phase priorities
  const long item_raw = REC ? ((long)(blockIdx.x / (unsigned)d.ncol) * WPB + wib) * d.ncol + blockIdx.x % (unsigned)d.ncol
                            : (long)blockIdx.x * WPB + wib;
  const bool valid = item_raw < nitem;
  const long item = valid ? item_raw : nitem - 1;
  const int chunk = (int)(item / d.ncol);
  const int col = (int)(item - (long)chunk * d.ncol);
  const int p = PairLbl::parity(lane);
  PairLds &s = s_col[wib];
  double(*st)[64] = s_stage[wib];

  RCM_PT_STAMP(0);
  int lane_p = lane;
  if constexpr (BR) RCM_PRIO_START(lane_p);
  lbl_pair_image(d, col, lane_p, s);
  __syncthreads();
  RCM_PT_STAMP(1);
  [[maybe_unused]] int pass = 0;  // (stamps only)

  double rTmin, rTmax;
  lbl_pair_trange(s, rTmin, rTmax);

  // even lane: acc1[j] = E_down[j] (j = 0..10), acc2[k] = E_up[k] (k = 0..9)
  // odd lane:  acc1[j] = E_up[20-j] (j = 0..10), acc2[k] = E_down[20-k] (k = 0..9)
  double acc1[NH + 1], acc2[NH];
#pragma unroll
  for (int j = 0; j <= NH; ++j) acc1[j] = 0.0;  // main.cpp:326-327
#pragma unroll
  for (int j = 0; j < NH; ++j) acc2[j] = 0.0;

  const long b0 = (long)chunk * d.chunk_len;
  const long b1 = (REC && !valid) ? b0 : min(b0 + (long)d.chunk_len, (long)d.nspec);  // (REC: a wave past the last chunk adds zeros)
  const LblAddr ad = lbl_pair_addr(d, p);

  // main.cpp:329: 32 bands per wave pass.  BR (one column): on odd iterations (Dev::flip) the passes run last first.
  const long bf = b0 + PairLbl::index(lane);
  const long blast = bf + ((b1 - 1 - bf) & ~31L);  // the lane's last band (< bf when it has none)
  const long bstep = (REC && d.flip) ? -32 : 32;
  for (long b = (REC && d.flip) ? blast : bf; REC ? (b >= bf && b < b1) : (b < b1); b += bstep) {
    // ---- optical depths of the lane's ten layers (lbl composition + cloud, main.cpp:266-274)
    double tau[NH];
    lbl_pair_tau(s, p, ad.sbase, ad.row0 + (unsigned)b * 8u, ad.rstep, st, lane);
#pragma unroll
    for (int k = 0; k < NH; ++k) tau[k] = st[k][lane];
    RCM_PT_STAMP(2 + 3 * (pass < 19 ? pass : 19));
    // ---- band Planck source of the lane's ten layers and of the surface (cplkavg.cpp:124-243)
    lbl_pair_planck(d, s, p, b, rTmin, rTmax, st, lane);
    double B[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) B[k] = st[k][lane];
    const double Linit = p ? st[NH][lane] : 0.0;  // main.cpp:300-301
    RCM_PT_STAMP(3 + 3 * (pass < 19 ? pass : 19));
    if constexpr (BR) {
      // one column: the short latency-bound phases of a wave (image, loads, Planck, reduction) run at wave priority 3,
      // its angle loop at 0 -- they no longer queue behind the partner wave's arithmetic (RT 56.4 - 58.0 -> 54.9 - 55.1 us,
      // three rounds of an A/B; the 10 000-column kernel, whose waves live for 150 passes, does not care: round 2)
      double Lin = Linit;
      RCM_PRIO_LO(Lin);
      pair_all_angles<PairLbl, CUBE>(d, tau, B, Lin, st, lane, acc1, acc2);
      RCM_PRIO_HI(acc1[0]);
    } else {
      pair_all_angles<PairLbl, CUBE>(d, tau, B, Linit, st, lane, acc1, acc2);
    }
    RCM_PT_STAMP(4 + 3 * (pass < 19 ? pass : 19));
    ++pass;
  }

  if constexpr (!REC) {
    store_pair_partial<PairLbl>(d, col, chunk, lane, valid, acc1, acc2);
  } else {
    static_assert(!REC || RCM_LBL_LAYOUT == 2, "the block reduction assumes Pair<2>: half-rows of 8 lanes alternate in parity");
    // [wave][half-row of 8 lanes][acc1[0..10], acc2[0..9]], aliased onto the staging rows (the angle loops are over)
    double(*red)[8][2 * NH + 1] = reinterpret_cast<double(*)[8][2 * NH + 1]>(&s_stage[0][0][0]);
    static_assert(sizeof(double) * WPB * 8 * (2 * NH + 1) <= sizeof(s_stage), "reduction rows fit the staging rows");
    __syncthreads();  // every wave of the block is done with its staging rows
#pragma unroll
    for (int j = 0; j <= NH; ++j) {
      const double v = PairLbl::template sum<3>(acc1[j]);
      if ((lane & 7) == 0) red[wib][lane >> 3][j] = v;
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const double v = PairLbl::template sum<3>(acc2[k]);
      if ((lane & 7) == 0) red[wib][lane >> 3][NH + 1 + k] = v;
    }
    __syncthreads();
    if (wib == 0 && lane < 2 * (2 * NH + 1)) {
      const int q = lane / (2 * NH + 1), v = lane - q * (2 * NH + 1);
      double acc = 0.0;
      for (int w = 0; w < WPB; ++w)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc += red[w][2 * g + q][v];  // half-row h holds parity h & 1 (Pair<2>)
      const int nrec = (d.nchunk + WPB - 1) / WPB;
      double *oEd = d.partial + ((size_t)col * nrec + blockIdx.x / (unsigned)d.ncol) * PSTRIDE, *oEu = oEd + NLEV;
      // q = 0 (layers 0..9): acc1[j] = E_down[j], acc2[k] = E_up[k];  q = 1: acc1[j] = E_up[20-j], acc2[k] = E_down[20-k]
      if (v <= NH) (q ? oEu[NL - v] : oEd[v]) = acc;
      else (q ? oEd[NL - (v - NH - 1)] : oEu[v - NH - 1]) = acc;
    }
  }
  RCM_PT_STAMP(63);
}

// ---------------------------------------------------------------------------------------------
// Thermal RT, repwvl reduced-wavelength set, lane-pair layout (production kernel for repwvl):
// read_tau's interpolation (repwvl_thermal.cpp:197-248) fused with radiative_transfer
// (main.cpp:320-344, monochromatic Planck main.cpp:186-204); tau never touches HBM.
// Same split as k_rt_lbl_pair: even lane = layers 0..9, odd lane = layers 19..10 of one wavelength.
// ---------------------------------------------------------------------------------------------
struct RepwvlPairLds {
  // ---- header, 592 B
  double T[2][NH + 1];   // T[p][k] = Tlayer[layer(p,k)], k = NH: T_surface
  double rT[2][NH + 1];  // 1/T
  double cloud[2][NH];   // grey cloud optical depth of the layer (main.cpp:266-274), 0 elsewhere
  unsigned xoff[2][NH];  // byte offset of xp[l][it][0][0] (pairs): the layer's temperature bracket (:229-232)
  // ---- coefficients, 320 B per SLOT j of Dev::act (species act[j]); the slots past Dev::nact hold zeros
  // Per layer and species the two coefficients of the temperature interpolation on the pressure-interpolated
  // table xp (derive_repwvl_layers): c[j][p][0] = numDens*(1-delT)*vmr, c[j][p][1] = numDens*delT*vmr, so that
  //   tau = sum_j c[j][p][0][k] * xp[l][it][act[j]][w] + c[j][p][1][k] * xp[l][it+1][act[j]][w]
  // is read_tau's numDens * sum_s vmr_s * (c0 + cT*delT + cP*delP + cPT*delT*delP) (:235-246) up to rounding
  // (rcm_get_tau, k_tau, keeps the reference's corner-difference form): one 16-byte load and 2 FMAs per species and
  // layer.  Slot-major, behind the header: what a kernel that visits NACT species needs is ONE contiguous prefix of
  // 592 + 320*NACT bytes (1 552 B for the three species of a CO2 sweep: two rounds of 16-byte pieces per wave
  // instead of four, and k_post_wave dirties 13 cache lines per column instead of 30).
  double c[RCM_NSPEC_REPWVL][2][2][NH];
  double wA[2][NH], wB[2][NH];  // numDens*(1-delT), numDens*delT (scratch of the in-place build; never copied)
};
static_assert(offsetof(RepwvlPairLds, c) == 592 && offsetof(RepwvlPairLds, c) % 16 == 0, "header: 37 pieces of 16 B");
constexpr int repwvl_image_prefix_bytes(int nact) { return 592 + 320 * nact; }

// The temperature bracket of read_tau for one layer (repwvl_thermal.cpp:229-239): it = LowerPos(t_ref[ip] +
// t_pert, midT) (:19-45: first change of sign between neighbours, else the last bracket) and delT; as the
// weights numDens*(1-delT), numDens*delT (numDens: :202, :246) and the byte offset of xp[layer][it][0][0].
// tp = the table's temperature perturbations in registers (<= 16; rcm_set_repwvl_tables checks).
constexpr int kMaxTpert = 16;

__device__ __forceinline__ void load_tpert(const Dev &d, double (&tp)[kMaxTpert]) {
#pragma unroll
  for (int j = 0; j < kMaxTpert; ++j) tp[j] = j < d.n_tpert ? d.t_pert[j] : 0.0;
}

__device__ __forceinline__ void repwvl_bracket(const Dev &d, const double (&tp)[kMaxTpert], int layer, double tr, double nd,
                                               double midT, double *wA, double *wB, unsigned *xoff) {
  int it = 0;
  double t0 = tr + tp[0], t1 = tr + tp[1];
  bool found = false;
  double gprev = tr + tp[0];
  int sign = sgn_dev(gprev - midT);
#pragma unroll
  for (int pidx = 1; pidx < kMaxTpert; ++pidx) {
    if (pidx < d.n_tpert && !found) {
      const double g = tr + tp[pidx];
      const int sg = sgn_dev(g - midT);
      if (sign != sg || pidx == d.n_tpert - 1) {
        found = true;
        it = pidx - 1;
        t0 = gprev, t1 = g;
      }
      sign = sg;
      gprev = g;
    }
  }
  const double dT = (midT - t0) / (t1 - t0);  // :239
  *wA = nd * (1.0 - dT), *wB = nd * dT;
  *xoff = (unsigned)((((size_t)layer * (d.n_tpert - 1) + it) * d.n_species) * d.n_wvl * 2 * sizeof(double));
}

// Start of an iteration, repwvl: a whole wave (lane l = layer l for l < NL) writes the image of column c for the
// optical depths of this iteration -- brackets at the temperatures midT, coefficients numDens*(1-delT)*vmr and
// numDens*delT*vmr, the adjusted temperatures and their reciprocals, T_surface, the cloud -- to d.img[c], in the layout
// the flattened kernel keeps in LDS.  h2o = this iteration's H2O mixing ratio of layer l (it may have been written to
// d.gas by this very wave a moment ago: it travels through LDS, not through HBM).  scr: 3*NL doubles of LDS owned by
// the wave, free at this point.
// W = lanes that work on this column (64: a whole wave; 32: half a wave, k_post_wave), lane = 0..W-1 among them.
template <int W = 64>
__device__ __forceinline__ void repwvl_image_store(const Dev &d, int c, int lane, double Tadj, double midT, double Tsfc,
                                                   double h2o, double *scr) {
  if (d.mode != MODE_REPWVL) return;
  RepwvlPairLds *g = reinterpret_cast<RepwvlPairLds *>(d.img) + c;
  double tp[kMaxTpert];
  load_tpert(d, tp);
  if (lane < NL) {
    const int l = lane, q = l < NH ? 0 : 1, k = q ? NL - 1 - l : l;
    double wA, wB;
    unsigned xoff;
    repwvl_bracket(d, tp, l, d.tref_l[l], d.numDens[l], midT, &wA, &wB, &xoff);
    scr[l] = wA, scr[NL + l] = wB, scr[2 * NL + l] = h2o;
    g->T[q][k] = Tadj;
    g->rT[q][k] = 1.0 / Tadj;
    g->xoff[q][k] = xoff;
    g->cloud[q][k] = (l == d.cloud_layer[c]) ? d.cloud_tau[c] : 0.0;
  } else if (lane == NL) {
    const double r = 1.0 / Tsfc;
    g->T[0][NH] = Tsfc, g->T[1][NH] = Tsfc;
    g->rT[0][NH] = r, g->rT[1][NH] = r;
  }
  __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations execute in program order
  // the coefficients of the ACTIVE species only (2 parities x nact x 10 layers: one round of the wave for three
  // species): the others are zero since the buffer was cleared (rcm_set_columns, whenever the species pattern changes)
  const int per_parity = d.nact * NH, items = 2 * per_parity;
  for (int i = lane; i < items; i += W) {
    const int q = i >= per_parity, r = i - q * per_parity;
    const int a = r / NH, k = r - a * NH;
    const int sp = (int)((d.act_nibbles >> (4 * a)) & 15u);
    const int layer = q ? NL - 1 - k : k;
    double vmr = scr[2 * NL + layer];  // H2O: this iteration's mixing ratio
    if (sp == d.merged_slot) vmr = 1.0;
    else if (sp != 0) vmr = d.gas[((size_t)c * RCM_NSPEC_REPWVL + sp) * NL + layer];
    g->c[a][q][0][k] = scr[layer] * vmr;  // :244 and :246 folded into the interpolation weights
    g->c[a][q][1][k] = scr[NL + layer] * vmr;
  }
}

// Fills the LDS image of column `col`: per (parity, position) the temperature bracket of read_tau
// (repwvl_thermal.cpp:229-239), the layer constants, the mixing ratios.  Called by a whole wave.
// Tl / Ttau [NL], Tsfc and h2o [NL] (species 0, the row the feedback rewrites) come from the caller: the
// column's rows in HBM, or the copies a fused step keeps in LDS.
__device__ __forceinline__ void repwvl_pair_image_from(const Dev &d, int col, int lane, RepwvlPairLds &s,
                                                       const double *Tl, const double *Ttau, double Tsfc,
                                                       const double *h2o) {
  // Everything the image needs from HBM is requested FIRST and together -- the mixing ratios (six per lane), the
  // temperature perturbations of the table (<= 16, wave-uniform), the layer's temperatures and constants -- so that
  // the build waits for memory once (it was a chain of five dependent round trips and cost the flattened kernel a
  // fifth of its time: ablation builds, profiles/README.md).
  constexpr int kPerLane = (2 * RCM_NSPEC_REPWVL * NH + 63) / 64;
  double vmr[kPerLane];
#pragma unroll
  for (int m = 0; m < kPerLane; ++m) {
    const int i = lane + 64 * m;
    const int q = i / (RCM_NSPEC_REPWVL * NH), r = i - q * (RCM_NSPEC_REPWVL * NH);
    const int a = min(r / NH, RCM_NSPEC_REPWVL - 1), k = r - (r / NH) * NH;  // slot a of Dev::act = species sp
    const int sp = (int)((d.act_nibbles >> (4 * a)) & 15u);
    const int layer = q ? NL - 1 - k : k;
    vmr[m] = (i < 2 * RCM_NSPEC_REPWVL * NH)
                 ? (!((d.prod_mask >> sp) & 1u) ? 0.0
                    : sp == 0 ? h2o[layer]
                    : sp == d.merged_slot ? 1.0 : d.gas[((size_t)col * RCM_NSPEC_REPWVL + sp) * NL + layer])
                 : 0.0;
  }
  double tp[kMaxTpert];
  load_tpert(d, tp);
  if (lane < 2 * (NH + 1)) {
    const int q = lane / (NH + 1), k = lane - q * (NH + 1);
    if (k == NH) {
      s.T[q][k] = Tsfc;
      s.rT[q][k] = 1.0 / Tsfc;
    } else {
      const int layer = q ? NL - 1 - k : k;
      const double Tlay = Tl[layer], midT = Ttau[layer];  // :217 (prop_at_Lev == 0)
      const double tr = d.tref_l[layer], nd = d.numDens[layer];
      const double cloud = (layer == d.cloud_layer[col]) ? d.cloud_tau[col] : 0.0;
      s.T[q][k] = Tlay;
      s.rT[q][k] = 1.0 / Tlay;
      repwvl_bracket(d, tp, layer, tr, nd, midT, &s.wA[q][k], &s.wB[q][k], &s.xoff[q][k]);
      s.cloud[q][k] = cloud;
    }
  }
  __builtin_amdgcn_wave_barrier();  // one wave builds an image: its LDS operations execute in program order
#pragma unroll
  for (int m = 0; m < kPerLane; ++m) {
    const int i = lane + 64 * m;
    if (i < 2 * RCM_NSPEC_REPWVL * NH) {
      const int q = i / (RCM_NSPEC_REPWVL * NH), r = i - q * (RCM_NSPEC_REPWVL * NH);
      const int a = r / NH, k = r - a * NH;
      s.c[a][q][0][k] = s.wA[q][k] * vmr[m];  // :244 and :246 folded into the interpolation weights
      s.c[a][q][1][k] = s.wB[q][k] * vmr[m];
    }
  }
}

// The image as k_pre / post_column stored it for this iteration (repwvl_image_store): a copy, 16 bytes per lane and
// load.  The flattened kernel brings every column's image into LDS 2.3 times per iteration (once per block that
// touches the column); search, divisions and products are done once per column.
template <int NACT>
__device__ __forceinline__ void repwvl_pair_image_copy(const Dev &d, int col, int lane, RepwvlPairLds &s) {
  static_assert(sizeof(RepwvlPairLds) % 16 == 0 && repwvl_image_prefix_bytes(NACT) % 16 == 0, "copied in 16-byte pieces");
  const double2 *src = reinterpret_cast<const double2 *>(reinterpret_cast<const RepwvlPairLds *>(d.img) + col);
  double2 *dst = reinterpret_cast<double2 *>(&s);
  constexpr int n16 = repwvl_image_prefix_bytes(NACT) / 16, rounds = (n16 + 63) / 64;  // header + the NACT visited slots
  double2 v[rounds];  // every load first, then every store, no branch in between (a lane past the end copies the last
#pragma unroll        // piece once more): ONE wait for memory, not one per round
  for (int m = 0; m < rounds; ++m) v[m] = src[min(lane + 64 * m, n16 - 1)];
#pragma unroll
  for (int m = 0; m < rounds; ++m) dst[min(lane + 64 * m, n16 - 1)] = v[m];
}

__device__ __forceinline__ void repwvl_pair_image(const Dev &d, int col, int lane, RepwvlPairLds &s) {
  repwvl_pair_image_from(d, col, lane, s, d.T + (size_t)col * NL, d.Ttau + (size_t)col * NL, d.Tsfc[col],
                         d.gas + (size_t)col * RCM_NSPEC_REPWVL * NL);
}

// Optical depths of the lane's ten layers at wavelength w (read_tau :210-246 + cloud, main.cpp:266-274), clamped
// once, into the lane's staging column st[0..9][lane].  THE optical-depth arithmetic of the production repwvl
// kernels; k_tau_pair dumps it.  s = the lane's column image.
// NACT: how many entries of Dev::act are visited (>= Dev::nact; the entries past nact add exactly +0).  A uniform
// BRANCH per species (an earlier version skipped the species outside the mask that way) makes every load wait for
// itself: 30 round trips to L2 in a row, a quarter of the wave's life (profiles/README.md); a fixed count lets the
// compiler send all loads of the stage first.
template <int NACT>
__device__ __forceinline__ void repwvl_pair_tau(const Dev &d, const RepwvlPairLds &s, int p, int w, double (*st)[64],
                                                int lane) {
  // xp[l][t][s][w] = {x(t), x(t+1)}: both temperature corners of a layer's bracket in one 16-byte load; the species
  // are n_wvl pairs apart: one scalar base + one 32-bit byte offset per (layer, species) address them (the table is
  // < 4 GiB, rcm_set_repwvl_tables checks)
  typedef double corner_pair __attribute__((ext_vector_type(2)));
  const unsigned ld16 = (unsigned)d.n_wvl * 16u;
  const char *bA = (const char *)d.xp;
#pragma unroll RCM_RW_TAU_UNROLL
  for (int k = 0; k < NH; ++k) {
#ifdef RCM_DBG_NOLOAD  // ablation builds only (tools/build_variants.py): every lane reads one address
    const unsigned vo = (s.xoff[p][k] + (unsigned)w * 16u) & 0u;
#else
    const unsigned vo = s.xoff[p][k] + (unsigned)w * 16u;
#endif
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < NACT; ++j) {
      const int sp = d.act[j];  // (wave-uniform)
      const corner_pair x = *(const corner_pair *)(bA + vo + (unsigned)sp * ld16);
      a = fma(s.c[j][p][0][k], x.x, fma(s.c[j][p][1][k], x.y, a));
    }
    st[k][lane] = clamp_tau(a + s.cloud[p][k]);  // main.cpp:266-274; + 0.0 leaves the other layers as they are
  }
}

// One wavelength on a lane pair: optical depths, monochromatic Planck (main.cpp:186-204), all angles.
// s = the lane's column image, st = the wave's staging column.
template <int CUBE, int NACT = RCM_NSPEC_REPWVL>
__device__ __forceinline__ void repwvl_pair_point(const Dev &d, const RepwvlPairLds &s, int p, int w, int lane,
                                                  double (*st)[64], double (&acc1)[NH + 1], double (&acc2)[NH]) {
  double tau[NH];
  RCM_PT_STAMP(20);
  repwvl_pair_tau<NACT>(d, s, p, w, st, lane);
#pragma unroll
  for (int k = 0; k < NH; ++k) tau[k] = st[k][lane];
  RCM_PT_STAMP(21);
  const double xc = d.rw_xc[w];  // h*c/(lambda*kB), main.cpp:70-72,191
  const double sc = d.rw_sc[w];  // weight*2*h*c^2/lambda^5/1e9, main.cpp:190-191
#ifdef RCM_DBG_NOPLANCK  // ablation builds only
  for (int k = 0; k <= NH; ++k) st[k][lane] = sc * s.rT[p][k] + xc;
#else
#pragma unroll RCM_RW_PLK_UNROLL
  for (int k = 0; k <= NH; ++k) st[k][lane] = planck_mono_fast(xc, sc, s.T[p][k], s.rT[p][k]);
#endif
  double B[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) B[k] = st[k][lane];
  const double Linit = p ? st[NH][lane] : 0.0;
  RCM_PT_STAMP(22);

#ifdef RCM_DBG_NOANGLES  // ablation builds only
#pragma unroll
  for (int k = 0; k < NH; ++k) acc1[k] += tau[k] * B[k] + Linit, acc2[k] += tau[k];
#else
  double Lin = Linit;
  RCM_PRIO_LO(Lin);
  pair_all_angles<PairRw, CUBE>(d, tau, B, Lin, st, lane, acc1, acc2);
  RCM_PRIO_HI(acc1[0]);
#endif
  RCM_PT_STAMP(23);
}

template <int CUBE>
__global__ __launch_bounds__(64 * WPB, RCM_PAIR_LB) void k_rt_repwvl_pair(Dev d) {
  __shared__ RepwvlPairLds s_col[WPB];
  __shared__ double s_stage[WPB][NH + 1][64];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // work item = (chunk, column), column fastest (see k_rt_lbl_pair)
  const long nitem = (long)d.ncol * d.nchunk;
  const long item_raw = (long)blockIdx.x * WPB + wib;
  const bool valid = item_raw < nitem;
  const long item = valid ? item_raw : nitem - 1;
  const int chunk = (int)(item / d.ncol);
  const int col = (int)(item - (long)chunk * d.ncol);
  const int p = PairRw::parity(lane);
  RepwvlPairLds &s = s_col[wib];
  double(*st)[64] = s_stage[wib];
  repwvl_pair_image(d, col, lane, s);
  __syncthreads();

  double acc1[NH + 1], acc2[NH];
#pragma unroll
  for (int j = 0; j <= NH; ++j) acc1[j] = 0.0;
#pragma unroll
  for (int j = 0; j < NH; ++j) acc2[j] = 0.0;
  const int w0 = chunk * d.chunk_len;
  const int w1 = min(w0 + d.chunk_len, d.nspec);
  for (int w = w0 + PairRw::index(lane); w < w1; w += 32) repwvl_pair_point<CUBE>(d, s, p, w, lane, st, acc1, acc2);
  store_pair_partial<PairRw>(d, col, chunk, lane, valid, acc1, acc2);
}

// ---------------------------------------------------------------------------------------------
// End of an iteration for column c, executed by ONE wave (lane l = layer l) once the column's fluxes E = {E_down[21],
// E_up[21]} stand in LDS: tail of radiative_transfer (main.cpp:337-341), calculate_timestep (:156-162), thermodynamics
// (:164-176), time (:581).  With next_pre the wave goes on with the START of the next iteration -- convective
// adjustment, equilibrium residual, water-vapour feedback (main.cpp:536-572; k_pre's work with first == 0).
// scr: 3*NL doubles of LDS scratch owned by the wave.  A wave's LDS operations execute in program order.
// ---------------------------------------------------------------------------------------------
struct PostIn {  // what post_column reads from HBM for (column c, layer l): loaded by the caller up front, so that
  double T, conv, solar, Tprev, rel_hum, player, h2o_ref;  // the latencies overlap the sum of the partial records
};

__device__ __forceinline__ PostIn post_load(const Dev &d, int c, int l, int next_pre) {
  PostIn in{};
  if (l < NL) {
    const size_t at = (size_t)c * NL + l;
    in.T = d.T[at];
    in.conv = d.conv[l];
    in.solar = d.solar_irr[c];
    if (next_pre) {
      in.Tprev = d.Tprev[at];
      in.rel_hum = d.rel_hum[at];
      in.player = d.player[l];
      in.h2o_ref = d.mode == MODE_LBL ? d.h2o_ref[l] : 1.0;
    }
  }
  return in;
}

// W = lanes that work on this column (64: a whole wave; 32: half a wave, two columns per wave in k_post_wave);
// l = 0..W-1 among them.  Lanes only ever exchange data with lanes of their own group.
template <int W = 64>
__device__ __forceinline__ void post_column(const Dev &d, int c, int l, const PostIn &in, const double *E, double *scr,
                                            int next_pre) {
  const double *Ed = E, *Eu = E + NLEV;
  double *s_dE = scr, *s_th = scr + NL, *s_sorted = scr + 2 * NL;
  const bool on = l < NL;
  const size_t at = (size_t)c * NL + (on ? l : 0);
  double dEl = 0.0;
  if (on) {
    dEl = Ed[l] - Ed[l + 1] + Eu[l + 1] - Eu[l];                     // main.cpp:338
    if (l == NL - 1) dEl += in.solar + Ed[NL] - Eu[NL];               // main.cpp:341
    s_dE[l] = dEl;
  }
  __builtin_amdgcn_wave_barrier();
  double Tn = 0.0;
  const double conv = in.conv;
  if (on) {
    double mx = s_dE[0];
#pragma unroll
    for (int i = 1; i < NL; ++i) mx = (mx < s_dE[i]) ? s_dE[i] : mx;  // signed max_element, main.cpp:157
    const double heat_cap = 1004.0 * d.dp * 100.0;               // c_air * dp * 100, main.cpp:157,169
    double dt = 5.0 / mx * heat_cap / 9.80665;                   // max_dT = 5 (float), main.cpp:82
    if (dt > 3600 * 12) dt = 3600 * 12;                          // main.cpp:158-160
    Tn = in.T + dEl * dt * 9.80665 / heat_cap;                   // main.cpp:169
    d.dE[at] = dEl;
    if (l == NL - 1) d.Tsfc[c] = Tn * conv;                      // main.cpp:173
    if (l == 0) {
      d.dt[c] = dt;
      d.time_h[c] += (float)dt / 3600;                           // main.cpp:581
    }
    if (!next_pre) d.T[at] = Tn;
  }
  if (!next_pre) return;
  double th = 0.0;
  if (on) {
    th = Tn * conv;                      // t_to_theta, main.cpp:123-129
    s_th[l] = th;
    s_sorted[l] = __builtin_nan("");     // a NaN among the inputs leaves ranks unassigned: the column stays NaN
  }
  __builtin_amdgcn_wave_barrier();
  if (on) {
    int rank = 0;
    for (int i = 0; i < NL; ++i) {
      const double ti = s_th[i];
      rank += (ti > th) || (ti == th && i < l);
    }
    s_sorted[rank] = th;                 // sort(theta, greater), main.cpp:538
  }
  __builtin_amdgcn_wave_barrier();
  double r = 0.0, Tadj = 0.0, h2o = 0.0;
  if (on) {
    const double T = s_sorted[l] / conv;  // theta_to_t, main.cpp:131-139
    d.T[at] = T;
    r = fabs(T - in.Tprev);
    d.Tprev[at] = T;
    d.Ttau[at] = T;
    Tadj = T;
    const double vmr = in.rel_hum * magnus_dev(T) / in.player;  // main.cpp:285-286
    h2o = vmr;
    double *gas = d.gas + (size_t)c * d.ngas * NL;
    if (d.mode == MODE_REPWVL)
      gas[0 * NL + l] = vmr;                    // H2O is read_tau's first species
    else
      gas[1 * NL + l] = vmr / in.h2o_ref;       // lbl: tau_H2O is linear in the mixing ratio
  }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {  // lanes 0..31 cover the 20 layers; a NaN layer makes the residual NaN
    const double o = __shfl_xor(r, off, 64);
    r = (o > r || o != o) ? o : r;
  }
  if (l == 0) d.resid[c] = r;
  // the next iteration's column image; T_surface was formed by lane NL-1 above; the scratch is free by now
  __builtin_amdgcn_wave_barrier();
  const int group0 = (int)(threadIdx.x & 63) - l;  // first lane of this column's group within the wave
  repwvl_image_store<W>(d, c, l, Tadj, Tadj, __shfl(Tn * conv, group0 + NL - 1, 64), h2o, scr);
}

// ---------------------------------------------------------------------------------------------
// The same, with the (column, wavelength) pairs FLATTENED over the lane pairs: wave i takes the 32
// consecutive items 32i..32i+31 of the column-major list, whichever columns they belong to.  100 wavelengths
// fill 100/128 of the lane pairs in the chunked kernel, 10 wavelengths 10/32; here every pass is full, and the
// work comes in 31 250 equal pieces for 10 000 x 100 (whole-column ownership leaves the last third of the run
// to a partly empty chip, measured with k_step_repwvl).  What a wave does besides its one pass is kept small:
//  * the column images are shared by the block: its 4 waves cover 128 consecutive items = at most 127/n_wvl + 2
//    columns, one image each, COPIED from HBM by the 4 waves in parallel (k_pre / k_post_wave store every column's
//    image once per iteration, repwvl_image_store; before: every wave built its own two);
//  * the per-column sums are masked (a lane contributes its sums to the column its item belongs to and zero to the
//    others): three DPP levels inside the half-rows of 8 lanes, then the four half-row sums of a parity meet in the
//    staging rows and one lane per output adds them -- not serial loops through LDS, no butterfly through ds_swizzle;
//  * 27 KB of LDS per block (Reduced100, CO2 sweep: 3 image prefixes of 1 552 B + 4 staging areas) and <= 128 VGPRs:
//    4 waves per SIMD.
// A column's partial record goes to slot ceil((first item of the segment - first item of the column)/32), so
// records never collide and the slots a column never uses stay zero (the buffer is zeroed once).
// d.nchunk = number of slots.
// ---------------------------------------------------------------------------------------------
#ifndef RCM_FLAT_WPB
#define RCM_FLAT_WPB 4  // waves per block of k_rt_repwvl_flat (they share the column images)
#endif
constexpr int FWPB = RCM_FLAT_WPB;
#ifndef RCM_FLAT_LB
#define RCM_FLAT_LB 4  // waves per SIMD k_rt_repwvl_flat is register-allocated for
#endif
template <int CUBE, int NACT>
__global__ __launch_bounds__(64 * FWPB, RCM_FLAT_LB) void k_rt_repwvl_flat(Dev d, int maxc) {
  extern __shared__ __align__(16) unsigned char s_dyn[];
  int lane = threadIdx.x & 63;
  RCM_PRIO_START(lane);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the block's columns: image sl at s_dyn + sl * kImg -- only the prefix this instantiation reads is kept in LDS
  // (1 552 of 3 792 B for NACT = 3: a 10-wavelength table's 14 images per block take 22 KB, not 53)
  constexpr int kImg = repwvl_image_prefix_bytes(NACT);
  auto img_at = [&](int sl) -> RepwvlPairLds & { return *reinterpret_cast<RepwvlPairLds *>(s_dyn + (size_t)sl * kImg); };
  double(*st)[64] = reinterpret_cast<double(*)[64]>(s_dyn + (size_t)maxc * kImg +
                                                    (size_t)wib * (NH + 1) * 64 * sizeof(double));
  const int p = PairRw::parity(lane);
  // 32-bit item arithmetic: ncol * n_wvl < 2^31 - 2^12 is a condition of this kernel (use_flat, rcm_capi.hip).  The
  // 64-bit forms cost four scalar divisions of ~130 instructions per wave and ONE PER LANE (130 of a wave's 4 490
  // VALU instructions); here: one 32-bit scalar division per wave, everything else from remainders < n_wvl + 128
  // by compare-and-count (n_wvl >= 8: a block spans at most 17 columns, a wave at most 5).
  const int nw = d.n_wvl;
  const int nflat = d.ncol * nw;
  const int blk0 = (int)blockIdx.x * (FWPB * 32);
  const int flat0 = blk0 + wib * 32;
  int cb_first = 0, ncb = 0, cb_rem = 0;  // first column the block touches, how many, and blk0's offset in the first
  if (blk0 < nflat) {
    cb_first = (int)((unsigned)blk0 / (unsigned)nw);
    cb_rem = blk0 - cb_first * nw;
    const int last_rem = cb_rem + min(FWPB * 32 - 1, nflat - 1 - blk0);  // offset of the block's last item from the
    ncb = 1;                                                             // start of column cb_first: < n_wvl + 128
    for (int t = nw; t <= last_rem; t += nw) ++ncb;                      // (wave-uniform: scalar, <= 16 rounds)
  }
  RCM_PT_STAMP(16);
#ifdef RCM_DBG_NOIMAGE  // ablation builds only: all-zero images (valid offsets, no work)
  for (int i = threadIdx.x; i < (int)((size_t)ncb * kImg / 8); i += blockDim.x)
    reinterpret_cast<double *>(s_dyn)[i] = 0.0;
#else
  for (int sl = wib; sl < ncb; sl += FWPB) repwvl_pair_image_copy<NACT>(d, cb_first + sl, lane, img_at(sl));
#endif
  __syncthreads();
  RCM_PT_STAMP(17);

  double acc1[NH + 1], acc2[NH];
#pragma unroll
  for (int j = 0; j <= NH; ++j) acc1[j] = 0.0;
#pragma unroll
  for (int j = 0; j < NH; ++j) acc2[j] = 0.0;
  const int flat = flat0 + PairRw::index(lane);
  // the lane's column and wavelength: offset from the start of column cb_first, minus whole columns (<= 16 of them)
  int col = -1, w = cb_rem + (flat - blk0), dc = 0;
#pragma unroll 1
  for (int t = nw; t <= w && dc < 16; t += nw) ++dc;  // dc = w / nw, w < n_wvl + 128, n_wvl >= 8
  w -= dc * nw;
  if (flat < nflat) {
    col = cb_first + dc;
    repwvl_pair_point<CUBE, NACT>(d, img_at(dc), p, w, lane, st, acc1, acc2);
  }
  if (flat0 >= nflat) return;  // (whole wave)

  // ---- segmented reduction: per column of the wave's 32 items, masked, in a fixed order
  // even lane: acc1[j] = E_down[j] (j = 0..10), acc2[k] = E_up[k] (k = 0..9)
  // odd lane:  acc1[j] = E_up[20-j],           acc2[k] = E_down[20-k]
#ifdef RCM_DBG_NOREDUCE  // ablation builds only
  if (acc1[3] == 1.2345) d.partial[0] = acc2[1];
  return;
#endif
  // columns of the wave's 32 items: those of its first and of its last live lane-pair
  int c_first = cb_first, c_last = cb_first;
  {
    const int r0 = cb_rem + wib * 32, r1 = cb_rem + min(flat0 + 31, nflat - 1) - blk0;
    for (int t = nw; t <= r1; t += nw) {
      c_first += t <= r0;
      ++c_last;
    }
  }
  // Per column of the wave's 32 items: the 21 sums of either parity, in a fixed order.  Three DPP levels sum inside
  // the half-rows of 8 lanes (Pair<2>: a half-row is one parity); the four half-row sums of a parity meet in the
  // staging rows, where lane e (0..41) adds the four that make up record element e and stores it: ONE store
  // instruction of 336 contiguous bytes per record (a store instruction costs the texture path the same whatever
  // its lane count), and no butterfly level through the LDS crossbar.
  static_assert(RCM_RW_LAYOUT == 2, "the reduction below assumes half-rows of 8 lanes");
  double *S = &st[0][0];  // [21 values][8 half-rows]: the staging columns are free after the angle loop
  // record element e = lane: E_down[e] (e <= 20) / E_up[e-21]; which parity holds it, and as which of its 21 values
  // (value v: acc1[v] for v <= 10, acc2[v-11] above; even parity: acc1[j] = E_down[j], acc2[k] = E_up[k]; odd parity:
  // acc1[j] = E_up[20-j], acc2[k] = E_down[20-k])
  int ep = 0, ev = 0;
  if (lane < PSTRIDE) {
    const int m = lane - NLEV;
    if (lane <= NH) ep = 0, ev = lane;
    else if (lane <= NL) ep = 1, ev = NH + 1 + (NL - lane);
    else if (m < NH) ep = 0, ev = NH + 1 + m;
    else ep = 1, ev = NL - m;
  }
  for (int c = c_first; c <= c_last; ++c) {
    const bool mine = col == c;
    const int seg0 = max(c * nw, flat0);
    const int slot = (seg0 - c * nw + 31) >> 5;
    double *out = d.partial + ((size_t)c * d.nchunk + slot) * PSTRIDE;  // E_down[21], E_up[21]
#pragma unroll
    for (int v = 0; v < NLEV; ++v) {
      const double x = PairRw::sum<3>(mine ? (v <= NH ? acc1[v <= NH ? v : 0] : acc2[v > NH ? v - NH - 1 : 0]) : 0.0);
      if ((lane & 7) == 0) S[v * 8 + (lane >> 3)] = x;  // lane >> 3 = 2 * (half-row of the parity) + parity
    }
    __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations execute in program order
    if (lane < PSTRIDE) {
      const double *q = S + ev * 8 + ep;
      out[lane] = ((q[0] + q[2]) + q[4]) + q[6];
    }
    __builtin_amdgcn_wave_barrier();
  }
  RCM_PT_STAMP(18);
}


// ---------------------------------------------------------------------------------------------
// Convective adjustment: theta = T*conv, sort descending, T = theta/conv (main.cpp:123-139,
// 536-540).  Odd-even transposition network on registers: static indices, no scratch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void adjust(const double *conv, double (&T)[NL], double (&th)[NL]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) th[i] = T[i] * conv[i];
#pragma unroll
  for (int r = 0; r < NL; ++r) {
#pragma unroll
    for (int i = (r & 1); i + 1 < NL; i += 2) {
      const double a = th[i], b = th[i + 1];
      th[i] = fmax(a, b);
      th[i + 1] = fmin(a, b);
    }
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) T[i] = th[i] / conv[i];
}

// One wave per column, lane l = layer l.  Start of loop iteration i (main.cpp:536-572) up to the
// optical-depth inputs: adjustment, and for i != 0 the water-vapour feedback (main.cpp:281-289).
// The sort of the 20 potential temperatures (main.cpp:538, descending) is a rank sort across the lanes:
// the sorted array of a set of numbers is unique, so it is the array std::sort leaves.  A thread per
// column (3 000 dependent instructions) took 10 us for ONE column and 18 us for 10 000.
__global__ __launch_bounds__(64 * WPB) void k_pre(Dev d, int first) {
  __shared__ double s_th[WPB][NL], s_sorted[WPB][NL], s_img[WPB][3 * NL];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int c = blockIdx.x * WPB + wib;
  const bool on = c < d.ncol && lane < NL;
  const size_t at = (size_t)(on ? c : 0) * NL + (on ? lane : 0);
  double T = 0.0, Tpre = 0.0, th = 0.0, conv = 1.0;
  if (on) {
    T = d.T[at];
    Tpre = T;
    conv = d.conv[lane];
    if (first) d.Ttau[at] = T;  // main.cpp:500: the first optical depths come from the pre-adjustment profile
    th = T * conv;              // t_to_theta, main.cpp:123-129
    s_th[wib][lane] = th;
    s_sorted[wib][lane] = __builtin_nan("");  // a NaN among the inputs leaves ranks unassigned: the column stays NaN
  }
  __syncthreads();
  if (on) {
    int rank = 0;
    for (int j = 0; j < NL; ++j) {
      const double tj = s_th[wib][j];
      rank += (tj > th) || (tj == th && j < lane);
    }
    s_sorted[wib][rank] = th;
  }
  __syncthreads();
  double r = 0.0;
  if (on) {
    T = s_sorted[wib][lane] / conv;  // theta_to_t, main.cpp:131-139
    d.T[at] = T;
    r = fabs(T - d.Tprev[at]);
    d.Tprev[at] = T;
  }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {  // lanes 0..31 cover the 20 layers; a NaN layer makes the residual NaN
    const double o = __shfl_xor(r, off, 64);  // (fmax would drop it and the column could pass for converged)
    r = (o > r || o != o) ? o : r;
  }
  if (on && lane == 0) d.resid[c] = first ? 1e300 : r;
  double h2o = 0.0;
  if (on && !first) {
    d.Ttau[at] = T;
    const double vmr = d.rel_hum[at] * magnus_dev(T) / d.player[lane];  // main.cpp:285-286
    h2o = vmr;
    double *gas = d.gas + (size_t)c * d.ngas * NL;
    if (d.mode == MODE_REPWVL)
      gas[0 * NL + lane] = vmr;                    // H2O is read_tau's first species
    else
      gas[1 * NL + lane] = vmr / d.h2o_ref[lane];  // lbl: tau_H2O is linear in the mixing ratio
  } else if (on && d.mode == MODE_REPWVL) {
    h2o = d.gas[(size_t)c * d.ngas * NL + lane];   // first iteration: the initial mixing ratio, no feedback yet
  }
  // this iteration's optical depths are interpolated at Ttau: T before (first) / after the adjustment
  if (c < d.ncol)
    repwvl_image_store(d, c, lane, T, first ? Tpre : T, lane == NL ? d.Tsfc[c] : 0.0, h2o, s_img[wib]);
}

// First stage of the chunk sum for FEW columns with MANY chunk records (one column x the full lbl
// spectrum: 2 048 records): one block of k_post pulls them through a single CU (18 us).  Block (slice,
// column) sums the records [slice*len, (slice+1)*len) in chunk order into out[column][slice][42]; k_post
// then sums the slices.  Fixed order, so still deterministic for a given chunking.
__global__ __launch_bounds__(1024) void k_reduce_partials(const double *partial, double *out, int nchunk, int len,
                                                          int nslice) {
  __shared__ double s_part[16][PSTRIDE];
  const int slice = blockIdx.x, c = blockIdx.y;
  const int j = threadIdx.x & 63, q = threadIdx.x >> 6, P = blockDim.x >> 6;
  const int k0 = slice * len, k1 = min(k0 + len, nchunk);
  if (j < PSTRIDE) {
    const double *p = partial + (size_t)c * nchunk * PSTRIDE + j;
    double acc = 0.0;
    for (int k = k0 + q; k < k1; k += 8 * P) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (k + u * P < k1) ? p[(size_t)(k + u * P) * PSTRIDE] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    s_part[q][j] = acc;
  }
  __syncthreads();
  if (q == 0 && j < PSTRIDE) {
    double acc = s_part[0][j];
    for (int r = 1; r < P; ++r) acc += s_part[r][j];
    out[((size_t)c * nslice + slice) * PSTRIDE + j] = acc;
  }
}

// One block of P waves per column (P = blockDim.x/64 <= 16).  Wave q sums the chunk partials
// q, q+P, q+2P, .. in that order, the P sums are combined in wave order (deterministic for a given
// chunking), then lane 0 runs the tail of radiative_transfer (main.cpp:337-341), calculate_timestep
// (:156-162), thermodynamics (:164-176) and the time accumulation (:581).
// With next_pre != 0 the block goes on with the START of the next iteration for its column -- convective
// adjustment, equilibrium residual, water-vapour feedback: k_pre's work with first == 0 -- so that inside
// rcm_step(h, n) only the first iteration launches k_pre (two launches per iteration instead of three; the last
// k_post of a call runs with next_pre == 0 and leaves the state as main.cpp:583 does, for the getters).
__global__ __launch_bounds__(1024) void k_post(Dev d, int next_pre) {
  __shared__ double s_part[16][PSTRIDE];
  __shared__ double s_E[PSTRIDE];
  const int c = blockIdx.x;
  const int j = threadIdx.x & 63, q = threadIdx.x >> 6, P = blockDim.x >> 6;
  const PostIn in = post_load(d, c, threadIdx.x, next_pre);  // (used by the first wave)
  if (j < PSTRIDE) {
    const double *p = d.partial + (size_t)c * d.nchunk * PSTRIDE + j;
    double acc = 0.0;
    for (int k = q; k < d.nchunk; k += 32 * P) {  // 32 loads in flight (one column x 512 block records: ONE round trip
      double v[32];                               // per wave), added in chunk order as before
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = (k + u * P < d.nchunk) ? p[(size_t)(k + u * P) * PSTRIDE] : 0.0;
#pragma unroll
      for (int u = 0; u < 32; ++u) acc += v[u];
    }
    s_part[q][j] = acc;
  }
  __syncthreads();
  if (q == 0 && j < PSTRIDE) {
    double acc = s_part[0][j];
    for (int r = 1; r < P; ++r) acc += s_part[r][j];
    s_E[j] = acc;
    if (j < NLEV)
      d.E_down[(size_t)c * NLEV + j] = acc;
    else
      d.E_up[(size_t)c * NLEV + (j - NLEV)] = acc;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;  // the rest is the first wave's: lane l = layer l (no block barrier below)
  __shared__ double s_scr[3 * NL];
  post_column(d, c, threadIdx.x, in, s_E, s_scr, next_pre);
}

// The same for few partial records per column (<= 16: the repwvl kernels, small lbl spectra): HALF A WAVE per column
// (lane l of the half = layer l; the 42 sums of a column are two per lane), eight columns per block, no block
// barrier.  The kernel is one dependent chain per column -- records -> fluxes -> time step -> sort -> feedback ->
// brackets -> image -- and nothing but latency: with a wave per column 10 000 waves needed 1.4 rounds of the chip's
// 7 168 slots at this register budget; two columns per wave run their chains side by side in ONE round.
constexpr int kPostCols = 2 * WPB;  // columns per block of k_post_wave
__global__ __launch_bounds__(64 * WPB) void k_post_wave(Dev d, int next_pre) {
  __shared__ double s_E[kPostCols][PSTRIDE];
  __shared__ double s_scr[kPostCols][3 * NL];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int l = lane & 31, slot = 2 * wib + (lane >> 5);
  const int c = blockIdx.x * kPostCols + slot;
  if (c >= d.ncol) return;  // (half a wave at most: the other half goes on alone, groups never exchange data)
  const PostIn in = post_load(d, c, l, next_pre);
  if (l < NLEV) {
    const double *p = d.partial + (size_t)c * d.nchunk * PSTRIDE + l;
    double acc_d = 0.0, acc_u = 0.0;
#pragma unroll
    for (int u0 = 0; u0 < 16; u0 += 8) {  // eight records of either flux in flight; added in chunk order, as k_post does
      if (u0 < d.nchunk) {
        double vd[8], vu[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool in_range = u0 + u < d.nchunk;
          vd[u] = in_range ? p[(size_t)(u0 + u) * PSTRIDE] : 0.0;
          vu[u] = in_range ? p[(size_t)(u0 + u) * PSTRIDE + NLEV] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc_d += vd[u], acc_u += vu[u];
      }
    }
    s_E[slot][l] = acc_d;
    s_E[slot][NLEV + l] = acc_u;
    d.E_down[(size_t)c * NLEV + l] = acc_d;
    d.E_up[(size_t)c * NLEV + l] = acc_u;
  }
  __builtin_amdgcn_wave_barrier();
  post_column<32>(d, c, l, in, s_E[slot], s_scr[slot], next_pre);
}

// output_conv view of one column (main.cpp:536-546) into scratch: adjusted T, sorted theta.
__global__ void k_profile(Dev d, int col, double *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double T[NL], th[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) T[i] = d.T[(size_t)col * NL + i];
  adjust(d.conv, T, th);
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    out[i] = T[i];
    out[NL + i] = th[i];
  }
}

// Optical depths of one column, [nspec][NL], for rcm_get_tau (parity of rows a5 / a9).
__global__ __launch_bounds__(64) void k_tau(Dev d, int col, double *out) {
  __shared__ RepwvlLds s;
  __shared__ double s_scale[RCM_NSPEC_LBL * NL];
  const int lane = threadIdx.x;
  if (d.mode == MODE_REPWVL) {
    repwvl_prologue(d, col, lane, s);
  } else {
    for (int i = lane; i < RCM_NSPEC_LBL * NL; i += 64) s_scale[i] = d.gas[(size_t)col * RCM_NSPEC_LBL * NL + i];
  }
  __syncthreads();
  const long p = (long)blockIdx.x * 64 + lane;
  if (p >= d.nspec) return;
  if (d.mode == MODE_REPWVL) {
#pragma unroll 1
    for (int l = 0; l < NL; ++l)
      out[(size_t)p * NL + l] = repwvl_tau_layer(d, s, (int)p, l, d.cloud_layer[col], d.cloud_tau[col]);
  } else {
#pragma unroll 4
    for (int l = 0; l < NL; ++l)
      out[(size_t)p * NL + l] = lbl_tau_layer(d, s_scale, p, l, d.cloud_layer[col], d.cloud_tau[col]);
  }
}

// The optical depths AS THE PRODUCTION KERNELS FORM THEM (rcm_get_tau_production), lbl and the chunked repwvl
// kernel (n_wvl < 8 or a pinned chunk count): the same column image (lbl_pair_image / repwvl_pair_image: T bracket,
// numDens-folded corner weights, built in place) and the same per-point function (lbl_pair_tau / repwvl_pair_tau:
// weighted-corner sum, cloud, clamp) that k_rt_lbl_pair and k_rt_repwvl_pair call, on the same lane-pair layout;
// one wave per 32 points.  The flattened repwvl kernel copies a STORED image instead: k_image / k_tau_img below.
__global__ __launch_bounds__(64) void k_tau_pair(Dev d, int col, double *out) {
  __shared__ union {
    PairLds lbl;
    RepwvlPairLds rw;
  } img;
  __shared__ double st[NH + 1][64];
  const int lane = threadIdx.x;
  const int p = d.mode == MODE_REPWVL ? PairRw::parity(lane) : PairLbl::parity(lane);
  if (d.mode == MODE_REPWVL)
    repwvl_pair_image(d, col, lane, img.rw);
  else
    lbl_pair_image(d, col, lane, img.lbl);
  __syncthreads();
  const long b = (long)blockIdx.x * 32 + (d.mode == MODE_REPWVL ? PairRw::index(lane) : PairLbl::index(lane));
  if (b >= d.nspec) return;
  if (d.mode == MODE_REPWVL) {
    repwvl_pair_tau<RCM_NSPEC_REPWVL>(d, img.rw, p, (int)b, st, lane);
  } else {
    const LblAddr ad = lbl_pair_addr(d, p);
    lbl_pair_tau(img.lbl, p, ad.sbase, ad.row0 + (unsigned)b * 8u, ad.rstep, st, lane);
  }
  for (int k = 0; k < NH; ++k) out[(size_t)b * NL + (p ? NL - 1 - k : k)] = st[k][lane];
}

// rcm_get_tau_production on the flattened repwvl path: the image k_rt_repwvl_flat really copies.
// k_image: (re)stores the image of ONE column for the optical depths of the most recent radiative transfer -- before
// the first step: the one it will use -- through repwvl_image_store, the coefficient code of k_pre / post_column
// (needed only when no stepping kernel has written d.img since the last change of the inputs: before step 0, after
// rcm_set_columns / a forcing change on a live handle, after rcm_checkpoint_load).  from_Ttau: the temperatures the
// brackets are taken at are d.Ttau (a step has run) or d.T (step 0: the pre-adjustment profile, main.cpp:500).
__global__ __launch_bounds__(64) void k_image(Dev d, int col, int from_Ttau) {
  __shared__ double scr[3 * NL];
  const int lane = threadIdx.x;
  const size_t at = (size_t)col * NL + (lane < NL ? lane : 0);
  const double T = d.T[at], midT = from_Ttau ? d.Ttau[at] : T;
  const double h2o = d.gas[(size_t)col * d.ngas * NL + (lane < NL ? lane : 0)];
  repwvl_image_store(d, col, lane, T, midT, d.Tsfc[col], h2o, scr);
}

// ... and the optical depths from it: repwvl_pair_image_copy of d.img[col] + repwvl_pair_tau<NACT> with the NACT the
// launch picks (launch_rt_pair), on the lane-pair layout of the production kernel; one wave per 32 wavelengths.
template <int NACT>
__global__ __launch_bounds__(64) void k_tau_img(Dev d, int col, double *out) {
  __shared__ __align__(16) RepwvlPairLds img;
  __shared__ double st[NH + 1][64];
  const int lane = threadIdx.x;
  const int p = PairRw::parity(lane);
  repwvl_pair_image_copy<NACT>(d, col, lane, img);
  __syncthreads();
  const long b = (long)blockIdx.x * 32 + PairRw::index(lane);
  if (b >= d.nspec) return;
  repwvl_pair_tau<NACT>(d, img, p, (int)b, st, lane);
  for (int k = 0; k < NH; ++k) out[(size_t)b * NL + (p ? NL - 1 - k : k)] = st[k][lane];
}

// doubling_adding + solar_radiative_transfer_setup (main.cpp:214-264), one thread per column:
// per-column optical thickness, cosine of the solar zenith angle and surface albedo.  Also sets the
// column's cloud optical depth to tau_s/2 as main.cpp:267 does.  out = [ncol][7] (may be null).
__global__ void k_solar(int ncol, const double *tau_s, const double *mu_s, const double *albedo, int doublings,
                        double g_asym, double daytime, double E_0, double *solar_irr, double *cloud_tau,
                        double *out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncol) return;
  const double mu = mu_s[c];
  double dtau = (1 - g_asym) * tau_s[c] / pow(2.0, (double)doublings);  // main.cpp:216-217
  double r = 0.5 * dtau / mu, t = 1.0 - r;                              // main.cpp:223-224
  double r_dir = dtau / mu * 0.5, s_dir = r_dir, t_dir = 1 - dtau / mu;  // main.cpp:225-227
  for (int i = 0; i < doublings; ++i) {                                  // main.cpp:232-250
    const double den = 1 - r * r;
    const double r2 = r + (r * t * t) / den;
    const double t2 = (t * t) / den;
    const double td2 = t_dir * t_dir;
    const double sd2 = (t * s_dir + t_dir * r_dir * r * t) / den + t_dir * s_dir;
    const double rd2 = (t * s_dir * r + t * t_dir * r) / den + r_dir;
    r = r2, t = t2, t_dir = td2, s_dir = sd2, r_dir = rd2;
  }
  const double r_total = r_dir + (t_dir + s_dir) / (1 - albedo[c] * r) * t * albedo[c];  // main.cpp:258
  const double irr = daytime * E_0 * mu * (1 - r_total);                                 // main.cpp:260
  solar_irr[c] = irr;
  cloud_tau[c] = tau_s[c] / 2.0;  // main.cpp:267
  if (out) {
    double *o = out + (size_t)c * 7;
    o[0] = r_dir, o[1] = s_dir, o[2] = t_dir, o[3] = r, o[4] = t, o[5] = r_total, o[6] = irr;
  }
}

// [ncol][NL+1] = (Tlayer, T_surface) for the one collective of the multi-GPU run.
__global__ void k_pack(Dev d, double *out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)d.ncol * NLEV) return;
  const size_t c = i / NLEV, l = i % NLEV;
  out[i] = (l < NL) ? d.T[c * NL + l] : d.Tsfc[c];
}

// lbl tables arrive [node][layer] (ASCII_file2xy2D layout); the kernels want band-fastest.
__global__ void k_transpose_lbl(const double *in, double *out, long n_nodes, long ld) {
  __shared__ double tile[NL][65];
  const long b0 = (long)blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * NL; i += blockDim.x) {
    const long b = b0 + i / NL;
    const int l = i % NL;
    tile[l][i / NL] = (b < n_nodes) ? in[b * NL + l] : 0.0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * NL; i += blockDim.x) {
    const int l = i / 64, k = i % 64;
    if (b0 + k < ld) out[(size_t)l * ld + b0 + k] = tile[l][k];
  }
}

#ifdef RCM_WITH_VARIANTS  // test-side library only (tests/_variants/librcm_hip_variants.so): the one-lane-per-point
#include "rcm_variants.hpp"  // reference kernels and the fused whole-step kernel; the product does not compile them
#endif

}  // namespace rcm

