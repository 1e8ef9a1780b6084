// rcm_capi.hip -- the ensemble handle and the C ABI (include/rcm.h) on top of the gfx950 kernels of
// rcm_kernels.hpp: device buffers, table upload and re-layout, launch geometry, step loop, getters,
// restart files.  Host-only pieces of the reference driver live in rcm_host.cpp.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rcm.h"
#include "rcm_internal.hpp"
#include "rcm_kernels.hpp"

#ifndef RCM_MERGE_SHARED
#define RCM_MERGE_SHARED 1  // repwvl: species with one profile for all columns share a merged table row set
#endif
#ifndef RCM_GEN15
#define RCM_GEN15 1  // root pairs 3m / 5m from one exponential (0: every root its own)
#endif

// =============================================================================================
// Host side: the handle and the C ABI.
// =============================================================================================
using rcm::Dev;
using rcm::fail;
using rcm::NL;
using rcm::NLEV;

#define HIPCK(call)                                                                       \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(RCM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
  } while (0)

struct rcm_ensemble {
  int ncol = 0, nlayer = 0, nangle = 0, device = 0, mode = rcm::MODE_NONE;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool have_grid = false, have_columns = false;
  long long step_index = 0;  // the reference's loop variable i (main.cpp:531)
  bool img_valid = false;    // repwvl: d.img holds, for every column, the image of the most recent radiative transfer
                             // (set by rcm_step; cleared by everything that changes what the image is made from)
  int user_chunks = 0;
  int fusion = -1;  // repwvl whole-step kernel k_step_repwvl: -1 automatic, 0 never, 1 always
  int exp_variant = 1, cube = 1, planck_fast = 1, pair = 1, flat = 1;
  int lds_pad = 0;  // -DRCM_EXPERIMENT builds: dynamic LDS bytes added to the lbl launch (bounds the blocks per CU)
  int block_records = 1;  // (-DRCM_EXPERIMENT builds read RCM_BLOCK_RECORDS: 0 = the two-stage sum with k_reduce_partials)
  int flip_passes = 1;    // (-DRCM_EXPERIMENT builds read RCM_FLIP: 0 = every launch takes its passes in the same order)
  int phase_prio = 1;     // (-DRCM_EXPERIMENT builds read RCM_FEW_PRIO: 0 = small ensembles run the plain instantiation)
  std::vector<double> plevel, player, conv;
  std::vector<void *> allocs;
  Dev d{};
  double *scratch = nullptr;  // 2*NL doubles
  double *partial_buf = nullptr;
  double *slice_buf = nullptr;  // [ncol][kPostSlices][42]: first stage of the chunk sum for few columns x many chunks
  double *gather_send = nullptr;  // rcm_gather: this rank's zero-padded block [gather_rows][21]
  int gather_rows = 0;
  // repwvl host copies needed to (re)derive per-layer constants
  std::vector<double> p_grid;
  std::vector<double> xsec_t;  // xsec re-laid out [t][s][p][w] (host copy: the per-layer table is re-derived with the grid)
  double *d_xp = nullptr;      // [layer][t][s][w][2]: xsec at the layer's mid pressure, pairs of temperature rows (derive_repwvl_layers)
  double *d_tref_l = nullptr;  // [layer]: t_ref[ip[layer]]
  std::vector<double> t_ref;   // host copy
  std::vector<double> shared_vmr;  // [9][NL]: the profiles of the species merged into d.merged_slot (rcm_set_columns)
  unsigned shared_set = 0;         // which species those are
  bool merge_species = true;       // rcm_set_species_merging
  int *d_ip = nullptr;
  double *d_delP = nullptr, *d_numDens = nullptr;
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
};

namespace {

template <typename T>
int dev_alloc(rcm_ensemble *h, T **p, size_t n) {
  void *q = nullptr;
  HIPCK(hipMalloc(&q, (n ? n : 1) * sizeof(T)));
  h->allocs.push_back(q);
  *p = (T *)q;
  return RCM_OK;
}

template <typename T>
int dev_upload(rcm_ensemble *h, T **p, const T *src, size_t n) {
  if (int rc = dev_alloc(h, p, n)) return rc;
  HIPCK(hipMemcpyAsync(*p, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return RCM_OK;
}

int use_device(rcm_ensemble *h) {
  HIPCK(hipSetDevice(h->device));
  return RCM_OK;
}

// Per-layer constants of read_tau that depend only on the pressure grid
// (repwvl_thermal.cpp:68-73, 201-202, 214, 226, 240), top-first layer index.
int derive_repwvl_layers(rcm_ensemble *h) {
  if (h->mode != rcm::MODE_REPWVL || !h->have_grid) return RCM_OK;
  const double avog = 6.02214076e23, molMassAir = 0.0289647, earthAccel = 9.80665;
  std::vector<int> ip(NL);
  std::vector<double> delP(NL), numDens(NL);
  const int np = (int)h->p_grid.size();
  for (int o = 0; o < NL; ++o) {
    const double p_up = h->plevel[o] * 100.0, p_dn = h->plevel[o + 1] * 100.0;  // Pa
    const double midP = (p_up + p_dn) / 2;                                       // :214
    // LowerPos(p_ref, midP), :19-45
    auto sgn = [](double v) { return (0.0 < v) - (v < 0.0); };
    int lp = 0, sign = sgn(h->p_grid[0] - midP);
    for (int p = 1; p != np; ++p) {
      const double delta = h->p_grid[p] - midP;
      if (sign != sgn(delta) || p == np - 1) {
        lp = p - 1;
        break;
      }
      sign = sgn(delta);
    }
    ip[o] = lp;
    delP[o] = (midP - h->p_grid[lp]) / (h->p_grid[lp + 1] - h->p_grid[lp]);  // :240
    numDens[o] = (p_dn - p_up) * avog / molMassAir / earthAccel;              // :202
  }
  // The pressure half of read_tau's bilinear interpolation (:235-241) does not depend on the column: every
  // column shares the pressure grid, so the layer's bracket ip and delP are per-handle constants.  The table is
  // therefore interpolated to each layer's mid pressure ONCE, here:
  //   xp[l][t][s][w] = x[t][s][ip_l][w] + delP_l * (x[t][s][ip_l+1][w] - x[t][s][ip_l][w])
  // (20 layers x 9 x 9 x n_wvl doubles: 1.3 MB for Reduced100, less than the table itself) and the kernels
  // interpolate in temperature only: 2 table loads and 2 FMAs per species and layer instead of 4 and 5.
  // c0 + cT*dT + cP*dP + cPT*dT*dP of the reference is (1-dT)*xp[it] + dT*xp[it+1] up to rounding
  // (rcm_get_tau keeps the reference's form; rcm_get_tau_production returns this one).
  {
    const Dev &d = h->d;
    const size_t nw = (size_t)d.n_wvl, rows = (size_t)d.n_tpert * d.n_species;
    std::vector<double> xp((size_t)NL * rows * nw);
    for (int l = 0; l < NL; ++l)
      for (size_t r = 0; r < rows; ++r) {
        const double *x0 = &h->xsec_t[(r * d.n_p + ip[l]) * nw], *x1 = x0 + nw;
        double *o = &xp[((size_t)l * rows + r) * nw];
        for (size_t w = 0; w < nw; ++w) o[w] = std::fma(delP[l], x1[w] - x0[w], x0[w]);
      }
    // species with one mixing-ratio profile for all columns (rcm_set_columns): their weighted sum takes the row set of
    // the first of them; the others' rows stay as they are and are not visited (Dev::prod_mask)
    if (d.merged_slot >= 0)
      for (int l = 0; l < NL; ++l)
        for (int t = 0; t < d.n_tpert; ++t) {
          double *slot = &xp[(((size_t)l * d.n_tpert + t) * d.n_species + d.merged_slot) * nw];
          std::vector<double> sum(nw, 0.0);
          for (int sp = 0; sp < d.n_species; ++sp)
            if ((h->shared_set >> sp) & 1u) {
              const double v = h->shared_vmr[(size_t)sp * NL + l];
              const double *x = &xp[(((size_t)l * d.n_tpert + t) * d.n_species + sp) * nw];
              for (size_t w = 0; w < nw; ++w) sum[w] = std::fma(v, x[w], sum[w]);
            }
          std::copy(sum.begin(), sum.end(), slot);
        }
    // what the kernels read: pairs {x(t), x(t+1)} of neighbouring temperature rows, one 16-byte load per bracket
    const size_t nt1 = (size_t)d.n_tpert - 1, ns = (size_t)d.n_species;
    std::vector<double> xq((size_t)NL * nt1 * ns * nw * 2);
    for (size_t l = 0; l < (size_t)NL; ++l)
      for (size_t t = 0; t < nt1; ++t)
        for (size_t sp = 0; sp < ns; ++sp) {
          const double *lo = &xp[((l * d.n_tpert + t) * ns + sp) * nw], *hi = lo + ns * nw;
          double *o = &xq[(((l * nt1 + t) * ns + sp) * nw) * 2];
          for (size_t w = 0; w < nw; ++w) o[2 * w] = lo[w], o[2 * w + 1] = hi[w];
        }
    HIPCK(hipMemcpyAsync(h->d_xp, xq.data(), xq.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));  // xq goes out of scope
  }
  std::vector<double> tref_l(NL);
  for (int l = 0; l < NL; ++l) tref_l[l] = h->t_ref[ip[l]];  // :229: the bracket's reference temperature per layer
  HIPCK(hipMemcpyAsync(h->d_tref_l, tref_l.data(), NL * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->d_ip, ip.data(), NL * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->d_delP, delP.data(), NL * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->d_numDens, numDens.data(), NL * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return RCM_OK;
}

// repwvl: which species the optical-depth stages visit, from the mixing ratios [ncol][9][NL] (rcm_set_columns, and
// rcm_checkpoint_load so that a resumed run forms its optical depths exactly as the run that wrote the checkpoint).
int classify_species(rcm_ensemble *h, const double *gas) {
  Dev &d = h->d;
  // species whose mixing ratio is 0 everywhere add exactly +0
  unsigned mask = 0;
  for (int c = 0; c < h->ncol; ++c)
    for (int s = 0; s < RCM_NSPEC_REPWVL; ++s)
      for (int l = 0; l < NL; ++l)
        if (gas[((size_t)c * RCM_NSPEC_REPWVL + s) * NL + l] != 0.0) mask |= 1u << s;
  d.species_mask = mask | 1u;  // H2O always: the feedback rewrites it
  // species (other than H2O) whose profile is the same in every column: one merged row set for all of them
  unsigned same = d.species_mask & ~1u;
  for (int s = 1; s < RCM_NSPEC_REPWVL; ++s)
    for (int c = 1; c < h->ncol && ((same >> s) & 1u); ++c)
      if (std::memcmp(&gas[((size_t)c * RCM_NSPEC_REPWVL + s) * NL], &gas[(size_t)s * NL], NL * sizeof(double)) != 0)
        same &= ~(1u << s);
  const unsigned old_set = h->shared_set;
  const std::vector<double> old_vmr = h->shared_vmr;
  if (RCM_MERGE_SHARED && h->merge_species && __builtin_popcount(same) >= 2) {
    h->shared_set = same;
    h->shared_vmr.assign(gas, gas + (size_t)RCM_NSPEC_REPWVL * NL);  // column 0 = every column, for these species
    d.merged_slot = __builtin_ctz(same);
    d.prod_mask = (d.species_mask & ~same) | (1u << d.merged_slot);
  } else {
    h->shared_set = 0;
    d.merged_slot = -1;
    d.prod_mask = d.species_mask;
  }
  d.nact = 0;
  for (int s = 0; s < RCM_NSPEC_REPWVL; ++s)
    if ((d.prod_mask >> s) & 1u) d.act[d.nact++] = s;
  for (int s = 0, j = d.nact; s < RCM_NSPEC_REPWVL && j < RCM_NSPEC_REPWVL; ++s)  // padding: coefficients 0 in the image
    if (!((d.prod_mask >> s) & 1u)) d.act[j++] = s;
  d.act_nibbles = 0;
  for (int j = 0; j < RCM_NSPEC_REPWVL; ++j) d.act_nibbles |= (unsigned long long)d.act[j] << (4 * j);
  // k_pre / k_post_wave write the coefficients of the active species only: everything else in the images is zero
  HIPCK(hipMemsetAsync(d.img, 0, (size_t)h->ncol * sizeof(rcm::RepwvlPairLds), h->stream));
  h->img_valid = false;
  bool changed = old_set != h->shared_set;
  for (int s = 0; !changed && s < RCM_NSPEC_REPWVL; ++s)
    if ((h->shared_set >> s) & 1u)
      changed = std::memcmp(&old_vmr[(size_t)s * NL], &h->shared_vmr[(size_t)s * NL], NL * sizeof(double)) != 0;
  if (changed) return derive_repwvl_layers(h);  // (a no-op until the grid is known; rcm_set_grid derives then)
  return RCM_OK;
}

// Spectral chunking.  Pure function of (ncol, nspec, user override); measured on the MI355X
// (tools/time_single.py, tools/chunk_sweep.sh):
//  * a wave needs ~8 passes of 32 points to amortise its column prologue (64 columns: 512 chunks beat
//    4096 by 7 %), so big ensembles use chunks of >= 8 passes, as many as give ~2.6e5 waves in flight
//    (10 000 columns: 27-32 chunks beat 4 by 4 %: shorter tail, better balance);
//  * a small ensemble must first fill the chip (3072 resident waves, a lone wave reaches ~1/6 of the
//    VALU issue rate): down to 2 passes per wave until there are ~4 rounds of waves.
void choose_chunks(int ncol, int nspec, int user_chunks, int *nchunk_out, int *chunk_len_out) {
  const int rows = (nspec + 31) / 32;  // a lane-pair wave covers 32 spectral points per pass
  long nchunk;
  if (user_chunks > 0) {
    nchunk = user_chunks;
  } else {
    const long fill = (12288L + ncol - 1) / ncol;          // chunks that give 4 rounds of resident waves
    const long amortised = rows / 8 > 0 ? rows / 8 : 1;    // chunks of 8 passes
    if (fill >= amortised) {
      nchunk = fill;
      if (nchunk > rows / 2) nchunk = rows / 2;            // keep two passes per wave
    } else {
      nchunk = (262144L + ncol - 1) / ncol;
      if (nchunk > amortised) nchunk = amortised;
    }
  }
  if (nchunk > rows) nchunk = rows;
  if (nchunk < 1) nchunk = 1;
  const int rows_per_chunk = (int)((rows + nchunk - 1) / nchunk);
  *chunk_len_out = rows_per_chunk * 32;
  *nchunk_out = (rows + rows_per_chunk - 1) / rows_per_chunk;
}

// repwvl with the (column, wavelength) pairs flattened over the lane pairs (k_rt_repwvl_flat): the
// production path unless the caller pins a chunk count or selects a reference variant.
constexpr int kPostSlices = 32;  // blocks per column of the first-stage chunk sum (k_reduce_partials)

// lbl, ONE column x many chunks (BASELINE configs[1]): the RT launch writes one record per block of 4 consecutive chunks
// (k_rt_lbl_pair<1, 1>) and k_post sums the nchunk / 4 records itself -- no k_reduce_partials.  Measured in one
// process (tools/ab_single.py, profiles/r04_lbl1_ab_block_records.txt): one column 67.8 - 71.1 -> 65.7 - 66.7 us per
// step; two to four columns are 3 - 5 % SLOWER that way (a block's waves then share a column instead of a chunk, and the
// chunk's table tile is what neighbouring waves share in L1/L2), sixteen the same: so one column only.
bool use_block_records(const rcm_ensemble *h) {
  return h->mode == rcm::MODE_LBL && h->pair && h->cube && h->exp_variant == 1 && h->planck_fast && h->block_records &&
         h->d.nchunk >= 512 && h->ncol == 1;
}

// lbl launches of at most this many waves run k_rt_lbl_pair<1, 2>: same arithmetic, but the short latency-bound phases of
// a wave (image, loads, Planck, reduction) at s_setprio 3 and its angle loop at 0.  Measured in one process
// (tools/ab_single.py, profiles/r04_lbl_few_columns_phase_priorities.txt), RT launch: 2 columns 92.5 -> 88.7 us (-4 %),
// 4: 164.7 -> 160.4, 8: 308.7 -> 304.0, 16: 598 -> 591, 64: 2 216 -> 2 199 (-0.8 %); 512 columns (262 144 waves) +0.25 %
// and 10 000 columns +-0.1 %: beyond ~65 000 waves a wave lives for 8 passes and more and does not care.
constexpr long kPhasePrioMaxWaves = 65536;

bool use_flat(const rcm_ensemble *h) {
  return h->mode == rcm::MODE_REPWVL && h->flat && h->pair && h->exp_variant == 1 && h->user_chunks == 0 &&
         (long long)h->ncol * h->d.n_wvl < (1ll << 31) - 4096 &&  // 32-bit item arithmetic in the kernel
         h->d.n_wvl >= 8;  // at most 127/8 + 2 = 17 column images per block, each the prefix the launch reads (1 552 B
                           // for the 3 visited species of a CO2 sweep, 3 472 B for all 9) + 22.5 KB staging: 48 KB / 80 KB of
                           // LDS at n_wvl = 8; Reduced10: 14 images, 44 KB (3 blocks per CU); Reduced100: 3 images, 27 KB
}

#ifdef RCM_WITH_VARIANTS
// The fused whole-step kernel of the repwvl path (k_step_repwvl): lane pairs per column LPC (a power of two, K =
// 32/LPC columns per wave).  Cost model in wave passes per SIMD: a wave runs ceil(n_wvl/LPC) passes (+ ~1 for the
// phases around them), ceil(ncol/K) waves share 1024 SIMDs.  10 000 x 100: LPC 16 (5 000 waves x 7 passes = 35
// per SIMD against 30.5 for perfect packing; 32: 40, 8: 39).  Returns 0 when the three-kernel path should run:
// fewer than two waves per SIMD would leave the chip to latency (small ensembles keep the flattened kernel,
// which cuts the work into 32-item waves).
int choose_lpc(const rcm_ensemble *h) {
  if (h->mode != rcm::MODE_REPWVL || !h->pair || h->exp_variant != 1 || h->user_chunks != 0 || h->fusion == 0) return 0;
  int best = 0;
  double best_cost = 1e300;
  for (int lpc = 32; lpc >= 2; lpc >>= 1) {
    const int K = 32 / lpc;
    const long waves = ((long)h->ncol + K - 1) / K;
    const long passes = (h->d.n_wvl + lpc - 1) / lpc;
    const double cost = (double)((waves + 1023) / 1024) * (double)(passes + 1);
    if (cost < best_cost) best_cost = cost, best = lpc;
  }
#ifdef RCM_LPC_ENV  // experiments only (tools/build_variants.py): force the lane pairs per column
  if (const char *v = std::getenv("RCM_LPC")) best = std::atoi(v);
#endif
  if (h->fusion == 1) return best;
  return 0;  // automatic = the three-kernel path: the fused kernel is slower at every size measured
}

#endif  // RCM_WITH_VARIANTS

int ensure_partials(rcm_ensemble *h) {
  if (use_flat(h)) {
    h->d.nchunk = (h->d.n_wvl + 31) / 32 + 1;  // record slots per column, see k_rt_repwvl_flat
    h->d.chunk_len = 0;
  } else {
    choose_chunks(h->ncol, h->d.nspec, h->user_chunks, &h->d.nchunk, &h->d.chunk_len);
  }
  if (h->partial_buf) {
    HIPCK(hipStreamSynchronize(h->stream));
    HIPCK(hipFree(h->partial_buf));
    h->partial_buf = nullptr;
  }
  const size_t bytes = (size_t)h->ncol * h->d.nchunk * rcm::PSTRIDE * sizeof(double);
  HIPCK(hipMalloc((void **)&h->partial_buf, bytes));
  HIPCK(hipMemsetAsync(h->partial_buf, 0, bytes, h->stream));  // slots a column never writes must read as 0
  h->d.partial = h->partial_buf;
  if (h->slice_buf) {
    HIPCK(hipFree(h->slice_buf));
    h->slice_buf = nullptr;
  }
  if (h->d.nchunk >= 512 && h->ncol <= 64 && !use_block_records(h))  // measured: one column x 2 048 records, k_post 18 -> 3 + 5 us
    HIPCK(hipMalloc((void **)&h->slice_buf, (size_t)h->ncol * kPostSlices * rcm::PSTRIDE * sizeof(double)));
  return RCM_OK;
}

// The production kernels: lane pairs, cube/root-pair angle chains (CUBE = 1), own exp and narrow-band Planck.
template <int CUBE>
void launch_rt_pair(rcm_ensemble *h, int blocks) {
  if (h->mode == rcm::MODE_LBL && CUBE && use_block_records(h)) {
    const int nrec = (h->d.nchunk + rcm::WPB - 1) / rcm::WPB;
    hipLaunchKernelGGL((rcm::k_rt_lbl_pair<1, 1>), dim3((unsigned)(nrec * h->ncol)), dim3(64 * rcm::WPB), 0, h->stream, h->d);
  } else if (h->mode == rcm::MODE_LBL && CUBE && h->phase_prio && (long)h->ncol * h->d.nchunk <= kPhasePrioMaxWaves) {
    // small ensembles (waves of a few passes): the instantiation whose phases around the angle loop run at wave priority 3
    hipLaunchKernelGGL((rcm::k_rt_lbl_pair<1, 2>), dim3(blocks), dim3(64 * rcm::WPB), 0, h->stream, h->d);
  } else if (h->mode == rcm::MODE_LBL) {
#ifdef RCM_EXPERIMENT
    if (h->lds_pad > 0)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rcm::k_rt_lbl_pair<CUBE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_pad);
#endif
    hipLaunchKernelGGL((rcm::k_rt_lbl_pair<CUBE>), dim3(blocks), dim3(64 * rcm::WPB), h->lds_pad, h->stream, h->d);
  } else if (use_flat(h)) {
    const int maxc = (rcm::FWPB * 32 - 1) / h->d.n_wvl + 2;  // columns the consecutive items of a block can touch
    const long waves = ((long)h->ncol * h->d.n_wvl + 31) / 32;
    // the instantiation that visits the fewest species >= the active ones (the entries past Dev::nact add +0)
    const int na = h->d.nact;
    const int nvisit = na <= 2 ? 2 : na <= 3 ? 3 : na <= 4 ? 4 : na <= 5 ? 5 : na <= 7 ? 7 : 9;
    const size_t lds = (size_t)maxc * rcm::repwvl_image_prefix_bytes(nvisit) +  // the images' prefixes (header + visited slots)
                       (size_t)rcm::FWPB * (rcm::NH + 1) * 64 * sizeof(double);
    const auto kern = na <= 2 ? rcm::k_rt_repwvl_flat<CUBE, 2> : na <= 3 ? rcm::k_rt_repwvl_flat<CUBE, 3>
                    : na <= 4 ? rcm::k_rt_repwvl_flat<CUBE, 4> : na <= 5 ? rcm::k_rt_repwvl_flat<CUBE, 5>
                    : na <= 7 ? rcm::k_rt_repwvl_flat<CUBE, 7> : rcm::k_rt_repwvl_flat<CUBE, 9>;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((waves + rcm::FWPB - 1) / rcm::FWPB)), dim3(64 * rcm::FWPB), lds, h->stream,
                       h->d, maxc);
  } else {
    hipLaunchKernelGGL((rcm::k_rt_repwvl_pair<CUBE>), dim3(blocks), dim3(64 * rcm::WPB), 0, h->stream, h->d);
  }
}

#ifdef RCM_WITH_VARIANTS
// Test-side library only (tests/_variants/librcm_hip_variants.so, -DRCM_WITH_VARIANTS): the one-lane-per-point
// kernels with the library exp / the branch-for-branch cplkavg / no cube chains, selected by environment
// variables at rcm_create.  The production library has neither these kernels nor the variables.
template <int EXPV, int CUBE>
void launch_rt_variant(rcm_ensemble *h, int blocks) {
  if (h->mode == rcm::MODE_LBL) {
    if (!h->planck_fast)
      hipLaunchKernelGGL((rcm::k_rt_lbl<EXPV, CUBE, 0>), dim3(blocks), dim3(64 * rcm::WPB), 0, h->stream, h->d);
    else
      hipLaunchKernelGGL((rcm::k_rt_lbl<EXPV, CUBE, 1>), dim3(blocks), dim3(64 * rcm::WPB), 0, h->stream, h->d);
  } else {
    hipLaunchKernelGGL((rcm::k_rt_repwvl<EXPV, CUBE>), dim3(blocks), dim3(64 * rcm::WPB), 0, h->stream, h->d);
  }
}
#endif

void launch_rt(rcm_ensemble *h, int blocks) {
#ifdef RCM_WITH_VARIANTS
  const bool production = h->exp_variant == 1 && h->pair && (h->mode != rcm::MODE_LBL || h->planck_fast);
  if (!production) {
    if (h->exp_variant == 0 && !h->cube)
      launch_rt_variant<0, 0>(h, blocks);
    else if (h->exp_variant == 0)
      launch_rt_variant<0, 1>(h, blocks);
    else if (!h->cube)
      launch_rt_variant<1, 0>(h, blocks);
    else
      launch_rt_variant<1, 1>(h, blocks);
    return;
  }
  if (!h->cube) {
    launch_rt_pair<0>(h, blocks);
    return;
  }
#endif
  launch_rt_pair<1>(h, blocks);
}

int timing_begin(rcm_ensemble *h, hipEvent_t *e1) {
  *e1 = nullptr;
  if (!h->timing) return RCM_OK;
  if (h->events_used == h->events.size()) {
    hipEvent_t a, b;
    HIPCK(hipEventCreate(&a));
    HIPCK(hipEventCreate(&b));
    h->events.emplace_back(a, b);
  }
  const hipEvent_t e0 = h->events[h->events_used].first;
  *e1 = h->events[h->events_used].second;
  ++h->events_used;
  HIPCK(hipEventRecord(e0, h->stream));
  return RCM_OK;
}

#ifdef RCM_WITH_VARIANTS
template <int CUBE, int LPC>
void launch_fused_lpc(rcm_ensemble *h, int first, int nsteps) {
  constexpr int K = 32 / LPC;
  const size_t lds = (size_t)K * (sizeof(rcm::RepwvlPairLds) + sizeof(rcm::StepLds)) + (size_t)(rcm::NH + 1) * 64 * sizeof(double) +
                     (K * 102 <= (rcm::NH + 1) * 64 ? 0 : (size_t)K * 102 * sizeof(double));
  const auto kern = rcm::k_step_repwvl<CUBE, LPC>;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)((h->ncol + K - 1) / K)), dim3(64), lds, h->stream, h->d, first, nsteps);
}

template <int CUBE>
void launch_fused(rcm_ensemble *h, int lpc, int first, int nsteps) {
  switch (lpc) {
    case 32: launch_fused_lpc<CUBE, 32>(h, first, nsteps); break;
    case 16: launch_fused_lpc<CUBE, 16>(h, first, nsteps); break;
    case 8: launch_fused_lpc<CUBE, 8>(h, first, nsteps); break;
    case 4: launch_fused_lpc<CUBE, 4>(h, first, nsteps); break;
    default: launch_fused_lpc<CUBE, 2>(h, first, nsteps); break;
  }
}

// `nsteps` iterations of the repwvl path in ONE launch (k_step_repwvl); long runs are cut into launches of at most
// 512 iterations so that no kernel runs for more than a few seconds.
int launch_fused_steps(rcm_ensemble *h, int lpc, int nsteps) {
  while (nsteps > 0) {
    const int n = nsteps < 512 ? nsteps : 512;
    const int first = (h->step_index == 0) ? 1 : 0;
    hipEvent_t e1 = nullptr;
    if (int rc = timing_begin(h, &e1)) return rc;
#ifdef RCM_WITH_VARIANTS
    if (!h->cube)
      launch_fused<0>(h, lpc, first, n);
    else
#endif
      launch_fused<1>(h, lpc, first, n);
    if (e1) HIPCK(hipEventRecord(e1, h->stream));
    HIPCK(hipGetLastError());
    h->step_index += n;
    nsteps -= n;
  }
  return RCM_OK;
}

#endif  // RCM_WITH_VARIANTS

// One iteration.  have_pre: the previous k_post of this rcm_step call already did this iteration's adjustment and
// feedback; next_pre: this iteration's k_post goes on with the next one's.
int launch_step(rcm_ensemble *h, bool have_pre, bool next_pre) {
  Dev &d = h->d;
  const int first = (h->step_index == 0) ? 1 : 0;
  if (!have_pre)
    hipLaunchKernelGGL(rcm::k_pre, dim3((h->ncol + rcm::WPB - 1) / rcm::WPB), dim3(64 * rcm::WPB), 0, h->stream, d, first);
  const int blocks = (int)(((long)h->ncol * d.nchunk + rcm::WPB - 1) / rcm::WPB);
  // lbl, one column x the full spectrum (every launch streams the 106 MB of tables once, the L2 holds 32 MiB): odd
  // iterations take each wave's passes in reverse, so a launch begins with what the previous one ended with (the order
  // of a wave's spectral sum alternates with the iteration: deterministic, last bits)
  d.flip = (use_block_records(h) && h->flip_passes) ? (int)(h->step_index & 1) : 0;
  hipEvent_t e1 = nullptr;
  if (int rc = timing_begin(h, &e1)) return rc;
  launch_rt(h, blocks);
  if (e1) HIPCK(hipEventRecord(e1, h->stream));
  Dev dp = d;  // what k_post sums: the chunk records, the block records, or the slice sums
  if (use_block_records(h)) {
    dp.nchunk = (d.nchunk + rcm::WPB - 1) / rcm::WPB;
  } else if (h->slice_buf) {
    const int len = (d.nchunk + kPostSlices - 1) / kPostSlices;
    hipLaunchKernelGGL(rcm::k_reduce_partials, dim3(kPostSlices, h->ncol), dim3(64 * 8), 0, h->stream, d.partial,
                       h->slice_buf, d.nchunk, len, kPostSlices);
    dp.partial = h->slice_buf;
    dp.nchunk = kPostSlices;
  }
  int post_waves = dp.nchunk / 16;  // waves per column in k_post: one per 16 chunk partials, 1..16
  post_waves = post_waves < 1 ? 1 : (post_waves > 16 ? 16 : post_waves);
  if (dp.nchunk <= 16)  // few records per column: half a wave per column, eight columns per block
    hipLaunchKernelGGL(rcm::k_post_wave, dim3((h->ncol + rcm::kPostCols - 1) / rcm::kPostCols), dim3(64 * rcm::WPB), 0, h->stream, dp,
                       next_pre ? 1 : 0);
  else
    hipLaunchKernelGGL(rcm::k_post, dim3(h->ncol), dim3(64 * post_waves), 0, h->stream, dp, next_pre ? 1 : 0);
  HIPCK(hipGetLastError());
  ++h->step_index;
  return RCM_OK;
}

}  // namespace

// ---- restart file: "RCMCKPT1", int64 {ncol, nlayer, ngas, mode, step_index}, then the arrays ----
namespace {
struct CkptField {
  void *dev;
  size_t bytes;
};
std::vector<CkptField> ckpt_fields(rcm_ensemble *h) {
  const Dev &d = h->d;
  const size_t nc = (size_t)h->ncol, nT = nc * NL;
  return {{d.T, nT * 8},           {d.Ttau, nT * 8},
          {d.Tsfc, nc * 8},        {d.dt, nc * 8},
          {d.time_h, nc * 4},      {d.gas, nT * d.ngas * 8},
          {(void *)d.rel_hum, nT * 8}, {(void *)d.solar_irr, nc * 8},
          {(void *)d.cloud_tau, nc * 8}, {(void *)d.cloud_layer, nc * 4},
          {d.E_down, nc * NLEV * 8}, {d.E_up, nc * NLEV * 8},
          {d.dE, nT * 8},          {d.resid, nc * 8},
          {d.Tprev, nT * 8}};
}
}  // namespace

extern "C" {

int rcm_create(int ncol, int nlayer, int nangle, int device, void *stream, rcm_ensemble **out) try {
  if (!out) return fail(RCM_ERR_ARG, "rcm_create: out is NULL");
  *out = nullptr;
  if (ncol < 1) return fail(RCM_ERR_ARG, "rcm_create: ncol must be >= 1");
  if (nlayer != NL) return fail(RCM_ERR_ARG, "rcm_create: kernels are specialised for nlayer == 20 (main.cpp:78)");
  if (nangle < 1 || nangle > 64) return fail(RCM_ERR_ARG, "rcm_create: nangle must be in 1..64");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(RCM_ERR_HIP, std::string("rcm_create: no HIP device (this library has no CPU path): hipGetDeviceCount -> ") +
                                 hipGetErrorString(e) + ", count " + std::to_string(ndev));
  if (device < 0 || device >= ndev) return fail(RCM_ERR_ARG, "rcm_create: device index out of range");
  rcm_ensemble *h = new (std::nothrow) rcm_ensemble();
  if (!h) return fail(RCM_ERR_NOMEM, "rcm_create: out of memory");
  h->ncol = ncol, h->nlayer = nlayer, h->nangle = nangle, h->device = device;
#ifdef RCM_EXPERIMENT  // A/B builds of tools/build_variants.py only
  if (const char *v = std::getenv("RCM_LDS_PAD")) h->lds_pad = std::atoi(v);
  if (const char *v = std::getenv("RCM_BLOCK_RECORDS")) h->block_records = std::atoi(v);
  if (const char *v = std::getenv("RCM_FLIP")) h->flip_passes = std::atoi(v);
  if (const char *v = std::getenv("RCM_FEW_PRIO")) h->phase_prio = std::atoi(v);
#endif
#ifdef RCM_WITH_VARIANTS  // kernel selection by environment: test-side library only
  if (const char *v = std::getenv("RCM_EXP_VARIANT")) h->exp_variant = std::atoi(v) ? 1 : 0;
  if (const char *v = std::getenv("RCM_CUBE")) h->cube = std::atoi(v) ? 1 : 0;
  if (const char *v = std::getenv("RCM_PLANCK_FAST")) h->planck_fast = std::atoi(v) ? 1 : 0;
  if (const char *v = std::getenv("RCM_PAIR")) h->pair = std::atoi(v) ? 1 : 0;
  if (const char *v = std::getenv("RCM_FLAT")) h->flat = std::atoi(v) ? 1 : 0;
#endif
  struct Guard {  // a half-built handle (and its device buffers) goes away on every way out, an exception included
    rcm_ensemble *h;
    ~Guard() {
      if (h) rcm_destroy(h);
    }
  } guard{h};
  if (int rc = use_device(h)) return rc;
  if (stream) {
    h->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreate(&h->stream) != hipSuccess) return fail(RCM_ERR_HIP, "hipStreamCreate failed");
    h->own_stream = true;
  }
  Dev &d = h->d;
  d.ncol = ncol, d.nangle = nangle, d.mode = rcm::MODE_NONE;
  d.merged_slot = -1;
  d.dp = 1000.0 / (double)nlayer;  // main.cpp:355
  d.sigdpi = 5.67032E-8 / M_PI;    // cplkavg.cpp:121,134
  const size_t nT = (size_t)ncol * NL, nE = (size_t)ncol * NLEV;
  int rc = 0;
  rc = rc ? rc : dev_alloc(h, &d.T, nT);
  rc = rc ? rc : dev_alloc(h, &d.Ttau, nT);
  rc = rc ? rc : dev_alloc(h, &d.Tsfc, (size_t)ncol);
  rc = rc ? rc : dev_alloc(h, &d.dt, (size_t)ncol);
  rc = rc ? rc : dev_alloc(h, &d.time_h, (size_t)ncol);
  rc = rc ? rc : dev_alloc(h, (double **)&d.rel_hum, nT);
  rc = rc ? rc : dev_alloc(h, &d.E_down, nE);
  rc = rc ? rc : dev_alloc(h, &d.E_up, nE);
  rc = rc ? rc : dev_alloc(h, &d.dE, nT);
  rc = rc ? rc : dev_alloc(h, &h->scratch, (size_t)2 * NL);
  rc = rc ? rc : dev_alloc(h, (double **)&d.solar_irr, (size_t)ncol);
  rc = rc ? rc : dev_alloc(h, (double **)&d.cloud_tau, (size_t)ncol);
  rc = rc ? rc : dev_alloc(h, (int **)&d.cloud_layer, (size_t)ncol);
  rc = rc ? rc : dev_alloc(h, &d.resid, (size_t)ncol);

  rc = rc ? rc : dev_alloc(h, &d.Tprev, nT);
  rc = rc ? rc : dev_alloc(h, (double **)&d.player, (size_t)NL);
  rc = rc ? rc : dev_alloc(h, (double **)&d.conv, (size_t)NL);
  rc = rc ? rc : dev_alloc(h, &h->d_ip, (size_t)NL);
  rc = rc ? rc : dev_alloc(h, &h->d_delP, (size_t)NL);
  rc = rc ? rc : dev_alloc(h, &h->d_numDens, (size_t)NL);
  if (rc) return rc;
  d.ip = h->d_ip, d.delP = h->d_delP, d.numDens = h->d_numDens;
  // angles: mu_i = dmu/2 + dmu*i (main.cpp:356,482); weight 2*pi*mu*dmu (main.cpp:302,308,313)
  std::vector<double> ang(2 * (size_t)nangle + 1, 0.0);
  const double dmu = 1.0 / (double)nangle;
  for (int i = 0; i < nangle; ++i) {
    const double mu = dmu / 2.0 + dmu * (double)i;
    ang[i] = 1.0 / mu;
    ang[nangle + i] = 2 * M_PI * mu * dmu;
    ang[2 * nangle] += ang[nangle + i];  // sum of the quadrature weights (= pi up to rounding)
  }
  // Root schedule of the lane-pair kernels (pair_all_angles): odd angle numbers n = 2i+1 with 3n > N need
  // an exponential of their own (the others follow by cubing); two roots 3m and 5m share the exponential
  // of the fictitious angle number 15m.  Entry = {a1 = -log2(e)/mu, bits of 4*n + mode}.
  {
    const int N = 2 * nangle - 1;
    const double log2e = 0x1.71547652b82fep+0;
    std::vector<char> used((size_t)N + 1, 0);
    auto is_root = [&](int n) { return n >= 1 && n <= N && (n & 1) && 3 * n > N; };
    auto push = [&](double a1, int n, int mode) {
      const long long code = 4LL * n + mode;
      double bits;
      std::memcpy(&bits, &code, sizeof bits);
      ang.push_back(a1), ang.push_back(bits);
    };
    d.nsched = 0;
    for (int m = N / 5; RCM_GEN15 && m >= 1; --m) {
      if (!(m & 1) || !is_root(3 * m) || !is_root(5 * m) || used[3 * m] || used[5 * m]) continue;
      used[3 * m] = used[5 * m] = 1;
      push(-(2.0 * nangle) / (15.0 * m) * log2e, 3 * m, 1);
      push(0.0, 5 * m, 2);
      d.nsched += 2;
    }
    for (int n = N; n >= 1; n -= 2)
      if (is_root(n) && !used[n]) push(-ang[(n - 1) >> 1] * log2e, n, 0), ++d.nsched;
  }
  if ((rc = dev_upload(h, (double **)&d.ang, ang.data(), ang.size()))) return rc;
  hipError_t me = hipMemsetAsync(d.time_h, 0, ncol * sizeof(float), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(d.dt, 0, ncol * sizeof(double), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(d.E_down, 0, nE * sizeof(double), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(d.E_up, 0, nE * sizeof(double), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(d.dE, 0, nT * sizeof(double), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(d.resid, 0, ncol * sizeof(double), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(d.Tprev, 0, nT * sizeof(double), h->stream);
  if (me == hipSuccess) me = hipStreamSynchronize(h->stream);
  if (me != hipSuccess) return fail(RCM_ERR_HIP, std::string("rcm_create: ") + hipGetErrorString(me));
  // the reference's compiled-in forcing until the caller says otherwise: cloud in layer 17 with
  // tau_s/2 = 1.0 (main.cpp:86,92,267), solar heating 0
  if ((rc = rcm_set_forcing(h, 0.0, 17, 2.0 / 2.0))) return rc;
  guard.h = nullptr;
  *out = h;
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_destroy(rcm_ensemble *h) try {
  if (!h) return fail(RCM_ERR_ARG, "rcm_destroy: NULL handle");
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void *p : h->allocs) (void)hipFree(p);
  if (h->partial_buf) (void)hipFree(h->partial_buf);
  if (h->slice_buf) (void)hipFree(h->slice_buf);
  if (h->gather_send) (void)hipFree(h->gather_send);
  for (auto &ev : h->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_set_grid(rcm_ensemble *h, const double *plevel_hPa) try {
  if (!h || !plevel_hPa) return fail(RCM_ERR_ARG, "rcm_set_grid: bad argument");
  if (int rc = use_device(h)) return rc;
  h->plevel.assign(plevel_hPa, plevel_hPa + NLEV);
  h->player.resize(NL);
  h->conv.resize(NL);
  for (int i = 0; i < NL; ++i) {
    h->player[i] = (plevel_hPa[i] + plevel_hPa[i + 1]) / 2.0;  // main.cpp:472
    if (!(h->player[i] > 0)) return fail(RCM_ERR_ARG, "rcm_set_grid: layer pressure must be positive");
    h->conv[i] = std::pow(1000.0 / h->player[i], 2.0 / 7.0);   // main.cpp:474, kappa main.cpp:66
  }
  HIPCK(hipMemcpyAsync((void *)h->d.player, h->player.data(), NL * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync((void *)h->d.conv, h->conv.data(), NL * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->have_grid = true;
  h->img_valid = false;
  return derive_repwvl_layers(h);
} RCM_ABI_CATCH

int rcm_set_forcing(rcm_ensemble *h, double solar_irr, int cloud_layer, double cloud_tau) try {
  if (!h || cloud_layer < -1 || cloud_layer >= NL) return fail(RCM_ERR_ARG, "rcm_set_forcing: bad argument");
  if (int rc = use_device(h)) return rc;
  std::vector<double> a((size_t)h->ncol, solar_irr), b((size_t)h->ncol, cloud_tau);
  std::vector<int> c((size_t)h->ncol, cloud_layer);
  return rcm_set_column_forcing(h, a.data(), c.data(), b.data());
} RCM_ABI_CATCH

int rcm_set_column_forcing(rcm_ensemble *h, const double *solar_irr, const int *cloud_layer, const double *cloud_tau) try {
  if (!h) return fail(RCM_ERR_ARG, "rcm_set_column_forcing: NULL handle");
  if (cloud_layer)
    for (int c = 0; c < h->ncol; ++c)
      if (cloud_layer[c] < -1 || cloud_layer[c] >= NL)
        return fail(RCM_ERR_ARG, "rcm_set_column_forcing: cloud_layer must be in -1..nlayer-1");
  if (int rc = use_device(h)) return rc;
  const size_t n = (size_t)h->ncol;
  if (solar_irr)
    HIPCK(hipMemcpyAsync((void *)h->d.solar_irr, solar_irr, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (cloud_layer)
    HIPCK(hipMemcpyAsync((void *)h->d.cloud_layer, cloud_layer, n * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (cloud_tau)
    HIPCK(hipMemcpyAsync((void *)h->d.cloud_tau, cloud_tau, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->img_valid = false;  // the cloud row is part of the stored column images (k_pre rewrites them at the next step)
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_solar_setup_columns(rcm_ensemble *h, const double *tau_s, const double *mu_s, const double *albedo,
                            int doublings, double g_asym, double daytime, double E_0, double *solar_out) try {
  if (!h || !tau_s || !mu_s || !albedo || doublings < 0)
    return fail(RCM_ERR_ARG, "rcm_solar_setup_columns: bad argument");
  for (int c = 0; c < h->ncol; ++c)
    if (!(mu_s[c] > 0)) return fail(RCM_ERR_ARG, "rcm_solar_setup_columns: mu_s must be positive");
  if (int rc = use_device(h)) return rc;
  const size_t n = (size_t)h->ncol;
  double *buf = nullptr;
  HIPCK(hipMalloc((void **)&buf, n * 10 * sizeof(double)));
  double *d_tau = buf, *d_mu = buf + n, *d_alb = buf + 2 * n, *d_out = buf + 3 * n;
  hipError_t e = hipMemcpyAsync(d_tau, tau_s, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_mu, mu_s, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_alb, albedo, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rcm::k_solar, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, h->stream, h->ncol, d_tau, d_mu,
                       d_alb, doublings, g_asym, daytime, E_0, (double *)h->d.solar_irr, (double *)h->d.cloud_tau,
                       d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess && solar_out)
    e = hipMemcpyAsync(solar_out, d_out, n * 7 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(RCM_ERR_HIP, std::string("rcm_solar_setup_columns: ") + hipGetErrorString(e));
  h->img_valid = false;  // k_solar rewrote the cloud optical depths
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_get_residual(rcm_ensemble *h, double *resid) try {
  if (!h || !resid) return fail(RCM_ERR_ARG, "rcm_get_residual: bad argument");
  if (int rc = use_device(h)) return rc;
  HIPCK(hipMemcpyAsync(resid, h->d.resid, (size_t)h->ncol * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_run_to_equilibrium(rcm_ensemble *h, double tol_K, int max_steps, int check_every, int *steps_done,
                           double *residual_K) try {
  if (!h || !(tol_K >= 0) || max_steps < 0 || check_every < 1)
    return fail(RCM_ERR_ARG, "rcm_run_to_equilibrium: bad argument");
  std::vector<double> r((size_t)h->ncol);
  int done = 0;
  double worst = HUGE_VAL;
  while (done < max_steps) {
    const int n = std::min(check_every, max_steps - done);
    if (int rc = rcm_step(h, n)) return rc;
    done += n;
    if (int rc = rcm_get_residual(h, r.data())) return rc;
    worst = 0.0;
    for (double v : r) worst = (v > worst || v != v) ? v : worst;
    if (worst <= tol_K) break;
  }
  if (steps_done) *steps_done = done;
  if (residual_K) *residual_K = worst;
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_set_repwvl_tables(rcm_ensemble *h, int n_tpert, int n_species, int n_wvl, int n_p,
                          const double *p_grid, const double *t_ref, const double *t_pert,
                          const double *wvl, const double *weight, const double *xsec) try {
  if (!h || !p_grid || !t_ref || !t_pert || !wvl || !weight || !xsec)
    return fail(RCM_ERR_ARG, "rcm_set_repwvl_tables: NULL argument");
  if (n_species != RCM_NSPEC_REPWVL) return fail(RCM_ERR_ARG, "rcm_set_repwvl_tables: n_species must be 9");
  if (n_tpert < 2 || n_p < 2 || n_wvl < 1) return fail(RCM_ERR_ARG, "rcm_set_repwvl_tables: bad dimensions");
  if (n_tpert > 16) return fail(RCM_ERR_ARG, "rcm_set_repwvl_tables: at most 16 temperature perturbations (the shipped tables have 9)");
  if (h->mode != rcm::MODE_NONE) return fail(RCM_ERR_STATE, "rcm_set_repwvl_tables: tables already set");
  if ((unsigned long long)n_tpert * n_species * n_p * n_wvl * sizeof(double) >= (1ull << 32))
    return fail(RCM_ERR_ARG, "rcm_set_repwvl_tables: the table must stay below 4 GiB (32-bit offsets in the kernel)");
  // ... and so must the table the production kernels really address: xp[layer][T bracket][species][wavelength][2],
  // through `unsigned xoff + w*16 + species*n_wvl*16` (repwvl_pair_tau); for n_p < 36 it is the larger of the two
  if ((unsigned long long)NL * (n_tpert - 1) * n_species * n_wvl * 2 * sizeof(double) >= (1ull << 32))
    return fail(RCM_ERR_ARG, "rcm_set_repwvl_tables: the pressure-interpolated table (20 layers x (n_tpert-1) x 9 x n_wvl "
                             "x 16 B) must stay below 4 GiB (32-bit offsets in the kernel)");
  if (int rc = use_device(h)) return rc;
  Dev &d = h->d;
  // re-lay out [t][s][w][p] -> [t][s][p][w] so that a wave's 64 wavelengths are contiguous
  std::vector<double> xt((size_t)n_tpert * n_species * n_p * n_wvl);
  for (int t = 0; t < n_tpert; ++t)
    for (int s = 0; s < n_species; ++s)
      for (int w = 0; w < n_wvl; ++w)
        for (int p = 0; p < n_p; ++p)
          xt[(((size_t)t * n_species + s) * n_p + p) * n_wvl + w] =
              xsec[(((size_t)t * n_species + s) * n_wvl + w) * n_p + p];
  // per-wavelength Planck constants in the reference's operation order (main.cpp:188-191)
  std::vector<double> pnum(n_wvl), lam5(n_wvl), lkb(n_wvl), rw_xc(n_wvl), rw_sc(n_wvl);
  const double hP = 6.62607e-34, cL = 299792458, kB = 1.380649e-23;  // main.cpp:70-72
  for (int w = 0; w < n_wvl; ++w) {
    double lam = wvl[w];
    lam *= 1e-9;
    pnum[w] = weight[w] * 2 * hP * std::pow(cL, 2);
    lam5[w] = std::pow(lam, 5);
    lkb[w] = lam * kB;
    rw_xc[w] = (6.62607e-34 * 299792458.0) / lkb[w];  // the lane-pair kernels' form (planck_mono_fast)
    rw_sc[w] = pnum[w] / lam5[w] / 1e9;
  }
  int rc = 0;
  rc = rc ? rc : dev_upload(h, (double **)&d.xsec, xt.data(), xt.size());
  rc = rc ? rc : dev_alloc(h, &h->d_xp, (size_t)NL * (n_tpert - 1) * n_species * n_wvl * 2);
  rc = rc ? rc : dev_alloc(h, &h->d_tref_l, (size_t)NL);
  rc = rc ? rc : dev_alloc(h, (char **)&d.img, (size_t)h->ncol * sizeof(rcm::RepwvlPairLds));  // 3.8 KB per column
  d.xp = h->d_xp;
  d.tref_l = h->d_tref_l;
  h->t_ref.assign(t_ref, t_ref + n_p);
  rc = rc ? rc : dev_upload(h, (double **)&d.t_ref, t_ref, (size_t)n_p);
  rc = rc ? rc : dev_upload(h, (double **)&d.t_pert, t_pert, (size_t)n_tpert);
  rc = rc ? rc : dev_upload(h, (double **)&d.pnum, pnum.data(), (size_t)n_wvl);
  rc = rc ? rc : dev_upload(h, (double **)&d.lam5, lam5.data(), (size_t)n_wvl);
  rc = rc ? rc : dev_upload(h, (double **)&d.lkb, lkb.data(), (size_t)n_wvl);
  rc = rc ? rc : dev_upload(h, (double **)&d.rw_xc, rw_xc.data(), (size_t)n_wvl);
  rc = rc ? rc : dev_upload(h, (double **)&d.rw_sc, rw_sc.data(), (size_t)n_wvl);
  if (rc) return rc;
  h->p_grid.assign(p_grid, p_grid + n_p);
  h->xsec_t = std::move(xt);
  d.n_tpert = n_tpert, d.n_species = n_species, d.n_p = n_p, d.n_wvl = n_wvl;
  d.nspec = n_wvl;
  d.ngas = RCM_NSPEC_REPWVL;
  d.mode = h->mode = rcm::MODE_REPWVL;
  d.species_mask = d.prod_mask = 0x1ff;
  d.nact = RCM_NSPEC_REPWVL;
  d.act_nibbles = 0;
  for (int s = 0; s < RCM_NSPEC_REPWVL; ++s) d.act[s] = s, d.act_nibbles |= (unsigned long long)s << (4 * s);
  d.merged_slot = -1;
  h->shared_set = 0;
  return derive_repwvl_layers(h);
} RCM_ABI_CATCH

int rcm_set_lbl_tables(rcm_ensemble *h, int n_nodes, const double *wvl,
                       const double *const tabs[RCM_NSPEC_LBL], const double *h2o_ref) try {
  if (!h || !wvl || !tabs || !h2o_ref) return fail(RCM_ERR_ARG, "rcm_set_lbl_tables: NULL argument");
  if (n_nodes < 2) return fail(RCM_ERR_ARG, "rcm_set_lbl_tables: need at least 2 wavelength nodes");
  for (int s = 0; s < RCM_NSPEC_LBL; ++s)
    if (!tabs[s]) return fail(RCM_ERR_ARG, "rcm_set_lbl_tables: NULL species table");
  for (int i = 0; i + 1 < n_nodes; ++i)  // cplkavg.cpp:144-146 would exit(1) on these
    if (!(wvl[i] > 0) || !(wvl[i] < wvl[i + 1]))
      return fail(RCM_ERR_ARG, "rcm_set_lbl_tables: wavelengths must be positive and strictly increasing");
  for (int l = 0; l < NL; ++l)
    if (!(h2o_ref[l] > 0)) return fail(RCM_ERR_ARG, "rcm_set_lbl_tables: h2o_ref must be positive");
  if (h->mode != rcm::MODE_NONE) return fail(RCM_ERR_STATE, "rcm_set_lbl_tables: tables already set");
  if (int rc = use_device(h)) return rc;
  Dev &d = h->d;
  const long nband = n_nodes - 1;
  const long ld = (nband + 63) / 64 * 64;
  if ((unsigned long long)RCM_NSPEC_LBL * NL * ld * sizeof(double) >= (1ull << 32))
    return fail(RCM_ERR_ARG, "rcm_set_lbl_tables: the five tables must stay below 4 GiB (32-bit offsets in the kernel)");
  double *tab = nullptr, *stage = nullptr;
  if (int rc = dev_alloc(h, &tab, (size_t)RCM_NSPEC_LBL * NL * ld)) return rc;
  HIPCK(hipMalloc((void **)&stage, (size_t)n_nodes * NL * sizeof(double)));
  for (int s = 0; s < RCM_NSPEC_LBL; ++s) {
    hipError_t e = hipMemcpyAsync(stage, tabs[s], (size_t)n_nodes * NL * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(rcm::k_transpose_lbl, dim3((unsigned)(ld / 64)), dim3(256), 0, h->stream, stage,
                         tab + (size_t)s * NL * ld, (long)n_nodes, ld);
      e = hipStreamSynchronize(h->stream);
    }
    if (e != hipSuccess) {
      (void)hipFree(stage);
      return fail(RCM_ERR_HIP, std::string("rcm_set_lbl_tables: ") + hipGetErrorString(e));
    }
  }
  (void)hipFree(stage);
  int rc = 0;
  rc = rc ? rc : dev_upload(h, (double **)&d.lbl_wvl, wvl, (size_t)n_nodes);
  std::vector<double> wnum((size_t)n_nodes);
  for (int i = 0; i < n_nodes; ++i) wnum[i] = 1.0E7 / wvl[i];  // cplkavg.cpp:141-142
  rc = rc ? rc : dev_upload(h, (double **)&d.lbl_wnum, wnum.data(), wnum.size());
  rc = rc ? rc : dev_upload(h, (double **)&d.h2o_ref, h2o_ref, (size_t)NL);
  if (rc) return rc;
  d.lbl_tab = tab;
  d.lbl_ld = ld;
  d.nspec = (int)nband;
  d.ngas = RCM_NSPEC_LBL;
  d.mode = h->mode = rcm::MODE_LBL;
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_set_columns(rcm_ensemble *h, const double *Tlayer0, const double *T_surface0,
                    const double *gas, const double *rel_hum) try {
  if (!h || !Tlayer0 || !T_surface0 || !gas || !rel_hum) return fail(RCM_ERR_ARG, "rcm_set_columns: NULL argument");
  if (h->mode == rcm::MODE_NONE) return fail(RCM_ERR_STATE, "rcm_set_columns: set the spectral tables first");
  if (int rc = use_device(h)) return rc;
  Dev &d = h->d;
  const size_t nT = (size_t)h->ncol * NL, nG = nT * d.ngas;
  if (!d.gas)
    if (int rc = dev_alloc(h, &d.gas, nG)) return rc;
  if (h->mode == rcm::MODE_REPWVL)
    if (int rc = classify_species(h, gas)) return rc;
  HIPCK(hipMemcpyAsync(d.T, Tlayer0, nT * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(d.Ttau, Tlayer0, nT * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(d.Tsfc, T_surface0, h->ncol * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(d.gas, gas, nG * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync((void *)d.rel_hum, rel_hum, nT * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemsetAsync(d.time_h, 0, h->ncol * sizeof(float), h->stream));
  HIPCK(hipMemsetAsync(d.dt, 0, h->ncol * sizeof(double), h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  h->step_index = 0;
  h->have_columns = true;
  h->img_valid = false;
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_set_species_merging(rcm_ensemble *h, int on) try {
  if (!h || (on != 0 && on != 1)) return fail(RCM_ERR_ARG, "rcm_set_species_merging: on must be 0 or 1");
  if (h->merge_species == (on != 0)) return RCM_OK;
  h->merge_species = on != 0;
  if (h->mode != rcm::MODE_REPWVL || !h->have_columns) return RCM_OK;  // takes effect at rcm_set_columns
  if (int rc = use_device(h)) return rc;
  std::vector<double> gas((size_t)h->ncol * RCM_NSPEC_REPWVL * NL);  // re-classify the columns the handle holds
  HIPCK(hipMemcpyAsync(gas.data(), h->d.gas, gas.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return classify_species(h, gas.data());
} RCM_ABI_CATCH

int rcm_set_spectral_chunks(rcm_ensemble *h, int nchunk) try {
  if (!h || nchunk < 0) return fail(RCM_ERR_ARG, "rcm_set_spectral_chunks: bad argument");
  h->user_chunks = nchunk;
  h->d.partial = nullptr;  // re-derive on the next step
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_get_spectral_chunks(rcm_ensemble *h) try {
  if (!h || h->mode == rcm::MODE_NONE) return fail(RCM_ERR_STATE, "rcm_get_spectral_chunks: tables not set");
  if (use_flat(h)) return (h->d.n_wvl + 31) / 32 + 1;
  int nchunk = 0, len = 0;
  choose_chunks(h->ncol, h->d.nspec, h->user_chunks, &nchunk, &len);
  return nchunk;
} RCM_ABI_CATCH

int rcm_step(rcm_ensemble *h, int nsteps) try {
  if (!h || nsteps < 0) return fail(RCM_ERR_ARG, "rcm_step: bad argument");
  if (h->mode == rcm::MODE_NONE || !h->have_grid || !h->have_columns)
    return fail(RCM_ERR_STATE, "rcm_step: grid, spectral tables and columns must be set first");
  if (int rc = use_device(h)) return rc;
#ifdef RCM_WITH_VARIANTS
  if (const int lpc = choose_lpc(h)) {
    h->img_valid = false;  // k_step_repwvl keeps its images in LDS: d.img is stale after it
    return launch_fused_steps(h, lpc, nsteps);
  }
#endif
  if (!h->d.partial)
    if (int rc = ensure_partials(h)) return rc;
  for (int i = 0; i < nsteps; ++i)
    if (int rc = launch_step(h, i > 0, i + 1 < nsteps)) return rc;
  // the last k_post of a call does not go on with the next iteration: d.img is the image of the last sweep
  if (nsteps > 0) h->img_valid = h->mode == rcm::MODE_REPWVL;
  return RCM_OK;
} RCM_ABI_CATCH

#ifdef RCM_DBG_STAMPS  // profiling builds only; not declared in include/rcm.h
int rcm_dbg_stamps(rcm_ensemble *h, unsigned long long *out, int n) try {
  if (!h->d.dbg) {
    HIPCK(hipMalloc((void **)&h->d.dbg, (size_t)n * sizeof(unsigned long long)));
    HIPCK(hipMemset(h->d.dbg, 0, (size_t)n * sizeof(unsigned long long)));
    HIPCK(hipMemcpyToSymbol(HIP_SYMBOL(rcm::g_stamps), &h->d.dbg, sizeof(h->d.dbg)));
    return RCM_OK;
  }
  HIPCK(hipStreamSynchronize(h->stream));
  HIPCK(hipMemcpy(out, h->d.dbg, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return RCM_OK;
} RCM_ABI_CATCH
#endif

#ifdef RCM_WITH_VARIANTS  // test-side build only; not declared in include/rcm.h
// Point-by-point evaluation of an elementary device function (k_eval_device_fn, rcm_variants.hpp); host arrays.
int rcm_dbg_eval(int fn, int n, const double *a, const double *b, const double *c, double *out) try {
  if (n < 1 || !a || !b || !c || !out || fn < 0 || fn > 5) return fail(RCM_ERR_ARG, "rcm_dbg_eval: bad argument");
  double *d = nullptr;
  const size_t bytes = (size_t)n * sizeof(double);
  HIPCK(hipMalloc((void **)&d, 4 * bytes));
  hipError_t e = hipMemcpy(d, a, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + n, b, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + 2 * (size_t)n, c, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(rcm::k_eval_device_fn, dim3((n + 255) / 256), dim3(256), 0, nullptr, fn, n, d, d + n,
                       d + 2 * (size_t)n, d + 3 * (size_t)n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RCM_ERR_HIP, std::string("rcm_dbg_eval: ") + hipGetErrorString(e));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_set_step_fusion(rcm_ensemble *h, int mode) try {
  if (!h || mode < -1 || mode > 1) return fail(RCM_ERR_ARG, "rcm_set_step_fusion: mode must be -1 (automatic), 0 or 1");
  h->fusion = mode;
  h->img_valid = false;
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_get_step_fusion(rcm_ensemble *h) try {
  if (!h || h->mode == rcm::MODE_NONE) return fail(RCM_ERR_STATE, "rcm_get_step_fusion: tables not set");
  return choose_lpc(h);
} RCM_ABI_CATCH
#endif

int rcm_sync(rcm_ensemble *h) try {
  if (!h) return fail(RCM_ERR_ARG, "rcm_sync: NULL handle");
  if (int rc = use_device(h)) return rc;
  HIPCK(hipStreamSynchronize(h->stream));
  return RCM_OK;
} RCM_ABI_CATCH

static int copy_out(rcm_ensemble *h, void *dst, const void *src, size_t bytes) {
  if (!dst) return RCM_OK;
  HIPCK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  return RCM_OK;
}

int rcm_get_state(rcm_ensemble *h, double *Tlayer, double *T_surface, float *time_h, double *timestep) try {
  if (!h) return fail(RCM_ERR_ARG, "rcm_get_state: NULL handle");
  if (int rc = use_device(h)) return rc;
  int rc = 0;
  rc = rc ? rc : copy_out(h, Tlayer, h->d.T, (size_t)h->ncol * NL * sizeof(double));
  rc = rc ? rc : copy_out(h, T_surface, h->d.Tsfc, h->ncol * sizeof(double));
  rc = rc ? rc : copy_out(h, time_h, h->d.time_h, h->ncol * sizeof(float));
  rc = rc ? rc : copy_out(h, timestep, h->d.dt, h->ncol * sizeof(double));
  if (rc) return rc;
  HIPCK(hipStreamSynchronize(h->stream));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_get_fluxes(rcm_ensemble *h, double *E_down, double *E_up, double *dE) try {
  if (!h) return fail(RCM_ERR_ARG, "rcm_get_fluxes: NULL handle");
  if (int rc = use_device(h)) return rc;
  int rc = 0;
  rc = rc ? rc : copy_out(h, E_down, h->d.E_down, (size_t)h->ncol * NLEV * sizeof(double));
  rc = rc ? rc : copy_out(h, E_up, h->d.E_up, (size_t)h->ncol * NLEV * sizeof(double));
  rc = rc ? rc : copy_out(h, dE, h->d.dE, (size_t)h->ncol * NL * sizeof(double));
  if (rc) return rc;
  HIPCK(hipStreamSynchronize(h->stream));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_get_profile(rcm_ensemble *h, int col, double *Tlayer_adj, double *theta, float *time_h) try {
  if (!h || col < 0 || col >= h->ncol) return fail(RCM_ERR_ARG, "rcm_get_profile: bad argument");
  if (!h->have_grid || !h->have_columns) return fail(RCM_ERR_STATE, "rcm_get_profile: grid/columns not set");
  if (int rc = use_device(h)) return rc;
  hipLaunchKernelGGL(rcm::k_profile, dim3(1), dim3(64), 0, h->stream, h->d, col, h->scratch);
  HIPCK(hipGetLastError());
  int rc = 0;
  rc = rc ? rc : copy_out(h, Tlayer_adj, h->scratch, NL * sizeof(double));
  rc = rc ? rc : copy_out(h, theta, h->scratch + NL, NL * sizeof(double));
  rc = rc ? rc : copy_out(h, time_h, h->d.time_h + col, sizeof(float));
  if (rc) return rc;
  HIPCK(hipStreamSynchronize(h->stream));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_nspectral(rcm_ensemble *h) try {
  if (!h || h->mode == rcm::MODE_NONE) return fail(RCM_ERR_STATE, "rcm_nspectral: tables not set");
  return h->d.nspec;
} RCM_ABI_CATCH

int rcm_get_tau(rcm_ensemble *h, int col, double *tau) try {
  if (!h || !tau || col < 0 || col >= h->ncol) return fail(RCM_ERR_ARG, "rcm_get_tau: bad argument");
  if (h->mode == rcm::MODE_NONE || !h->have_grid || !h->have_columns)
    return fail(RCM_ERR_STATE, "rcm_get_tau: grid, tables and columns must be set first");
  if (int rc = use_device(h)) return rc;
  const size_t n = (size_t)h->d.nspec * NL;
  double *buf = nullptr;
  HIPCK(hipMalloc((void **)&buf, n * sizeof(double)));
  hipLaunchKernelGGL(rcm::k_tau, dim3((h->d.nspec + 63) / 64), dim3(64), 0, h->stream, h->d, col, buf);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(tau, buf, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(RCM_ERR_HIP, std::string("rcm_get_tau: ") + hipGetErrorString(e));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_get_tau_production(rcm_ensemble *h, int col, double *tau) try {
  if (!h || !tau || col < 0 || col >= h->ncol) return fail(RCM_ERR_ARG, "rcm_get_tau_production: bad argument");
  if (h->mode == rcm::MODE_NONE || !h->have_grid || !h->have_columns)
    return fail(RCM_ERR_STATE, "rcm_get_tau_production: grid, tables and columns must be set first");
  if (int rc = use_device(h)) return rc;
  const size_t n = (size_t)h->d.nspec * NL;
  double *buf = nullptr;
  HIPCK(hipMalloc((void **)&buf, n * sizeof(double)));
  if (use_flat(h)) {
    // what k_rt_repwvl_flat reads: the image stored in d.img (k_pre / post_column; here k_image when no step has
    // written it since the inputs last changed), copied, and the instantiation launch_rt_pair picks
    if (!h->img_valid)
      hipLaunchKernelGGL(rcm::k_image, dim3(1), dim3(64), 0, h->stream, h->d, col, h->step_index > 0 ? 1 : 0);
    const int na = h->d.nact;
    const auto kern = na <= 2 ? rcm::k_tau_img<2> : na <= 3 ? rcm::k_tau_img<3> : na <= 4 ? rcm::k_tau_img<4>
                    : na <= 5 ? rcm::k_tau_img<5> : na <= 7 ? rcm::k_tau_img<7> : rcm::k_tau_img<9>;
    hipLaunchKernelGGL(kern, dim3((h->d.nspec + 31) / 32), dim3(64), 0, h->stream, h->d, col, buf);
  } else {
    hipLaunchKernelGGL(rcm::k_tau_pair, dim3((h->d.nspec + 31) / 32), dim3(64), 0, h->stream, h->d, col, buf);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(tau, buf, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(buf);
  if (e != hipSuccess) return fail(RCM_ERR_HIP, std::string("rcm_get_tau_production: ") + hipGetErrorString(e));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_get_gas(rcm_ensemble *h, double *gas) try {
  if (!h || !gas) return fail(RCM_ERR_ARG, "rcm_get_gas: bad argument");
  if (!h->have_columns) return fail(RCM_ERR_STATE, "rcm_get_gas: columns not set");
  if (int rc = use_device(h)) return rc;
  if (int rc = copy_out(h, gas, h->d.gas, (size_t)h->ncol * h->d.ngas * NL * sizeof(double))) return rc;
  HIPCK(hipStreamSynchronize(h->stream));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_pack_state(rcm_ensemble *h, void *dev_out) try {
  if (!h || !dev_out) return fail(RCM_ERR_ARG, "rcm_pack_state: bad argument");
  if (!h->have_columns) return fail(RCM_ERR_STATE, "rcm_pack_state: columns not set");
  if (int rc = use_device(h)) return rc;
  const size_t n = (size_t)h->ncol * NLEV;
  hipLaunchKernelGGL(rcm::k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d, (double *)dev_out);
  HIPCK(hipGetLastError());
  return RCM_OK;
} RCM_ABI_CATCH

// ncclAllGather of the caller's RCCL.  The communicator comes from the caller, so the call must go to the
// very librccl the caller created it with: the symbol is looked up in the running process (global scope
// first, then the already loaded librccl by name) instead of linking a second copy in -- PyTorch ships its
// own librccl.so next to /opt/rocm's, and a communicator of one is garbage to the other.
namespace {
using allgather_fn = int (*)(const void *, void *, size_t, int, void *, hipStream_t);
allgather_fn find_allgather(std::string *where) {
  if (void *f = dlsym(RTLD_DEFAULT, "ncclAllGather")) {
    *where = "global scope";
    return (allgather_fn)f;
  }
  for (const char *name : {"librccl.so.1", "librccl.so"}) {
    if (void *lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD))
      if (void *f = dlsym(lib, "ncclAllGather")) {
        *where = name;
        return (allgather_fn)f;
      }
  }
  return nullptr;
}
}  // namespace

int rcm_gather(rcm_ensemble *h, void *nccl_comm, void *dev_recv, int ncol_pad) try {
  if (!h || !nccl_comm || !dev_recv || ncol_pad < h->ncol)
    return fail(RCM_ERR_ARG, "rcm_gather: bad argument (ncol_pad must be >= this handle's ncol)");
  if (!h->have_columns) return fail(RCM_ERR_STATE, "rcm_gather: columns not set");
  if (int rc = use_device(h)) return rc;
  std::string where;
  const allgather_fn allgather = find_allgather(&where);
  if (!allgather)
    return fail(RCM_ERR_STATE, "rcm_gather: no RCCL in this process (ncclAllGather not found): the communicator is the "
                               "caller's, so the caller's librccl must be loaded before this call");
  if (h->gather_rows < ncol_pad) {
    if (h->gather_send) {
      HIPCK(hipStreamSynchronize(h->stream));
      HIPCK(hipFree(h->gather_send));
      h->gather_send = nullptr;
    }
    HIPCK(hipMalloc((void **)&h->gather_send, (size_t)ncol_pad * NLEV * sizeof(double)));
    h->gather_rows = ncol_pad;
  }
  const size_t blk = (size_t)ncol_pad * NLEV;
  if (ncol_pad > h->ncol)  // the padding rows of the last shard travel as zeros
    HIPCK(hipMemsetAsync(h->gather_send + (size_t)h->ncol * NLEV, 0, (size_t)(ncol_pad - h->ncol) * NLEV * sizeof(double),
                         h->stream));
  if (int rc = rcm_pack_state(h, h->gather_send)) return rc;
  const int nrc = allgather(h->gather_send, dev_recv, blk, /* ncclFloat64 */ 8, nccl_comm, h->stream);
  if (nrc != 0) return fail(RCM_ERR_HIP, "rcm_gather: ncclAllGather (" + where + ") returned " + std::to_string(nrc));
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_checkpoint_save(rcm_ensemble *h, const char *path) try {
  if (!h || !path) return fail(RCM_ERR_ARG, "rcm_checkpoint_save: bad argument");
  if (!h->have_columns) return fail(RCM_ERR_STATE, "rcm_checkpoint_save: columns not set");
  if (int rc = use_device(h)) return rc;
  rcm::File f(std::fopen(path, "wb"));  // closed on every way out
  if (!f) return fail(RCM_ERR_IO, std::string("rcm_checkpoint_save: cannot open ") + path);
  const long long hdr[5] = {h->ncol, NL, h->d.ngas, h->mode, h->step_index};
  bool ok = std::fwrite("RCMCKPT1", 1, 8, f) == 8 && std::fwrite(hdr, sizeof(long long), 5, f) == 5;
  std::vector<char> buf;
  for (const auto &fld : ckpt_fields(h)) {
    buf.resize(fld.bytes);
    hipError_t e = hipMemcpyAsync(buf.data(), fld.dev, fld.bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess)
      return fail(RCM_ERR_HIP, std::string("rcm_checkpoint_save: ") + hipGetErrorString(e));
    ok = ok && std::fwrite(buf.data(), 1, fld.bytes, f) == fld.bytes;
  }
  ok = (f.close() == 0) && ok;
  return ok ? RCM_OK : fail(RCM_ERR_IO, std::string("rcm_checkpoint_save: write failed: ") + path);
} RCM_ABI_CATCH

int rcm_checkpoint_load(rcm_ensemble *h, const char *path) try {
  if (!h || !path) return fail(RCM_ERR_ARG, "rcm_checkpoint_load: bad argument");
  if (h->mode == rcm::MODE_NONE || !h->have_grid)
    return fail(RCM_ERR_STATE, "rcm_checkpoint_load: set the grid and the spectral tables first");
  if (int rc = use_device(h)) return rc;
  rcm::File f(std::fopen(path, "rb"));  // closed on every way out
  if (!f) return fail(RCM_ERR_IO, std::string("rcm_checkpoint_load: cannot open ") + path);
  char magic[8];
  long long hdr[5];
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "RCMCKPT1", 8) != 0 ||
      std::fread(hdr, sizeof(long long), 5, f) != 5)
    return fail(RCM_ERR_FORMAT, std::string("rcm_checkpoint_load: not an RCMCKPT1 file: ") + path);
  if (hdr[0] != h->ncol || hdr[1] != NL || hdr[2] != h->d.ngas || hdr[3] != h->mode || hdr[4] < 0)
    return fail(RCM_ERR_ARG, "rcm_checkpoint_load: checkpoint does not match this ensemble (ncol/nlayer/species/mode)");
  if (!h->d.gas)
    if (int rc = dev_alloc(h, &h->d.gas, (size_t)h->ncol * NL * h->d.ngas))
      return rc;
  std::vector<char> buf;
  for (const auto &fld : ckpt_fields(h)) {
    buf.resize(fld.bytes);
    if (std::fread(buf.data(), 1, fld.bytes, f) != fld.bytes)
      return fail(RCM_ERR_FORMAT, std::string("rcm_checkpoint_load: truncated file: ") + path);
    hipError_t e = hipMemcpyAsync(fld.dev, buf.data(), fld.bytes, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess)
      return fail(RCM_ERR_HIP, std::string("rcm_checkpoint_load: ") + hipGetErrorString(e));
    if (fld.dev == (void *)h->d.gas && h->mode == rcm::MODE_REPWVL)  // the species pattern of the run that saved it
      if (int rc = classify_species(h, reinterpret_cast<const double *>(buf.data())))
        return rc;
  }
  h->step_index = hdr[4];
  h->have_columns = true;
  h->img_valid = false;  // (classify_species cleared the images; the next step or the optical-depth getter rebuilds them)
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_set_rt_timing(rcm_ensemble *h, int enable) try {
  if (!h) return fail(RCM_ERR_ARG, "rcm_set_rt_timing: NULL handle");
  h->timing = enable != 0;
  h->events_used = 0;
  return RCM_OK;
} RCM_ABI_CATCH

int rcm_get_rt_timing(rcm_ensemble *h, double *ms_total, int *launches) try {
  if (!h) return fail(RCM_ERR_ARG, "rcm_get_rt_timing: NULL handle");
  if (int rc = use_device(h)) return rc;
  HIPCK(hipStreamSynchronize(h->stream));
  double total = 0.0;
  for (size_t i = 0; i < h->events_used; ++i) {
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, h->events[i].first, h->events[i].second));
    total += ms;
  }
  if (ms_total) *ms_total = total;
  if (launches) *launches = (int)h->events_used;
  h->events_used = 0;
  return RCM_OK;
} RCM_ABI_CATCH

// ---- measured HBM bandwidth (bench.py: roofline.peak_measured) -------------------------------------------------
namespace {
// Geometries found fastest on gfx950 by tools/ubench/stream_bw.hip (2 GiB buffers): the copy as ONE 16-byte piece per
// lane and no loop (6.42 TB/s read + write; grid-stride loops of any unroll: 4.4 - 5.8); the read-only pass as
// non-temporal loads, 8 independent pieces per lane and iteration, 16 384 blocks (6.56 TB/s; plain loads 5.0 - 5.9).
typedef double rcm_d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_stream_copy(const rcm_d2 *__restrict__ src, rcm_d2 *__restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_stream_read(const rcm_d2 *__restrict__ src, double *sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double a = 0.0, b = 0.0;
  for (; i + 7 * stride < n; i += 8 * stride) {
    rcm_d2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u].x, b += v[u].y;
  }
  for (; i < n; i += stride) a += src[i].x, b += src[i].y;
  if (a + b == 1.2345e300) sink[0] = a;  // keeps the loads; never true for the zero-filled buffer
}
}  // namespace

int rcm_measure_hbm_bandwidth(int device, size_t bytes, int reps, double *copy_gbs, double *read_gbs) try {
  if (bytes < ((size_t)64 << 20) || reps < 1 || reps > 1000) return fail(RCM_ERR_ARG, "rcm_measure_hbm_bandwidth: bytes >= 64 MiB, 1 <= reps <= 1000");
  HIPCK(hipSetDevice(device));
  struct Bufs {  // freed on every way out
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Bufs() {
      if (a) (void)hipFree(a);
      if (b) (void)hipFree(b);
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } m;
  const size_t n = bytes / sizeof(rcm_d2);
  if ((n + 255) / 256 > 0x7fffffffull) return fail(RCM_ERR_ARG, "rcm_measure_hbm_bandwidth: at most 8 TiB");
  HIPCK(hipMalloc(&m.a, n * sizeof(rcm_d2)));
  HIPCK(hipMalloc(&m.b, n * sizeof(rcm_d2)));
  HIPCK(hipMemset(m.a, 0, n * sizeof(rcm_d2)));
  HIPCK(hipMemset(m.b, 0, n * sizeof(rcm_d2)));
  HIPCK(hipEventCreate(&m.e0));
  HIPCK(hipEventCreate(&m.e1));
  double best_copy = 0.0, best_read = 0.0;
  for (int r = 0; r <= reps; ++r) {  // r = 0 warms up
    float ms = 0.f;
    HIPCK(hipEventRecord(m.e0, nullptr));
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (const rcm_d2 *)m.a,
                       (rcm_d2 *)m.b, n);
    HIPCK(hipEventRecord(m.e1, nullptr));
    HIPCK(hipEventSynchronize(m.e1));
    HIPCK(hipEventElapsedTime(&ms, m.e0, m.e1));
    if (r && ms > 0.f) best_copy = std::max(best_copy, 2.0 * (double)(n * sizeof(rcm_d2)) / (ms * 1e-3) / 1e9);
    HIPCK(hipEventRecord(m.e0, nullptr));
    hipLaunchKernelGGL(k_stream_read, dim3(16384), dim3(256), 0, nullptr, (const rcm_d2 *)m.a, (double *)m.b, n);
    HIPCK(hipEventRecord(m.e1, nullptr));
    HIPCK(hipEventSynchronize(m.e1));
    HIPCK(hipEventElapsedTime(&ms, m.e0, m.e1));
    if (r && ms > 0.f) best_read = std::max(best_read, (double)(n * sizeof(rcm_d2)) / (ms * 1e-3) / 1e9);
  }
  HIPCK(hipGetLastError());
  if (copy_gbs) *copy_gbs = best_copy;
  if (read_gbs) *read_gbs = best_read;
  return RCM_OK;
} RCM_ABI_CATCH

}  // extern "C"
