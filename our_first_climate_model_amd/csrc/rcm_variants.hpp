// rcm_variants.hpp -- TEST-SIDE KERNELS.  Included by rcm_kernels.hpp (inside namespace rcm, after everything it
// declares) only under -DRCM_WITH_VARIANTS, i.e. only by tests/_variants/librcm_hip_variants.so
// (our_first_climate_model_amd.build.build_variants); the production library librcm_hip.so never compiles this file.
//
//   sweep / all_angles / store_partial, k_rt_lbl<EXPV,CUBE,PLK>, k_rt_repwvl<EXPV,CUBE>
//       the first implementation of the path: one lane per spectral point, 41 flux accumulators per lane, library
//       exp / own exp, branch-for-branch cplkavg / narrow-band Planck, with and without the cube chains.  The tests
//       cross-check the production lane-pair kernels against them (test_chunking_and_kernel_variants_agree, ...).
//   k_step_repwvl<CUBE,LPC>
//       the whole time-loop body of the repwvl path in one kernel: built, measured, slower; a
//       complete second implementation of the step that the tests compare with the oracle.
#pragma once

// ---------------------------------------------------------------------------------------------
// Schwarzschild sweep for one angle (main.cpp:297-315) on the lane's spectral point.
// t[l] = exp(-tau_l/mu).  The reference's L = (1-alpha)*L + alpha*B with alpha = 1-t is
// evaluated as B + t*(L-B).
// ---------------------------------------------------------------------------------------------
// B = the lane's Planck column in LDS (B[l][0] for layer l, B[NL][0] for the surface): the level
// grid lives in LDS, the 41 flux accumulators and the transmissions in VGPRs.
__device__ __forceinline__ void sweep(const double (&t)[NL], const double (*B)[64], double cw,
                                      double (&Ed)[NLEV], double (&Eu)[NLEV]) {
  double L = 0.0;  // main.cpp:300
#pragma unroll
  for (int l = 0; l < NL; ++l) {  // main.cpp:306-309
    const double Bl = B[l][0];
    L = fma(t[l], L - Bl, Bl);
    Ed[l + 1] = fma(cw, L, Ed[l + 1]);
  }
  double Lu = B[NL][0];           // main.cpp:301
  Eu[NL] = fma(cw, Lu, Eu[NL]);   // main.cpp:302
#pragma unroll
  for (int l = NL - 1; l >= 0; --l) {  // main.cpp:311-314
    const double Bl = B[l][0];
    Lu = fma(t[l], Lu - Bl, Bl);
    Eu[l] = fma(cw, Lu, Eu[l]);
  }
}

// t_l = exp(-tau_l/mu) for one angle (main.cpp:207-212).  EXPV 1 expects clamp_tau()'d optical depths.
template <int EXPV>
__device__ __forceinline__ void transmissions(const double (&tau)[NL], double inv_mu, double (&t)[NL]) {
  if constexpr (EXPV == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) t[l] = exp(-tau[l] * inv_mu);
  } else {
    const double a1 = -inv_mu * 0x1.71547652b82fep+0;
#pragma unroll
    for (int l = 0; l < NL; ++l) t[l] = exp_trans(tau[l], a1);
  }
}

// All angles for one spectral point.  CUBE: mu_i = (2i+1)/(2 nangle), so the transmission at
// odd n = 2i+1 obeys t_{n/3} = t_n^3; only the angles with 3n > 2 nangle - 1 need an exp, the
// rest follow by two multiplications per layer (20 instead of 30 exp per layer at nangle 30).
template <int EXPV, int CUBE>
__device__ __forceinline__ void all_angles(const Dev &d, const double (&tau)[NL],
                                           const double (*B)[64], double (&Ed)[NLEV],
                                           double (&Eu)[NLEV]) {
  const int na = d.nangle;
  if constexpr (!CUBE) {
    for (int i = 0; i < na; ++i) {
      const double inv_mu = d.ang[i], cw = d.ang[na + i];
      double t[NL];
      transmissions<EXPV>(tau, inv_mu, t);  // main.cpp:209
      sweep(t, B, cw, Ed, Eu);
    }
  } else {
    const int N = 2 * na - 1;
    for (int n = N; 3 * n > N; n -= 2) {
      const double inv_mu = d.ang[(n - 1) >> 1];
      double t[NL];
      transmissions<EXPV>(tau, inv_mu, t);
      int m = n;
      for (;;) {
        sweep(t, B, d.ang[na + ((m - 1) >> 1)], Ed, Eu);
        if (m % 3 != 0) break;
        m /= 3;
#pragma unroll
        for (int l = 0; l < NL; ++l) t[l] = t[l] * t[l] * t[l];
      }
    }
  }
}

// Butterfly-reduce the accumulators of a wave and let lane 0 store the partial record.
__device__ __forceinline__ void store_partial(const Dev &d, int col, int chunk, int lane, bool valid,
                                              double (&Ed)[NLEV], double (&Eu)[NLEV]) {
  double *out = d.partial + ((size_t)col * d.nchunk + chunk) * PSTRIDE;
#pragma unroll
  for (int j = 0; j < NLEV; ++j) {
    const double a = wave_sum(Ed[j]);
    const double b = wave_sum(Eu[j]);
    if (lane == 0 && valid) {
      out[j] = a;
      out[NLEV + j] = b;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Thermal RT, lbl spectra: radiative_transfer (main.cpp:320-344) with the band Planck source
// cplkavg (cplkavg.cpp:124-243) and optical depths assembled on the fly.
// grid = ceil(nchunk * ncol / 4) blocks of 4 waves; consecutive waves work on the SAME spectral chunk
// of different columns, so a table tile is fetched from HBM once and then served from L2.
// ---------------------------------------------------------------------------------------------
template <int EXPV, int CUBE, int PLK>
__global__ __launch_bounds__(64 * WPB, RCM_LB) void k_rt_lbl(Dev d) {
  __shared__ double s_col[WPB][200];
  __shared__ double s_stage[WPB][NLEV][64];  // per-lane scratch column, indexable at run time
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // work item = (chunk, column), column fastest: the waves of a block and neighbouring blocks share
  // the chunk (its table tile stays in L2); a small ensemble still fills every wave slot
  const long nitem = (long)d.ncol * d.nchunk;
  const long item_raw = (long)blockIdx.x * WPB + wib;
  const bool valid = item_raw < nitem;
  const long item = valid ? item_raw : nitem - 1;
  const int chunk = (int)(item / d.ncol);
  const int col = (int)(item - (long)chunk * d.ncol);
  double(*st)[64] = s_stage[wib];
  // LDS: [0..19] Tlayer, [20] T_surface, [21..41] sigdpi*T^4, [48..147] gas multipliers [5][20],
  //      [150..170] sigdpi*T^4*conc, [171..191] 1/T
  double *sc = s_col[wib];
  const int cloud_layer = d.cloud_layer[col];
  const double cloud_tau = d.cloud_tau[col];
  if (lane < NL) sc[lane] = d.T[(size_t)col * NL + lane];
  if (lane == NL) sc[NL] = d.Tsfc[col];
  for (int i = lane; i < RCM_NSPEC_LBL * NL; i += 64) sc[48 + i] = d.gas[(size_t)col * RCM_NSPEC_LBL * NL + i];
  __syncthreads();
  // cplkavg.cpp:181,237: sigdpi * pow(t,4) (* conc on the Simpson path), once per column-level
  if (lane <= NL) {
    const double t = sc[lane];
    const double tp = d.sigdpi * pow(t, 4.0);
    sc[NLEV + lane] = tp;
    sc[150 + lane] = tp * kConc;
    sc[171 + lane] = 1.0 / t;
  }
  __syncthreads();

  double Ed[NLEV], Eu[NLEV];
#pragma unroll
  for (int j = 0; j < NLEV; ++j) Ed[j] = Eu[j] = 0.0;  // main.cpp:326-327

  const long b0 = (long)chunk * d.chunk_len;
  const long b1 = min(b0 + (long)d.chunk_len, (long)d.nspec);
  for (long b = b0 + lane; b < b1; b += 64) {  // main.cpp:329, lanes over spectral points
    double tau[NL];
#pragma unroll 4
    for (int l = 0; l < NL; ++l) st[l][lane] = lbl_tau_layer(d, sc + 48, b, l, cloud_layer, cloud_tau);
#pragma unroll
    for (int l = 0; l < NL; ++l) tau[l] = EXPV ? clamp_tau(st[l][lane]) : st[l][lane];
    const double wnumhi = 1.0E7 / d.lbl_wvl[b];      // cplkavg.cpp:141
    const double wnumlo = 1.0E7 / d.lbl_wvl[b + 1];  // cplkavg.cpp:142
    const double c2lo = 1.438786 * wnumlo, c2hi = 1.438786 * wnumhi;  // cplkavg.cpp:152-153
    const bool narrow = PLK && (wnumhi - wnumlo) / wnumhi < 1.e-2;     // cplkavg.cpp:155
#pragma unroll RCM_PLK_UNROLL
    for (int l = 0; l <= NL; ++l) {
      double Bl;
      if (!(narrow && cplkavg_fast(c2lo, c2hi, sc[l], sc[171 + l], sc[150 + l], &Bl)))
        Bl = cplkavg_dev(wnumlo, wnumhi, sc[l], sc[NLEV + l]);
      st[l][lane] = Bl;
    }
    all_angles<EXPV, CUBE>(d, tau, (const double(*)[64]) & st[0][lane], Ed, Eu);
  }
  store_partial(d, col, chunk, lane, valid, Ed, Eu);
}

// ---------------------------------------------------------------------------------------------
// Thermal RT, repwvl reduced-wavelength set (one lane per wavelength: reference variants): read_tau interpolation (repwvl_thermal.cpp:197-248)
// fused with radiative_transfer (main.cpp:320-344, monochromatic Planck main.cpp:186-204).
// tau never touches HBM.
// ---------------------------------------------------------------------------------------------
template <int EXPV, int CUBE>
__global__ __launch_bounds__(64 * WPB, 2) void k_rt_repwvl(Dev d) {
  __shared__ RepwvlLds s_col[WPB];
  __shared__ double s_stage[WPB][NLEV][64];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // work item = (chunk, column), column fastest: the waves of a block and neighbouring blocks share
  // the chunk (its table tile stays in L2); a small ensemble still fills every wave slot
  const long nitem = (long)d.ncol * d.nchunk;
  const long item_raw = (long)blockIdx.x * WPB + wib;
  const bool valid = item_raw < nitem;
  const long item = valid ? item_raw : nitem - 1;
  const int chunk = (int)(item / d.ncol);
  const int col = (int)(item - (long)chunk * d.ncol);
  RepwvlLds &s = s_col[wib];
  double(*st)[64] = s_stage[wib];
  repwvl_prologue(d, col, lane, s);
  __syncthreads();

  double Ed[NLEV], Eu[NLEV];
#pragma unroll
  for (int j = 0; j < NLEV; ++j) Ed[j] = Eu[j] = 0.0;

  const int w0 = chunk * d.chunk_len;
  const int w1 = min(w0 + d.chunk_len, d.nspec);
  for (int w = w0 + lane; w < w1; w += 64) {
    double tau[NL];
#pragma unroll 1
    for (int l = 0; l < NL; ++l) st[l][lane] = repwvl_tau_layer(d, s, w, l, d.cloud_layer[col], d.cloud_tau[col]);
#pragma unroll
    for (int l = 0; l < NL; ++l) tau[l] = EXPV ? clamp_tau(st[l][lane]) : st[l][lane];
    const double pnum = d.pnum[w], lam5 = d.lam5[w], lkb = d.lkb[w];
#pragma unroll 1
    for (int l = 0; l <= NL; ++l) st[l][lane] = planck_mono(pnum, lam5, lkb, s.T[l]);  // main.cpp:331, :301
    all_angles<EXPV, CUBE>(d, tau, (const double(*)[64]) & st[0][lane], Ed, Eu);
  }
  store_partial(d, col, chunk, lane, valid, Ed, Eu);
}

// ---------------------------------------------------------------------------------------------
// THE WHOLE TIME-LOOP BODY main.cpp:531-583 FOR THE repwvl PATH IN ONE KERNEL -- built, measured, NOT the production
// path: test-side build only (-DRCM_WITH_VARIANTS; DESIGN.md 5 has the numbers).  10 000 x Reduced100: 525 us per
// step against 400 us for k_pre + k_rt_repwvl_flat + k_post.  Whole-column ownership cuts the work into 5 000
// waves of 7 passes for 3 072 wave slots; the second "round" leaves 37 % of the chip idle (a lone wave is latency
// bound: it takes as long per pass as one of three), and a pass of this kernel is no faster than a pass of the
// flattened one.  Kept because it is a complete second implementation of the step that the tests compare with
// the oracle and with the three-kernel path.
//
// A wave OWNS K = 32/LPC columns for the whole launch and runs `nsteps` iterations on them without leaving the
// kernel: columns are independent (no horizontal coupling anywhere in main.cpp), so nothing but the wave itself
// ever waits for a column's previous step.  Per iteration:
//   A  convective adjustment (rank sort of theta over 20 lanes per column, main.cpp:536-540), equilibrium
//      residual, water-vapour feedback (main.cpp:281-289)                                   [k_pre's work]
//   B  the column images of read_tau (T brackets, corner weights; repwvl_thermal.cpp:229-239)
//   C  radiative transfer (main.cpp:320-344): LPC lane pairs per column, lane pair j of a column takes the
//      wavelengths j, j+LPC, ..; a lane never changes column, so its 21 flux sums stay in VGPRs for all passes
//   D  butterfly over the column's LPC lane pairs -> E_down[21], E_up[21]
//   E  dE, time step, temperature update, time (main.cpp:337-341, 156-176, 581)            [k_post's work]
// Against k_pre + k_rt_repwvl_flat + k_post: no per-32-item column images, no segmented reduction through LDS, no
// partial records in HBM, no launch boundaries (3 per step), and the latency-bound phases A, B, D, E of one wave
// overlap the VALU-bound phase C of its neighbours instead of running in lock-step at every kernel start.
// State flows through LDS from one iteration to the next; HBM holds the same arrays as ever for the getters.
// One wave per block: waves of a block would only wait for each other.
// ---------------------------------------------------------------------------------------------
struct StepLds {     // per owned column
  double T[NL];      // Tlayer as the last thermodynamics left it (main.cpp:583)
  double Ttau[NL];   // temperatures the optical depths are interpolated at (main.cpp:500 / :564)
  double Tprev[NL];  // adjusted profile of the previous iteration (equilibrium residual)
  double h2o[NL];    // H2O mixing ratio (the row the feedback rewrites)
  double Tsfc;
  float time_h;      // the column's clock (a float, main.cpp:359)
};

#ifdef RCM_DBG_STAMPS  // profiling builds only: lane 0 of every 61st wave writes s_memtime at phase boundaries of its last step
#define RCM_STAMP(i)                                                                                        \
  do {                                                                                                      \
    if (d.dbg && step == nsteps - 1 && threadIdx.x == 0 && blockIdx.x % 61 == 0)                            \
      d.dbg[(size_t)(blockIdx.x / 61) * 64 + (i)] = __builtin_amdgcn_s_memtime();                           \
  } while (0)
#else
#define RCM_STAMP(i) \
  do {               \
  } while (0)
#endif

template <int CUBE, int LPC>
__global__ __launch_bounds__(64, RCM_PAIR_LB) void k_step_repwvl(Dev d, int first_step, int nsteps) {
  constexpr int K = 32 / LPC;  // columns per wave
  constexpr int kLog2Lpc = LPC == 32 ? 5 : LPC == 16 ? 4 : LPC == 8 ? 3 : LPC == 4 ? 2 : LPC == 2 ? 1 : 0;
  extern __shared__ __align__(16) unsigned char s_dyn[];
  RepwvlPairLds *img = reinterpret_cast<RepwvlPairLds *>(s_dyn);
  StepLds *stp = reinterpret_cast<StepLds *>(s_dyn + K * sizeof(RepwvlPairLds));
  double(*st)[64] = reinterpret_cast<double(*)[64]>(s_dyn + K * (sizeof(RepwvlPairLds) + sizeof(StepLds)));
  // scratch of phases A, D and E: per column theta[20] (later dE), sorted[20], E_down[21] + E_up[21] (in phase A:
  // the residuals) = 102 doubles, aliased onto the staging columns, which are only busy in phase C
  // (K * 102 <= 11 * 64 holds for K <= 6; more columns per wave get their own block behind st)
  double *scr = (K * 102 <= (NH + 1) * 64) ? &st[0][0] : reinterpret_cast<double *>(st + (NH + 1));
  const int lane = threadIdx.x;
  const int c0 = blockIdx.x * K;
  const int p = PairRw::parity(lane);
  const int slot = PairRw::index(lane) / LPC;       // which of the wave's columns this lane pair works for
  const int j = PairRw::index(lane) - slot * LPC;   // its index among the column's lane pairs
  const bool col_ok = c0 + slot < d.ncol;
  // lanes of phases A / E: (column q, layer l), three columns per round
  const int aq = lane / NL, al = lane - aq * NL;

  // ---- load the owned columns' state
  for (int i = lane; i < K * NL; i += 64) {
    const int q = i / NL, l = i - q * NL, c = min(c0 + q, d.ncol - 1);
    stp[q].T[l] = d.T[(size_t)c * NL + l];
    stp[q].Ttau[l] = d.Ttau[(size_t)c * NL + l];
    stp[q].Tprev[l] = d.Tprev[(size_t)c * NL + l];
    stp[q].h2o[l] = d.gas[(size_t)c * RCM_NSPEC_REPWVL * NL + l];
    if (l == 0) {
      stp[q].Tsfc = d.Tsfc[c];
      stp[q].time_h = d.time_h[c];
    }
  }
  __syncthreads();

  for (int step = 0; step < nsteps; ++step) {
    const bool first = first_step && step == 0;
    RCM_STAMP(0);
    // ================= A: adjustment, residual, feedback (main.cpp:536-572) =================
    for (int base = 0; base < K; base += 3) {
      const int q = base + aq;
      const bool on = lane < 3 * NL && q < K && c0 + q < d.ncol;
      double *th = scr + (size_t)(on ? q : 0) * 102, *sorted = th + NL, *rr = th + 2 * NL;
      double T = 0.0, thv = 0.0, conv = 1.0;
      if (on) {
        T = stp[q].T[al];
        conv = d.conv[al];
        if (first) stp[q].Ttau[al] = T;  // main.cpp:500: the first optical depths come from the pre-adjustment profile
        thv = T * conv;                  // t_to_theta, main.cpp:123-129
        th[al] = thv;
        sorted[al] = __builtin_nan("");  // a NaN among the inputs leaves ranks unassigned: the column stays NaN
      }
      __syncthreads();
      if (on) {
        int rank = 0;
        for (int i = 0; i < NL; ++i) {
          const double ti = th[i];
          rank += (ti > thv) || (ti == thv && i < al);
        }
        sorted[rank] = thv;  // sort(theta, greater), main.cpp:538
      }
      __syncthreads();
      if (on) {
        T = sorted[al] / conv;  // theta_to_t, main.cpp:131-139
        rr[al] = fabs(T - stp[q].Tprev[al]);
        stp[q].Tprev[al] = T;
        stp[q].T[al] = T;
        const size_t at = (size_t)(c0 + q) * NL + al;
        d.Tprev[at] = T;
        if (!first) {
          stp[q].Ttau[al] = T;
          const double vmr = d.rel_hum[at] * magnus_dev(T) / d.player[al];  // main.cpp:285-286
          stp[q].h2o[al] = vmr;
          d.gas[(size_t)(c0 + q) * RCM_NSPEC_REPWVL * NL + al] = vmr;  // H2O is read_tau's first species
        }
        d.Ttau[at] = stp[q].Ttau[al];
      }
      __syncthreads();
      if (on && al == 0) {
        double r = rr[0];
        for (int i = 1; i < NL; ++i) r = (rr[i] > r || rr[i] != rr[i]) ? rr[i] : r;  // NaN-aware max
        d.resid[c0 + q] = first ? 1e300 : r;
      }
      __syncthreads();
    }
    RCM_STAMP(1);
    // ================= B: column images =================
    for (int q = 0; q < K; ++q)
      if (c0 + q < d.ncol)
        repwvl_pair_image_from(d, c0 + q, lane, img[q], stp[q].T, stp[q].Ttau, stp[q].Tsfc, stp[q].h2o);
    __syncthreads();
    RCM_STAMP(2);
    // ================= C: radiative transfer =================
    double acc1[NH + 1], acc2[NH];
#pragma unroll
    for (int i = 0; i <= NH; ++i) acc1[i] = 0.0;  // main.cpp:326-327
#pragma unroll
    for (int i = 0; i < NH; ++i) acc2[i] = 0.0;
    for (int w = j; w < d.n_wvl; w += LPC) {
      // both lanes of a pair share w and col_ok: the hand-over sees both active
      if (col_ok) repwvl_pair_point<CUBE>(d, img[slot], p, w, lane, st, acc1, acc2);
      RCM_STAMP(8 + (w - j) / LPC);
    }
    RCM_STAMP(3);
    // ================= D: sum over the column's lane pairs =================
    // even lane: acc1[i] = E_down[i] (i = 0..10), acc2[k] = E_up[k]; odd lane: acc1[i] = E_up[20-i], acc2[k] = E_down[20-k]
    __syncthreads();  // the staging columns are free again: scr may be written
#pragma unroll
    for (int i = 0; i <= NH; ++i) {
      const double v = PairRw::sum<kLog2Lpc>(acc1[i]);
      if (j == 0 && col_ok) (scr + (size_t)slot * 102 + 2 * NL)[p ? NLEV + NL - i : i] = v;
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const double v = PairRw::sum<kLog2Lpc>(acc2[k]);
      if (j == 0 && col_ok) (scr + (size_t)slot * 102 + 2 * NL)[p ? NL - k : NLEV + k] = v;
    }
    __syncthreads();
    RCM_STAMP(4);
    // ================= E: dE, time step, thermodynamics (main.cpp:337-341, 156-176, 581) =================
    for (int base = 0; base < K; base += 3) {
      const int q = base + aq;
      const bool on = lane < 3 * NL && q < K && c0 + q < d.ncol;
      const double *E = scr + (size_t)(on ? q : 0) * 102 + 2 * NL;  // E_down[21], E_up[21]
      double *sdE = scr + (size_t)(on ? q : 0) * 102;                // aliases theta (dead by now)
      const double *Ed = E, *Eu = E + NLEV;
      const int c = c0 + q;
      double dEl = 0.0;
      if (on) {
        dEl = Ed[al] - Ed[al + 1] + Eu[al + 1] - Eu[al];                // main.cpp:338
        if (al == NL - 1) dEl += d.solar_irr[c] + Ed[NL] - Eu[NL];      // main.cpp:341
        sdE[al] = dEl;
        d.E_down[(size_t)c * NLEV + al] = Ed[al];
        d.E_up[(size_t)c * NLEV + al] = Eu[al];
        if (al == 0) {
          d.E_down[(size_t)c * NLEV + NL] = Ed[NL];
          d.E_up[(size_t)c * NLEV + NL] = Eu[NL];
        }
      }
      __syncthreads();
      if (on) {
        double mx = sdE[0];
#pragma unroll
        for (int i = 1; i < NL; ++i) mx = (mx < sdE[i]) ? sdE[i] : mx;  // signed max_element, main.cpp:157
        const double heat_cap = 1004.0 * d.dp * 100.0;                  // c_air * dp * 100, main.cpp:157,169
        double dt = 5.0 / mx * heat_cap / 9.80665;                      // max_dT = 5 (float), main.cpp:82
        if (dt > 3600 * 12) dt = 3600 * 12;                             // main.cpp:158-160
        const double Tn = stp[q].T[al] + dEl * dt * 9.80665 / heat_cap; // main.cpp:169
        stp[q].T[al] = Tn;
        d.T[(size_t)c * NL + al] = Tn;
        d.dE[(size_t)c * NL + al] = dEl;
        if (al == NL - 1) {
          const double ts = Tn * d.conv[NL - 1];                        // main.cpp:173
          stp[q].Tsfc = ts;
          d.Tsfc[c] = ts;
        }
        if (al == 0) {
          d.dt[c] = dt;
          stp[q].time_h += (float)dt / 3600;                            // main.cpp:581
          d.time_h[c] = stp[q].time_h;
        }
      }
      __syncthreads();
    }
    RCM_STAMP(5);
  }
}


// ---------------------------------------------------------------------------------------------
// The elementary device functions of the production kernels, evaluated point by point for the accuracy pins of
// tests/test_gpu_parity.py: a different v_rcp_f64 precision or a regenerated coefficient table must
// fail THERE, loudly, instead of moving flux parity by a few 1e-11.  Same sources, same compile-time defaults
// (RCM_EXP_DEG, RCM_EXP_HALF, RCM_EXPPOS_HALF, RCM_RCP_NEWTON) as the product.
//   fn 0  exp_trans(a, b)                  a = tau (clamped by the caller's range), b = a1 = -s*log2(e)
//   fn 1  planck_mono_fast(a, b, c, 1/c)   a = xc, b = sc, c = T
//   fn 2  cplkavg_fast_mid(a, b, c)        a = xm, b = hh, c = pc
//   fn 3  exp_pos(a)
//   fn 4  rcp_newton(a)
//   fn 5  clamp_tau(a)
// ---------------------------------------------------------------------------------------------
__global__ void k_eval_device_fn(int fn, int n, const double *a, const double *b, const double *c, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  switch (fn) {
    case 0: r = exp_trans(a[i], b[i]); break;
    case 1: r = planck_mono_fast(a[i], b[i], c[i], 1.0 / c[i]); break;
    case 2: r = cplkavg_fast_mid(a[i], b[i], c[i]); break;
    case 3: r = exp_pos(a[i]); break;
    case 4: r = rcp_newton(a[i]); break;
    default: r = clamp_tau(a[i]); break;
  }
  out[i] = r;
}
