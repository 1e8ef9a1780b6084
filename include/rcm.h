/*
 * rcm.h -- C ABI of the MI355X radiative-convective column ensemble (librcm_hip.so).
 *
 * Drop-in boundary for ONE path of pabloconrat/our_first_climate_model: the body of
 * the time loop main.cpp:531-583 (convective adjustment, water-vapour feedback,
 * optical depths, thermal radiative transfer, time step) for an ensemble of
 * independent columns.  The reference has no plugin/FFI interface; its seam is a
 * set of C++ free functions in one translation unit.  Each entry point below names
 * the reference function(s) it replaces.  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary; no exceptions cross it; every function returns
 * RCM_OK (0) or a negative RCM_ERR_* code and leaves a message in rcm_last_error().
 *
 * Conventions: layer 0 = top of atmosphere; arrays are row-major, caller-owned,
 * host memory unless a parameter is documented as a DEVICE pointer; all numbers are
 * fp64 except `time_h` (float, main.cpp:359).  One handle = one device; calls on a
 * handle must be serialised by the caller; different handles are independent.
 *
 * The compute entry points need a gfx950 GPU.  There is no CPU fallback: without a
 * usable device rcm_create() fails with RCM_ERR_HIP.
 */
#ifndef RCM_H
#define RCM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RCM_ABI_VERSION 3 /* 3: + rcm_measure_hbm_bandwidth, rcm_set_species_merging (additions only: a caller of 2 runs unchanged) */

#define RCM_OK 0
#define RCM_ERR_ARG (-1)    /* bad argument / unsupported size            */
#define RCM_ERR_STATE (-2)  /* call order violated (tables/columns unset) */
#define RCM_ERR_HIP (-3)    /* HIP runtime error, no device               */
#define RCM_ERR_IO (-4)     /* file not found / unreadable                */
#define RCM_ERR_NOMEM (-5)
#define RCM_ERR_FORMAT (-6) /* malformed input file                       */

#define RCM_NSPEC_REPWVL 9 /* H2O,CO2,O3,N2O,CO,CH4,O2,HNO3,N2: read_tau argument order, repwvl_thermal.h:3-7 */
#define RCM_NSPEC_LBL 5    /* CO2,H2O,O3,CH4,N2O: lbl.arts/README:5                                          */

typedef struct rcm_ensemble rcm_ensemble;

int rcm_abi_version(void);
const char *rcm_last_error(void); /* thread-local, never NULL */

/* ------------------------------------------------------------------------------------------
 * Ensemble life cycle.
 * nlayer must be 20 (main.cpp:78-80 compiles 20 layers / 30 angles in; the kernels are specialised
 * for nlayer == 20 and take any 1 <= nangle <= 64).  `stream` is a hipStream_t passed as
 * void* (e.g. torch.cuda.current_stream().cuda_stream); NULL makes the library create its own.
 * ------------------------------------------------------------------------------------------ */
int rcm_create(int ncol, int nlayer, int nangle, int device, void *stream, rcm_ensemble **out);
int rcm_destroy(rcm_ensemble *h);

/* Replaces the grid set-up of main.cpp:355-356, 472-476, 481-483: plevel[nlayer+1] in hPa,
 * shared by all columns.  Derives player, theta conversion factors, dp = 1000/nlayer. */
int rcm_set_grid(rcm_ensemble *h, const double *plevel_hPa);

/* Scalars the reference compiles in: solar_irr (main.cpp:515, added to the lowest layer at
 * :341), the grey cloud of cloud_into_tau (main.cpp:266-274; cloud_layer 17, cloud_tau
 * tau_s/2 = 1.0).  cloud_layer = -1 disables the cloud. */
int rcm_set_forcing(rcm_ensemble *h, double solar_irr, int cloud_layer, double cloud_tau);

/* Per-column forcing: the same three quantities as arrays [ncol]; a NULL
 * pointer leaves that quantity as it is.  cloud_layer[c] = -1 disables the cloud of column c. */
int rcm_set_column_forcing(rcm_ensemble *h, const double *solar_irr, const int *cloud_layer,
                           const double *cloud_tau);

/* doubling_adding + solar_radiative_transfer_setup (main.cpp:214-264) on the device, one thread per
 * column, for per-column tau_s[ncol], mu_s[ncol] (cosine of the solar zenith angle), albedo[ncol];
 * sets each column's solar_irr and, as main.cpp:267 does, its cloud optical depth tau_s/2.
 * solar_out (may be NULL) receives [ncol][7] = {r_dir, s_dir, t_dir, r, t, r_total, solar_irr}. */
int rcm_solar_setup_columns(rcm_ensemble *h, const double *tau_s, const double *mu_s, const double *albedo,
                            int doublings, double g_asym, double daytime, double E_0, double *solar_out);

/* Replaces the table side of read_tau (repwvl_thermal.cpp:113-176): the datasets of a
 * Reduced*Forcing.nc file, xsec in FILE order [n_tpert][n_species][n_wvl][n_p].  The library
 * re-lays it out wavelength-fastest on the device.  n_species must be 9; the table must stay
 * below 4 GiB (the shipped ones: 0.3 - 2.7 MB). */
int rcm_set_repwvl_tables(rcm_ensemble *h, int n_tpert, int n_species, int n_wvl, int n_p,
                          const double *p_grid, const double *t_ref, const double *t_pert,
                          const double *wvl, const double *weight, const double *xsec);

/* lbl.arts spectra (lbl.arts/README:5-17; the reference holds the loader ascii.cpp:1631-1691
 * but no driver).  wvl[n_nodes] in nm, strictly increasing; tabs[s] = [n_nodes][nlayer] layer
 * optical depths exactly as ASCII_file2xy2D returns them, s in RCM_NSPEC_LBL order; h2o_ref
 * [nlayer] = the H2O mixing ratio the H2O table was computed at (absolute).  Spectral band b
 * (0 <= b < n_nodes-1) spans wvl[b]..wvl[b+1], takes node b's optical depth and the
 * band-integrated Planck source cplkavg(wvl[b], wvl[b+1], T) (cplkavg.cpp:124-243).  The five
 * tables together must stay below 4 GiB (n_nodes < 5.3 million). */
int rcm_set_lbl_tables(rcm_ensemble *h, int n_nodes, const double *wvl,
                       const double *const tabs[RCM_NSPEC_LBL], const double *h2o_ref);

/* Per-column initial state (main.cpp:357, 439-469, 473):
 *   Tlayer0   [ncol][nlayer]  K
 *   T_surface0[ncol]          K       (main.cpp:357: 288.2)
 *   gas       [ncol][ngas][nlayer]    repwvl: absolute mixing ratios, 9 species;
 *                                      lbl: multipliers of the 5 species tables (1 = as tabulated)
 *   rel_hum   [ncol][nlayer]          frozen relative humidity of main.cpp:467-468
 * Resets the step counter and `time` to 0.
 * repwvl: species whose profile is the same in EVERY column handed to this handle are summed into one table row set
 * (DESIGN.md 5).  Which species those are depends on the columns the handle gets: a rank's shard of an ensemble, or a
 * one-column handle, may merge a different set than the whole ensemble would, so the same column integrated under
 * another sharding agrees to rounding (optical depths <= 1e-12 relative), not bit for bit. */
int rcm_set_columns(rcm_ensemble *h, const double *Tlayer0, const double *T_surface0,
                    const double *gas, const double *rel_hum);

/* repwvl: on = 0 turns the merging of shared species off (every species keeps its own table rows: ~6 % slower for a CO2
 * sweep), so that a column's OPTICAL DEPTHS are bit-identical under any sharding of the ensemble; on = 1 (the default)
 * merges.  May be called before or after rcm_set_columns.  (Fluxes still depend on the cut in their last bits: the
 * flattened kernel's wave boundaries, i.e. the order of the spectral sum, move with it.) */
int rcm_set_species_merging(rcm_ensemble *h, int on);

/* Runs `nsteps` iterations of the loop body main.cpp:531-583 for every column, entirely on the
 * device, asynchronously on the handle's stream (call a getter or rcm_sync to wait):
 *   convective adjustment (:536-540) -> [i != 0: water_vapor_feedback (:562), optical depths
 *   (:564-568)] -> radiative_transfer (:574) -> calculate_timestep (:576) -> thermodynamics
 *   (:578) -> time += (float)dt/3600 (:581).
 * The first ever step uses optical depths of the PRE-adjustment profile (main.cpp:500). */
int rcm_step(rcm_ensemble *h, int nsteps);
int rcm_sync(rcm_ensemble *h);

/* The reference runs a fixed n_steps (main.cpp:83,531) and has no equilibrium criterion; this one is
 * the build's: a column's residual is max_l |change of the convectively ADJUSTED
 * profile| -- the Tlayer column output_conv prints -- between its last two iterations [K per step].
 * (The raw step always moves the warmest-heated layer by max_dT = 5 K, main.cpp:156-162, and the
 * model's top-of-atmosphere budget does not close, so neither can serve; the adjusted profile
 * converges monotonically: 0.16 K/step after 50 steps, 2.7e-3 after 2000 for test.atm/Reduced10.)
 * Steps `check_every` at a time until every column's residual is <= tol_K or max_steps have run.
 * steps_done / residual_K (worst column) may be NULL. */
int rcm_run_to_equilibrium(rcm_ensemble *h, double tol_K, int max_steps, int check_every, int *steps_done,
                           double *residual_K);
int rcm_get_residual(rcm_ensemble *h, double *resid /* [ncol] */);

/* Binary restart file (the reference has no resume): step counter, Tlayer,
 * T_surface, time, last timestep, gas block (H2O follows the feedback), rel_hum, forcing.  Load into a
 * handle of the same ncol whose grid and spectral tables are already set; stepping on continues
 * bit-identically. */
int rcm_checkpoint_save(rcm_ensemble *h, const char *path);
int rcm_checkpoint_load(rcm_ensemble *h, const char *path);

/* State after the last step (post-thermodynamics, pre-adjustment, as main.cpp:583 leaves it).
 * Any output pointer may be NULL.  Tlayer [ncol][nlayer], T_surface/time_h/timestep [ncol]. */
int rcm_get_state(rcm_ensemble *h, double *Tlayer, double *T_surface, float *time_h,
                  double *timestep);
/* Fluxes of the last radiative_transfer call: E_down/E_up [ncol][nlayer+1], dE [ncol][nlayer]. */
int rcm_get_fluxes(rcm_ensemble *h, double *E_down, double *E_up, double *dE);
/* What output_conv (main.cpp:102-114) would print for column `col` at the top of the next
 * iteration: adjusted Tlayer[nlayer], sorted theta[nlayer], time.  Does not modify the state. */
int rcm_get_profile(rcm_ensemble *h, int col, double *Tlayer_adj, double *theta, float *time_h);
/* Layer optical depths of column `col`, [nspectral][nlayer], cloud included -- what read_tau +
 * cloud_into_tau (or the lbl assembly) handed to the most recent radiative_transfer; before the first
 * step: what they will hand to it (the initial, not yet adjusted profile, main.cpp:500). */
int rcm_get_tau(rcm_ensemble *h, int col, double *tau);
/* The same optical depths AS THE PRODUCTION KERNELS FORM THEM (diagnostic; parity of rows a5 / a9): rcm_get_tau
 * evaluates read_tau's corner-DIFFERENCE form (repwvl_thermal.cpp:235-241); the thermal-RT kernels interpolate
 * with numDens-folded corner WEIGHTS and clamp to [-10, 1e7].  This getter follows the path the next
 * rcm_step launch would take: on the flattened repwvl kernel (n_wvl >= 8, no pinned chunk count) it copies the
 * column image STORED in device memory by the stepping kernels (rebuilt by the same store function when no step has
 * written it since the inputs last changed) and visits the species count of that launch; on the chunked repwvl
 * kernel and on lbl it builds the image in place, as those kernels do.  Returns what they feed to the sweep, same
 * layout and same timing rules as rcm_get_tau. */
int rcm_get_tau_production(rcm_ensemble *h, int col, double *tau);
/* Current gas block [ncol][ngas][nlayer] (H2O row follows the feedback). */
int rcm_get_gas(rcm_ensemble *h, double *gas);
int rcm_nspectral(rcm_ensemble *h); /* n_wvl (repwvl) or n_nodes-1 (lbl); <0 if unset */

/* Multi-GPU hand-off: writes [ncol][nlayer+1] = (Tlayer[0..nlayer-1], T_surface) to the DEVICE
 * buffer `dev_out` on the handle's stream.  The caller all-gathers it (RCCL; one collective,
 * SURVEY section 8e).  No host copy.
 * ORDERING: the pack kernel runs on the HANDLE's stream.  A handle created with stream = NULL owns a
 * private stream (torch's default stream has the handle value 0 = NULL, so passing
 * torch.cuda.current_stream().cuda_stream on the default stream also means "create your own"): call
 * rcm_sync(h) -- or record an event on that stream -- before a collective on any OTHER stream reads
 * `dev_out`, or the collective races the pack. */
int rcm_pack_state(rcm_ensemble *h, void *dev_out);

/* The one collective of the multi-GPU run, done by the library: packs this
 * handle's [ncol][nlayer+1] block, zero-pads it to ncol_pad rows (every rank passes the same ncol_pad >=
 * its own ncol, normally ceil(N_total/ranks)) and enqueues ONE ncclAllGather of ncol_pad*(nlayer+1)
 * doubles per rank on the handle's stream.  `nccl_comm` is the caller's ncclComm_t (RCCL), passed as void*;
 * `dev_recv` is a DEVICE buffer [ranks][ncol_pad][nlayer+1].  Asynchronous: rcm_sync(h) before reading it.
 * The library calls the ncclAllGather of the RCCL already loaded in the process (the one the communicator
 * was made with) and fails with RCM_ERR_STATE when there is none.  With several handles in ONE process
 * (one per GPU), bracket the calls with ncclGroupStart/ncclGroupEnd as for any RCCL collective. */
int rcm_gather(rcm_ensemble *h, void *nccl_comm, void *dev_recv, int ncol_pad);

/* Launch geometry override for the thermal-RT kernel (0 = automatic): number of spectral
 * chunks each column is split into. */
int rcm_set_spectral_chunks(rcm_ensemble *h, int nchunk);
int rcm_get_spectral_chunks(rcm_ensemble *h);

/* Kernel timing with HIP events on the handle's stream, around every thermal-RT launch (the fused repwvl
 * kernel counts as ONE launch however many iterations it covers).
 * rcm_get_rt_timing synchronises, returns the accumulated milliseconds and launch count since
 * the last reset, and resets. */
int rcm_set_rt_timing(rcm_ensemble *h, int enable);
int rcm_get_rt_timing(rcm_ensemble *h, double *ms_total, int *launches);

/* Measurement utility, no handle needed (SURVEY.md section 8d asks for the HBM peak MEASURED on the box, next to
 * the vendor's 8 TB/s): streams two device buffers of `bytes` each (>= 64 MiB; use >= 1 GiB so that the 256 MiB
 * Infinity Cache cannot serve them) `reps` times with 16-byte accesses per lane and returns the best pass:
 *   copy_gbs  dst = src, bytes read + bytes written per second   [GB/s]  (the figure a "float4 copy" quotes)
 *   read_gbs  read-only pass (a sum), bytes read per second       [GB/s]  (what streaming absorption tables is)
 * Either output may be NULL. */
int rcm_measure_hbm_bandwidth(int device, size_t bytes, int reps, double *copy_gbs, double *read_gbs);

/* ------------------------------------------------------------------------------------------
 * Host-side pieces of the reference driver, kept so that rcm_main is a drop-in for main().
 * ------------------------------------------------------------------------------------------ */

/* main.cpp:396-430: 4 header lines, then whitespace-separated columns z p T air H2O O3 [CO2 CH4
 * N2O] (ppm).  cols = [9][nlevel_max]; 6-column files (lbl.arts/fpda.lbl.atm) get CO2/CH4/N2O =
 * 400/1.7/0.315 ppm (lbl.arts/README:14-17).  *nlevel_out = rows read. */
int rcm_atm_load(const char *path, int nlevel_max, double *cols, int *nlevel_out,
                 int *ncols_found);

/* main.cpp:439-485 for one column: layer means, ppm -> absolute, CO2 factor (main.cpp:452),
 * relative humidity from LEVEL p/T and LAYER vmr (main.cpp:467-468), layer T.  rh_scale
 * multiplies the initial H2O (1 = reference).  vmr9 = [9][nlayer] in read_tau order. */
int rcm_column_init(int nlevel, const double *atm_cols, double co2_factor, double rh_scale,
                    double *plevel, double *player, double *Tlayer, double *conv, double *vmr9,
                    double *rel_hum);

/* doubling_adding + solar_radiative_transfer_setup, main.cpp:214-264.
 * out = {r_dir, s_dir, t_dir, r, t, r_total, solar_irr}. */
int rcm_solar_setup(double tau_s, double mu_s, int doublings, double g_asym, double albedo,
                    double daytime, double E_0, double out[7]);
/* The values main.cpp:86-91 compiles in. */
int rcm_default_solar(double out[7]);

/* ASCII_file2xy2D (lbl.arts/ascii.cpp:1631-1691, ascii.h:34-38,63): whitespace matrix with '%'
 * or '#' comments; x = first column [nx], y = remaining columns, returned FLAT [nx][ny]
 * (malloc'd; free with rcm_free).  Lines are read with the reference's 1 MiB fgets buffer (ascii.cpp:155,255:
 * a longer physical line continues as the next row).  Returns the reference's codes: 0, -1 file not found,
 * -2 no memory, -5 not a rectangular matrix. */
int rcm_ascii_file2xy2D(const char *path, int *nx, int *ny, double **x, double **y);

/* One-time binary cache of the five lbl.xxx.asc tables, stored
 * [species][layer][node] behind the wavelength nodes.  read returns the layout rcm_set_lbl_tables
 * takes: wvl[n_nodes] and ONE block tabs[5][n_nodes][nlayer] (both malloc'd; rcm_free). */
int rcm_lbl_cache_write(const char *path, int n_nodes, int nlayer, const double *wvl,
                        const double *const tabs[RCM_NSPEC_LBL]);
int rcm_lbl_cache_read(const char *path, int *n_nodes, int *nlayer, double **wvl, double **tabs);

/* The datasets read_tau takes from a Reduced*Forcing.nc file (repwvl_thermal.cpp:113-176), either
 * from the file itself -- NetCDF-4 = HDF5; read without either library, for the layout netcdf-c
 * writes (contiguous, uncompressed little-endian f64 variables; anything else: RCM_ERR_FORMAT with
 * the feature named) -- or from the .rtab blob tools/flatten_repwvl.py makes of it (h5dump).
 * dims = {n_tpert, n_species, n_wvl, n_p}; *blob (malloc'd; rcm_free) holds p_grid, t_ref,
 * t_pert, vmrs_ref, wvl, weight, xsec back to back. */
int rcm_rtab_load(const char *path, long long dims[4], double **blob);
void rcm_free(void *p);

/* output.txt writers with the reference's exact formats (main.cpp:513-519, 528, 109-110).
 * `f` is a FILE* passed as void*. */
int rcm_write_banner(void *f, double tau_s, double mu_s, const double solar[7]);
int rcm_write_profile(void *f, int nlayer, const double *player, const double *Tlayer,
                      const double *theta, float time_h);

#ifdef __cplusplus
}
#endif
#endif /* RCM_H */
