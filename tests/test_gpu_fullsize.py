"""GPU: the BASELINE.json configurations at their REAL ensemble sizes, the production optical-depth arithmetic, the
N = 2 path with HIP compute, rcm_gather, the lbl equilibrium run.

The 10 000-column launch geometries (27 spectral chunks / 67 500 blocks for lbl, 31 250 flattened repwvl waves,
k_post over 10 000 columns, the big-ensemble branch of choose_chunks) are compared with the CPU oracle on sampled
columns -- the oracle runs those columns alone, on identical per-column inputs, which is the same computation
because columns are independent (no horizontal coupling anywhere in main.cpp) -- and through size-independent
identities on EVERY column."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

FLUX_TIGHT = 2e-9  # W/m^2 (contract: 1e-6)
TEMP_TIGHT = 1e-9  # K     (contract: 1e-6)


@pytest.fixture(scope="module")
def pkg(rcm):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return rcm


def _columns(inp, idx):
    """The input description restricted to the columns `idx` (what the oracle runs)."""
    idx = np.asarray(idx)
    out = dict(inp)
    out["ncol"] = len(idx)
    out["cols"] = {k: (v[idx] if getattr(v, "ndim", 0) >= 1 and v.shape[0] == inp["ncol"] else v)
                   for k, v in inp["cols"].items()}
    return out


def _compare_sampled(pkg, olib, inp, idx, steps, library=None, fusion=-1):
    g = H.gpu_ensemble(pkg, inp, library=library)
    if fusion != -1:
        H.set_step_fusion(g, fusion)
    o = H.oracle_ensemble(olib, _columns(inp, idx))
    g.step(steps), o.step(steps)
    fg, fo, sg, so = g.get_fluxes(), o.get_fluxes(), g.get_state(), o.get_state()
    for k in ("E_down", "E_up", "dE"):
        assert H.max_abs(fg[k][idx], fo[k]) <= FLUX_TIGHT, (k, H.max_abs(fg[k][idx], fo[k]))
    assert H.max_abs(sg["Tlayer"][idx], so["Tlayer"]) <= TEMP_TIGHT
    assert H.max_abs(sg["T_surface"][idx], so["T_surface"]) <= TEMP_TIGHT
    np.testing.assert_allclose(sg["timestep"][idx], so["timestep"], rtol=1e-10)
    np.testing.assert_allclose(sg["time_h"][idx], so["time_h"], rtol=2e-7)  # float, main.cpp:359,581
    # every column: finite, and the identities of main.cpp:337-341, 156-162, 173 between its own outputs
    Ed, Eu, dE = fg["E_down"], fg["E_up"], fg["dE"]
    assert np.isfinite(Ed).all() and np.isfinite(Eu).all() and np.isfinite(sg["Tlayer"]).all()
    want = Ed[:, :-1] - Ed[:, 1:] + Eu[:, 1:] - Eu[:, :-1]
    want[:, -1] += inp["solar"] + Ed[:, -1] - Eu[:, -1]
    np.testing.assert_allclose(dE, want, rtol=0, atol=1e-10)
    dt = np.minimum(5.0 / dE.max(axis=1) * (1004 * 50.0 * 100.0) / 9.80665, 43200.0)
    np.testing.assert_allclose(sg["timestep"], dt, rtol=1e-14)
    np.testing.assert_allclose(sg["T_surface"], sg["Tlayer"][:, -1] * inp["cols"]["conv"][-1], rtol=1e-15)
    assert np.all(Ed[:, 0] == 0.0) and np.all(Ed[:, 1:] > 0) and np.all(Eu > 0)
    g.close()
    return fg, sg


def test_config2_10000_columns_reduced100_sampled_against_oracle(pkg, olib, variants):
    """BASELINE configs[2] at size: 10 000 columns (CO2 x RH x T_surface perturbed) x Reduced100, 2 steps: the
    production path (k_pre, 31 250 waves of k_rt_repwvl_flat in 7 813 blocks, k_post over 10 000 columns) -- what
    `bench.py --workload repwvl100` runs -- and the test-side fused whole-step kernel at the same size."""
    inp = H.make_inputs(pkg, "repwvl", 10000, table="Reduced100")
    idx = [0, 1, 2, 3, 3333, 4999, 5000, 7777, 9996, 9997, 9998, 9999]  # incl. the first and the last block
    g = H.gpu_ensemble(pkg, inp)
    assert H.get_step_fusion(g) == 0 and g.get_spectral_chunks() == 5  # flattened kernel, 5 record slots per column
    g.close()
    _compare_sampled(pkg, olib, inp, idx, steps=2)
    _compare_sampled(pkg, olib, inp, idx, steps=2, library=variants, fusion=1)


def test_lbl_2000_columns_big_ensemble_chunk_branch_sampled_against_oracle(pkg, olib):
    """2 000 columns x 4 096 bands: ncol*rows > 98k, the `fill < amortised` branch of choose_chunks (16 chunks of
    8 passes), which the 10 000-column bench launch also takes and no smaller test reaches."""
    inp = H.make_inputs(pkg, "lbl", 2000, n_nodes=4097, atm="fpda.lbl.atm")
    g = H.gpu_ensemble(pkg, inp)
    assert g.get_spectral_chunks() == 16
    g.close()
    _compare_sampled(pkg, olib, inp, [0, 1, 999, 1000, 1998, 1999], steps=2)


def test_config3_10000_columns_full_lbl_spectrum_sampled_against_oracle(pkg, olib):
    """BASELINE configs[3] at size: 10 000 columns x 131 071 bands (the bench workload, same generator), 2 steps;
    columns 0 / 4 999 / 9 999 against a 3-column oracle run (0.6 s per column-step on a core)."""
    inp = H.make_inputs(pkg, "lbl", 10000, n_nodes=131072, atm="fpda.lbl.atm")
    g = H.gpu_ensemble(pkg, inp)
    assert g.get_spectral_chunks() == 27
    g.close()
    _compare_sampled(pkg, olib, inp, [0, 4999, 9999], steps=2)


def test_config4_one_gpus_share_of_the_100000_column_run_sampled_against_oracle(pkg, olib):
    """BASELINE configs[4] at the size ONE of its 8 GPUs sees: rank 3's block of the 100 000-column ensemble (columns
    37 500 .. 49 999 of the CO2 x RH x T_surface sweep, cut exactly as bench.py / equilibrium_ensemble.py cut it:
    distributed.shard_bounds + perturbation_sweep(first, total)) x 131 071 bands, 2 steps; six of its columns against
    the oracle, the identities of main.cpp:337-341 / 156-162 / 173 on all 12 500.  (All eight blocks to equilibrium,
    one after another on one GPU: profiles/r02_equilibrium_100000_shards.json.)"""
    from our_first_climate_model_amd.distributed import shard_bounds
    lo, hi = shard_bounds(100000, 3, 8)
    assert (lo, hi) == (37500, 50000)
    atm_cols, _ = pkg.atm_load(os.path.join(H.GOLDEN, "fpda.lbl.atm"))
    spec = pkg.perturbation_sweep(hi - lo, perturb_rh=True, perturb_ts=True, first=lo, total=100000)
    cols = pkg.build_columns(atm_cols, spec, "lbl")
    wvl, tabs = pkg.synth_lbl_tables(131072, plevel=cols["plevel"])
    inp = dict(mode="lbl", ncol=hi - lo, cols=cols, solar=pkg.default_solar()["solar_irr"], wvl=wvl, tabs=tabs)
    full = pkg.perturbation_sweep(100000, perturb_rh=True, perturb_ts=True)          # the window IS that part of the sweep
    assert np.array_equal(spec.co2_factor, full.co2_factor[lo:hi]) and np.array_equal(spec.T_surface0, full.T_surface0[lo:hi])
    _compare_sampled(pkg, olib, inp, [0, 1, 6249, 6250, 12498, 12499], steps=2)


# ---- the fused whole-step kernel of the repwvl path (k_step_repwvl; TEST-SIDE build: measured slower than the
# three-kernel path, kept as a second complete implementation of the step) ------------------------------------
@pytest.mark.parametrize("table", ["Reduced10", "Reduced20", "Reduced100"])
def test_fused_step_kernel_against_oracle_ragged_sizes(pkg, olib, variants, table):
    """k_step_repwvl forced on for small ensembles: every lane-pairs-per-column choice the cost model makes for
    these sizes, waves whose last columns do not exist (ncol not a multiple of the columns per wave), several
    iterations inside ONE launch (incl. the first one with its pre-adjustment optical depths, main.cpp:500)."""
    seen = set()
    for ncol in (1, 2, 3, 5, 7, 33, 130):
        inp = H.make_inputs(pkg, "repwvl", ncol, table=table)
        inp["cols"]["Tlayer0"][ncol // 2, 15] += 25.0  # one unstable column: the sort reorders its layers
        g, o = H.gpu_ensemble(pkg, inp, library=variants), H.oracle_ensemble(olib, inp)
        H.set_step_fusion(g, 1)
        seen.add(H.get_step_fusion(g))
        g.step(1), o.step(1)
        for k, v in o.get_fluxes().items():
            assert H.max_abs(g.get_fluxes()[k], v) <= FLUX_TIGHT, (ncol, k)
        assert np.all(g.get_residual() == 1e300)
        g.step(4), o.step(4)  # four iterations, one launch
        fg, fo, sg, so = g.get_fluxes(), o.get_fluxes(), g.get_state(), o.get_state()
        for k in fo:
            assert H.max_abs(fg[k], fo[k]) <= FLUX_TIGHT, (ncol, k)
        assert H.max_abs(sg["Tlayer"], so["Tlayer"]) <= TEMP_TIGHT and H.max_abs(sg["T_surface"], so["T_surface"]) <= TEMP_TIGHT
        np.testing.assert_allclose(sg["timestep"], so["timestep"], rtol=1e-10)
        np.testing.assert_allclose(sg["time_h"], so["time_h"], rtol=2e-7)
        np.testing.assert_allclose(g.get_residual(), o.get_residual(), rtol=0, atol=1e-9)
        np.testing.assert_allclose(g.get_gas(), o.get_gas(), rtol=1e-12, atol=0)  # the feedback's H2O row
        np.testing.assert_allclose(g.get_tau_production(ncol - 1), o.get_tau(ncol - 1), rtol=1e-12, atol=0)
        T, th, tm = g.get_profile(ncol // 2)
        To, tho, tmo = o.get_profile(ncol // 2)
        assert H.max_abs(T, To) <= TEMP_TIGHT and H.max_abs(th, tho) <= 1e-8
        g.close()
    assert len(seen) >= 1 and all(2 <= v <= 32 for v in seen), seen


def test_fused_and_three_kernel_paths_agree_and_interleave(pkg, olib, variants, tmp_path):
    """The same ensemble stepped by k_step_repwvl and by k_pre + k_rt_repwvl_flat + k_post: equal to the order of the
    spectral sum; a run may switch between them at any iteration (state lives in the same HBM arrays), also across a
    checkpoint; 600 iterations cross the 512-iterations-per-launch cut."""
    inp = H.make_inputs(pkg, "repwvl", 9, table="Reduced20")
    a, c = (H.gpu_ensemble(pkg, inp, library=variants) for _ in range(2))
    b = H.gpu_ensemble(pkg, inp)  # the product: three kernels per iteration, nothing else to select
    H.set_step_fusion(a, 1)
    assert H.get_step_fusion(a) > 0 and H.get_step_fusion(b) == 0 and H.get_step_fusion(c) == 0  # automatic = off
    with pytest.raises(pkg.RcmError):
        H.set_step_fusion(b, 1)
    a.step(6), b.step(6)
    H.set_step_fusion(c, 1), c.step(2), H.set_step_fusion(c, 0), c.step(3), H.set_step_fusion(c, 1), c.step(1)
    for x in (b, c):
        for k, v in a.get_fluxes().items():
            assert H.max_abs(x.get_fluxes()[k], v) <= FLUX_TIGHT, k
        assert H.max_abs(x.get_state()["Tlayer"], a.get_state()["Tlayer"]) <= TEMP_TIGHT
        np.testing.assert_array_equal(x.get_state()["time_h"], a.get_state()["time_h"])
    ck = str(tmp_path / "f.ck")
    a.checkpoint_save(ck)
    d = H.gpu_ensemble(pkg, inp, library=variants)
    d.checkpoint_load(ck)
    H.set_step_fusion(d, 1)
    a.step(600), d.step(600)
    np.testing.assert_array_equal(a.get_state()["Tlayer"], d.get_state()["Tlayer"])  # resume is bit-identical
    o = H.oracle_ensemble(olib, inp)
    o.step(606)
    assert H.max_abs(a.get_state()["Tlayer"], o.get_state()["Tlayer"]) <= 1e-6
    assert H.max_abs(a.get_state()["T_surface"], o.get_state()["T_surface"]) <= 1e-6


def test_fused_equilibrium_and_nan_column(pkg, olib, variants):
    inp = H.make_inputs(pkg, "repwvl", 6, table="Reduced10")
    g, o = H.gpu_ensemble(pkg, inp, library=variants), H.oracle_ensemble(olib, inp)
    H.set_step_fusion(g, 1)
    n_g, r_g = g.run_to_equilibrium(0.12, 400, 25)
    n_o, r_o = o.run_to_equilibrium(0.12, 400, 25)
    assert n_g == n_o and 25 <= n_g < 400 and abs(r_g - r_o) <= 1e-9
    inp["cols"]["Tlayer0"][4, 3] = np.nan
    g = H.gpu_ensemble(pkg, inp, library=variants)
    H.set_step_fusion(g, 1)
    g.step(3)
    s = g.get_state()
    assert np.isnan(s["Tlayer"][4]).all() and np.isnan(s["T_surface"][4]) and np.isnan(g.get_residual()[4])
    assert np.isfinite(np.delete(s["Tlayer"], 4, axis=0)).all()


# ---- a5: the optical depths the PRODUCTION kernels form ------------------------------------------
@pytest.mark.parametrize("table", ["Reduced10", "Reduced20", "Reduced100"])
def test_production_tau_interpolation_against_read_tau(pkg, olib, table):
    """rcm_get_tau_production runs the device functions of the production kernels (repwvl_pair_image: T bracket,
    numDens-folded corner weights; repwvl_pair_tau: weighted-corner sum, cloud, clamp), not the side kernel k_tau
    with the reference's corner-difference form.  Against orc_read_tau (repwvl_thermal.cpp:197-248), <= 1e-12
    relative, for columns inside the table and for columns whose temperatures leave the table's t_ref + t_pert
    range on either side, so that LowerPos clamps to the first / last bracket and read_tau EXTRAPOLATES
    (repwvl_thermal.cpp:19-45: delT < 0 resp. > 1)."""
    inp = H.make_inputs(pkg, "repwvl", 6, table=table)
    T0 = inp["cols"]["Tlayer0"]
    T0[2] += 140.0   # every layer above t_ref + 120: last bracket, delT up to 1.4
    T0[3] -= 135.0   # below t_ref - 120 in most layers: first bracket, delT < 0
    T0[4, ::2] += 150.0  # alternating: neighbouring layers in different regimes (and an unstable profile)
    T0[5] = np.linspace(95.0, 420.0, 20)
    g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)
    t = inp["table"]
    extrap = 0
    for c in range(6):
        # the production clamps tau to [-10, 1e7] (extrapolated cross sections go negative in places; the
        # reference then forms exp(+|tau|/mu) = inf and NaN fluxes, the kernel inf or finite: garbage either way)
        tp, to = g.get_tau_production(c), np.clip(o.get_tau(c), -10.0, 1e7)
        np.testing.assert_allclose(tp, to, rtol=1e-12, atol=0)
        np.testing.assert_allclose(tp, np.clip(g.get_tau(c), -10.0, 1e7), rtol=1e-12, atol=0)
        for l in range(20):  # count the extrapolating (column, layer) brackets the comparison covered
            midp = 0.5 * (inp["cols"]["plevel"][l] + inp["cols"]["plevel"][l + 1]) * 100.0
            ip = olib.orc_lower_pos(t["p_grid"].ctypes.data_as(C.POINTER(C.c_double)), t["n_p"], midp)
            grid = t["t_ref"][ip] + t["t_pert"]
            extrap += int(T0[c, l] < grid[0] or T0[c, l] > grid[-1])
    assert extrap >= 40
    # after steps the feedback has rewritten H2O and the brackets follow the new temperatures
    g.step(3), o.step(3)
    for c in (0, 1):  # (the extrapolating columns have negative optical depths somewhere: NaN after a step)
        np.testing.assert_allclose(g.get_tau_production(c), o.get_tau(c), rtol=1e-12, atol=0)


def test_production_tau_comes_from_the_stored_column_images(pkg, olib, tmp_path):
    """The flattened repwvl kernel COPIES the image repwvl_image_store wrote to d.img (k_pre / post_column; active
    species only, the rest zero since the memset) and runs repwvl_pair_tau<NACT> with NACT < 9; the chunked kernel
    builds its image in place (repwvl_pair_image_from, all 9 species).  rcm_get_tau_production follows whichever
    the launch would take: here both routes, on a live handle whose species pattern changes,
    after a cloud change between steps, and after rcm_checkpoint_load -- each <= 1e-12 from read_tau, and the two
    routes BIT-identical to each other (the species a smaller NACT skips contribute fma(0, x, a) = a)."""
    rng = np.random.default_rng(3)
    inp = H.make_inputs(pkg, "repwvl", 9, table="Reduced20")
    g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)

    def both_routes(tag):
        for c in (0, 4, 8):
            stored = g.get_tau_production(c)          # flat: k_image (if needed) + copy of d.img + NACT of the launch
            g.set_spectral_chunks(2)                  # a pinned chunk count selects k_rt_repwvl_pair: image built in place
            in_place = g.get_tau_production(c)
            g.set_spectral_chunks(0)
            assert np.array_equal(stored, in_place), (tag, c)
            np.testing.assert_allclose(stored, o.get_tau(c), rtol=1e-12, atol=0, err_msg=tag)

    both_routes("before the first step")
    g.step(3), o.step(3)
    both_routes("after 3 steps")                      # d.img as post_column / k_pre left it
    # a cloud change between steps: the getter shows the new cloud at once (as the oracle's does)
    tau_c = np.linspace(0.2, 1.8, 9)
    lay = np.array([17, 16, 15, -1, 3, 17, 0, 19, 10], dtype=np.int32)
    g.set_column_forcing(cloud_layer=lay, cloud_tau=tau_c), o.set_column_forcing(cloud_layer=lay, cloud_tau=tau_c)
    both_routes("after a forcing change")
    g.step(2), o.step(2)
    both_routes("2 steps later")
    # checkpoint -> a fresh handle: classify_species clears the images, the getter rebuilds from Ttau and the gas block
    path = str(tmp_path / "img.ckpt")
    g.checkpoint_save(path)
    want = [g.get_tau_production(c) for c in range(9)]
    b = pkg.Ensemble(9)
    b.set_grid(inp["cols"]["plevel"])
    b.set_repwvl_tables(inp["table"])
    b.checkpoint_load(path)
    for c in range(9):
        assert np.array_equal(b.get_tau_production(c), want[c]), c
    b.step(2), g.step(2), o.step(2)
    for c in range(9):
        assert np.array_equal(b.get_tau_production(c), g.get_tau_production(c)), c
    both_routes("after the resumed steps")
    # another species pattern on the live handle: CH4 now varies, O3 doubled; N2O alone stays shared -> no merged slot
    gas = inp["cols"]["gas"]
    gas[:, 5] *= rng.uniform(0.5, 2.0, (9, 1))
    gas[:, 2] *= 2.0
    gas[:, 3] *= rng.uniform(0.5, 2.0, (9, 1))
    c_ = inp["cols"]
    g.set_columns(c_["Tlayer0"], c_["T_surface0"], c_["gas"], c_["rel_hum"])
    o = H.oracle_ensemble(olib, inp)
    o.set_column_forcing(cloud_layer=lay, cloud_tau=tau_c)   # (the handle keeps its forcing across rcm_set_columns)
    both_routes("new species pattern")
    g.step(2), o.step(2)
    both_routes("new species pattern, stepped")
    fg, fo = g.get_fluxes(), o.get_fluxes()
    for k in fg:
        assert H.max_abs(fg[k], fo[k]) <= FLUX_TIGHT, k


def test_production_tau_of_a_table_with_fewer_than_8_wavelengths(pkg, olib):
    """n_wvl < 8 keeps the chunked kernel (17 column images per block would not pay): the getter takes the
    in-place image route there, against read_tau to 1e-12, and the fluxes agree."""
    inp = H.make_inputs(pkg, "repwvl", 5, table="Reduced10")
    t = dict(inp["table"])
    t["n_wvl"] = 5
    t["wvl"], t["weight"] = t["wvl"][:5].copy(), t["weight"][:5].copy()
    t["xsec"] = np.ascontiguousarray(t["xsec"][:, :, :5, :])
    inp["table"] = t
    g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)
    g.step(2), o.step(2)
    for c in range(5):
        np.testing.assert_allclose(g.get_tau_production(c), o.get_tau(c), rtol=1e-12, atol=0)
    fg, fo = g.get_fluxes(), o.get_fluxes()
    for k in fg:
        assert H.max_abs(fg[k], fo[k]) <= FLUX_TIGHT, k


def test_repwvl_table_whose_derived_layers_would_overflow_32_bit_offsets_is_refused(pkg):
    """the kernels address xp[20][n_tpert-1][9][n_wvl][2] through 32-bit byte offsets; a table that passes
    the raw 4 GiB check but whose pressure-interpolated form does not (few pressures, many wavelengths) is refused."""
    n_t, n_p, n_w = 9, 3, 190000          # raw: 9*9*3*190000*8 = 0.37 GB; derived: 20*8*9*190000*16 = 4.4 GB
    t = dict(n_tpert=n_t, n_species=9, n_wvl=n_w, n_p=n_p, p_grid=np.array([1.1e5, 5e4, 1e2]),
             t_ref=np.array([288.0, 250.0, 220.0]), t_pert=np.linspace(-120, 120, n_t),
             wvl=np.linspace(4000.0, 1e5, n_w), weight=np.ones(n_w), xsec=np.zeros((n_t, 9, n_w, n_p)))
    e = pkg.Ensemble(1)
    with pytest.raises(pkg.RcmError) as ei:
        e.set_repwvl_tables(t)
    assert ei.value.code == -1 and "pressure-interpolated table" in str(ei.value)


@pytest.mark.parametrize("table", ["Reduced10", "Reduced100"])
def test_gpu_tau_reproduces_the_tables_own_nodes(pkg, olib, table):
    """a5 on the GPU against reference-held DATA (the oracle-free twin of
    tests/test_reference_pins.py::test_read_tau_reproduces_the_tables_own_nodes): columns whose layer temperatures sit
    exactly on temperature nodes of the table, one absorbing species with mixing ratio 1 -- both optical-depth getters
    (the reference-form kernel and the production image/prefix route with its merged and padded species slots) must
    return the 1-D pressure interpolation of the table's own entries: species order, wavelength row, perturbation
    index, pressure bracket and direction, layer order, number density.  One column per species in ONE ensemble, so
    that every species varies between columns (nothing merged) and the launch visits all nine slots; a second
    ensemble of identical columns makes every species shared (everything merged into one slot)."""
    import test_reference_pins as P
    rng = np.random.default_rng(37)
    inp = H.make_inputs(pkg, "repwvl", 9, table=table, perturb=False)
    t, c = inp["table"], inp["cols"]
    ci = dict(plevel=c["plevel"])
    want, scale = [], []
    for sp in range(9):
        T, vmr, w, sc = P._node_case(t, ci, rng, sp, olib, 1e-6)   # (tau stays below the production clamp of 1e7)
        c["Tlayer0"][sp], c["gas"][sp] = T, vmr
        want.append(w), scale.append(sc)
    inp["cols"]["rel_hum"][:] = 0.0
    g = H.gpu_ensemble(pkg, inp)
    g.set_forcing(inp["solar"], cloud_layer=-1, cloud_tau=0.0)
    for sp in range(9):
        for tau in (g.get_tau(sp), g.get_tau_production(sp)):
            assert np.all(np.abs(tau - want[sp]) <= 1e-12 * scale[sp] + 1e-300), (table, sp)
    for sp in (0, 1, 6):                      # identical columns: all species shared -> one merged slot (+ H2O's)
        one = H.make_inputs(pkg, "repwvl", 3, table=table, perturb=False)
        T, vmr, w, sc = P._node_case(t, ci, rng, sp, olib, 1e-6)
        one["cols"]["Tlayer0"][:], one["cols"]["gas"][:] = T, vmr
        g1 = H.gpu_ensemble(pkg, one)
        g1.set_forcing(one["solar"], cloud_layer=-1, cloud_tau=0.0)
        for col in (0, 2):
            for tau in (g1.get_tau(col), g1.get_tau_production(col)):
                assert np.all(np.abs(tau - w) <= 1e-12 * sc + 1e-300), (table, sp, col)


def test_production_tau_outside_the_pressure_grid_measured_bound(pkg, olib):
    """Layers whose mid pressure lies ABOVE the table's largest pressure (110 000 Pa): read_tau's LowerPos walks
    the descending p_grid without a sign change and returns the LAST bracket (the 1e-3 Pa end), so delP is about
    -1.8e8 and the 'interpolation' is a difference of nearly equal table rows times 1e8 -- the reference's own
    result is conditioned 1e7 there (tau up to 7e6, some negative).  The two algebraically equal forms (corner
    differences vs corner weights) then agree only to the conditioning: measured 6.4e-9 relative (Reduced100),
    3.8e-10 (Reduced10).  Stated, not hidden: in-table layers of the same column still meet 1e-12."""
    inp = H.make_inputs(pkg, "repwvl", 2, table="Reduced100", perturb=False)
    inp["cols"]["plevel"] = inp["cols"]["plevel"] * 1.3  # surface at 1300 hPa: layers 17..19 leave the grid
    g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)
    tp, to = g.get_tau_production(0), o.get_tau(0)
    inside = [l for l in range(20) if 0.5 * (inp["cols"]["plevel"][l] + inp["cols"]["plevel"][l + 1]) * 100 < 110000.0]
    outside = [l for l in range(20) if l not in inside]
    assert len(outside) >= 2
    np.testing.assert_allclose(tp[:, inside], to[:, inside], rtol=1e-12, atol=0)
    rel = np.abs(tp[:, outside] - np.clip(to[:, outside], -10.0, 1e7)) / np.abs(to[:, outside])
    assert rel.max() <= 1e-7, rel.max()
    tr = g.get_tau(0)  # the reference-form getter (k_tau): the same conditioning applies to FMA contraction
    np.testing.assert_allclose(tr[:, inside], to[:, inside], rtol=1e-12, atol=0)
    assert (np.abs(tr[:, outside] - to[:, outside]) / np.abs(to[:, outside])).max() <= 1e-7


def test_production_tau_lbl_and_clamp(pkg, olib):
    """lbl: the production composition (lbl_pair_tau) equals the oracle's; an absurd optical depth is clamped to
    1e7 (exp(-1e7) == 0 for every angle, as exp(-1e9) is) and a NaN one stays NaN."""
    inp = H.make_inputs(pkg, "lbl", 3, n_nodes=500, atm="fpda.lbl.atm")
    inp["tabs"][0][7, 4] = 3e9
    inp["tabs"][2][11, 13] = np.nan
    g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)
    tp, to = g.get_tau_production(1), o.get_tau(1)
    assert tp[7, 4] == 1e7 and to[7, 4] > 1e9 and np.isnan(tp[11, 13]) and np.isnan(to[11, 13])
    m = np.ones_like(tp, dtype=bool)
    m[7, 4] = m[11, 13] = False
    np.testing.assert_allclose(tp[m], to[m], rtol=1e-13, atol=0)


def test_nan_mixing_ratio_stays_nan_and_isolated(pkg, olib):
    """A NaN mixing ratio (or relative humidity) in one column: NaN optical depth -> NaN fluxes and temperatures
    for that column, as in the reference arithmetic (the oracle agrees) -- not a clamped tau = -10 with finite
    garbage; neighbours untouched; the equilibrium loop never counts the column as converged."""
    for mode, kw in (("repwvl", dict(table="Reduced20")), ("lbl", dict(n_nodes=200, atm="fpda.lbl.atm"))):
        inp = H.make_inputs(pkg, mode, 4, **kw)
        inp["cols"]["gas"][2, 1, 7] = np.nan
        g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)
        g.step(2), o.step(2)
        sg, so, fg = g.get_state(), o.get_state(), g.get_fluxes()
        assert np.isnan(so["Tlayer"][2]).all()
        assert np.isnan(sg["Tlayer"][2]).all() and np.isnan(sg["T_surface"][2]) and np.isnan(fg["E_up"][2, 0])
        for c in (0, 1, 3):
            assert H.max_abs(sg["Tlayer"][c], so["Tlayer"][c]) <= TEMP_TIGHT
        n, r = g.run_to_equilibrium(1e9, 6, 3)  # every finite residual passes 1e9 K: only the NaN holds the run
        assert n == 6 and r != r
        assert np.isnan(g.get_residual()[2]) and np.isfinite(g.get_residual()[[0, 1, 3]]).all()


# ---- configs[4] in small: the equilibrium criterion on the lbl path --------------------------------
def test_lbl_run_to_equilibrium_matches_oracle(pkg, olib):
    """rcm_run_to_equilibrium on the lbl path (calibrated synthetic spectra): GPU and oracle stop at the same
    iteration with the same residual, and the criterion -- not the step cap -- ends the run."""
    inp = H.make_inputs(pkg, "lbl", 4, n_nodes=600, atm="fpda.lbl.atm")
    g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)
    n_g, r_g = g.run_to_equilibrium(0.05, 1000, 20)
    n_o, r_o = o.run_to_equilibrium(0.05, 1000, 20)
    assert n_g == n_o and 100 <= n_g < 1000, (n_g, n_o)
    assert abs(r_g - r_o) <= 1e-9 and r_g <= 0.05
    sg, so = g.get_state(), o.get_state()
    assert H.max_abs(sg["Tlayer"], so["Tlayer"]) <= 1e-6 and H.max_abs(sg["T_surface"], so["T_surface"]) <= 1e-6
    assert np.all((sg["T_surface"] > 280.0) & (sg["T_surface"] < 320.0))  # a physical steady state


# ---- multi-GPU path with HIP compute ---------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _bench(args, nproc=None, timeout=900):
    cmd = [sys.executable]
    if nproc:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(H.ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_with_hip_compute_equal_one_rank(pkg, tmp_path):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank; both ranks on
    this box's only GPU, collective over gloo): shard -> HIP steps -> pack -> all-gather.  The gathered [64][21]
    state equals the one-rank run to 1e-10 K (another cut = another fixed order of the spectral sum)."""
    common = ["--ncol", "64", "--nodes", "4097", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    d1 = _bench(common + ["--dump-state", one])
    d2 = _bench(common + ["--gpus", "2", "--backend", "gloo", "--dump-state", two], nproc=2)
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d2["config"]["columns_per_gpu"] == 32
    assert d2["scaling"] == "strong" and d2["config"]["columns_total"] == 64
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (64, 21) and np.isfinite(a).all()
    assert H.max_abs(a, b) <= 1e-10
    assert np.ptp(a[:, 20]) > 0.1  # the perturbed columns differ: a rank-order mix-up would show


def test_a_rank_without_columns_creates_no_handle_and_joins_the_collectives(pkg, tmp_path):
    """3 columns on 4 ranks (HIP compute, gloo): rank 3's block is empty.  rcm_create refuses zero columns
    (test_gpu_parity), so that rank makes no handle (distributed.EmptyShard) but still joins the warm-up gather, the
    barriers, the final all-gather and the all-reduce of the time: rc 0, one JSON line, the gathered [3][21] state
    equal to the one-rank run.  The same for the equilibrium driver's residual all-reduce.  (9 columns on 8 ranks
    -- ranks 5..7 empty -- runs on CPU over gloo in tests/test_multirank_cpu.py; the box allows 6 GPU processes.)"""
    common = ["--ncol", "3", "--nodes", "2049", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one, four = str(tmp_path / "one.npy"), str(tmp_path / "four.npy")
    d1 = _bench(common + ["--dump-state", one])
    d4 = _bench(common + ["--gpus", "4", "--backend", "gloo", "--dump-state", four], nproc=4)
    assert d4["n_gpus"] == 4 and d4["config"]["columns_per_gpu"] == 1 and d4["config"]["columns_total"] == 3
    a, b = np.load(one), np.load(four)
    assert a.shape == b.shape == (3, 21) and np.isfinite(a).all() and H.max_abs(a, b) <= 1e-10
    eq = ["--ncol", "3", "--nodes", "1025", "--tol", "0.2", "--check-every", "10", "--max-steps", "600"]
    e1 = _equilibrium_example(eq + ["--save-state", one])
    e4 = _equilibrium_example(eq + ["--backend", "gloo", "--save-state", four], nproc=4)
    assert e1["converged"] and e4["converged"] and e4["ranks"] == 4 and e1["steps"] == e4["steps"]
    assert H.max_abs(np.load(one), np.load(four)) <= 1e-10


def test_bench_gpus_2_without_a_launcher_starts_its_own(pkg, tmp_path):
    """`python bench.py --gpus 2 ...` with NO torch.distributed.run in front (the form the driver used for N = 1 in
    round 2): bench.py starts the launcher itself as a child process before touching the GPU, the child's one JSON
    line comes through on stdout, rc 0, and the gathered state equals the one-rank run."""
    common = ["--ncol", "64", "--nodes", "4097", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    d1 = _bench(common + ["--dump-state", one])
    d2 = _bench(common + ["--gpus", "2", "--backend", "gloo", "--dump-state", two])   # nproc=None: no launcher
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d2["config"]["columns_per_gpu"] == 32
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (64, 21) and np.isfinite(a).all()
    assert H.max_abs(a, b) <= 1e-10
    assert np.ptp(a[:, 20]) > 0.1


def test_nccl_leg_with_the_one_rank_this_box_allows(pkg, tmp_path):
    """The code the 8-GPU SCALE run executes -- init_process_group("nccl", device_id=...), the packed state
    all-gathered ON DEVICE by RCCL, all_reduce(max) of the time -- with the one rank a one-GPU box can give RCCL:
    rc 0, one JSON line, and the gathered [64][21] state BIT-IDENTICAL to the run without a launcher (same cut,
    same chunking, so the same order of every sum)."""
    common = ["--ncol", "64", "--nodes", "4097", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    plain, nccl = str(tmp_path / "plain.npy"), str(tmp_path / "nccl.npy")
    d0 = _bench(common + ["--dump-state", plain])
    d1 = _bench(common + ["--gpus", "1", "--backend", "nccl", "--dump-state", nccl], nproc=1)
    assert d0["n_gpus"] == d1["n_gpus"] == 1 and d1["config"]["columns_per_gpu"] == 64
    assert d1["roofline"]["launches"] == 2 and d1["value"] > 0
    a, b = np.load(plain), np.load(nccl)
    assert a.shape == (64, 21) and np.array_equal(a, b)


def _equilibrium_example(args, nproc=None, timeout=900):
    cmd = [sys.executable]
    if nproc:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(H.ROOT, "examples", "equilibrium_ensemble.py"), "--progress", "0"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_equilibrium_stop_rule_across_two_ranks_with_hip_compute(pkg, tmp_path):
    """BASELINE configs[4]'s multi-rank path with HIP compute: examples/equilibrium_ensemble.py
    as two torch.distributed.run ranks on this box's one GPU (gloo: collectives on CPU tensors), each integrating its
    8 of 16 columns on the lbl kernels, distributed.run_to_equilibrium's all-reduce(max) of the residual keeping them
    on one iteration count.  Against the one-rank run of the same 16 columns: the SAME number of steps, the residual
    equal to 1e-9, the gathered state to 1e-10 K (another cut = another fixed order of the spectral sums), converged.
    Then a NaN column on rank 1 only: neither rank may report convergence (the NaN travels as a flag: max() is not
    NaN-propagating in every backend) and both run to max_steps."""
    common = ["--ncol", "16", "--nodes", "1025", "--tol", "0.05", "--check-every", "10", "--max-steps", "1500"]
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    d1 = _equilibrium_example(common + ["--save-state", one])
    d2 = _equilibrium_example(common + ["--backend", "gloo", "--save-state", two], nproc=2)
    assert d1["converged"] and d2["converged"] and d2["ranks"] == 2 and d2["backend"] == "gloo"
    assert d1["steps"] == d2["steps"] and 100 <= d1["steps"] < 1500 and d1["steps"] % 10 == 0
    assert abs(d1["residual_K_per_step"] - d2["residual_K_per_step"]) <= 1e-9 and d2["residual_K_per_step"] <= 0.05
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (16, 21) and np.isfinite(a).all() and H.max_abs(a, b) <= 1e-10
    assert np.all((a[:, 20] > 280.0) & (a[:, 20] < 320.0)) and np.ptp(a[:, 20]) > 0.1
    # column 12 lives on rank 1; rank 0's own columns all converge, the run must not
    d3 = _equilibrium_example(["--ncol", "16", "--nodes", "1025", "--tol", "1e9", "--check-every", "5", "--max-steps",
                               "20", "--backend", "gloo", "--nan-column", "12", "--save-state", two], nproc=2)
    assert not d3["converged"] and d3["steps"] == 20 and d3["residual_K_per_step"] != d3["residual_K_per_step"]
    c = np.load(two)
    assert np.isnan(c[12]).all() and np.isfinite(np.delete(c, 12, axis=0)).all()
    d4 = _equilibrium_example(["--ncol", "16", "--nodes", "1025", "--tol", "1e9", "--check-every", "5", "--max-steps",
                               "20", "--backend", "gloo"], nproc=2)
    assert d4["converged"] and d4["steps"] == 5       # the same run without the NaN stops at the first check


def test_repwvl_bench_workload_contract(pkg):
    d = _bench(["--workload", "repwvl100", "--ncol", "500", "--steps", "5", "--warmup", "1"])
    assert d["config"]["wavelengths"] == 100 and d["roofline"]["kernel"] == "k_rt_repwvl_flat"
    assert d["roofline"]["launches"] == 5
    assert d["roofline"]["algorithmic_bytes_per_launch"] == (9 * 9 * 41 * 8 + 16) * 100 * 500
    # the nominal numerator never crosses HBM for this workload (the 2.66 MB table lives in L2): a fraction above 1 is
    # not a roofline, so the machine-read line says null and why, and carries the two fractions that are measured
    r = d["roofline"]
    assert r["frac"] is None and "frac is null on purpose" in r["note"] and r["achieved"] > 0
    assert "hbm_traffic_frac" in r and "actual_frac" in r and r["actual_bound"].startswith("fp64_valu_issue+")
    if r["fp64_valu_issue"] is not None:   # counters of THIS kernel build are committed
        assert r["actual_frac"] == r["fp64_valu_issue"]["issue_frac_at_2.4GHz"] and 0.0 < r["actual_frac"] <= 1.0
    else:
        assert r["actual_frac"] is None
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["single_core_rt_only_value"] > 0
    assert d["cpu_baseline"]["single_core_reread_table_per_step_value"] < d["cpu_baseline"]["single_core_value"]


def test_rcm_gather_one_rank_over_rccl(pkg):
    """rcm_gather: pack + zero padding + ONE ncclAllGather on the handle's stream, with a
    communicator the CALLER made -- here a one-rank communicator from the RCCL PyTorch loaded.  The received
    block is the packed state followed by the padding rows."""
    import torch
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    rccl = C.CDLL(os.path.join(libdir, "librccl.so"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        inp = H.make_inputs(pkg, "repwvl", 5, table="Reduced10")
        g = H.gpu_ensemble(pkg, inp)
        g.step(2)
        recv = torch.full((8, 21), -1.0, dtype=torch.float64, device="cuda")
        g.gather(comm.value, recv.data_ptr(), 8)
        g.sync()
        s = g.get_state()
        got = recv.cpu().numpy()
        np.testing.assert_array_equal(got[:5, :20], s["Tlayer"])
        np.testing.assert_array_equal(got[:5, 20], s["T_surface"])
        assert np.all(got[5:] == 0.0)
        with pytest.raises(pkg.RcmError) as e:
            g.gather(comm.value, recv.data_ptr(), 4)  # ncol_pad < ncol
        assert e.value.code == -1
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_two_handles_on_two_streams_equal_one_handle(pkg):
    """Columns never interact, so an ensemble may be cut into handles that run concurrently on their own HIP streams
    (what --devices does across GPUs, here on one): bit-identical to the same columns in one handle when the cut
    keeps the spectral chunking (repwvl: the flattened list restarts at the cut, so the wave boundaries inside the
    second part move: last bits only)."""
    import torch
    ncol, steps = 600, 7
    for mode, kw in (("repwvl", dict(table="Reduced100")), ("lbl", dict(n_nodes=2049, atm="fpda.lbl.atm"))):
        inp = H.make_inputs(pkg, mode, ncol, **kw)
        one = H.gpu_ensemble(pkg, inp)
        one.step(steps)
        want = one.get_state()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        cut = 257
        parts = [pkg.configure(pkg.Ensemble(len(idx), stream=s.cuda_stream), _columns(inp, idx))
                 for s, idx in zip(streams, (np.arange(0, cut), np.arange(cut, ncol)))]
        if mode == "lbl":
            for p in parts:
                p.set_spectral_chunks(one.get_spectral_chunks())
        for p in parts:  # both enqueue before either is waited for
            p.step(steps)
        got = [p.get_state() for p in parts]
        for k in ("Tlayer", "T_surface", "timestep"):
            both = np.concatenate([g[k] for g in got])
            if mode == "lbl":
                assert np.array_equal(both, want[k]), (mode, k)
            else:
                assert H.max_abs(both, want[k]) <= 1e-10 * max(1.0, float(np.max(np.abs(want[k])))), (mode, k)
        for p in parts:
            p.close()
        one.close()


def test_repwvl_result_under_another_sharding_agrees_to_the_documented_bound(pkg, olib):
    """include/rcm.h (rcm_set_columns): which repwvl species are summed into one table row set is decided from the columns
    a HANDLE holds, so a rank's shard may merge another set than the whole ensemble would.  12 columns in
    which N2O varies only among the last four and CH4 only among the first three: the whole ensemble merges {O3}; shard
    A (columns 0..5) merges {O3, N2O}, shard B (6..11) {O3, CH4} -- a species stops being shared in each.  The same
    column under either sharding: optical depths <= 1e-12 relative (both also against read_tau), and after 6 steps
    fluxes <= 2e-9 W/m2, temperatures <= 1e-9 K -- agreement to rounding, not bit for bit, as documented."""
    rng = np.random.default_rng(11)
    inp = H.make_inputs(pkg, "repwvl", 12, table="Reduced20")
    gas = inp["cols"]["gas"]            # [12][9][20]: H2O 0, CO2 1, O3 2, N2O 3, CO 4, CH4 5
    gas[8:, 3] *= rng.uniform(0.5, 2.0, (4, 1))
    gas[:3, 5] *= rng.uniform(0.5, 2.0, (3, 1))
    one = H.gpu_ensemble(pkg, inp)
    parts = [H.gpu_ensemble(pkg, _columns(inp, np.arange(0, 6))), H.gpu_ensemble(pkg, _columns(inp, np.arange(6, 12)))]
    o = H.oracle_ensemble(olib, inp)
    differs = 0
    for c in range(12):
        a, b = one.get_tau_production(c), parts[c // 6].get_tau_production(c % 6)
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=0)
        np.testing.assert_allclose(b, o.get_tau(c), rtol=1e-12, atol=0)
        differs += int(not np.array_equal(a, b))
    assert differs > 0     # the shards really summed other species sets (else this test checks nothing)
    one.step(6), parts[0].step(6), parts[1].step(6)
    s1, f1 = one.get_state(), one.get_fluxes()
    sp = [p.get_state() for p in parts]
    fp = [p.get_fluxes() for p in parts]
    for k in ("Tlayer", "T_surface"):
        assert H.max_abs(np.concatenate([x[k] for x in sp]), s1[k]) <= TEMP_TIGHT, k
    for k in f1:
        assert H.max_abs(np.concatenate([x[k] for x in fp]), f1[k]) <= FLUX_TIGHT, k
    # rcm_set_species_merging(0) on the live handles: every species keeps its own rows -> a column's optical depths
    # no longer know the sharding (the states differ in their last bits by now: compare handles that hold the same state)
    for e in (one, parts[0], parts[1]):
        e.set_species_merging(False)
    one.step(2), parts[0].step(2), parts[1].step(2)
    s1 = one.get_state()
    sp = [p.get_state() for p in parts]
    assert H.max_abs(np.concatenate([x["Tlayer"] for x in sp]), s1["Tlayer"]) <= TEMP_TIGHT
    # ... and before rcm_set_columns, from the initial state: bit-identical under either sharding, 1e-12 from read_tau
    def unmerged(sub):
        e = pkg.Ensemble(sub["ncol"])
        e.set_species_merging(False)
        return pkg.configure(e, sub)
    one0 = unmerged(inp)
    parts0 = [unmerged(_columns(inp, np.arange(0, 6))), unmerged(_columns(inp, np.arange(6, 12)))]
    for c in range(12):
        a, b = one0.get_tau_production(c), parts0[c // 6].get_tau_production(c % 6)
        assert np.array_equal(a, b), c
        np.testing.assert_allclose(a, o.get_tau(c), rtol=1e-12, atol=0)
    live = H.gpu_ensemble(pkg, _columns(inp, np.arange(0, 6)))   # merged {O3, N2O} ...
    live.set_species_merging(False)                              # ... re-classified on the handle: as if never merged
    for c in range(6):
        assert np.array_equal(live.get_tau_production(c), parts0[0].get_tau_production(c)), c


def test_handles_give_their_device_memory_back(pkg):
    """create -> tables -> columns -> step -> destroy, twenty times per path: the free device memory ends where it began
    (rcm_destroy frees every buffer of the handle, the partial records and the column images included)."""
    import torch
    inputs = [H.make_inputs(pkg, "repwvl", 300, table="Reduced100"),
              H.make_inputs(pkg, "lbl", 300, n_nodes=4097, atm="fpda.lbl.atm")]
    for inp in inputs:  # once unmeasured: the runtime's own first-use allocations (kernel code objects, streams)
        g = H.gpu_ensemble(pkg, inp)
        g.step(1), g.sync(), g.close()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(20):
        for inp in inputs:
            g = H.gpu_ensemble(pkg, inp)
            g.step(2)
            g.get_fluxes()
            g.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 <= 8 << 20, f"{(free0 - free1) / 2**20:.1f} MiB of device memory did not come back"


@pytest.mark.parametrize("case", ["all_shared", "some_vary", "only_one_shared", "all_vary", "new_columns"])
def test_species_with_one_profile_for_all_columns_are_merged_correctly(pkg, olib, case):
    """The production optical-depth stage sums the species whose mixing-ratio profile is the same in every column into
    one table row set (Dev::merged_slot, derive_repwvl_layers); which species those are depends on the ensemble.  Whatever
    the pattern, optical depths (<= 1e-12 relative) and fluxes equal the oracle's, which knows nothing of merging."""
    rng = np.random.default_rng(7)
    ncol = 1 if case == "all_shared" else 12
    inp = H.make_inputs(pkg, "repwvl", ncol, table="Reduced20")
    gas = inp["cols"]["gas"]  # [ncol][9][20]; H2O 0, CO2 1, O3 2, N2O 3, CO 4, CH4 5, O2 6, HNO3 7, N2 8
    if case == "some_vary":        # N2O and CO differ from column to column: merged = {O3, CH4}
        gas[:, 3] *= rng.uniform(0.5, 2.0, (ncol, 1))
        gas[:, 4] = 1e-7 * rng.uniform(0.5, 2.0, (ncol, 20))
    elif case == "only_one_shared":  # everything but CH4 varies: nothing to merge
        for s in (2, 3):
            gas[:, s] *= rng.uniform(0.5, 2.0, (ncol, 1))
    elif case == "all_vary":
        for s in (2, 3, 5):
            gas[:, s] *= rng.uniform(0.5, 2.0, (ncol, 20))
    g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)
    if case == "new_columns":      # the same handle gets other columns: the merged rows follow (O3 doubled, N2O varies)
        g.step(2)
        gas[:, 2] *= 2.0
        gas[:, 3] *= rng.uniform(0.5, 2.0, (ncol, 1))
        c = inp["cols"]
        g.set_columns(c["Tlayer0"], c["T_surface0"], c["gas"], c["rel_hum"])
        o = H.oracle_ensemble(olib, inp)
    for c in range(ncol):
        np.testing.assert_allclose(g.get_tau_production(c), o.get_tau(c), rtol=1e-12, atol=0)
    g.step(3), o.step(3)
    fg, fo = g.get_fluxes(), o.get_fluxes()
    for k in fg:
        assert H.max_abs(fg[k], fo[k]) <= FLUX_TIGHT, (case, k)
    assert H.max_abs(g.get_state()["Tlayer"], o.get_state()["Tlayer"]) <= TEMP_TIGHT
    for c in (0, ncol - 1):
        np.testing.assert_allclose(g.get_tau_production(c), o.get_tau(c), rtol=1e-12, atol=0)
