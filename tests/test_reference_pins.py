"""CPU: the oracle against the reference's OWN CODE, compiled here.

oracle/_ref/libref_main.so is /root/reference/main.cpp's netCDF-free body (:12-21 standard includes, :25-344
Consts + every free function above main()), piped unedited from where it lies into g++ together with
oracle/ref_main_shim.inc (C-linkage call-throughs).  No stand-in header, no edited line; main.cpp:22 (<netcdf>) is
used by main() alone.  These tests assert  oracle/rcm_oracle.c == that compiled code BIT FOR BIT  on

  a1  cplkavg (monochromatic Planck x weight)      main.cpp:186-204
  a2  emissivity                                   main.cpp:207-212
  a3  monochromatic_radiative_transfer             main.cpp:291-318
  a4  radiative_transfer                           main.cpp:320-344
  a5' cloud_into_tau                               main.cpp:266-274
  a6  calculate_timestep                           main.cpp:156-162
  a7  thermodynamics                               main.cpp:164-176
  a8  t_to_theta / sort / theta_to_t, magnus, water_vapor_feedback   main.cpp:123-139, 277-289, 536-540
  solar set-up                                     main.cpp:214-264
  a5  LowerPos (repwvl_thermal.cpp:19-45, libref_lowerpos.so) and read_tau itself: libref_readtau.so = every
      statement of repwvl_thermal.cpp:55-262 that names no netCDF type (:55-111, 124-141, 153-168, 178-206, 210-262:
      unit conversion, reversals, numDens, brackets, the corner-difference interpolation :235-241, accumulation,
      store order, transposition), compiled unedited; oracle/ref_readtau_shim.inc hands it the VALUES the skipped
      getDim / getVar lines (:113-123, 143-149, 171-176, 208) would have delivered, from the same file's datasets.

On the GPU box /root/reference does not exist; the prebuilt oracle/_ref/*.so travel with gpurun.  Without them
(a fresh clone with no reference tree) the tests skip and say so.
"""
import math
import os

import numpy as np
import pytest

import oracle_lib as ol

_p = ol._p


@pytest.fixture(scope="module")
def ref():
    L = ol.load_ref_main()
    if L is None:
        pytest.skip("oracle/_ref/libref_main.so not built (reference tree absent at build time)")
    return L


@pytest.fixture(scope="module")
def rtau():
    L = ol.load_ref_readtau()
    if L is None:
        pytest.skip("oracle/_ref/libref_readtau.so not built (reference tree absent at build time)")
    return L


def _ref_tau(rtau, ref, t, ci, Tlayer, vmr9, cloud=True):
    """The reference's read_tau (+ its cloud_into_tau): what main.cpp:500-504 / :564-568 produce."""
    tau, wvl, weight = ol.ref_read_tau(rtau, t, ci["plevel"], Tlayer, vmr9)
    assert wvl.tolist() == t["wvl"].tolist() and weight.tolist() == t["weight"].tolist()
    if cloud:
        ref.ref_cloud_into_tau(_p(tau), t["n_wvl"])
    return tau


def _column(olib, golden, co2=1.0):
    atm, _ = ol.atm_load(olib, os.path.join(golden, "test.atm"))
    return atm, ol.column_init(olib, atm, co2, 1.0)


def _orc_tau(olib, t, ci, Tlayer, vmr9, cloud=True):
    tau = np.zeros((t["n_wvl"], 20))
    xs = np.ascontiguousarray(t["xsec"])
    olib.orc_read_tau(t["n_tpert"], t["n_species"], t["n_wvl"], t["n_p"], _p(t["p_grid"]), _p(t["t_ref"]),
                      _p(t["t_pert"]), _p(xs), 21, _p(ci["plevel"]), _p(np.ascontiguousarray(Tlayer)),
                      _p(np.ascontiguousarray(vmr9)), _p(tau))
    if cloud:
        olib.orc_cloud_into_tau(_p(tau), t["n_wvl"], 20, 17, 1.0)
    return tau


def _orc_rt(olib, tau, wvl, weight, T, Ts, solar):
    Ed, Eu, dE = np.zeros(21), np.zeros(21), np.zeros(20)
    tau = np.ascontiguousarray(tau)
    olib.orc_radiative_transfer(tau.shape[0], 20, 30, _p(tau), _p(np.ascontiguousarray(wvl)),
                                _p(np.ascontiguousarray(weight)), 0, _p(np.ascontiguousarray(T)), float(Ts),
                                float(solar), _p(Ed), _p(Eu), _p(dE))
    return Ed, Eu, dE


def test_compiled_in_constants_are_the_ones_the_build_uses(ref, olib, rcm):
    """Consts (main.cpp:66-91) as compiled, against the defaults of the oracle and of the product's host side."""
    c = np.zeros(20)
    ref.ref_consts(_p(c))
    assert c[0] == 2.0 / 7.0 and c[1] == 1004 and c[2] == 9.80665 and c[3] == 1361.0
    assert (c[4], c[5], c[6], c[7]) == (6.62607e-34, 299792458.0, 1.380649e-23, 273.15)
    assert tuple(c[8:12]) == (20, 21, 30, 5.0)
    assert c[12] == 2.0 and c[13] == np.cos(60 * np.pi / 180.0) and c[14] == 20 and c[15] == 0.85
    assert c[16] == 0.12 and c[17] == 0.5 and c[18] == 17 and c[19] == 10
    sol_ref, sol_orc = np.zeros(7), np.zeros(7)
    ref.ref_solar(_p(sol_ref))
    olib.orc_default_solar(_p(sol_orc))
    assert sol_ref.tolist() == sol_orc.tolist()                      # doubling_adding + setup, bit for bit
    assert list(rcm.default_solar().values()) == sol_ref.tolist()    # ... and the product's host function
    sol2 = np.zeros(7)
    olib.orc_solar_setup(c[12], c[13], int(c[14]), c[15], c[16], c[17], c[3], _p(sol2))
    assert sol2.tolist() == sol_ref.tolist()


def test_planck_bit_identical(ref, olib):
    """a1: both cplkavg overloads of main.cpp:186-204 over the repwvl wavelength range and beyond, weights of both
    signs (Reduced100's ChosenWeights reach -46 250), temperatures 100-400 K and the underflow/overflow ends."""
    rng = np.random.default_rng(11)
    for _ in range(3000):
        wvl = float(np.exp(rng.uniform(np.log(300.0), np.log(5e5))))
        w = float(rng.choice([1.0, rng.normal() * 10 ** rng.uniform(-3, 6)]))
        T = float(rng.uniform(100.0, 400.0)) if rng.random() < 0.9 else float(10 ** rng.uniform(-1, 4))
        assert olib.orc_planck_mono(wvl, w, T) == ref.ref_cplkavg_scalar(wvl, w, T), (wvl, w, T)
    for _ in range(200):
        wvl = float(np.exp(rng.uniform(np.log(3000.0), np.log(2e5))))
        w = float(rng.normal() * 1e4)
        T = rng.uniform(150.0, 330.0, 20)
        B = np.zeros(20)
        ref.ref_cplkavg_layers(wvl, w, _p(T), _p(B))
        assert B.tolist() == [olib.orc_planck_mono(wvl, w, float(t)) for t in T]


def test_emissivity_and_single_wavelength_sweep_bit_identical(ref, olib):
    """a2 + a3: one wavelength through the reference's monochromatic_radiative_transfer against the oracle's
    radiative_transfer with nspec = 1 (whose E_down/E_up start from zero like the reference's fill)."""
    rng = np.random.default_rng(5)
    mu, dmu = ol.ref_mu()
    for case in range(300):
        tau = 10 ** rng.uniform(-8, 4.5, 20)
        if case % 7 == 0:
            tau[rng.integers(0, 20)] = 0.0
        wvl = np.array([float(np.exp(rng.uniform(np.log(3000.0), np.log(2e5))))])
        wt = float(rng.normal() * 10 ** rng.uniform(0, 5))
        T = rng.uniform(180.0, 320.0, 20)
        Ts = float(rng.uniform(250.0, 330.0))
        # emissivity alone: 1 - exp(-tau/mu) as the reference's libm call forms it
        t30, a30 = np.concatenate([tau, np.zeros(10)]), np.zeros(30)
        ref.ref_emissivity(_p(t30), float(mu[case % 30]), _p(a30))
        assert a30[:20].tolist() == [1.0 - math.exp(-x / mu[case % 30]) for x in tau]   # the same glibc exp
        assert a30[20:].tolist() == [0.0] * 10
        B = np.zeros(20)
        ref.ref_cplkavg_layers(float(wvl[0]), wt, _p(T), _p(B))
        Ed, Eu = np.zeros(21), np.zeros(21)
        ref.ref_monochromatic_radiative_transfer(_p(B), 0, _p(tau), wt, 1, _p(wvl), _p(mu), dmu, _p(T), Ts,
                                                 _p(Ed), _p(Eu))
        Ed_o, Eu_o, _ = _orc_rt(olib, tau[None], wvl, np.array([wt]), T, Ts, 0.0)
        assert Ed.tolist() == Ed_o.tolist() and Eu.tolist() == Eu_o.tolist(), case


def test_radiative_transfer_bit_identical_on_random_cases(ref, olib):
    """a4 (ii): 200 random (tau[nwvl][20], T, weight) cases: tau in [1e-8, 3e4], negative weights, 1-120
    wavelengths, any solar term."""
    rng = np.random.default_rng(20260104)
    for case in range(200):
        nw = int(rng.integers(1, 121))
        tau = 10 ** rng.uniform(-8, np.log10(3e4), (nw, 20))
        wvl = np.sort(np.exp(rng.uniform(np.log(3000.0), np.log(2e5), nw)))
        wt = rng.normal(size=nw) * 10 ** rng.uniform(0, 5.6, nw)
        wt[rng.integers(0, nw)] = -46250.7
        T = rng.uniform(180.0, 320.0, 20)
        Ts, solar = float(rng.uniform(250.0, 330.0)), float(rng.uniform(0.0, 400.0))
        got = _orc_rt(olib, tau, wvl, wt, T, Ts, solar)
        want = ol.ref_radiative_transfer(ref, tau, wvl, wt, T, Ts, solar)
        for g, w, name in zip(got, want, ("E_down", "E_up", "dE")):
            assert g.tolist() == w.tolist(), (case, name)


@pytest.mark.parametrize("table", ["Reduced10", "Reduced20", "Reduced100"])
@pytest.mark.parametrize("co2", [1.0, 2.0])
def test_radiative_transfer_bit_identical_on_the_reference_inputs(ref, olib, golden, table, co2):
    """a4 (i): test.atm, the reference's tables, step-0 state (tau from the oracle's read_tau: a5 is the one row
    the compiled reference cannot supply)."""
    _, ci = _column(olib, golden, co2)
    t = ol.read_rtab(os.path.join(golden, f"{table}Forcing.rtab"))
    tau = _orc_tau(olib, t, ci, ci["Tlayer"], ci["vmr9"])
    sol = np.zeros(7)
    ref.ref_solar(_p(sol))
    T, theta = ci["Tlayer"].copy(), np.zeros(20)      # tau from the pre-adjustment profile (main.cpp:500), then
    ref.ref_convective_adjustment(_p(T), _p(theta), _p(ci["conv"]))   # :536-540: T*f/f moves last bits even when
    got = _orc_rt(olib, tau, t["wvl"], t["weight"], T, 288.2, sol[6])  # the order of the layers does not change
    want = ol.ref_radiative_transfer(ref, tau, t["wvl"], t["weight"], T, 288.2, sol[6])
    for g, w in zip(got, want):
        assert g.tolist() == w.tolist()
    # and through the oracle's ensemble handle, the thing every GPU parity test compares with
    e = ol.OracleEnsemble(olib, 1)
    e.set_grid(ci["plevel"])
    e.set_forcing(sol[6])
    e.set_repwvl_tables(t)
    e.set_columns(ci["Tlayer"][None], np.array([288.2]), ci["vmr9"][None], ci["rel_hum"][None])
    e.step(1)
    f = e.get_fluxes()
    assert f["E_down"][0].tolist() == want[0].tolist() and f["E_up"][0].tolist() == want[1].tolist()
    assert f["dE"][0].tolist() == want[2].tolist()
    assert e.get_state()["timestep"][0] == ref.ref_calculate_timestep(50.0, _p(want[2]))


def test_cloud_into_tau_bit_identical(ref, olib):
    rng = np.random.default_rng(2)
    tau = 10 ** rng.uniform(-8, 4, (37, 20))
    a, b = tau.copy(), tau.copy()
    ref.ref_cloud_into_tau(_p(a), 37)
    olib.orc_cloud_into_tau(_p(b), 37, 20, 17, 2.0 / 2.0)
    assert a.tolist() == b.tolist() and (a[:, 17] == tau[:, 17] + 1.0).all() and (np.delete(a, 17, 1) == np.delete(tau, 17, 1)).all()


def test_time_step_and_thermodynamics_bit_identical(ref, olib):
    """a6 + a7: signed maximum (negative and tiny maxima included), the 12 h cap, T update and T_surface."""
    rng = np.random.default_rng(9)
    conv = (1000.0 / (np.arange(20) * 50.0 + 25.0)) ** (2.0 / 7.0)
    for case in range(2000):
        dE = rng.normal(size=20) * 10 ** rng.uniform(-3, 3)
        if case % 5 == 0:
            dE = -np.abs(dE)            # all negative: signed max gives a negative time step (main.cpp:157)
        if case % 11 == 0:
            dE *= 1e-4                  # tiny maximum: the 43 200 s cap binds
        dt_r = ref.ref_calculate_timestep(50.0, _p(dE))
        dt_o = olib.orc_calculate_timestep(50.0, _p(dE), 20)
        assert dt_r == dt_o, case
        T = rng.uniform(180.0, 320.0, 20)
        Tr, To = T.copy(), T.copy()
        Ts_o = np.zeros(1)
        Ts_r = ref.ref_thermodynamics(_p(Tr), 50.0, _p(dE), dt_r, _p(conv))
        olib.orc_thermodynamics(_p(To), 20, 50.0, _p(dE), dt_o, _p(Ts_o), _p(conv))
        assert Tr.tolist() == To.tolist() and Ts_r == Ts_o[0], case


def test_convective_adjustment_and_feedback_bit_identical(ref, olib, golden):
    """a8: theta = T * conv, std::sort(greater), back (main.cpp:536-540); magnus + H2O from relative humidity."""
    rng = np.random.default_rng(13)
    _, ci = _column(olib, golden)
    for case in range(1000):
        T = ci["Tlayer"] + rng.normal(size=20) * (0.0 if case == 0 else 10 ** rng.uniform(-6, 1.5))
        if case % 9 == 0:
            T[5] = T[6] * ci["conv"][6] / ci["conv"][5]     # (nearly) equal potential temperatures
        Tr, To, thr, tho = T.copy(), T.copy(), np.zeros(20), np.zeros(20)
        ref.ref_convective_adjustment(_p(Tr), _p(thr), _p(ci["conv"]))
        olib.orc_convective_adjustment(_p(To), _p(tho), _p(ci["conv"]), 20)
        assert Tr.tolist() == To.tolist() and thr.tolist() == tho.tolist(), case
        vr, vo = np.zeros(20), np.zeros(20)
        ref.ref_water_vapor_feedback(_p(Tr), _p(ci["rel_hum"]), _p(ci["player"]), _p(vr))
        olib.orc_water_vapor_feedback(_p(To), _p(ci["rel_hum"]), _p(ci["player"]), _p(vo), 20)
        assert vr.tolist() == vo.tolist(), case
    for T in np.concatenate([np.linspace(150.0, 350.0, 401), [29.0, 30.11, 30.12, 1e3]]):
        assert ref.ref_magnus(float(T)) == olib.orc_magnus(float(T))
    lev = rng.uniform(0, 400, 21)
    a, b = np.zeros(20), np.zeros(20)
    ref.ref_VMR_level_to_layer(_p(lev), _p(a))
    assert a.tolist() == ((lev[:-1] + lev[1:]) / 2.0).tolist()


@pytest.mark.parametrize("table,co2", [("Reduced10", 1.0), ("Reduced100", 1.0), ("Reduced20", 2.0)])
def test_101_chained_steps_bit_identical(ref, rtau, olib, golden, table, co2):
    """(iii): the loop body of main.cpp:531-583 driven HERE with the reference's compiled functions ONLY -- adjustment,
    feedback, read_tau, cloud, radiative_transfer, time step, thermodynamics -- against the oracle's own orc_step:
    state, fluxes and time step identical after every one of the 101 iterations i = 0..100."""
    _, ci = _column(olib, golden, co2)
    t = ol.read_rtab(os.path.join(golden, f"{table}Forcing.rtab"))
    sol = np.zeros(7)
    ref.ref_solar(_p(sol))
    e = ol.OracleEnsemble(olib, 1)
    e.set_grid(ci["plevel"])
    e.set_forcing(sol[6])
    e.set_repwvl_tables(t)
    e.set_columns(ci["Tlayer"][None], np.array([288.2]), ci["vmr9"][None], ci["rel_hum"][None])

    T, Ts, vmr = ci["Tlayer"].copy(), 288.2, ci["vmr9"].copy()
    time_h = np.float32(0.0)
    tau = _ref_tau(rtau, ref, t, ci, T, vmr)              # main.cpp:500-504: from the PRE-adjustment profile
    theta = np.zeros(20)
    for i in range(101):
        ref.ref_convective_adjustment(_p(T), _p(theta), _p(ci["conv"]))             # :536-540
        if i != 0:                                                                    # :549-572
            h2o = np.zeros(20)
            ref.ref_water_vapor_feedback(_p(T), _p(ci["rel_hum"]), _p(ci["player"]), _p(h2o))
            vmr[0] = h2o
            tau = _ref_tau(rtau, ref, t, ci, T, vmr)                                  # :564-568
        Ed, Eu, dE = ol.ref_radiative_transfer(ref, tau, t["wvl"], t["weight"], T, Ts, sol[6])   # :574
        dt = ref.ref_calculate_timestep(50.0, _p(dE))                                 # :576
        Ts = ref.ref_thermodynamics(_p(T), 50.0, _p(dE), dt, _p(ci["conv"]))          # :578
        time_h = np.float32(time_h + np.float32(dt) / np.float32(3600))               # :581, float arithmetic
        e.step(1)
        s, f = e.get_state(), e.get_fluxes()
        assert s["Tlayer"][0].tolist() == T.tolist(), i
        assert s["T_surface"][0] == Ts and s["timestep"][0] == dt, i
        assert f["E_down"][0].tolist() == Ed.tolist() and f["E_up"][0].tolist() == Eu.tolist(), i
        assert f["dE"][0].tolist() == dE.tolist(), i
        assert s["time_h"][0] == time_h, i
    assert 295.0 < Ts < 305.0


def test_lower_pos_bit_identical(olib, golden):
    """a5, the bracket search only: LowerPos (repwvl_thermal.cpp:19-45) compiled from the reference against
    orc_lower_pos on the tables' descending pressure grid, ascending temperature rows, values outside either
    end, exact grid hits (sgn == 0) and random grids with repeated values."""
    L = ol.load_ref_lowerpos()
    if L is None:
        pytest.skip("oracle/_ref/libref_lowerpos.so not built")
    t = ol.read_rtab(os.path.join(golden, "Reduced10Forcing.rtab"))
    rng = np.random.default_rng(4)
    grids = [t["p_grid"], t["p_grid"][::-1].copy()]
    grids += [t["t_ref"][k] + t["t_pert"] for k in range(0, 41, 5)]
    for _ in range(60):
        g = np.sort(rng.normal(size=int(rng.integers(2, 12)))) * (1 if rng.random() < 0.5 else -1)
        if rng.random() < 0.3:
            g[1] = g[0]
        grids.append(np.ascontiguousarray(g))
    n = 0
    for g in grids:
        g = np.ascontiguousarray(g, dtype=np.float64)
        lo, hi = g.min(), g.max()
        xs = list(g) + list((g[:-1] + g[1:]) / 2) + [lo - 1.0, hi + 1.0, lo * 0.999, hi * 1.001]
        xs += list(rng.uniform(lo - 0.1 * (hi - lo) - 1e-9, hi + 0.1 * (hi - lo) + 1e-9, 50))
        for x in xs:
            assert L.ref_lower_pos(_p(g), g.size, float(x)) == olib.orc_lower_pos(_p(g), g.size, float(x)), (g, x)
            n += 1
    assert n > 4000


@pytest.mark.parametrize("table", ["Reduced10", "Reduced20", "Reduced100"])
def test_read_tau_bit_identical_to_the_compiled_reference(rtau, olib, golden, table):
    """a5: orc_read_tau == the reference's read_tau statements compiled as they stand (libref_readtau.so), BIT FOR BIT:
    test.atm at 1x and 2x CO2; profiles perturbed by 5 / 40 / 150 K (temperatures leave the table on both sides:
    delT < 0 and > 1); every species present incl. the ones main.cpp zeroes; random positive and zero mixing ratios;
    random monotone pressure grids with 5..40 levels, among them surface pressures above the table's largest
    (LowerPos returns the last bracket, delP ~ -1e8) and a top at 0 hPa; the wavelength and weight vectors handed
    back are the table's.  prop_at_Lev = 0 as main.cpp:495 fixes it (the level variant is not on the path and the
    oracle does not restate it)."""
    atm, _ = ol.atm_load(olib, os.path.join(golden, "test.atm"))
    t = ol.read_rtab(os.path.join(golden, f"{table}Forcing.rtab"))
    xs = np.ascontiguousarray(t["xsec"])
    rng = np.random.default_rng(45)

    def both(plevel, T, vmr):
        plevel, T, vmr = (np.ascontiguousarray(a, dtype=np.float64) for a in (plevel, T, vmr))
        want, wvl, weight = ol.ref_read_tau(rtau, t, plevel, T, vmr)
        assert wvl.tolist() == t["wvl"].tolist() and weight.tolist() == t["weight"].tolist()
        got = np.zeros_like(want)
        olib.orc_read_tau(t["n_tpert"], t["n_species"], t["n_wvl"], t["n_p"], _p(t["p_grid"]), _p(t["t_ref"]),
                          _p(t["t_pert"]), _p(xs), plevel.size, _p(plevel), _p(T), _p(vmr), _p(got))
        assert np.isfinite(want).all()
        assert got.tolist() == want.tolist()
        return want

    n = 0
    for co2 in (1.0, 2.0, 0.5, 4.0):
        ci = ol.column_init(olib, atm, co2, 1.0)
        tau = both(ci["plevel"], ci["Tlayer"], ci["vmr9"])
        n += 1
        if table == "Reduced100" and co2 == 1.0:                      # SURVEY.md 8(c) samples
            assert abs(tau[0, 0] - 9.94215e-05) < 1e-10 and abs(tau[99, 19] - 113.779) < 1e-3
    ci = ol.column_init(olib, atm, 1.0, 1.0)
    for case in range(60):
        T = ci["Tlayer"] + rng.normal(size=20) * (5.0, 40.0, 150.0)[case % 3]
        vmr = ci["vmr9"] * rng.uniform(0.3, 3.0, (9, 1))
        if case % 2:
            vmr = vmr + 10 ** rng.uniform(-9, -2, (9, 20)) * (rng.random((9, 20)) < 0.7)   # CO, O2, HNO3, N2 too
        both(ci["plevel"], np.maximum(T, 1.0), vmr)
        n += 1
    for case in range(60):                                             # other grids
        nlev = int(rng.integers(5, 41))
        top = 0.0 if case % 3 == 0 else float(10 ** rng.uniform(-4, 1))
        sfc = float(rng.uniform(600.0, 1099.0)) if case % 4 else float(rng.uniform(1100.5, 1300.0))  # table ends at 1100 hPa
        pl = np.sort(np.concatenate([[top, sfc], rng.uniform(top, sfc, nlev - 2)]))
        if np.any(np.diff(pl) <= 0):
            continue
        T = rng.uniform(150.0, 330.0, nlev - 1)
        vmr = 10 ** rng.uniform(-8, -1.5, (9, nlev - 1)) * (rng.random((9, 1)) < 0.8)
        both(pl, T, vmr)
        n += 1
    assert n > 100


def test_read_tau_against_an_independent_array_statement(olib, golden):
    """Second opinion beside the compiled pin above (kept from the rounds in which a5 had none): the same rules written again, array-wise and
    from the text of repwvl_thermal.cpp:197-248 (not from oracle/rcm_oracle.c) -- brackets by a vectorised sign-change
    search, the four corners gathered with fancy indexing, the corner-difference form, species sum, number density --
    against orc_read_tau on test.atm x Reduced10/20/100 and on perturbed profiles that leave the table on both sides.
    Corroboration, not a pin.  Agreement to 1e-14 relative (numpy sums the nine species in the same order; the
    only freedom is FMA-free rounding, identical here), in practice bit for bit."""
    atm, _ = ol.atm_load(olib, os.path.join(golden, "test.atm"))
    ci = ol.column_init(olib, atm, 1.0, 1.0)
    rng = np.random.default_rng(8)

    def lower_pos(grid, x):                     # :19-45, vectorised over x; grid 1-D
        d = np.sign(grid[None, :] - x[:, None]).astype(int)
        change = d[:, 1:] != d[:, :-1]          # first p >= 1 whose sign differs from its predecessor's
        first = np.where(change.any(axis=1), change.argmax(axis=1), grid.size - 2)
        return np.minimum(first, grid.size - 2)

    for table in ("Reduced10", "Reduced20", "Reduced100"):
        t = ol.read_rtab(os.path.join(golden, f"{table}Forcing.rtab"))
        for case in range(4):
            T = ci["Tlayer"] + (0.0 if case == 0 else rng.normal(size=20) * (5.0, 40.0, 150.0)[case - 1])
            vmr = ci["vmr9"] * (1.0 if case == 0 else rng.uniform(0.3, 3.0, (9, 1)))
            want = _orc_tau(olib, t, ci, T, vmr, cloud=False)
            pPa = ci["plevel"][::-1] * 100.0                     # :68-73, :89: surface first
            Tr, vr = T[::-1], vmr[:, ::-1]
            midP = (pPa[1:] + pPa[:-1]) / 2                      # :214
            nd = (pPa[:-1] - pPa[1:]) * 6.02214076e23 / 0.0289647 / 9.80665   # :202
            ip = lower_pos(t["p_grid"], midP)                    # :226
            tgrid = t["t_ref"][ip][:, None] + t["t_pert"][None, :]             # :229-231, [layer][pert]
            it = np.array([lower_pos(tgrid[l], Tr[l:l + 1])[0] for l in range(20)])   # :232
            L = np.arange(20)
            delT = (Tr - tgrid[L, it]) / (tgrid[L, it + 1] - tgrid[L, it])     # :239
            delP = (midP - t["p_grid"][ip]) / (t["p_grid"][ip + 1] - t["p_grid"][ip])   # :240
            x = t["xsec"]                                        # [pert][species][wvl][p]
            tau = np.zeros((t["n_wvl"], 20))
            for s in range(9):                                   # :212, species order of :102-110
                c0 = np.stack([x[it[l], s, :, ip[l]] for l in range(20)], axis=1)      # [wvl][layer], :235
                cT = np.stack([x[it[l] + 1, s, :, ip[l]] for l in range(20)], axis=1) - c0
                cP = np.stack([x[it[l], s, :, ip[l] + 1] for l in range(20)], axis=1) - c0
                cPT = np.stack([x[it[l] + 1, s, :, ip[l] + 1] for l in range(20)], axis=1) - cP - cT - c0
                xs = c0 + cT * delT[None, :] + cP * delP[None, :] + cPT * delT[None, :] * delP[None, :]   # :241
                tau += xs * vr[s][None, :]                       # :244
            tau *= nd[None, :]                                   # :246
            tau = tau[:, ::-1]                                   # stored top first (:244 index)
            np.testing.assert_allclose(tau, want, rtol=1e-14, atol=0, err_msg=f"{table} case {case}")


def test_oracle_lbl_sweep_against_the_reference_pieces_wired_together(ref, olib, rcm):
    """a9 has no reference driver, but its pieces are reference code: cplkavg.cpp (band Planck) and main.cpp's
    monochromatic_radiative_transfer + dE formula.  Wired together by the documented composition, on the
    oracle's optical depths, they reproduce the oracle's lbl radiative transfer.  Not bit for bit, by construction:
    the reference's sweep forms the surface radiance as weight x monochromatic Planck (main.cpp:301), so the band
    integral at T_surface enters through a weight = band / monochromatic -- one extra rounding on one term.  Layers:
    the reference's own cplkavg.cpp values, identical to the oracle's (test_oracle_pins.py)."""
    import ctypes as C
    import helpers as H
    p = os.path.join(ol.REF_DIR, "libref_cplkavg.so")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/libref_cplkavg.so not built")
    band = getattr(C.CDLL(p), "_Z7cplkavgddd")
    band.restype, band.argtypes = C.c_double, [C.c_double] * 3
    inp = H.make_inputs(rcm, "lbl", 2, n_nodes=300, atm="fpda.lbl.atm")
    o = H.oracle_ensemble(olib, inp, threads=1)
    wvl, c = inp["wvl"], inp["cols"]
    mu, dmu = ol.ref_mu()
    taus = [o.get_tau(k) for k in range(2)]
    o.step(1)
    f = o.get_fluxes()
    for col in range(2):
        T0, Ts = c["Tlayer0"][col].copy(), float(c["T_surface0"][col])
        theta = np.zeros(20)
        ref.ref_convective_adjustment(_p(T0), _p(theta), _p(c["conv"]))
        Ed, Eu = np.zeros(21), np.zeros(21)
        for b in range(wvl.size - 1):
            B = np.array([band(wvl[b], wvl[b + 1], float(t)) for t in T0])
            w_sfc = band(wvl[b], wvl[b + 1], Ts) / ref.ref_cplkavg_scalar(float(wvl[b]), 1.0, Ts)
            ref.ref_monochromatic_radiative_transfer(_p(B), b, _p(np.ascontiguousarray(taus[col][b])), w_sfc,
                                                     wvl.size - 1, _p(wvl), _p(mu), dmu, _p(T0), Ts, _p(Ed), _p(Eu))
        dE = Ed[:-1] - Ed[1:] + Eu[1:] - Eu[:-1]
        dE[-1] += inp["solar"] + Ed[-1] - Eu[-1]
        assert f["E_down"][col].tolist() == Ed.tolist()          # no surface term in E_down: bit for bit
        np.testing.assert_allclose(f["E_up"][col], Eu, rtol=0, atol=1e-12)
        np.testing.assert_allclose(f["dE"][col], dE, rtol=0, atol=1e-12)


def _node_case(t, ci, rng, species, olib, vmr_value=1.0):
    """A column whose layer temperatures sit EXACTLY on temperature nodes of the table (t_ref[ip] + t_pert[k], the very
    sum read_tau forms at repwvl_thermal.cpp:230), one absorbing species with mixing ratio `vmr_value`: read_tau's bilinear
    interpolation then collapses, whatever its algebraic form, to the 1-D interpolation in pressure of the table's own
    entries  xsec[k][species][w][ip], xsec[k][species][w][ip+1]  -- reference-held DATA."""
    pPa = ci["plevel"] * 100.0
    midP = (pPa[1:] + pPa[:-1]) / 2                       # top first
    ip = np.array([olib.orc_lower_pos(_p(t["p_grid"]), t["n_p"], float(m)) for m in midP])
    k = rng.integers(0, t["n_tpert"], 20)
    T = t["t_ref"][ip] + t["t_pert"][k]                   # exact nodes
    vmr = np.zeros((9, 20))
    vmr[species] = vmr_value
    delP = (midP - t["p_grid"][ip]) / (t["p_grid"][ip + 1] - t["p_grid"][ip])
    nd = (pPa[1:] - pPa[:-1]) * 6.02214076e23 / 0.0289647 / 9.80665       # repwvl_thermal.cpp:124-126, 202
    x = t["xsec"]                                          # [t_pert][species][wvl][p]
    lo = np.stack([x[k[l], species, :, ip[l]] for l in range(20)], axis=1)      # [wvl][layer]
    hi = np.stack([x[k[l], species, :, ip[l] + 1] for l in range(20)], axis=1)
    want = (lo + (hi - lo) * delP[None, :]) * vmr_value * nd[None, :]
    scale = np.maximum(np.abs(lo), np.abs(hi)) * vmr_value * nd[None, :]
    return T, vmr, want, scale


@pytest.mark.parametrize("table", ["Reduced10", "Reduced100"])
def test_read_tau_reproduces_the_tables_own_nodes(olib, golden, table):
    """a5 pinned by reference-held data as far as data can pin it: every INDEX convention of read_tau -- species order
    (:102-110), wavelength row (the reversals at :206 and :244 cancel), temperature-perturbation index, pressure
    bracket and its direction on the descending grid, layer order (inputs reversed at :89-99, results stored back top
    first) and the number density (:202) -- checked against the table entries themselves, for each of the 9 species,
    random temperature nodes incl. the first and last perturbation.  What this cannot pin is the rounding of the
    corner-difference form between nodes (:235-241); tolerance 1e-12 of the larger corner."""
    atm, _ = ol.atm_load(olib, os.path.join(golden, "test.atm"))
    ci = ol.column_init(olib, atm, 1.0, 1.0)
    t = ol.read_rtab(os.path.join(golden, f"{table}Forcing.rtab"))
    rng = np.random.default_rng(31)
    for species in range(9):
        for rep in range(3):
            T, vmr, want, scale = _node_case(t, ci, rng, species, olib)
            got = _orc_tau(olib, t, ci, T, vmr, cloud=False)
            assert np.all(np.abs(got - want) <= 1e-12 * scale + 1e-300), (species, rep)
        assert np.abs(want).max() > 0 or species in ()     # every species has entries in these tables


def test_oracle_against_the_committed_reference_vectors(olib, golden):
    """The same pin WITHOUT the reference tree or oracle/_ref: tests/golden/ref_main_vectors.npz holds seeded inputs
    and the outputs of the reference's own compiled functions (tests/golden/make_ref_vectors.py, run once where the
    reference tree is; data only).  The oracle must reproduce every number bit for bit: Planck, radiative_transfer on 40
    random cases, time step + thermodynamics, adjustment + feedback, magnus, solar set-up, constants, LowerPos, 600
    cplkavg.cpp values (all branches), read_tau's optical depths for 12 columns x Reduced10/20/100 (a5; test.atm at
    1x / 2x CO2 and perturbed profiles that leave the table on both sides), and the state / fluxes at iterations 0, 1,
    2, 10, 50, 100 of the 101-iteration chain for Reduced10 and Reduced100 (every function of the chain, read_tau
    included, was the reference's when the vectors were made).  Never skips."""
    v = np.load(os.path.join(golden, "ref_main_vectors.npz"))
    wvl, wgt, T = v["planck_in"]
    assert [olib.orc_planck_mono(float(a), float(b), float(c)) for a, b, c in zip(wvl, wgt, T)] == v["planck_out"].tolist()
    for case in range(int(v["rt_cases"][0])):
        Ts, solar = v[f"rt{case}_scalars"]
        got = _orc_rt(olib, v[f"rt{case}_tau"], v[f"rt{case}_wvl"], v[f"rt{case}_weight"], v[f"rt{case}_T"], Ts, solar)
        assert np.concatenate(got).tolist() == v[f"rt{case}_out"].tolist(), case
    conv = np.ascontiguousarray(v["step_conv"])
    for dE, Tl, want in zip(v["step_dE"], v["step_T"], v["step_out"]):
        dE, Tn, Ts = np.ascontiguousarray(dE), np.ascontiguousarray(Tl).copy(), np.zeros(1)
        dt = olib.orc_calculate_timestep(50.0, _p(dE), 20)
        olib.orc_thermodynamics(_p(Tn), 20, 50.0, _p(dE), dt, _p(Ts), _p(conv))
        assert [dt, Ts[0]] + Tn.tolist() == want.tolist()
    conv, rh, pl = (np.ascontiguousarray(v[k]) for k in ("adj_conv", "adj_rel_hum", "adj_player"))
    for Tl, want in zip(v["adj_T"], v["adj_out"]):
        Ta, th, h2o = np.ascontiguousarray(Tl).copy(), np.zeros(20), np.zeros(20)
        olib.orc_convective_adjustment(_p(Ta), _p(th), _p(conv), 20)
        olib.orc_water_vapor_feedback(_p(Ta), _p(rh), _p(pl), _p(h2o), 20)
        assert np.concatenate([Ta, th, h2o]).tolist() == want.tolist()
    assert [olib.orc_magnus(float(t)) for t in v["magnus_in"]] == v["magnus_out"].tolist()
    sol = np.zeros(7)
    olib.orc_default_solar(_p(sol))
    assert sol.tolist() == v["solar_out"].tolist()
    c = v["consts_out"]
    assert (c[8], c[9], c[10], c[12], c[18]) == (20, 21, 30, 2.0, 17)
    for i in range(int(v["lp_cases"][0])):
        g = np.ascontiguousarray(v[f"lp{i}_grid"])
        assert [olib.orc_lower_pos(_p(g), g.size, float(x)) for x in v[f"lp{i}_x"]] == v[f"lp{i}_out"].tolist(), i
    for lo, hi, t, want in v["cplkavg"]:                       # a10: cplkavg.cpp, all three branches
        assert olib.orc_cplkavg(float(lo), float(hi), float(t)) == want, (lo, hi, t)
    atm, _ = ol.atm_load(olib, os.path.join(golden, "test.atm"))
    ci = ol.column_init(olib, atm, 1.0, 1.0)
    assert ci["plevel"].tolist() == v["tau_plevel"].tolist()
    for table in ("Reduced10", "Reduced20", "Reduced100"):     # a5: read_tau
        t = ol.read_rtab(os.path.join(golden, f"{table}Forcing.rtab"))
        for Tl, vm, want in zip(v[f"tau_{table}_T"], v[f"tau_{table}_vmr"], v[f"tau_{table}_out"]):
            assert _orc_tau(olib, t, ci, Tl, vm, cloud=False).tolist() == want.tolist(), table
    for table in ("Reduced10", "Reduced100"):
        e = ol.OracleEnsemble(olib, 1)
        e.set_grid(ci["plevel"])
        e.set_forcing(sol[6])
        e.set_repwvl_tables(ol.read_rtab(os.path.join(golden, f"{table}Forcing.rtab")))
        e.set_columns(ci["Tlayer"][None], np.array([288.2]), ci["vmr9"][None], ci["rel_hum"][None])
        done = 0
        for row in v[f"chain_{table}"]:
            i = int(row[0])
            e.step(i + 1 - done)
            done = i + 1
            s, f = e.get_state(), e.get_fluxes()
            got = np.concatenate([[i, s["timestep"][0], s["T_surface"][0]], s["Tlayer"][0], f["E_down"][0], f["E_up"][0],
                                  f["dE"][0]])
            assert got.tolist() == row.tolist(), (table, i)


def test_build_ref_refuses_a_reference_whose_lines_have_moved(tmp_path):
    """the reference is compiled by LINE SLICES, so oracle/build_ref.sh hashes every slice (and every range it
    deliberately skips) against oracle/ref_slices.sha256 before g++ sees anything.  A tree in which repwvl_thermal.cpp
    is another file (a symlink to the reference's cplkavg.cpp: nothing is copied) must fail loudly, name the slices
    that differ and leave NO library behind -- not even a stale one: the output directory is removed first."""
    import subprocess
    REF = "/root/reference"
    if not os.path.isfile(os.path.join(REF, "main.cpp")):
        pytest.skip("reference tree absent")
    script = os.path.join(ol.ROOT, "oracle", "build_ref.sh")
    fake = tmp_path / "ref"
    (fake / "repwvl_V2.01_cpp").mkdir(parents=True)
    (fake / "lbl.arts").mkdir()
    os.symlink(os.path.join(REF, "main.cpp"), fake / "main.cpp")
    os.symlink(os.path.join(REF, "cplkavg.cpp"), fake / "cplkavg.cpp")
    os.symlink(os.path.join(REF, "lbl.arts", "ascii.cpp"), fake / "lbl.arts" / "ascii.cpp")
    os.symlink(os.path.join(REF, "cplkavg.cpp"), fake / "repwvl_V2.01_cpp" / "repwvl_thermal.cpp")   # the "moved" file
    out = tmp_path / "out"
    out.mkdir()
    (out / "libref_readtau.so").write_bytes(b"stale")
    env = dict(os.environ, RCM_REF_OUT=str(out))
    r = subprocess.run(["bash", script, str(fake)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "not the revision the line slices were cut for" in r.stderr
    for name in ("rt_prelude", "rt_210_262", "rt_skipped"):
        assert name in r.stderr
    assert not out.exists()                       # the stale library went with the directory
    # the unmodified reference builds into the same place (five libraries)
    r = subprocess.run(["bash", script, REF], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sorted(os.listdir(out)) == ["libref_ascii.so", "libref_cplkavg.so", "libref_lowerpos.so", "libref_main.so",
                                       "libref_readtau.so"]
