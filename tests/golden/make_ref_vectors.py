#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Writes tests/golden/ref_main_vectors.npz: inputs and OUTPUTS OF THE REFERENCE'S OWN COMPILED CODE
(oracle/_ref/libref_main.so = /root/reference/main.cpp:12-21,25-344 piped unedited into g++; libref_readtau.so =
the netCDF-free statements of repwvl_thermal.cpp:55-262 the same way; oracle/build_ref.sh) for
the functions on the hot path, so that the oracle stays pinned in a checkout that has no reference tree (and no
oracle/_ref): tests/test_reference_pins.py::test_oracle_against_the_committed_reference_vectors compares the oracle
with these numbers bit for bit.  Data only: seeded inputs, the reference's outputs.  Needs the reference tree once:

    make -C oracle && python tests/golden/make_ref_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402

_p = ol._p
ref = ol.load_ref_main()
lp = ol.load_ref_lowerpos()
rt = ol.load_ref_readtau()
assert None not in (ref, lp, rt), "build oracle/_ref first (make -C oracle, reference tree present)"
olib = ol.load()  # only for the inputs that come from the atmosphere file (the loader / column set-up of main())
rng = np.random.default_rng(20260104)
out = {}

# a1: Planck, scalar overload (main.cpp:197-204)
n = 400
wvl = np.exp(rng.uniform(np.log(300.0), np.log(5e5), n))
wgt = rng.normal(size=n) * 10 ** rng.uniform(-3, 6, n)
T = np.where(rng.random(n) < 0.9, rng.uniform(100.0, 400.0, n), 10 ** rng.uniform(-1, 4, n))
out["planck_in"] = np.stack([wvl, wgt, T])
out["planck_out"] = np.array([ref.ref_cplkavg_scalar(float(a), float(b), float(c)) for a, b, c in zip(wvl, wgt, T)])

# a2-a4: radiative_transfer (main.cpp:320-344) on random cases, 1..24 wavelengths
cases = []
for case in range(40):
    nw = int(rng.integers(1, 25))
    tau = 10 ** rng.uniform(-8, np.log10(3e4), (nw, 20))
    w = np.sort(np.exp(rng.uniform(np.log(3000.0), np.log(2e5), nw)))
    wt = rng.normal(size=nw) * 10 ** rng.uniform(0, 5.6, nw)
    wt[rng.integers(0, nw)] = -46250.7
    Tl = rng.uniform(180.0, 320.0, 20)
    Ts, solar = float(rng.uniform(250.0, 330.0)), float(rng.uniform(0.0, 400.0))
    Ed, Eu, dE = ol.ref_radiative_transfer(ref, tau, w, wt, Tl, Ts, solar)
    out[f"rt{case}_tau"], out[f"rt{case}_wvl"], out[f"rt{case}_weight"], out[f"rt{case}_T"] = tau, w, wt, Tl
    out[f"rt{case}_scalars"] = np.array([Ts, solar])
    out[f"rt{case}_out"] = np.concatenate([Ed, Eu, dE])
out["rt_cases"] = np.array([40])

# a6 + a7: time step and thermodynamics (main.cpp:156-176)
conv = (1000.0 / (np.arange(20) * 50.0 + 25.0)) ** (2.0 / 7.0)
dEs, Ts_in, res = [], [], []
for case in range(200):
    dE = rng.normal(size=20) * 10 ** rng.uniform(-3, 3)
    if case % 5 == 0:
        dE = -np.abs(dE)
    if case % 11 == 0:
        dE *= 1e-4
    Tl = rng.uniform(180.0, 320.0, 20)
    dt = ref.ref_calculate_timestep(50.0, _p(dE))
    Tn = Tl.copy()
    Ts = ref.ref_thermodynamics(_p(Tn), 50.0, _p(dE), dt, _p(conv))
    dEs.append(dE), Ts_in.append(Tl), res.append(np.concatenate([[dt, Ts], Tn]))
out["step_dE"], out["step_T"], out["step_conv"], out["step_out"] = np.array(dEs), np.array(Ts_in), conv, np.array(res)

# a8: adjustment (main.cpp:536-540), feedback (:277-289), magnus
atm, _ = ol.atm_load(olib, os.path.join(ROOT, "tests", "golden", "test.atm"))
ci = ol.column_init(olib, atm, 1.0, 1.0)
Tin, res = [], []
for case in range(200):
    Tl = ci["Tlayer"] + rng.normal(size=20) * (0.0 if case == 0 else 10 ** rng.uniform(-6, 1.5))
    if case % 9 == 0:
        Tl[5] = Tl[6] * ci["conv"][6] / ci["conv"][5]
    Ta, th, v = Tl.copy(), np.zeros(20), np.zeros(20)
    ref.ref_convective_adjustment(_p(Ta), _p(th), _p(ci["conv"]))
    ref.ref_water_vapor_feedback(_p(Ta), _p(ci["rel_hum"]), _p(ci["player"]), _p(v))
    Tin.append(Tl), res.append(np.concatenate([Ta, th, v]))
out["adj_T"], out["adj_out"] = np.array(Tin), np.array(res)
out["adj_conv"], out["adj_rel_hum"], out["adj_player"] = ci["conv"], ci["rel_hum"], ci["player"]
mT = np.concatenate([np.linspace(150.0, 350.0, 201), [29.0, 30.11, 30.12, 1e3]])
out["magnus_in"], out["magnus_out"] = mT, np.array([ref.ref_magnus(float(t)) for t in mT])

# solar set-up and constants (main.cpp:66-91, 214-264)
sol, cst = np.zeros(7), np.zeros(20)
ref.ref_solar(_p(sol)), ref.ref_consts(_p(cst))
out["solar_out"], out["consts_out"] = sol, cst

# a5: LowerPos (repwvl_thermal.cpp:19-45)
t = ol.read_rtab(os.path.join(ROOT, "tests", "golden", "Reduced10Forcing.rtab"))
grids = [t["p_grid"], t["t_ref"][7] + t["t_pert"], t["t_ref"][30] + t["t_pert"]]
for _ in range(12):
    g = np.sort(rng.normal(size=int(rng.integers(2, 12)))) * (1 if rng.random() < 0.5 else -1)
    grids.append(np.ascontiguousarray(g))
for i, g in enumerate(grids):
    g = np.ascontiguousarray(g, dtype=np.float64)
    xs = np.concatenate([g, (g[:-1] + g[1:]) / 2, [g.min() - 1.0, g.max() + 1.0],
                         rng.uniform(g.min() - 0.1 * np.ptp(g) - 1e-9, g.max() + 0.1 * np.ptp(g) + 1e-9, 30)])
    out[f"lp{i}_grid"], out[f"lp{i}_x"] = g, xs
    out[f"lp{i}_out"] = np.array([lp.ref_lower_pos(_p(g), g.size, float(x)) for x in xs], dtype=np.int32)
out["lp_cases"] = np.array([len(grids)])

# a5: read_tau (repwvl_thermal.cpp:55-262 as compiled into libref_readtau.so), no cloud.  Per table: test.atm at 1x and
# 2x CO2, then 10 perturbed columns: temperatures off by 5 / 40 / 150 K (the last leave t_ref + t_pert on both sides),
# every gas scaled, CO / O2 / HNO3 / N2 present in half of them.  Inputs stored with the outputs.
out["tau_plevel"] = ci["plevel"]
for table in ("Reduced10", "Reduced20", "Reduced100"):
    tt = ol.read_rtab(os.path.join(ROOT, "tests", "golden", f"{table}Forcing.rtab"))
    Ts_, vs_, taus_ = [], [], []
    for case in range(12):
        if case < 2:
            c2 = ol.column_init(olib, atm, float(case + 1), 1.0)
            Tl, v = c2["Tlayer"].copy(), c2["vmr9"].copy()
        else:
            Tl = np.maximum(ci["Tlayer"] + rng.normal(size=20) * (5.0, 40.0, 150.0)[case % 3], 1.0)
            v = ci["vmr9"] * rng.uniform(0.3, 3.0, (9, 1))
            if case % 2:
                v = v + 10 ** rng.uniform(-9, -3, (9, 20)) * (rng.random((9, 20)) < 0.7)
        tau, _, _ = ol.ref_read_tau(rt, tt, ci["plevel"], Tl, v)
        Ts_.append(Tl), vs_.append(v), taus_.append(tau)
    out[f"tau_{table}_T"], out[f"tau_{table}_vmr"], out[f"tau_{table}_out"] = np.array(Ts_), np.array(vs_), np.array(taus_)

# 101 chained iterations of main.cpp:536-581 with the reference's functions ALONE (read_tau included since round 4)
for table in ("Reduced10", "Reduced100"):
    tt = ol.read_rtab(os.path.join(ROOT, "tests", "golden", f"{table}Forcing.rtab"))

    def tau_of(Tl, v):
        tau, _, _ = ol.ref_read_tau(rt, tt, ci["plevel"], Tl, v)          # main.cpp:500 / :564
        ref.ref_cloud_into_tau(_p(tau), tt["n_wvl"])                     # :504 / :568
        return tau
    Tl, Ts, vmr = ci["Tlayer"].copy(), 288.2, ci["vmr9"].copy()
    tau = tau_of(Tl, vmr)
    th, hist = np.zeros(20), []
    for i in range(101):
        ref.ref_convective_adjustment(_p(Tl), _p(th), _p(ci["conv"]))
        if i != 0:
            h2o = np.zeros(20)
            ref.ref_water_vapor_feedback(_p(Tl), _p(ci["rel_hum"]), _p(ci["player"]), _p(h2o))
            vmr[0] = h2o
            tau = tau_of(Tl, vmr)
        Ed, Eu, dE = ol.ref_radiative_transfer(ref, tau, tt["wvl"], tt["weight"], Tl, Ts, sol[6])
        dt = ref.ref_calculate_timestep(50.0, _p(dE))
        Ts = ref.ref_thermodynamics(_p(Tl), 50.0, _p(dE), dt, _p(ci["conv"]))
        if i in (0, 1, 2, 10, 50, 100):
            hist.append(np.concatenate([[i, dt, Ts], Tl, Ed, Eu, dE]))
    out[f"chain_{table}"] = np.array(hist)

# a10: cplkavg.cpp as a whole translation unit (oracle/_ref/libref_cplkavg.so): close bounds (Simpson), wide bands at
# thermal temperatures (exponential series), any temperature (power series included)
import ctypes as C  # noqa: E402
band = getattr(C.CDLL(os.path.join(ol.REF_DIR, "libref_cplkavg.so")), "_Z7cplkavgddd")
band.restype, band.argtypes = C.c_double, [C.c_double] * 3
rows = []
for _ in range(600):
    lo = float(np.exp(rng.uniform(np.log(300.0), np.log(2e5))))
    kind = rng.integers(0, 3)
    hi = lo * (1 + 10 ** rng.uniform(-6, -2.2)) if kind == 0 else lo * rng.uniform(1.011, 50.0)
    t = float(rng.uniform(150.0, 350.0)) if kind < 2 else float(10 ** rng.uniform(0.5, 4))
    rows.append([lo, hi, t, band(lo, hi, t)])
out["cplkavg"] = np.array(rows)

path = os.path.join(ROOT, "tests", "golden", "ref_main_vectors.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes,", len(out), "arrays")
