#!/usr/bin/env python3
"""How far are the GPU fluxes and the CPU oracle's from the EXACT value of the reference's formulas?
The sweep main.cpp:291-344 (30 angles, alpha = 1 - exp(-tau/mu), the two recurrences, the weighted sums) is
re-evaluated in numpy long double (64-bit mantissa) from the same optical depths and the same Planck values
(taken in double from the oracle, so only the sweep is under test); fluxes of one step are compared.

    python tools/accuracy_vs_longdouble.py [n_nodes=4000]      (on the MI355X box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import helpers as H
import oracle_lib as ol
import our_first_climate_model_amd as pkg

LD = np.longdouble


def run(n_nodes=4000, log=print, library=None):
    """Returns [(|GPU - exact|, |oracle - exact|, |GPU - oracle|)] per column, W/m^2."""
    out = []
    olib = ol.load()
    inp = H.make_inputs(pkg, "lbl", 2, n_nodes=n_nodes, atm="fpda.lbl.atm")
    g, o = H.gpu_ensemble(pkg, inp, library=library), H.oracle_ensemble(olib, inp)
    g.step(1), o.step(1)
    fg, fo = g.get_fluxes(), o.get_fluxes()

    nangle, nl = 30, 20
    for col in range(2):
        tau = o.get_tau(col).astype(LD)                       # [bands][20], cloud included
        # the state has moved on by one step: the adjusted profile and surface temperature of step 0 come from a fresh handle
        o2 = H.oracle_ensemble(olib, inp)
        Tadj, _, _ = o2.get_profile(col)
        Ts = float(inp["cols"]["T_surface0"][col])
        wvl = inp["wvl"]
        nb = n_nodes - 1
        B = np.array([[olib.orc_cplkavg(float(wvl[b]), float(wvl[b + 1]), float(Tadj[l])) for l in range(nl)]
                      for b in range(nb)], dtype=LD)
        Bs = np.array([olib.orc_cplkavg(float(wvl[b]), float(wvl[b + 1]), Ts) for b in range(nb)], dtype=LD)
        Ed, Eu = np.zeros(nl + 1, LD), np.zeros(nl + 1, LD)
        dmu = LD(1) / LD(nangle)
        for i in range(nangle):
            mu = dmu / 2 + dmu * i
            alpha = 1 - np.exp(-tau / mu)                      # [bands][20]
            w = 2 * LD(np.pi) * mu * dmu
            L = np.zeros(nb, LD)
            for l in range(nl):
                L = (1 - alpha[:, l]) * L + alpha[:, l] * B[:, l]
                Ed[l + 1] += w * L.sum()
            L = Bs.copy()
            Eu[nl] += w * L.sum()
            for l in range(nl - 1, -1, -1):
                L = (1 - alpha[:, l]) * L + alpha[:, l] * B[:, l]
                Eu[l] += w * L.sum()
        eg = max(float(np.max(np.abs(fg["E_down"][col] - Ed))), float(np.max(np.abs(fg["E_up"][col] - Eu))))
        eo = max(float(np.max(np.abs(fo["E_down"][col] - Ed))), float(np.max(np.abs(fo["E_up"][col] - Eu))))
        ego = max(H.max_abs(fg["E_down"][col], fo["E_down"][col]), H.max_abs(fg["E_up"][col], fo["E_up"][col]))
        out.append((eg, eo, ego))
        log(f"column {col}, {nb} bands: |GPU - exact| {eg:.2e}  |oracle - exact| {eo:.2e}  |GPU - oracle| {ego:.2e} W/m^2 "
              f"(E_up[0] = {float(Eu[0]):.6f})")
    return out


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 4000)
