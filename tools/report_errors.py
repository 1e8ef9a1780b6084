#!/usr/bin/env python3
"""Prints the measured GPU-vs-oracle deviations (fluxes, temperatures) for a few configurations."""
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

olib = ol.load()
cases = [("repwvl R100, 1 col", dict(mode="repwvl", ncol=1, table="Reduced100", perturb=False), (1, 100)),
         ("repwvl R100, 37 cols", dict(mode="repwvl", ncol=37, table="Reduced100"), (1, 20)),
         ("lbl 1000 nodes, 5 cols", dict(mode="lbl", ncol=5, n_nodes=1000, atm="fpda.lbl.atm"), (1, 20)),
         ("lbl 20000 nodes, 3 cols", dict(mode="lbl", ncol=3, n_nodes=20000, atm="fpda.lbl.atm"), (1, 10))]
for name, kw, (s1, s2) in cases:
    mode, ncol = kw.pop("mode"), kw.pop("ncol")
    inp = H.make_inputs(pkg, mode, ncol, **kw)
    g, o = H.gpu_ensemble(pkg, inp), H.oracle_ensemble(olib, inp)
    g.step(s1), o.step(s1)
    fg, fo = g.get_fluxes(), o.get_fluxes()
    e1 = max(H.max_abs(fg[k], fo[k]) for k in fg)
    g.step(s2), o.step(s2)
    fg, fo = g.get_fluxes(), o.get_fluxes()
    e2 = max(H.max_abs(fg[k], fo[k]) for k in fg)
    et = H.max_abs(g.get_state()["Tlayer"], o.get_state()["Tlayer"])
    print(f"{name:28s} max|dFlux| step0 {e1:.2e}  after {s1 + s2} steps {e2:.2e} W/m^2   max|dT| {et:.2e} K")
    g.close()
