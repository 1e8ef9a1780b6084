#!/usr/bin/env python3
"""A/B of library builds on the Planck-sensitive checks: narrow-band slice (fluxes vs oracle, 3 W/m^2 total) and the
full-range lbl flux error, plus the lbl kernel time.  usage (GPU box): python tools/ab_planck.py [ncol=2048] [rounds=3]"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import helpers as H
import oracle_lib as ol
import our_first_climate_model_amd as pkg
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
olib = ol.load()
libs = {"product": pkg.lib()}
for so in sorted(glob.glob(os.path.join(ROOT, "our_first_climate_model_amd", "variants", "librcm_hip_*.so"))):
    libs[os.path.basename(so)[len("librcm_hip_"):-3]] = pkg.load_library(so)
inp = H.make_inputs(pkg, "lbl", 3, n_nodes=3000, atm="fpda.lbl.atm")
inp["wvl"] = np.linspace(9900.0, 10100.0, 3000)
o = H.oracle_ensemble(olib, inp); o.step(1); fo = o.get_fluxes()
inp2 = H.make_inputs(pkg, "lbl", 3, n_nodes=6000, atm="fpda.lbl.atm")
o2 = H.oracle_ensemble(olib, inp2); o2.step(1); fo2 = o2.get_fluxes()
for name, L in libs.items():
    g = H.gpu_ensemble(pkg, inp, library=L); g.step(1); fg = g.get_fluxes()
    e1 = max(H.max_abs(fg[k], fo[k]) for k in fo)
    g = H.gpu_ensemble(pkg, inp2, library=L); g.step(1); fg = g.get_fluxes()
    e2 = max(H.max_abs(fg[k], fo2[k]) for k in fo2)
    print(f"{name:10s} narrow-band slice |GPU-oracle| {e1:.2e} W/m^2 (test bound 1e-12)   full range {e2:.2e}", flush=True)
big = H.make_inputs(pkg, "lbl", ncol, n_nodes=131072, atm="fpda.lbl.atm")
ens = {}
for name, L in libs.items():
    g = H.gpu_ensemble(pkg, big, library=L); g.step(1); g.sync(); g.set_rt_timing(True); ens[name] = g
t = {n: [] for n in libs}
for _ in range(rounds):
    for name, g in ens.items():
        g.step(1); t[name].append(g.get_rt_timing()[0])
for name in libs:
    a = np.array(t[name]); print(f"{name:10s} lbl {ncol} cols: median {np.median(a):8.3f} ms (min {a.min():8.3f})", flush=True)
