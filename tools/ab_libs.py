#!/usr/bin/env python3
"""A/B of library builds in ONE process on ONE box (boxes differ by 3-5 %): for every
our_first_climate_model_amd/variants/librcm_hip_<name>.so (tools/build_variants.py) the flux error against the
long-double evaluation of the reference's sweep (tools/accuracy_vs_longdouble.py) and the kernel time of the lbl
and repwvl workloads, rounds interleaved.
    usage (GPU box): python tools/ab_libs.py [lbl_ncol=2048] [rounds=3] [acc_nodes=4000]"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa: F401
import helpers as H
import our_first_climate_model_amd as pkg
import accuracy_vs_longdouble as acc

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
acc_nodes = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
libs = {"product": pkg.lib()}
for so in sorted(glob.glob(os.path.join(ROOT, "our_first_climate_model_amd", "variants", "librcm_hip_*.so"))):
    libs[os.path.basename(so)[len("librcm_hip_"):-3]] = pkg.load_library(so)

print(f"== flux error against long double, {acc_nodes - 1} bands, 2 columns (W/m^2): |GPU-exact| |oracle-exact| |GPU-oracle|")
for name, L in libs.items():
    r = acc.run(acc_nodes, lambda *_: None, library=L)
    print(f"{name:14s} " + "  ".join(f"{eg:.2e} {eo:.2e} {ego:.2e}" for eg, eo, ego in r), flush=True)

inp = H.make_inputs(pkg, "lbl", ncol, n_nodes=131072, atm="fpda.lbl.atm")
inr = H.make_inputs(pkg, "repwvl", 10000, table="Reduced100")
ens = {}
for name, L in libs.items():
    g = H.gpu_ensemble(pkg, inp, library=L)
    r = H.gpu_ensemble(pkg, inr, library=L)
    H.set_step_fusion(r, 0)
    g.step(1), r.step(20)
    g.sync(), r.sync()
    g.set_rt_timing(True), r.set_rt_timing(True)
    ens[name] = (g, r)
tl, tr = {n: [] for n in libs}, {n: [] for n in libs}
for _ in range(rounds):
    for name, (g, r) in ens.items():
        g.step(1)
        tl[name].append(g.get_rt_timing()[0])
        r.step(100)
        tr[name].append(r.get_rt_timing()[0] / 100)
print(f"== kernel time: lbl {ncol} columns x 131071 bands per launch; repwvl 10000 x Reduced100 per step (RT kernel, or the "
      "whole fused step)")
for name in libs:
    a, b = np.array(tl[name]), np.array(tr[name])
    print(f"{name:14s} lbl median {np.median(a):8.3f} ms (min {a.min():8.3f})   repwvl median {np.median(b) * 1e3:7.1f} us "
          f"(min {b.min() * 1e3:7.1f})  fusion={H.get_step_fusion(ens[name][1])}", flush=True)
