#!/usr/bin/env python3
"""A/B of library builds on the repwvl workload (10 000 x Reduced100) in ONE process: the three-kernel path
(flattened RT kernel) and the fused whole-step kernel at forced lane-pairs-per-column (builds with -DRCM_LPC_ENV).
    usage (GPU box): python tools/ab_repwvl.py [ncol=10000] [steps=100] [rounds=3] [table=Reduced100]"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import helpers as H
import our_first_climate_model_amd as pkg

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
table = sys.argv[4] if len(sys.argv) > 4 else "Reduced100"
libs = {"product": pkg.lib()}
for so in sorted(glob.glob(os.path.join(ROOT, "our_first_climate_model_amd", "variants", "librcm_hip_*.so"))):
    libs[os.path.basename(so)[len("librcm_hip_"):-3]] = pkg.load_library(so)
inp = H.make_inputs(pkg, "repwvl", ncol, table=table)
# the fused kernel exists only in builds with -DRCM_WITH_VARIANTS (add -DRCM_LPC_ENV to force its lane pairs per column)
cases = [("flat", 0, None)] + ([("fused16", 1, "16"), ("fused32", 1, "32")] if os.environ.get("AB_FUSED") else [])
ens = {}
for name, L in libs.items():
    for cname, fusion, lpc in cases:
        g = H.gpu_ensemble(pkg, inp, library=L)
        H.set_step_fusion(g, fusion)
        ens[(name, cname)] = (g, lpc)
import time
times = {k: [] for k in ens}
walls = {k: [] for k in ens}
for r in range(rounds + 1):
    for k, (g, lpc) in ens.items():
        if lpc:
            os.environ["RCM_LPC"] = lpc
        else:
            os.environ.pop("RCM_LPC", None)
        g.set_rt_timing(True)
        g.step(steps)
        ms, n = g.get_rt_timing()
        g.set_rt_timing(False)  # the whole step without the timing events
        g.sync()
        t0 = time.perf_counter()
        g.step(steps)
        g.sync()
        wall = (time.perf_counter() - t0) / steps
        if r:
            times[k].append(ms / steps)
            walls[k].append(wall * 1e3)
for k, t in times.items():
    t = np.array(t)
    print(f"{k[0]:10s} {k[1]:8s} kernel median {np.median(t) * 1e3:7.1f} us/step (min {t.min() * 1e3:7.1f})   "
          f"whole step median {np.median(walls[k]) * 1e3:7.1f} us (min {min(walls[k]) * 1e3:7.1f})", flush=True)
