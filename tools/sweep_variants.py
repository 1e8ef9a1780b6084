#!/usr/bin/env python3
"""A/B the thermal-RT kernel variants of the TEST-SIDE build (tests/_variants, -DRCM_WITH_VARIANTS: the production
library has no variants) in ONE process (interleaved rounds, HIP-event timing of the RT launches).
usage: sweep_variants.py [ncol] [nodes] [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import helpers as H
import our_first_climate_model_amd as pkg

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
variants = [v for v in os.environ.get("SWEEP", "0:0:0:0,1:1:0:0,1:0:0:1,1:1:0:1,1:1:16:1").split(",")]

from our_first_climate_model_amd import build as _b
VLIB = pkg.load_library(os.environ.get("RCM_LIB_PATH") or _b.build_variants())
inp = H.make_inputs(pkg, "lbl", ncol, n_nodes=nodes, atm="fpda.lbl.atm")
ens = {}
for v in variants:
    e, c, k, pl = (int(x) for x in v.split(":"))
    os.environ["RCM_EXP_VARIANT"], os.environ["RCM_CUBE"], os.environ["RCM_PLANCK_FAST"] = str(e), str(c), str(pl)
    g = H.gpu_ensemble(pkg, inp, library=VLIB)
    g.set_spectral_chunks(k)
    g.step(1)
    g.sync()
    g.set_rt_timing(True)
    ens[v] = g
times = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        ens[v].step(1)
        ms, n = ens[v].get_rt_timing()
        times[v].append(ms / n)
pts = ncol * (nodes - 1)
for v in variants:
    t = np.array(times[v])
    print(f"exp:cube:chunks:planck={v:10s} median {np.median(t):9.3f} ms  min {t.min():9.3f} ms  "
          f"{pts / np.median(t) / 1e6:8.2f} Gpoint/s  alg {808 * pts / np.median(t) / 1e6:8.1f} GB/s "
          f"chunks={ens[v].get_spectral_chunks()}")
