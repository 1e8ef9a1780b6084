#!/usr/bin/env python3
"""A/B of launch geometries for FEW columns x the full lbl spectrum on ONE box: for each library variant and column count,
ms per step (wall, 200 steps in one rcm_step call) and the RT launch (HIP events).
usage (GPU box): ab_single.py name[:ENV=VAL,..] ...   (name = product | a variant of tools/build_variants.py)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import helpers as H
import our_first_climate_model_amd as pkg

inputs = {n: H.make_inputs(pkg, "lbl", n, n_nodes=131072, atm="fpda.lbl.atm") for n in (1, 2, 3, 4, 8, 16, 64)}
for rep in range(2):
    for spec in sys.argv[1:]:
        name, _, envs = spec.partition(":")
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=")
            os.environ[k] = v
        L = None if name == "product" else pkg.load_library(os.path.join(ROOT, "our_first_climate_model_amd", "variants", f"librcm_hip_{name}.so"))
        row = []
        for n, inp in inputs.items():
            g = H.gpu_ensemble(pkg, inp, library=L)
            g.step(20); g.sync(); g.set_rt_timing(True)
            t0 = time.perf_counter(); g.step(200); g.sync(); dt = (time.perf_counter() - t0) / 200
            ms, k = g.get_rt_timing()
            row.append(f"{n} col: {dt * 1e6:6.1f} us/step RT {ms / k * 1e3:6.1f} us ({g.get_spectral_chunks()} rec)")
            g.close()
        for kv in filter(None, envs.split(",")):
            os.environ.pop(kv.split("=")[0])
        print(f"{spec:28s} " + " | ".join(row), flush=True)
