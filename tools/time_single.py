#!/usr/bin/env python3
"""BASELINE.json configs[1]: ONE column, full lbl spectrum: ms per time step on the GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import helpers as H
import our_first_climate_model_amd as pkg
cases = [(1, 0), (1, 1024), (3, 0), (8, 0), (8, 1024), (64, 0), (64, 512), (500, 0)]
for ncol, chunks in cases:
    inp = H.make_inputs(pkg, "lbl", ncol, n_nodes=131072, atm="fpda.lbl.atm")
    g = H.gpu_ensemble(pkg, inp)
    g.set_spectral_chunks(chunks)
    g.step(3); g.sync(); g.set_rt_timing(True)
    t0 = time.perf_counter(); g.step(20); g.sync(); dt = time.perf_counter() - t0
    ms, n = g.get_rt_timing()
    print(f"ncol={ncol}: {dt / 20 * 1e3:.3f} ms/step wall, RT kernel {ms / n:.3f} ms, chunks={g.get_spectral_chunks()}, "
          f"{ncol * 20 / dt:.1f} column-timesteps/s")
