#!/usr/bin/env python3
"""Times the repwvl path (BASELINE.json configs[2]: 10 000 columns, Reduced100) per step for the three launch
forms: the fused whole-step kernel (k_step_repwvl), the three-kernel path with the flattened RT kernel, and the
chunked lane-pair kernel.   usage (GPU box): python tools/time_repwvl.py [ncol] [table] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import helpers as H
import our_first_climate_model_amd as pkg

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
table = sys.argv[2] if len(sys.argv) > 2 else "Reduced100"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
from our_first_climate_model_amd import build as _b
VLIB = pkg.load_library(_b.build_variants())  # the fused kernel lives in the test-side build
inp = H.make_inputs(pkg, "repwvl", ncol, table=table)
nw = inp["table"]["n_wvl"]
for name, fusion, chunks in (("fused", 1, 0), ("flat", 0, 0), ("chunked", 0, 1)):
    g = H.gpu_ensemble(pkg, inp, library=VLIB if fusion == 1 else None)
    H.set_step_fusion(g, fusion)
    if chunks:
        g.set_spectral_chunks(chunks)
    g.step(20)
    g.sync()
    g.set_rt_timing(True)
    t0 = time.perf_counter()
    g.step(steps)
    g.sync()
    dt = time.perf_counter() - t0
    ms, n = g.get_rt_timing()
    print(f"{table} ncol={ncol} {name:8s}: {dt / steps * 1e3:.4f} ms/step wall, kernel {ms / steps:.4f} ms/step ({n} launches), "
          f"{ncol * steps / dt / 1e6:.2f} M column-timesteps/s, {ncol * nw / (ms / steps) / 1e6:.2f} Gpoint/s, "
          f"lane pairs/column={H.get_step_fusion(g)}", flush=True)
    g.close()
