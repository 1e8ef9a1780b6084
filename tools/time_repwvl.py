#!/usr/bin/env python3
"""Times the repwvl path (BASELINE.json configs[2]: 10 000 columns, Reduced100) per step, RT kernel and whole step."""
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
steps = 50
inp = H.make_inputs(pkg, "repwvl", ncol, table=table)
for pair in (0, 1):
    os.environ["RCM_PAIR"] = str(pair)
    g = H.gpu_ensemble(pkg, inp)
    g.step(5)
    g.sync()
    g.set_rt_timing(True)
    t0 = time.perf_counter()
    g.step(steps)
    g.sync()
    dt = time.perf_counter() - t0
    ms, n = g.get_rt_timing()
    nw = inp["table"]["n_wvl"]
    print(f"{table} ncol={ncol} pair={pair}: {dt / steps * 1e3:.3f} ms/step wall, RT kernel {ms / n:.3f} ms, "
          f"{ncol * steps / dt:.0f} column-timesteps/s, {ncol * nw / (ms / n) / 1e6:.2f} Gpoint/s, "
          f"chunks={g.get_spectral_chunks()}")
    g.close()
