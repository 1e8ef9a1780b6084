#!/usr/bin/env python3
"""Host-to-device hand-over of the default bench workload (10 000 columns x 131 072 nodes): the time of
each set-up call of the C ABI and of the read-back, i.e. everything that crosses PCIe outside the time loop."""
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
inp = H.make_inputs(pkg, "lbl", ncol, n_nodes=131072, atm="fpda.lbl.atm")
c = inp["cols"]
for rep in range(2):  # the second round is the warm one
    t = [time.perf_counter()]
    e = pkg.Ensemble(ncol); t.append(time.perf_counter())
    e.set_grid(c["plevel"]); e.set_forcing(inp["solar"]); t.append(time.perf_counter())
    e.set_lbl_tables(inp["wvl"], inp["tabs"], c["h2o_ref"]); e.sync(); t.append(time.perf_counter())
    e.set_columns(c["Tlayer0"], c["T_surface0"], c["gas"], c["rel_hum"]); e.sync(); t.append(time.perf_counter())
    e.step(1); e.sync(); t.append(time.perf_counter())
    e.step(4); e.sync(); t.append(time.perf_counter())
    s = e.get_state(); f = e.get_fluxes(); t.append(time.perf_counter())
    names = ["create", "grid+forcing", "set_lbl_tables (105.9 MB H2D + transpose)", "set_columns", "first step (allocs partials)",
             "4 steps", "get_state + get_fluxes (D2H)"]
    print(f"round {rep}: " + "; ".join(f"{n} {1e3 * (b - a):.1f} ms" for n, a, b in zip(names, t[:-1], t[1:])))
    tot = t[-1] - t[0]
    print(f"   whole job incl. hand-over, 5 steps: {tot:.3f} s = {ncol * 5 / tot:.0f} column-timesteps/s; "
          f"in-loop rate {ncol * 4 / (t[6] - t[5]):.0f}")
    e.close()
