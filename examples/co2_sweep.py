#!/usr/bin/env python3
"""What the reference answers for one column, for a whole CO2 sweep at once: surface temperature of
the radiative-convective steady state as a function of the CO2 mixing ratio (main.cpp:452 `factor`).

    python examples/co2_sweep.py [--ncol 64] [--table Reduced100] [--tol 1e-3]

Needs an MI355X (the product has no CPU path)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import our_first_climate_model_amd as rcm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ncol", type=int, default=64)
ap.add_argument("--table", default="Reduced100")
ap.add_argument("--tol", type=float, default=1e-3, help="K per step on the adjusted profile")
ap.add_argument("--max-steps", type=int, default=6000)
a = ap.parse_args()

golden = os.path.join(ROOT, "tests", "golden")
atm, _ = rcm.atm_load(os.path.join(golden, "test.atm"))
spec = rcm.perturbation_sweep(a.ncol)                      # CO2 factor linspace(0.5, 4.0)
cols = rcm.build_columns(atm, spec, "repwvl")
table = rcm.read_rtab(os.path.join(golden, f"{a.table}Forcing.rtab"))

ens = rcm.Ensemble(a.ncol)
ens.set_grid(cols["plevel"])
ens.set_forcing(rcm.default_solar()["solar_irr"])          # tau_s = 2.0, cloud in layer 17 (main.cpp:86-92)
ens.set_repwvl_tables(table)
ens.set_columns(cols["Tlayer0"], cols["T_surface0"], cols["gas"], cols["rel_hum"])
t0 = time.perf_counter()
steps, resid = ens.run_to_equilibrium(a.tol, a.max_steps, 100)
dt = time.perf_counter() - t0
s = ens.get_state()
print(f"{a.ncol} columns, {steps} steps in {dt:.2f} s ({a.ncol * steps / dt:.0f} column-timesteps/s), "
      f"residual {resid:.2e} K/step")
ppm = 400.0 * spec.co2_factor
print("  CO2 [ppm]   T_surface [K]   model time [days]")
for c in np.linspace(0, a.ncol - 1, min(a.ncol, 8)).round().astype(int):
    print(f"  {ppm[c]:9.1f}   {s['T_surface'][c]:12.4f}   {s['time_h'][c] / 24:10.1f}")
k = np.polyfit(np.log2(spec.co2_factor), s["T_surface"], 1)[0]
print(f"sensitivity of this model: {k:.3f} K per CO2 doubling (fixed relative humidity, fixed cloud)")
