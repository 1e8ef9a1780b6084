#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (run on the MI355X box): ensemble size, spectrum length, angle
count, spectral chunking, cloud position/depth and step count are drawn from a seeded generator; fluxes,
optical depths and temperatures are compared with the CPU oracle at the tolerances of the tests.

    python tools/fuzz_parity.py [cases=120] [seed=1] [big]

Prints one line per case and a summary; exit code 1 if any case is outside the tolerance."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)
import helpers as H
import oracle_lib as ol
import our_first_climate_model_amd as pkg

FLUX_TOL, T_TOL, TAU_RTOL = 2e-9, 1e-9, 1e-12  # tests/test_gpu_parity.py


def run(ncases=120, seed=1, log=print, big=False):
    """Returns (cases outside the tolerance, worst deviations)."""
    rng = np.random.default_rng(seed)
    olib = ol.load()
    worst = dict(flux=0.0, T=0.0, tau=0.0)
    bad = flips = 0
    t_start = time.time()
    for case in range(ncases):
        mode = "lbl" if rng.random() < 0.6 else "repwvl"
        nangle = int(rng.choice([30, 30, 30, 1, 2, 3, 5, 7, 13, 29, 31, 45, 64]))
        if mode == "lbl" and big and rng.random() < 0.5:
            # few columns x long spectra: many chunk records per column (two-stage chunk sum, k_reduce_partials)
            n_nodes = int(rng.integers(16000, 40000))
            ncol = int(rng.integers(1, 6))
            inp = H.make_inputs(pkg, "lbl", ncol, n_nodes=n_nodes, atm="fpda.lbl.atm", perturb=True)
            nspec = n_nodes - 1
        elif mode == "lbl":
            n_nodes = int(rng.choice([2, 3, 33, 34, 64, 65, 66, 97, 129, int(rng.integers(2, 6000))]))
            ncol = int(rng.integers(1, max(2, min(200, 150000 // max(n_nodes, 1)))))
            inp = H.make_inputs(pkg, "lbl", ncol, n_nodes=n_nodes, atm="fpda.lbl.atm", perturb=bool(rng.random() < 0.8))
            nspec = n_nodes - 1
        else:
            table = str(rng.choice(["Reduced10", "Reduced20", "Reduced100"]))
            ncol = int(rng.integers(1, 120))
            inp = H.make_inputs(pkg, "repwvl", ncol, table=table, perturb=bool(rng.random() < 0.8))
            if rng.random() < 0.5:  # a run of the table's wavelengths: below 8 the flattened kernel hands over to the chunked
                t = dict(inp["table"])  # one; odd counts put the column boundaries of the flattened list at odd lanes
                nw = int(rng.integers(1, (min(13, t["n_wvl"]) if rng.random() < 0.5 else t["n_wvl"]) + 1))
                w0 = int(rng.integers(0, t["n_wvl"] - nw + 1))
                t["n_wvl"] = nw
                t["wvl"], t["weight"] = t["wvl"][w0:w0 + nw].copy(), t["weight"][w0:w0 + nw].copy()
                t["xsec"] = np.ascontiguousarray(t["xsec"][:, :, w0:w0 + nw, :])
                inp["table"] = t
            nspec = inp["table"]["n_wvl"]
        inp["cloud_layer"] = int(rng.integers(-1, 20))
        inp["cloud_tau"] = float(rng.choice([0.0, 1.0, 0.3, 25.0]))
        steps = int(rng.integers(1, 4))
        g = H.gpu_ensemble(pkg, inp, nangle=nangle)
        chunks = 0
        if rng.random() < 0.5:
            chunks = int(rng.integers(1, max(2, (nspec + 31) // 32 + 1)))
            g.set_spectral_chunks(chunks)
        o = H.oracle_ensemble(olib, inp, nangle=nangle)
        percol = rng.random() < 0.3
        if percol:  # per-column forcing: solar heating, cloud layer (-1: none) and optical depth
            sol = inp["solar"] * rng.uniform(0.8, 1.2, ncol)
            cl = rng.integers(-1, 20, ncol).astype(np.int32)
            ct = rng.choice([0.0, 0.5, 1.0, 8.0], ncol)
            g.set_column_forcing(sol, cl, ct), o.set_column_forcing(sol, cl, ct)
        split = steps > 1 and rng.random() < 0.3
        if split:  # the same iterations in two calls
            g.step(1), o.step(1)
            steps -= 1
        g.step(steps), o.step(steps)
        fg, fo = g.get_fluxes(), o.get_fluxes()
        e_flux = max(H.max_abs(fg[k], fo[k]) for k in fg)
        sg, so = g.get_state(), o.get_state()
        e_T = max(H.max_abs(sg["Tlayer"], so["Tlayer"]), H.max_abs(sg["T_surface"], so["T_surface"]))
        col = int(rng.integers(0, ncol))
        tg, to = g.get_tau(col), o.get_tau(col)
        e_tau = float(np.max(np.abs(tg - to) / np.maximum(np.abs(to), 1e-300)))
        # ... and the optical depths as the production kernels form them (clamped to [-10, 1e7] there)
        tp, tc = g.get_tau_production(col), np.clip(to, -10.0, 1e7)
        e_tau = max(e_tau, float(np.max(np.abs(tp - tc) / np.maximum(np.abs(tc), 1e-300))))
        # `time` is a float sum of double steps (main.cpp:359,581): a step that differs in its last bits can flip ONE float
        # rounding, so equality is the rule (3 exceptions in 30 000 cases) and 2 ulp of float per step the requirement
        same_time = bool(np.array_equal(sg["time_h"], so["time_h"]))
        near_time = bool(np.allclose(sg["time_h"], so["time_h"], rtol=2.4e-7 * (steps + split), atol=0.0))
        flips += not same_time
        ok = e_flux <= FLUX_TOL and e_T <= T_TOL and e_tau <= TAU_RTOL and near_time
        bad += not ok
        worst["flux"], worst["T"], worst["tau"] = max(worst["flux"], e_flux), max(worst["T"], e_T), max(worst["tau"], e_tau)
        log(f"{case:4d} {mode:6s} ncol={ncol:4d} nspec={nspec:5d} nangle={nangle:2d} chunks={chunks or 'auto':>5} "
            f"used={g.get_spectral_chunks():4d} cloud=({inp['cloud_layer']:2d},{inp['cloud_tau']:4.1f}){'*' if percol else ' '} steps={steps}{'+1' if split else '  '} "
            f"dFlux={e_flux:.2e} dT={e_T:.2e} dtau={e_tau:.1e} time={'==' if same_time else ('~=' if near_time else '!=')} {'ok' if ok else 'FAIL'}")
        g.close()
    log(f"{ncases} cases in {time.time() - t_start:.0f} s: {bad} outside tolerance, {flips} with a float `time` one rounding apart; worst dFlux {worst['flux']:.2e} W/m^2, "
        f"dT {worst['T']:.2e} K, dtau {worst['tau']:.1e} (relative)")
    return bad, worst


if __name__ == "__main__":
    n_bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                   big=len(sys.argv) > 3 and sys.argv[3] == "big")
    sys.exit(1 if n_bad else 0)
