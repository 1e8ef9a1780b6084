#!/usr/bin/env python3
"""Puts the eight one-GPU legs of tools/eq100k_shards.sh together: BASELINE configs[4] (100 000 columns x full lbl
spectrum to equilibrium, 8 ranks) rehearsed on one GPU, rank block by rank block, no collective.
Reads gpurun_out/eq100k_shard{0..7}.{json,log,npy}; writes profiles/r02_equilibrium_100000_shards.json."""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
legs, states, hist = [], [], []
for r in range(8):
    j = json.loads(open(os.path.join(O, f"eq100k_shard{r}.json")).read().strip().splitlines()[-1])
    s = np.load(os.path.join(O, f"eq100k_shard{r}.npy"))
    h = {}
    for ln in open(os.path.join(O, f"eq100k_shard{r}.log")):
        m = re.match(r"\[equilibrium\] (\d+) iterations, worst residual ([0-9.eE+-]+|nan|inf) K/iteration", ln)
        if m:
            h[int(m.group(1))] = float(m.group(2))
    legs.append(j), states.append(s), hist.append(h)
full = np.concatenate(states)
assert full.shape == (100000, 21), full.shape
stop = max(j["steps"] for j in legs)  # the all-reduce(max) of the residual makes every rank stop when the slowest block has
# residual of every block at the checks up to its own stop (10 iterations apart); the blocks that stopped earlier stay
# below the tolerance (the residual falls monotonically in every log)
mono = all(all(h[a] >= h[b] for a, b in zip(sorted(h)[5:], sorted(h)[6:])) for h in hist)
out = {
    "what": "BASELINE configs[4] rehearsed on ONE MI355X: the 8 rank blocks (12 500 columns each) of the 100 000-column "
            "ensemble x 131 071 lbl bands, each integrated alone to its own equilibrium (tol 0.01 K/iteration on the worst "
            "column), one after another (examples/equilibrium_ensemble.py --shard R/8); no collective ran",
    "legs": [{"rank": r, "steps": j["steps"], "converged": j["converged"], "residual_K_per_step": j["residual_K_per_step"],
              "seconds": j["seconds"], "column_timesteps_per_sec": j["column_timesteps_per_sec"],
              "T_surface_K": j["T_surface_K"]} for r, j in enumerate(legs)],
    "all_converged": all(j["converged"] for j in legs),
    "all_finite": bool(np.isfinite(full).all()),
    "global_stop_iteration_of_an_8_rank_run": stop,
    "residual_monotone_after_50_iterations_in_every_leg": bool(mono),
    "T_surface_K_all_columns": {"min": float(full[:, 20].min()), "median": float(np.median(full[:, 20])),
                                "max": float(full[:, 20].max())},
    "seconds_total_one_gpu": sum(j["seconds"] for j in legs),
    "column_timesteps_total": sum(12500 * j["steps"] for j in legs),
    "seconds_slowest_leg": max(j["seconds"] for j in legs),
}
out["column_timesteps_per_sec_one_gpu"] = out["column_timesteps_total"] / out["seconds_total_one_gpu"]
p = os.path.join(ROOT, "profiles", "r02_equilibrium_100000_shards.json")
json.dump(out, open(p, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "legs"}, indent=1))
for l in out["legs"]:
    print(l["rank"], l["steps"], l["converged"], f"{l['seconds']:.0f} s", l["T_surface_K"])
