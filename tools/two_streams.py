#!/usr/bin/env python3
"""Does the repwvl step gain from running the ensemble as independent parts on their own HIP streams (columns
never interact, so RT of one part can overlap the latency-bound end-of-iteration kernel and the tail of the other)?
One handle of N columns against P handles of N/P columns on P streams, same process, interleaved rounds.
usage (GPU box): python tools/two_streams.py [ncol] [steps] [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H
import our_first_climate_model_amd as pkg

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4


def part(inp, lo, hi):
    out = dict(inp)
    out["ncol"] = hi - lo
    out["cols"] = {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == inp["ncol"] else v)
                   for k, v in inp["cols"].items()}
    return out


inp = H.make_inputs(pkg, "repwvl", ncol, table="Reduced100")
cases = {}
for P in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(P)]
    ens = []
    for i, s in enumerate(streams):
        lo, hi = ncol * i // P, ncol * (i + 1) // P
        e = pkg.configure(pkg.Ensemble(hi - lo, stream=s.cuda_stream), part(inp, lo, hi))
        e.step(5)
        ens.append(e)
    cases[P] = ens
torch.cuda.synchronize()
res = {P: [] for P in cases}
for r in range(rounds):
    for P, ens in cases.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e in ens:
            e.step(steps)
        torch.cuda.synchronize()
        res[P].append((time.perf_counter() - t0) / steps * 1e3)
for P, t in res.items():
    print(f"{P} part(s): median {np.median(t):.4f} ms/step  min {min(t):.4f}  ({ncol / np.median(t) / 1e3:.2f} M column-timesteps/s)", flush=True)
