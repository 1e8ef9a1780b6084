#!/usr/bin/env python3
"""One column x full lbl spectrum, 60 steps: target for `rocprofv3 --kernel-trace --stats` (per-kernel times of configs[1])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import helpers as H
import our_first_climate_model_amd as pkg
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = H.gpu_ensemble(pkg, H.make_inputs(pkg, "lbl", ncol, n_nodes=131072, atm="fpda.lbl.atm"))
g.step(60); g.sync()
