#!/usr/bin/env python3
"""Times every our_first_climate_model_amd/variants/librcm_hip_*.so with tools/sweep_variants.py
(one subprocess per library, same box, back to back).  usage: run_variants.py [ncol] [nodes] [rounds]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:] or ["2048", "131072", "3"]
for so in sorted(glob.glob(os.path.join(ROOT, "our_first_climate_model_amd", "variants", "librcm_hip_*.so"))):
    env = dict(os.environ, RCM_LIB_PATH=so)
    env.setdefault("SWEEP", "1:0:0:1,1:1:0:1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_variants.py")] + args, env=env,
                       capture_output=True, text=True)
    name = os.path.basename(so)[len("librcm_hip_"):-3]
    for ln in r.stdout.splitlines():
        if "median" in ln:
            print(f"{name:24s} {ln}", flush=True)
    if r.returncode != 0:
        print(f"{name}: FAILED\n{r.stderr[-1500:]}", flush=True)
