#!/usr/bin/env python3
"""Builds experimental variants of librcm_hip.so (compile-time knobs of rcm_kernels.hpp) into
our_first_climate_model_amd/variants/ for A/B runs on the GPU:
    python tools/build_variants.py name:-DRCM_PAIR_LB=2,-DRCM_EXP_DEG=8 name2:...
and tools/run_variants.py times each with tools/sweep_variants.py."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "our_first_climate_model_amd")
OUT = os.path.join(PKG, "variants")
os.makedirs(OUT, exist_ok=True)


def build(spec):
    name, _, flags = spec.partition(":")
    flags = [f for f in flags.split(",") if f]
    so = os.path.join(OUT, f"librcm_hip_{name}.so")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-shared", "-o", so,
           os.path.join(PKG, "csrc", "rcm_capi.hip"), os.path.join(PKG, "csrc", "rcm_host.cpp"),
           os.path.join(PKG, "csrc", "rcm_nc.cpp"),
           "-Rpass-analysis=kernel-resource-usage"] + flags
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return name, "BUILD FAILED\n" + r.stderr[-2000:]
    info = []
    lines = r.stderr.splitlines()
    for i, ln in enumerate(lines):
        if any(k in ln for k in ("Function Name: _ZN3rcm13k_rt_lbl_pairILi1", "Function Name: _ZN3rcm16k_rt_repwvl_flatILi1",
                                 "Function Name: _ZN3rcm13k_step_repwvlILi1ELi16")):
            blk = " ".join(x.split("remark:")[1].split("[-R")[0].strip() for x in lines[i:i + 12]
                           if any(k in x for k in ("Function Name", " VGPRs:", "AGPRs", "ScratchSize", "VGPRs Spill")))
            info.append(blk)
    return name, "\n   ".join(info)


if __name__ == "__main__":
    with ThreadPoolExecutor(4) as ex:
        for name, info in ex.map(build, sys.argv[1:]):
            print(f"{name}:\n   {info}")
