#!/usr/bin/env python3
"""Phase durations inside the fused repwvl kernel from s_memtime stamps of sampled waves (a library built with
-DRCM_DBG_STAMPS: python tools/build_variants.py stamps:-DRCM_DBG_STAMPS,-DRCM_WITH_VARIANTS).
usage (GPU box): phase_stamps.py [ncol] [fused|flat] [variant name = stamps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import helpers as H
import our_first_climate_model_amd as pkg

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
vname = sys.argv[3] if len(sys.argv) > 3 else "stamps"
L = pkg.load_library(os.path.join(ROOT, "our_first_climate_model_amd", "variants", f"librcm_hip_{vname}.so"))
L.rcm_dbg_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
inp = H.make_inputs(pkg, "repwvl", ncol, table="Reduced100")
g = H.gpu_ensemble(pkg, inp, library=L)
if mode == "flat":
    H.set_step_fusion(g, 0)
    nblocks = (ncol * 100 + 127) // 128
    nsamp = (nblocks + 60) // 61
    buf = np.zeros(nsamp * 64, dtype=np.uint64)
    g.step(5)
    assert L.rcm_dbg_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0  # allocates
    g.step(3)
    assert L.rcm_dbg_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0
    s = buf.reshape(nsamp, 64).astype(np.float64)
    s = s[s[:, 18] > 0]
    print(f"flattened kernel, {ncol} columns, {len(s)} sampled waves; durations in shader-clock cycles (median / p10 / p90)")
    for n, a, b in (("column images + barrier", 16, 17), ("optical depths", 20, 21), ("Planck", 21, 22), ("angles", 22, 23),
                    ("segmented reduction", 23, 18), ("whole wave", 16, 18)):
        dd = s[:, b] - s[:, a]
        print(f"  {n:26s} {np.median(dd):9.0f} {np.percentile(dd, 10):9.0f} {np.percentile(dd, 90):9.0f}")
    sys.exit(0)
H.set_step_fusion(g, 1)
lpc = H.get_step_fusion(g)
nwaves = (ncol + 32 // lpc - 1) // (32 // lpc)
nsamp = (nwaves + 60) // 61
buf = np.zeros(nsamp * 64, dtype=np.uint64)
g.step(5)
assert L.rcm_dbg_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0  # allocates
g.step(20)
assert L.rcm_dbg_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0
s = buf.reshape(nsamp, 64).astype(np.float64)
ok = s[:, 5] > 0
s = s[ok]
tick = 1.0  # s_memtime counts shader-clock cycles here (the whole-step total matches the kernel time at ~2.4 GHz)
names = [("A adjust+feedback", 0, 1), ("B images", 1, 2), ("C RT, all passes", 2, 3), ("D reduce", 3, 4), ("E time step", 4, 5),
         ("whole step", 0, 5)]
print(f"{ncol} columns, {lpc} lane pairs per column, {len(s)} sampled waves; durations in shader-clock cycles (median / p10 / p90)")
for n, a, b in names:
    d = (s[:, b] - s[:, a]) * tick
    print(f"  {n:22s} {np.median(d):8.2f} {np.percentile(d, 10):8.2f} {np.percentile(d, 90):8.2f}")
npass = (100 + lpc - 1) // lpc
prev = s[:, 2]
for i in range(npass):
    cur = s[:, 8 + i]
    d = (cur - prev) * tick
    print(f"  pass {i}                 {np.median(d):8.2f} {np.percentile(d, 10):8.2f} {np.percentile(d, 90):8.2f}")
    prev = cur
