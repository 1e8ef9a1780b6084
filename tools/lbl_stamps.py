#!/usr/bin/env python3
"""Phase durations inside k_rt_lbl_pair from s_memtime stamps of sampled waves (wave 0 of every 61st block), for the
few-column launch geometry (BASELINE configs[1]: one column x the full spectrum).  Needs a library built with
-DRCM_DBG_STAMPS:  python tools/build_variants.py stamps:-DRCM_DBG_STAMPS
usage (GPU box): lbl_stamps.py [ncol=1] [chunks=0] [variant=stamps]"""
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

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 1
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
vname = sys.argv[3] if len(sys.argv) > 3 else "stamps"
L = pkg.load_library(os.path.join(ROOT, "our_first_climate_model_amd", "variants", f"librcm_hip_{vname}.so"))
L.rcm_dbg_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
inp = H.make_inputs(pkg, "lbl", ncol, n_nodes=131072, atm="fpda.lbl.atm")
g = H.gpu_ensemble(pkg, inp, library=L)
g.set_spectral_chunks(chunks)
nch = g.get_spectral_chunks()
nblocks = (ncol * nch + 3) // 4
nsamp = (nblocks + 60) // 61
buf = np.zeros(nsamp * 64, dtype=np.uint64)
g.step(5)
assert L.rcm_dbg_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0  # allocates
g.step(8)
g.sync()
g.set_rt_timing(True)
g.step(20)
rt_ms, n = g.get_rt_timing()
assert L.rcm_dbg_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), buf.size) == 0
s = buf.reshape(nsamp, 64).astype(np.float64)
s = s[s[:, 63] > 0]
npass = int(np.ceil(np.ceil(131071 / 32) / nch))
print(f"{ncol} column(s), {nch} chunks = {nblocks} blocks, {npass} pass(es) per wave, {len(s)} sampled waves; RT launch "
      f"{rt_ms / n * 1e3:.1f} us; s_memtime ticks (median / p10 / p90); kernel span first start -> last end: "
      f"{s[:, 63].max() - s[:, 0].min():.0f}")


def show(name, d):
    print(f"  {name:30s} {np.median(d):9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 90):9.0f}")


show("start offset in the launch", s[:, 0] - s[:, 0].min())
show("column image + barrier", s[:, 1] - s[:, 0])
prev = s[:, 1]
for j in range(min(npass, 19)):
    show(f"pass {j}: optical depths", s[:, 2 + 3 * j] - prev)
    show(f"pass {j}: Planck", s[:, 3 + 3 * j] - s[:, 2 + 3 * j])
    show(f"pass {j}: angles", s[:, 4 + 3 * j] - s[:, 3 + 3 * j])
    prev = s[:, 4 + 3 * j]
show("reduce + store", s[:, 63] - prev)
show("whole wave", s[:, 63] - s[:, 0])
