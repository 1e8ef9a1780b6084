#!/usr/bin/env python3
"""what the hand-over of REAL lbl.*.asc spectra costs, at the size the synthetic stand-in has.
Writes lbl.{co2,h2o,o3,ch4,n2o}.asc -- 131 072 rows x (wavelength + 20 layers), lbl.arts/README:5-9 -- from the seeded
synthetic tables into a scratch directory, then times, on this host:
  * the product's ASCII_file2xy2D (rcm_ascii_file2xy2D, lbl.arts/ascii.cpp:1631-1691 semantics), per file;
  * the reference's own ASCII_file2xy2D compiled from where it lies (oracle/_ref/libref_ascii.so), when it is built;
  * rcm_lbl_cache_write / rcm_lbl_cache_read of the binary cache;
  * with a GPU: `rcm_main --lbl-dir D --lbl-cache F --steps 0` twice (parse + write the cache; read the cache).
usage: time_lbl_loader.py [scratch_dir=/tmp/rcm_lbl_asc] [n_nodes=131072]"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import our_first_climate_model_amd as rcm  # noqa: E402

d = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rcm_lbl_asc"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
os.makedirs(d, exist_ok=True)
atm, _ = rcm.atm_load(os.path.join(ROOT, "tests", "golden", "fpda.lbl.atm"))
wvl, tabs = rcm.synth_lbl_tables(n, plevel=rcm.column_init(atm)["plevel"])
names = ["co2", "h2o", "o3", "ch4", "n2o"]
t0 = time.perf_counter()
for nm, t in zip(names, tabs):
    with open(os.path.join(d, f"lbl.{nm}.asc"), "w") as f:
        f.write("% wavelength [nm], then the optical depth of the 20 layers, top first (synthetic stand-in)\n")
        np.savetxt(f, np.column_stack([wvl, t]), fmt="%.10e")
size = sum(os.path.getsize(os.path.join(d, f"lbl.{nm}.asc")) for nm in names)
print(f"wrote 5 files, {n} rows x 21 columns, {size / 1e6:.0f} MB of ASCII in {time.perf_counter() - t0:.1f} s (numpy.savetxt)")

t0 = time.perf_counter()
got = []
for nm in names:
    t1 = time.perf_counter()
    st, x, y = rcm.ascii_file2xy2D(os.path.join(d, f"lbl.{nm}.asc"))
    assert st == 0 and x.shape == (n,) and y.shape == (n, 20)
    got.append(y)
    print(f"  rcm_ascii_file2xy2D lbl.{nm}.asc: {time.perf_counter() - t1:.2f} s")
t_parse = time.perf_counter() - t0
print(f"product loader, five files: {t_parse:.2f} s = {size / 1e6 / t_parse:.0f} MB/s")
ref = os.path.join(ROOT, "oracle", "_ref", "libref_ascii.so")
if os.path.exists(ref):
    L = C.CDLL(ref)
    f = getattr(L, "ASCII_file2xy2D", None)
    if f is not None:
        dp = C.POINTER(C.c_double)
        f.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(dp), C.POINTER(C.POINTER(dp))]
        t1 = time.perf_counter()
        nx, ny, x, y = C.c_int(0), C.c_int(0), dp(), C.POINTER(dp)()
        st = f(os.path.join(d, "lbl.co2.asc").encode(), C.byref(nx), C.byref(ny), C.byref(x), C.byref(y))
        dt = time.perf_counter() - t1
        print(f"reference ASCII_file2xy2D (compiled ascii.cpp) on lbl.co2.asc: status {st}, {nx.value} x {ny.value}, {dt:.2f} s "
              f"(five files: ~{5 * dt:.1f} s)")
cache = os.path.join(d, "lbl.cache")
t1 = time.perf_counter()
rcm.lbl_cache_write(cache, wvl, got)
t_w = time.perf_counter() - t1
t1 = time.perf_counter()
w2, t2 = rcm.lbl_cache_read(cache)
t_r = time.perf_counter() - t1
assert np.array_equal(w2, wvl) and all(np.array_equal(a, b) for a, b in zip(t2, got))
print(f"binary cache {os.path.getsize(cache) / 1e6:.0f} MB: write {t_w:.2f} s, read {t_r:.2f} s; values identical to the parse")
# the ASCII round trip loses digits (%.10e): the cache holds what the parser read, as the reference's reader would
print(f"max relative deviation of the parsed tables from the generator's doubles: "
      f"{max(float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) for a, b in zip(got, tabs)):.1e}")
import torch  # noqa: E402
if torch.cuda.is_available():
    exe = os.path.join(ROOT, "our_first_climate_model_amd", "rcm_main")
    os.remove(cache)
    for what in ("parse the five ASCII files + write the cache", "read the cache"):
        t1 = time.perf_counter()
        r = subprocess.run([exe, "--atm", os.path.join(ROOT, "tests", "golden", "fpda.lbl.atm"), "--lbl-dir", d, "--lbl-cache",
                            cache, "--steps", "0", "--output", os.path.join(d, "output.txt")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        print(f"rcm_main, whole process, {what}: {time.perf_counter() - t1:.2f} s")
