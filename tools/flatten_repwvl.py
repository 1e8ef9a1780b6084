#!/usr/bin/env python3
"""Flatten a repwvl ``Reduced*Forcing.nc`` lookup table (NetCDF-4 = HDF5) into the
raw little-endian ``.rtab`` blob consumed by ``rcm_load_repwvl_rtab`` / the oracle.

The reference reads these datasets through netcdf-cxx4
(/root/reference/repwvl_V2.01_cpp/repwvl_thermal.cpp:113-176, :208); neither
netcdf nor h5py exists in this image or on the GPU box, so the conversion is
done once, here, with the image's own ``h5dump`` and the blobs are committed.

.rtab layout (all little endian, no padding):
    char   magic[8]            = "RWVLTAB1"
    int64  n_tpert, n_species, n_wvl, n_p
    f64    p_grid[n_p]         [Pa], descending
    f64    t_ref[n_p]          [K]
    f64    t_pert[n_tpert]     [K]
    f64    vmrs_ref[n_species][n_p]   (read by the reference, unused)
    f64    wvl[n_wvl]          [nm]   (ChosenWvls)
    f64    weight[n_wvl]              (ChosenWeights)
    f64    xsec[n_tpert][n_species][n_wvl][n_p]   (file order, unchanged)

usage: flatten_repwvl.py IN.nc OUT.rtab
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

H5DUMP = os.environ.get("H5DUMP", "/opt/conda/bin/h5dump")


def dump(nc, name):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "d.bin")
        subprocess.run([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", out, nc],
                       check=True, stdout=subprocess.DEVNULL)
        return np.fromfile(out, dtype="<f8")


def main(nc, out):
    p_grid = dump(nc, "p_grid")
    t_ref = dump(nc, "t_ref")
    t_pert = dump(nc, "t_pert")
    vmrs_ref = dump(nc, "vmrs_ref")
    wvl = dump(nc, "ChosenWvls")
    weight = dump(nc, "ChosenWeights")
    xsec = dump(nc, "xsec")
    n_p, n_t, n_w = p_grid.size, t_pert.size, wvl.size
    n_s = vmrs_ref.size // n_p
    assert t_ref.size == n_p and weight.size == n_w
    assert xsec.size == n_t * n_s * n_w * n_p, (xsec.size, n_t, n_s, n_w, n_p)
    with open(out, "wb") as f:
        f.write(b"RWVLTAB1")
        np.array([n_t, n_s, n_w, n_p], dtype="<i8").tofile(f)
        for a in (p_grid, t_ref, t_pert, vmrs_ref, wvl, weight, xsec):
            a.astype("<f8").tofile(f)
    print(f"{out}: n_tpert={n_t} n_species={n_s} n_wvl={n_w} n_p={n_p} "
          f"({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
