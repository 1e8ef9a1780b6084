"""MI355X-native radiative-convective column ensemble.

Host-side Python mirror of the C ABI in include/rcm.h (librcm_hip.so: hand-written gfx950 HIP
kernels).  There is no CPU fallback: importing works without a GPU (so that the ABI can be
inspected), creating an Ensemble does not.
"""
from .api import (Ensemble, RcmError, configure, load_library, measure_hbm_bandwidth, ascii_file2xy2D, atm_load, column_init, default_solar, lib, lbl_cache_read,
                  lbl_cache_write, read_rtab, solar_setup, LIB_PATH)
from .setup_ensemble import EnsembleSpec, build_columns, synth_lbl_tables, perturbation_sweep

__all__ = ["Ensemble", "RcmError", "configure", "load_library", "measure_hbm_bandwidth", "ascii_file2xy2D", "atm_load", "column_init", "default_solar", "lib",
           "read_rtab", "solar_setup", "lbl_cache_read", "lbl_cache_write", "LIB_PATH", "EnsembleSpec", "build_columns", "synth_lbl_tables",
           "perturbation_sweep"]
