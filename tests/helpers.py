"""Shared set-up for the parity tests: builds the same ensemble on the oracle and on the GPU."""
import os

import numpy as np

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def make_inputs(pkg, mode, ncol, table="Reduced100", n_nodes=1000, perturb=True, atm="test.atm"):
    """Inputs shared by both sides, produced by the product's own host functions."""
    atm_cols, _ = pkg.atm_load(os.path.join(GOLDEN, atm))
    spec = pkg.perturbation_sweep(ncol, perturb_rh=perturb, perturb_ts=perturb)
    cols = pkg.build_columns(atm_cols, spec, mode)
    solar = pkg.default_solar()["solar_irr"]
    inp = dict(mode=mode, ncol=ncol, cols=cols, solar=solar)
    if mode == "repwvl":
        inp["table"] = ol.read_rtab(os.path.join(GOLDEN, f"{table}Forcing.rtab"))
    else:
        inp["wvl"], inp["tabs"] = pkg.synth_lbl_tables(n_nodes, plevel=cols["plevel"])
    return inp


def configure(e, inp):
    """The product's own our_first_climate_model_amd.configure (duck-typed: it also drives the oracle handle)."""
    from our_first_climate_model_amd import configure as _configure
    return _configure(e, inp)


def oracle_ensemble(olib, inp, threads=8, nangle=30):
    e = ol.OracleEnsemble(olib, inp["ncol"], nangle=nangle)
    configure(e, inp)
    e.set_threads(threads)
    return e


def gpu_ensemble(pkg, inp, **kw):
    e = pkg.Ensemble(inp["ncol"], **kw)
    return configure(e, inp)


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


# ---- TEST-SIDE library only (tests/_variants/librcm_hip_variants.so, -DRCM_WITH_VARIANTS) --------------------------
def set_step_fusion(e, mode):
    """The fused whole-step repwvl kernel k_step_repwvl of the variants build: 1 on / 0 or -1 off.  The product
    library has neither the kernel nor the entry point: asking it for mode 1 is an error, 0 / -1 a no-op."""
    if not hasattr(e._L, "rcm_set_step_fusion"):
        if mode == 1:
            from our_first_climate_model_amd import RcmError
            raise RcmError(-1, "the fused repwvl step kernel exists in the test-side build only")
        return
    import ctypes as C
    e._L.rcm_set_step_fusion.argtypes = [C.c_void_p, C.c_int]
    e._ck(e._L.rcm_set_step_fusion(e.h, mode))


def get_step_fusion(e):
    """Lane pairs per column of the fused repwvl kernel in use; 0 = the three-kernel path (always, in the product)."""
    if not hasattr(e._L, "rcm_get_step_fusion"):
        return 0
    import ctypes as C
    e._L.rcm_get_step_fusion.argtypes = [C.c_void_p]
    n = e._L.rcm_get_step_fusion(e.h)
    if n < 0:
        e._ck(n)
    return n
