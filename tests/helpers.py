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
