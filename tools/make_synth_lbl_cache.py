#!/usr/bin/env python3
"""Writes the seeded synthetic lbl spectra (setup_ensemble.synth_lbl_tables) as the binary cache that
`rcm_main --lbl-cache FILE` reads, so the headline workload can be run from the C++ driver alone:

    python tools/make_synth_lbl_cache.py synth.lbl [n_nodes]
    ./our_first_climate_model_amd/rcm_main --atm tests/golden/fpda.lbl.atm --lbl-cache synth.lbl --ncol 10000 --steps 9
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import our_first_climate_model_amd as rcm  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
atm, _ = rcm.atm_load(os.path.join(ROOT, "tests", "golden", "fpda.lbl.atm"))
plevel = rcm.column_init(atm)["plevel"]
wvl, tabs = rcm.synth_lbl_tables(n, plevel=plevel)
rcm.lbl_cache_write(out, wvl, tabs)
print(f"{out}: {n} nodes, {os.path.getsize(out) / 1e6:.1f} MB")
