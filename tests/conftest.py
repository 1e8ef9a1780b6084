import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def olib():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def rcm():
    """The product package with its HIP library built (hipcc cross-compiles without a GPU)."""
    from our_first_climate_model_amd import build as _b
    _b.build()
    import our_first_climate_model_amd as pkg
    pkg.lib()
    return pkg


@pytest.fixture(scope="session")
def variants(rcm):
    """The TEST-SIDE build with the one-lane-per-point reference kernels (-DRCM_WITH_VARIANTS), loaded beside
    the product; kernel selection by RCM_EXP_VARIANT / RCM_CUBE / RCM_PLANCK_FAST / RCM_PAIR / RCM_FLAT works
    in this library only."""
    from our_first_climate_model_amd import build as _b
    return rcm.load_library(_b.build_variants())


@pytest.fixture(scope="session")
def probe(golden):
    import json
    with open(os.path.join(golden, "survey_probe.json")) as f:
        return json.load(f)
