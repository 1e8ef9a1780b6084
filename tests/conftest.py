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
def probe(golden):
    import json
    with open(os.path.join(golden, "survey_probe.json")) as f:
        return json.load(f)
