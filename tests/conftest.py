import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def manifest(golden_dir):
    import json
    with open(os.path.join(golden_dir, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def state_dict():
    from efficientsam3_amd import schema
    return schema.synthetic_state_dict("efficientvit", "b1", seed=0)
