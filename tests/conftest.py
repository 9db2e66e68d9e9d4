import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The fp32 CPU oracle is the checker of most GPU tests.  torch takes one thread per physical core (128 on the 256-core GPU hosts) and
    # the oracle's convolutions / GEMMs run 3.5 x SLOWER that way than on 32 threads (efficientvit-b1 distribution test: 51 s at 128 threads,
    # 27 s at 64, 15 s at 32 -- profiles/r06/host_threads_oracle.txt); the suite's wall time is mostly oracle time.
    try:
        import torch
        if torch.get_num_threads() > 32 and not os.environ.get("OMP_NUM_THREADS"):
            torch.set_num_threads(32)
    except Exception:  # noqa: BLE001  (torch missing: nothing to configure)
        pass


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
