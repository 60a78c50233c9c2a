import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU oracle checks")


@pytest.fixture(scope="session")
def oracle():
    """Compiled CPU oracle (test infrastructure only)."""
    from oracle import bindings
    bindings.build()
    return bindings


@pytest.fixture(scope="session")
def meso_lib():
    from meso_amd import _lib
    return _lib.load()


DP_RUN = dict(a0=15.0, gamma=4.5, sigma=3.0, expw=1.0, cut=1.0, seed=419084618, skin=0.3, dt=0.005)
