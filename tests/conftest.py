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


def join_ranks(threads, errs=None, deadline=300.0):
    """Wait for the rank threads of an in-process (LOCAL transport) run with ONE overall deadline.  A rank that failed leaves
    its peers waiting in the transport's barrier for good: as soon as `errs` (list, or sequence with None for "no error") holds
    an entry the wait ends after a short grace period instead of running into the deadline once per thread."""
    import time
    t_end = time.time() + deadline
    t_err = None
    while any(t.is_alive() for t in threads) and time.time() < t_end:
        for t in threads:
            t.join(0.05)
        failed = errs is not None and any(e for e in errs)
        if failed and t_err is None:
            t_err = time.time()
        if t_err is not None and time.time() - t_err > 3.0:
            break


DP_RUN = dict(a0=15.0, gamma=4.5, sigma=3.0, expw=1.0, cut=1.0, seed=419084618, skin=0.3, dt=0.005)
