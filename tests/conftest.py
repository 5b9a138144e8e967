import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box: pytest -m gpu)")


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(ROOT, "tests", "golden")
    return {name: np.load(os.path.join(d, name + ".npz")) for name in ("nodes", "spectrum", "k7_regression", "k7_golden", "wsola_golden", "swr_golden")}


@pytest.fixture(scope="session")
def nae():
    import naeload
    return naeload.load()


@pytest.fixture(scope="session")
def ctx(nae):
    """GPU context through the C ABI.  No fallback: a missing library or device is an error, not a skip."""
    c = nae.Context(0)
    yield c
    c.close()


def rel_rms(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))
