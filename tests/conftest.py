import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def relerr(a, b):
    """Relative max-norm difference."""
    a = np.asarray(a); b = np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
