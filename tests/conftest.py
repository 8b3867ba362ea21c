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
    # the oracle's multi-colour sweeps (order=1) on several host threads: bit for bit the single-thread result
    # (test_oracle_kernels.py::test_colour_sweep_threads_are_bit_identical); the full-size checks then take seconds
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    orc.set_threads(min(32, max(1, cores)))
    return orc


def assert_norms_close(got, ref, rtol=2e-9, floor=1e-14, strict_rtol=1e-10, strict_above=1e-5):
    """Per-cycle residual norms.

    * Every cycle whose residual is still above ``strict_above`` x the source norm (``ref[0]``) must agree to
      the north star's ``strict_rtol`` = 1e-10 RELATIVE TO ITSELF (measured: <= 1e-12 there).
    * Later cycles: relative agreement ``rtol`` plus ``floor`` relative to the source norm -- a residual
      ||s - A e|| that has dropped to 1e-7 ||s|| carries the cancellation error of the subtraction, which is
      relative to ||s||, not to itself (2e-9 of such a norm is 2e-16 of the source norm)."""
    got = np.asarray(got, dtype=float)
    ref = np.asarray(ref, dtype=float)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    tol = rtol * np.abs(ref) + floor * abs(ref[0])
    strict = np.abs(ref) > strict_above * abs(ref[0])
    tol[strict] = strict_rtol * np.abs(ref[strict]) + 1e-16 * abs(ref[0])
    bad = np.abs(got - ref) > tol
    assert not bad.any(), (got[bad], ref[bad], (np.abs(got - ref) / np.abs(ref))[bad])


@pytest.fixture
def lab():
    """The LAB build of the library (-DEMG3D_LAB: superseded kernel variants + one environment variable per tuning knob)
    for the duration of a test; the product library afterwards."""
    from emg3d_amd import _lib
    prev = _lib.use(_lib.LAB_PATH)
    yield _lib
    _lib.use(prev)
