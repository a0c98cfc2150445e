import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)      # tests/host_models.py

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout, when installed)")


def pytest_collection_modifyitems(config, items):
    """A hung GPU test must fail, not stall the whole run: default per-test limit (inert without pytest-timeout)."""
    for it in items:
        if it.get_closest_marker("gpu") is not None and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(480))


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def relerr(a, b):
    """The norm every stated tolerance of this suite is in: max |a - b| / max |b| (error relative to the LARGEST entry of the
    reference array).  Small entries may be off by more in relative terms: see relerr_elementwise."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    den = max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b))) / den


def relerr_elementwise(a, b, floor=1e-3):
    """max_ij |a_ij - b_ij| / |b_ij| over the entries with |b_ij| >= floor * max |b| (entries nearer to zero have no meaningful
    relative error), and the share of the entries that covers.  Reported beside `relerr` at the full sizes."""
    a = np.asarray(a, dtype=float).reshape(-1)
    b = np.asarray(b, dtype=float).reshape(-1)
    big = np.abs(b) >= floor * max(float(np.max(np.abs(b))), 1e-300)
    if not big.any():
        return 0.0, 0.0
    return float(np.max(np.abs(a[big] - b[big]) / np.abs(b[big]))), float(big.mean())
