"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` tests: oracle vs golden vectors, host logic, C-ABI symbol table, gloo
multi-process tests.  `-m gpu` tests: the HIP engine through the C-ABI vs the oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    """Load a committed golden fixture (plain arrays only, no pickles)."""
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


@pytest.fixture(scope="session")
def gold():
    return golden
