import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def sign_align(V, Vref):
    """Principal components are defined up to a per-row sign."""
    V2 = V.reshape(V.shape[0], -1)
    R2 = Vref.reshape(Vref.shape[0], -1)
    s = np.sign(np.sum(V2 * R2, axis=1))
    s[s == 0] = 1
    return (V2 * s[:, None]).reshape(V.shape)
