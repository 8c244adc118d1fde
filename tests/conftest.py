import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "slow: longer CPU test (still run by default)")
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def fits():
    return golden("G1_spline_fits.npz")


@pytest.fixture(scope="session")
def rings():
    g = golden("G1_rings.npz")
    return g["ringL"], g["ringR"]


def spline(fits, tag):
    return fits[f"{tag}_t"], fits[f"{tag}_cx"], fits[f"{tag}_cy"], int(fits[f"{tag}_k"]), float(fits[f"{tag}_length"])
