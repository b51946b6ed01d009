import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def gpu():
    """The product library on a box with a usable device; fails loudly (no skip, no fallback)."""
    import cbird_amd

    cbird_amd.require_device()
    return cbird_amd


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name))


def golden_cases():
    return ["vptree_n4096.npz", "vptree_n32768.npz"]
