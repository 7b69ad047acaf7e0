import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.fixture(scope="session")
def ctx():
    """One libilsx context on cuda:0 for the whole GPU session."""
    import ilswiss_amd
    c = ilswiss_amd.Context(0, seed=1234)
    yield c
    c.close()
