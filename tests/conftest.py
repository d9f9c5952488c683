import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_dataset(name):
    d = np.load(os.path.join(DATA, name + ".npz"))
    return d["n_node"], d["n_edge"], d["senders"], d["receivers"]


@pytest.fixture(scope="session")
def grid_small():
    return load_dataset("grid_small")


@pytest.fixture(scope="session")
def community_medium():
    return load_dataset("community_medium")
