import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def g_synth():
    return load_golden("synth_rdf.npz")


@pytest.fixture(scope="session")
def g_c1():
    return load_golden("c1_rdf.npz")


@pytest.fixture(scope="session")
def g_small():
    return load_golden("small_md.npz")


@pytest.fixture(scope="session")
def g_acf():
    return load_golden("acf.npz")


def sorted_frame(frame, id_col=0):
    """Rows of one dump table ordered by ascending id (rdf_cn.py:192, diffusion.py:176)."""
    return frame[np.argsort(frame[:, id_col], kind="stable")]
