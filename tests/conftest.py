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


def golden(name):
    """Load a committed fixture (data only; numpy's safe loader)."""
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def sd_default():
    from speechseparation_amd import weights
    return weights.synth_state_dict(None, seed=0)


@pytest.fixture(scope="session")
def sd_hot():
    from speechseparation_amd import weights
    return weights.synth_state_dict(None, seed=1, lstm_gain=3.0)
