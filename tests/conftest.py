import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def weights_mod(pkg):
    import importlib
    return importlib.import_module("asr_2pass_amd.weights")


def synth_pcm(index, n, rng):
    """SURVEY.md §8d synthetic utterance (same recipe as bench.py)."""
    import numpy as np
    t = np.arange(n, dtype=np.float64) / 16000.0
    f = 110.0 * 2.0 ** ((index % 24) / 12.0)
    x = 8000.0 * (0.6 * np.sin(2 * np.pi * f * t) + 0.4 * rng.standard_normal(n))
    return (np.clip(np.round(x), -32768, 32767) / 32768.0).astype(np.float32)
