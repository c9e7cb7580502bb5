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


def assert_ids_match(got_ids, ref, tie_gap=1e-4):
    """Greedy ids against the oracle's: identical, except on rows where the oracle's own two best log-probs are closer than
    `tie_gap` (far below north_star's 1e-3 log-prob tolerance, and of the size of fp32 summation-order noise through 66 residual
    blocks): there the runner-up is accepted.  Token-for-token agreement cannot mean more than that between two fp32 pipelines
    that sum in different orders."""
    import numpy as np
    got, want = [int(x) for x in got_ids], [int(x) for x in ref["ids"]]
    assert len(got) == len(want), (len(got), len(want))
    lp = np.asarray(ref["logp"])
    for r, (a, b) in enumerate(zip(got, want)):
        if a == b:
            continue
        order = np.argsort(lp[r])
        gap = float(lp[r][order[-1]] - lp[r][order[-2]])
        assert gap < tie_gap and a == int(order[-2]), f"row {r}: got {a}, oracle {b} (top-2 gap {gap:.2e})"
