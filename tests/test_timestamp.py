"""CPU (host logic): TimestampOnnx (csrc/host/timestamp.cpp through the C ABI) against the oracle restatement of
onnxruntime/src/util.cpp:838-963 and hand-derived cases."""
import numpy as np
import pytest

from oracle import timestamp as T

RATE = 10.0 * 6 / 1000 / 3          # util.cpp:851: 20 ms per upsampled frame


def peaks_at(n, idx):
    p = np.zeros(n, np.float32)
    p[list(idx)] = 1.0
    return p


def test_exact_peak_count(pkg):
    """3 tokens -> 4 peaks at 10, 30, 60, 100: token i spans peak i..i+1, offset -1.5 frames, leading/trailing <sil>."""
    n = 150
    p = peaks_at(n, [10, 30, 60, 100])
    a = np.full(n, 4 / n, np.float32)
    got = pkg.timestamp_onnx(a, p, 3)
    ref = T.timestamp_onnx(a, p, 3)
    assert got == pytest.approx(ref)
    toks = [s for s in got if not s[2]]
    assert len(toks) == 3
    assert toks[0][0] == pytest.approx(8.5 * RATE, rel=1e-5) and toks[0][1] == pytest.approx(28.5 * RATE, rel=1e-5)
    assert got[0][2] and got[0][0] == 0.0                        # leading <sil> (first peak > 5 frames in)
    assert got[-1][2] and got[-1][1] == pytest.approx(n * RATE)   # trailing <sil>
    assert toks[-1][1] == pytest.approx((n + 98.5) / 2 * RATE, rel=1e-5)


def test_peak_count_mismatch_rebuilds_from_alphas(pkg):
    n = 120
    p = peaks_at(n, [20, 50])                                   # 2 peaks but 4 tokens -> rebuild (:872-904)
    rng = np.random.default_rng(0)
    a = rng.uniform(0.0, 0.1, n).astype(np.float32)
    got = pkg.timestamp_onnx(a, p, 4)
    ref = T.timestamp_onnx(a, p, 4)
    assert got == pytest.approx(ref)
    assert len([s for s in got if not s[2]]) == 4


def test_long_gap_is_split_into_token_and_sil(pkg):
    n = 200
    p = peaks_at(n, [5, 100, 120])
    a = np.full(n, 3 / n, np.float32)
    got = pkg.timestamp_onnx(a, p, 2)
    assert got == pytest.approx(T.timestamp_onnx(a, p, 2))
    assert [s[2] for s in got[:3]] == [False, True, False]       # first token cut at 30 frames, then <sil>
    assert got[0][1] - got[0][0] == pytest.approx(30 * RATE, rel=1e-4)


def test_begin_time_offset_and_degenerate(pkg):
    n = 60
    p = peaks_at(n, [3, 30, 58])
    a = np.full(n, 0.05, np.float32)
    g0 = pkg.timestamp_onnx(a, p, 2)
    g1 = pkg.timestamp_onnx(a, p, 2, begin_time=1500.0)
    assert [s[0] + 1.5 for s in g0] == pytest.approx([s[0] for s in g1], rel=1e-5)
    assert pkg.timestamp_onnx(a, p, 0) == [] == T.timestamp_onnx(a, p, 0)
    assert pkg.timestamp_onnx(np.zeros(n, np.float32), np.zeros(n, np.float32), 2) == []     # scale == 0 (:875-877)


def test_random_against_oracle(pkg):
    rng = np.random.default_rng(1)
    for _ in range(20):
        n = int(rng.integers(30, 400))
        k = int(rng.integers(1, 12))
        a = rng.uniform(0, 2.0 * (k + 1) / n, n).astype(np.float32)
        idx = sorted(rng.choice(n, size=int(rng.integers(0, k + 3)), replace=False))
        p = peaks_at(n, idx)
        assert pkg.timestamp_onnx(a, p, k) == pytest.approx(T.timestamp_onnx(a, p, k), rel=1e-5, abs=1e-6)
