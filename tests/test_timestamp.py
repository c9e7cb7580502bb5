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


# ---- PostProcess (util.cpp:720-836): text + stamps assembly --------------------------------------------------------------
def _post_process_abi(pkg, chars, stamps):
    import ctypes
    lib = pkg.load_lib()
    n = len(chars)
    arr = (ctypes.c_char_p * max(n, 1))(*[c.encode("utf-8") for c in chars])
    st = np.ascontiguousarray(np.asarray(stamps, np.float32).reshape(-1))
    out = ctypes.create_string_buffer(4096)
    n_out = ctypes.c_int(0)
    lib.pfhip_post_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    rc = lib.pfhip_post_process(arr, st.ctypes.data if n else None, n, out, 4096, ctypes.byref(n_out))
    assert rc == 0
    return out.value.decode("utf-8")


def test_post_process_known_answers(pkg):
    from oracle import timestamp as T
    cases = [
        # CJK characters: no spaces, one stamp each; special tokens dropped
        (["<s>", "你", "好", "</s>"], [[0, 0.1], [0.1, 0.3], [0.3, 0.5], [0.5, 0.6]], "你好 | 0.100000, 0.300000,0.300000, 0.500000"),
        # BPE pieces: "hel@@" + "lo" -> "hello", stamp from the first piece's begin to the last piece's end; words spaced
        (["hel@@", "lo", "wor@@", "ld"], [[0.0, 0.2], [0.2, 0.4], [0.5, 0.7], [0.7, 1.0]], "hello world | 0.000000, 0.400000,0.500000, 1.000000"),
        # Latin after CJK: no space in between; CJK after Latin resets the spacing
        (["我", "love", "you", "们"], [[0, 1], [1, 2], [2, 3], [3, 4]], "我love you们 | 0.000000, 1.000000,1.000000, 2.000000,2.000000, 3.000000,3.000000, 4.000000"),
        # a piece ending in "@@" right before a CJK character (or at the very end) is closed with a blank
        (["lo@@", "中"], [[0, 1], [1, 2]], "lo 中 | 0.000000, 1.000000,1.000000, 2.000000"),
        (["ab@@", "cd@@"], [[0, 1], [1, 2]], "abcd  | 0.000000, 2.000000"),
        ([], [], " | "),
    ]
    for chars, stamps, want in cases:
        assert T.post_process(chars, stamps) == want, chars
        assert _post_process_abi(pkg, chars, stamps) == want, chars


def test_post_process_random_agreement(pkg):
    from oracle import timestamp as T
    rng = np.random.default_rng(5)
    vocab = ["你", "好", "世", "界", "a", "bc", "de@@", "f@@", "ghi", "<unk>", "</s>", "x@@", "é", "中"]
    for _ in range(200):
        n = int(rng.integers(0, 12))
        chars = [vocab[int(k)] for k in rng.integers(0, len(vocab), n)]
        t = np.cumsum(rng.uniform(0.01, 0.4, 2 * n)).astype(np.float32).reshape(n, 2) if n else np.zeros((0, 2), np.float32)
        assert _post_process_abi(pkg, chars, t) == T.post_process(chars, t.tolist()), chars
