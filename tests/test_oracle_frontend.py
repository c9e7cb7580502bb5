"""CPU: pins the oracle's front end against the reference's golden data.
(i) committed vectors produced by the reference's own knf sources (tests/golden/make_golden.py),
(ii) the knf known-answer test third_party/kaldi-native-fbank/.../csrc/test-rfft.cc:32-50,
(iii) live comparison with oracle/_ref/libknf_ref.so when it has been built,
(iv) hand-derived known answers for LfrCmvn / LoadCmvn / GetPosEmb."""
import ctypes
import math
import os

import numpy as np
import pytest

from oracle import frontend as fe

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
# oracle vs knf: identical operation order; differences come only from numpy's SIMD logf (<= 1 ulp at
# |x| ~ 20 -> 1.9e-6) and pocketfft-vs-Ooura rounding in the double FFT
FBANK_TOL = 4e-6


@pytest.mark.parametrize("name", ["fbank_synth", "fbank_xmov", "fbank_floor", "fbank_stevejobs_10s", "fbank_number", "fbank_fullscale"])
def test_fbank_matches_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    got = fe.fbank(g["pcm"].astype(np.float32) / 32768)
    assert got.shape == g["fbank"].shape
    assert np.abs(got - g["fbank"]).max() <= FBANK_TOL


def test_floor_value_is_log_flt_epsilon():
    g = np.load(os.path.join(GOLD, "fbank_floor.npz"))
    # frames that contain only zeros hit log(FLT_EPSILON) (feature-fbank.cc:102-107)
    assert np.isclose(g["fbank"][0, 0], math.log(np.finfo(np.float32).eps), atol=1e-6)
    assert fe.fbank(np.zeros(400, np.float32))[0, 0] == np.float32(math.log(np.finfo(np.float32).eps))


def test_rfft_known_answer():
    """test-rfft.cc:32-50."""
    d = fe.rfft_packed(np.array([1, -1, 3, 8, 20, 6, 0, 2], np.float32))
    assert d[0] == 39 and d[1] == 9
    assert abs(d[2] - -28.1924) < 1e-3 and abs(-d[3] - -2.2929) < 1e-3
    assert abs(d[4] - 18) < 1e-3 and abs(-d[5] - 5) < 1e-3
    assert abs(d[6] - -9.8076) < 1e-3 and abs(-d[7] - 3.7071) < 1e-3


def test_num_frames_snip_edges():
    """feature-window.cc:84-87 and SURVEY §8 sizes: 30 s -> 2998, 5 s -> 498."""
    assert fe.num_frames(399) == 0 and fe.num_frames(400) == 1 and fe.num_frames(559) == 1 and fe.num_frames(560) == 2
    assert fe.num_frames(480000) == 2998 and fe.num_frames(80000) == 498


def test_live_reference_build_if_present():
    so = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libknf_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built on this machine")
    lib = ctypes.CDLL(so)
    lib.knf_ref_fbank.restype = ctypes.c_int
    lib.knf_ref_fbank.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(7)
    for n in (400, 1234, 16000):
        w = (rng.standard_normal(n) * 0.2).clip(-1, 1).astype(np.float32)
        nf = lib.knf_ref_fbank(w.ctypes.data, n, 80, None, 0)
        ref = np.empty((nf, 80), np.float32)
        lib.knf_ref_fbank(w.ctypes.data, n, 80, ref.ctypes.data, nf)
        got = fe.fbank(w)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= FBANK_TOL


def test_mel_banks_shape():
    off, size, w = fe.mel_banks(80)
    assert off[0] >= 1 and off[-1] + size[-1] <= 256 and size.max() <= 32
    assert all(0 < x.max() <= 1.0 for x in w)


# ---- LfrCmvn: paraformer.cpp:421-461 -------------------------------------------------------------
def test_lfr_known_answer_frame_index():
    """feat[f, k] = f  =>  row i = clamp(6i-3 .. 6i+3, 0, F-1) repeated over the 80 dims."""
    for F in (1, 5, 6, 7, 12, 13, 20):
        feats = np.repeat(np.arange(F, dtype=np.float32)[:, None], 80, axis=1)
        out = fe.lfr_cmvn(feats, np.zeros(560, np.float32), np.ones(560, np.float32))
        T = math.ceil(F / 6)
        assert out.shape == (T, 560)
        for i in range(T):
            want = np.clip(np.arange(6 * i - 3, 6 * i + 4), 0, F - 1)
            assert np.array_equal(out[i].reshape(7, 80)[:, 0], want.astype(np.float32)), (F, i)


def test_cmvn_is_add_then_scale():
    feats = np.ones((6, 80), np.float32)
    mean = np.full(560, -8.0, np.float32)
    istd = np.full(560, 0.25, np.float32)
    out = fe.lfr_cmvn(feats, mean, istd)
    assert np.allclose(out, (1.0 - 8.0) * 0.25)


def test_parse_cmvn_kaldi_nnet_text():
    """LoadCmvn paraformer.cpp:325-360: tokens [3..n-1) of the line after <AddShift>/<Rescale>."""
    txt = "<Nnet>\n<AddShift> 3 3\n<LearnRateCoef> 0 [ -1.5 -2.5 -3.5 ]\n<Rescale> 3 3\n<LearnRateCoef> 0 [ 0.1 0.2 0.3 ]\n</Nnet>\n"
    m, s = fe.parse_cmvn(txt)
    assert np.allclose(m, [-1.5, -2.5, -3.5]) and np.allclose(s, [0.1, 0.2, 0.3])


# ---- GetPosEmb: paraformer-online.cpp:240-268 ---------------------------------------------------------
def test_pos_emb_first_row():
    pe = fe.pos_emb(3, 560)
    assert pe.shape == (3, 560)
    # i = 0: timescale 1 -> sin(1), cos(1) at position 1
    assert np.isclose(pe[0, 0], math.sin(1.0), atol=1e-7) and np.isclose(pe[0, 280], math.cos(1.0), atol=1e-7)
    assert np.isclose(pe[2, 0], math.sin(3.0), atol=1e-6)
    # last frequency: exp(279 * scale) = 1e-4
    assert np.isclose(pe[0, 279], math.sin(1e-4), rtol=1e-3)
    # streaming continuation: rows of a later chunk equal the rows of one long table
    assert np.array_equal(fe.pos_emb(2, 560, start=1), pe[1:3])
