"""CPU: the end-point detector pinned to the REFERENCE compiled here (SURVEY §8 rows c, f1).

tests/golden/vadseg_*.npz were produced by `funasr::E2EVadModel` itself (onnxruntime/src/e2e-vad.h, compiled in place into
oracle/_ref/libe2evad_ref.so; generator tests/golden/make_vadseg_golden.py).  Both the oracle restatement
(oracle/e2e_vad.py) and the product's C++ detector (csrc/host/vad_segmenter.cpp through pfhip_vadseg_feed) must reproduce
every call's segments bit-exactly (integers).  Where the compiled reference is present it is also driven directly on
random plans (it travels to the GPU box with the snapshot but these are CPU tests)."""
import glob
import importlib.util
import os

import numpy as np
import pytest

from oracle import e2e_vad as E

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = sorted(glob.glob(os.path.join(HERE, "golden", "vadseg_*.npz")))
REF_SO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libe2evad_ref.so")

spec = importlib.util.spec_from_file_location("make_vadseg_golden", os.path.join(HERE, "golden", "make_vadseg_golden.py"))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)


def test_fixture_set_is_complete():
    names = {os.path.basename(p)[7:-4] for p in GOLD}
    assert {"all_silence", "single_burst", "max_segment_split", "online_1s_feeds", "low_energy", "reuse_offline", "reuse_online",
            "online_irregular", "speech_to_eof_offline", "speech_to_eof_online", "online_empty_final", "offline_chunked"} <= names


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_oracle_reproduces_reference_segments(path):
    c = np.load(path)
    m = E.E2EVadModel()
    got = G.run_plan(lambda *a: m(*a), c["sil"], G.waveform(c["amp"], c["pattern"]), c["calls"], c["params"])
    assert got.tolist() == c["segs"].tolist()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[7:-4] for p in GOLD])
def test_product_reproduces_reference_segments(pkg, path):
    c = np.load(path)
    m = pkg.E2EVadModelHost()
    got = G.run_plan(lambda *a: m(*a), c["sil"], G.waveform(c["amp"], c["pattern"]), c["calls"], c["params"])
    m.close()
    assert got.tolist() == c["segs"].tolist()


def test_known_consequences_of_the_goldens():
    """Sanity on the fixtures themselves (hand-derivable from e2e-vad.h's constants): forced splits are exactly
    max_single_segment_time + one frame apart (:672-690: `cur_frm_idx - confirmed_start_frame + 1 > 600` fires at the 601st
    frame); online open/close markers alternate; an empty final call closes nothing."""
    c = np.load(os.path.join(HERE, "golden", "vadseg_max_segment_split.npz"))
    s = c["segs"]
    assert [int(e - b) for _, b, e in s[:3]] == [6010, 6010, 6010] and all(s[i][2] == s[i + 1][1] for i in range(3))
    c = np.load(os.path.join(HERE, "golden", "vadseg_online_1s_feeds.npz"))
    marks = [(int(b) >= 0, int(e) >= 0) for _, b, e in c["segs"]]
    assert marks == [(True, False), (False, True)] * 3
    c = np.load(os.path.join(HERE, "golden", "vadseg_online_empty_final.npz"))
    assert c["segs"].tolist() == [[1, 280, -1]]


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref/libe2evad_ref.so not built (make -C oracle)")
def test_random_plans_against_the_compiled_reference(pkg):
    lib = G.load_ref()
    rng = np.random.default_rng(77)
    pattern = (0.5 * rng.standard_normal(160)).astype(np.float32)
    for trial in range(60):
        runs = [(int(rng.integers(5, 400)), bool(k % 2 == int(trial % 2))) for k in range(int(rng.integers(1, 9)))]
        sil = G.scores(rng, runs, noise=float(rng.uniform(0.01, 0.2)), flip=float(rng.uniform(0, 0.1)))
        T = len(sil)
        amp = (10.0 ** rng.uniform(-5, 0, T + 2)).astype(np.float32)
        amp[rng.random(T + 2) < 0.05] = 0.0
        online = bool(trial % 3 != 0)
        sizes = [int(v) for v in rng.integers(1, 200, 4)] if trial % 4 else [T]
        calls = G.chunks(T, sizes, online, final_empty=bool(trial % 5 == 0 and online))
        params = (int(rng.choice([200, 500, 800, 1500])), int(rng.choice([1000, 6000, 15000, 60000])), float(rng.choice([0.5, 0.8, 0.9])))
        w = G.waveform(amp, pattern)
        h = lib.e2evad_ref_create()

        def ref_feed(s, wv, fin, on, tail, mx, thr):
            s = np.ascontiguousarray(s, np.float32)
            wv = np.ascontiguousarray(wv, np.float32)
            pairs = np.zeros((len(s) + 8, 2), np.int32)
            n = lib.e2evad_ref_feed(h, s.ctypes.data, len(s), wv.ctypes.data, len(wv), int(fin), int(on), tail, mx, thr, 16000,
                                    pairs.ctypes.data, len(s) + 8)
            return [list(map(int, p)) for p in pairs[:n]]

        want = G.run_plan(ref_feed, sil, w, calls, params).tolist()
        lib.e2evad_ref_destroy(h)
        mo, mp = E.E2EVadModel(), pkg.E2EVadModelHost()
        assert G.run_plan(lambda *a: mo(*a), sil, w, calls, params).tolist() == want, ("oracle", trial)
        assert G.run_plan(lambda *a: mp(*a), sil, w, calls, params).tolist() == want, ("product", trial)
        mp.close()
