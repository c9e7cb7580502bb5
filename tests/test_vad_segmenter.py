"""CPU (host logic): the C++ end-point detector (csrc/host/vad_segmenter.cpp, through the C ABI) against the oracle
restatement of funasr::E2EVadModel (onnxruntime/src/e2e-vad.h), plus hand-derived consequences of the reference's
constants (200-ms window, 150-ms hysteresis, 200-ms look-back, 800-ms end silence, 15-s max segment)."""
import numpy as np
import pytest

from oracle import e2e_vad as E


def make_scores(rng, runs, noise=0.02):
    """runs: list of (n_frames, is_speech).  Silence posterior ~0.97 in silence, ~0.03 in speech, plus jitter and a few flips."""
    sil = []
    for n, sp in runs:
        base = 0.03 if sp else 0.97
        v = np.clip(base + noise * rng.standard_normal(n), 1e-4, 1 - 1e-4)
        flips = rng.random(n) < 0.03
        v[flips] = 1 - v[flips]
        sil.append(v)
    return np.concatenate(sil).astype(np.float32)


def wave_for(n_frames, rng):
    return (0.05 * rng.standard_normal(400 + 160 * (n_frames - 1))).astype(np.float32)


@pytest.fixture()
def seg(pkg):
    s = pkg.E2EVadModelHost()
    yield s
    s.close()


def test_offline_one_shot_matches_oracle_and_is_sane(seg):
    rng = np.random.default_rng(1)
    runs = [(120, False), (300, True), (150, False), (80, True), (40, False), (500, True), (200, False)]
    sil = make_scores(rng, runs)
    w = wave_for(len(sil), rng)
    ref = E.E2EVadModel()(sil, w, True, False, 800, 15000, 0.9)
    got = seg(sil, w, True, False, 800, 15000, 0.9)
    assert got == ref and len(got) >= 2
    # first speech run starts at frame 120: start = onset - look-back (200 ms) within the 150-ms detection delay
    assert 1200 - 200 - 60 <= got[0][0] <= 1200
    for s, e in got:
        assert 0 <= s < e <= len(sil) * 10 + 10


def test_online_chunked_feeds_match_oracle_and_merge_to_offline(seg):
    rng = np.random.default_rng(2)
    runs = [(90, False), (400, True), (130, False), (250, True), (100, False)]
    sil = make_scores(rng, runs)
    T = len(sil)
    w = wave_for(T, rng)
    ref_m, pos = E.E2EVadModel(), 0
    all_ref, all_got = [], []
    while pos < T:
        n = min(100, T - pos)                    # the reference feeds 1-s slices (audio.cpp:1183-1196)
        fin = pos + n >= T
        ws = w[pos * 160: (pos + n - 1) * 160 + 400]
        r = ref_m(sil[pos:pos + n], ws, fin, True, 800, 15000, 0.9)
        g = seg(sil[pos:pos + n], ws, fin, True, 800, 15000, 0.9)
        assert g == r, pos
        all_ref += r
        all_got += g
        pos += n
    # merge the [-1] markers the way Audio::CutSplit does (audio.cpp:1199-1223)
    merged, s_i, e_i = [], -1, -1
    for s, e in all_got:
        if s != -1:
            s_i = s
        if e != -1:
            e_i = e
        if s_i != -1 and e_i != -1:
            merged.append([s_i, e_i])
            s_i = e_i = -1
    off = E.E2EVadModel()(sil, w, True, False, 800, 15000, 0.9)
    assert merged == off


def test_max_single_segment_time_splits(seg):
    rng = np.random.default_rng(3)
    sil = make_scores(rng, [(50, False), (2500, True), (100, False)], noise=0.005)
    w = wave_for(len(sil), rng)
    ref = E.E2EVadModel()(sil, w, True, False, 800, 6000, 0.9)
    got = seg(sil, w, True, False, 800, 6000, 0.9)
    assert got == ref
    assert len(got) >= 4 and all(e - s <= 6000 + 20 for s, e in got)


def test_all_silence_and_reuse_after_final(seg):
    rng = np.random.default_rng(4)
    sil = make_scores(rng, [(300, False)], noise=0.005)
    w = wave_for(300, rng)
    assert seg(sil, w, True, False) == E.E2EVadModel()(sil, w, True, False) == []
    # the detector resets itself on is_final (e2e-vad.h:357-359) and can take the next file
    sil2 = make_scores(rng, [(60, False), (200, True), (120, False)])
    w2 = wave_for(len(sil2), rng)
    ref_m = E.E2EVadModel()
    ref_m(sil, w, True, False)
    assert seg(sil2, w2, True, False, 800, 15000, 0.9) == ref_m(sil2, w2, True, False, 800, 15000, 0.9)


def test_random_stress_matches_oracle(seg):
    rng = np.random.default_rng(5)
    for trial in range(6):
        runs = [(int(rng.integers(5, 400)), bool(i % 2)) for i in range(int(rng.integers(3, 12)))]
        sil = make_scores(rng, runs, noise=0.2)          # noisy: many threshold crossings
        w = wave_for(len(sil), rng)
        thr = float(rng.choice([0.6, 0.8, 0.9]))
        tail = int(rng.choice([400, 800, 1200]))
        assert seg(sil, w, True, False, tail, 15000, thr) == E.E2EVadModel()(sil, w, True, False, tail, 15000, thr), trial


def test_bad_arguments(pkg, seg):
    with pytest.raises(pkg.PfhipError):
        seg(np.full(100, 0.5, np.float32), np.zeros(1000, np.float32))        # waveform too short for 100 frames
