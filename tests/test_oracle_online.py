"""CPU: the streaming oracle's host-side control flow against hand-derived consequences of
onnxruntime/src/paraformer-online.cpp (chunk scheduling, caches, PE continuity)."""
import numpy as np

from conftest import synth_pcm
from oracle import frontend as fe
from oracle import paraformer as P
from oracle import paraformer_online as PO


def make(weights_mod, **over):
    cfg = weights_mod.small_config(enc_layers=1, dec_layers=1, vocab=200, **over)
    man, blob = weights_mod.synth_weights(cfg)
    return P.Weights(man, blob)


def test_compute_frame_num():
    """paraformer-online.h:25-31."""
    assert PO.compute_frame_num(399, 400, 160) == 0 and PO.compute_frame_num(400, 400, 160) == 1
    assert PO.compute_frame_num(9600, 400, 160) == 58 and PO.compute_frame_num(9840, 400, 160) == 60


def test_streaming_features_equal_offline_features(weights_mod):
    """Across 9600-sample steps the streamed LFR/CMVN rows must equal the offline LfrCmvn of the same audio
    (the splice cache exists exactly for that), except that the stream has emitted one row less until flush."""
    W = make(weights_mod)
    rng = np.random.default_rng(3)
    pcm = synth_pcm(0, 9600 * 5 + 1234, rng)
    on = PO.ParaformerOnline(W)
    rows = []
    pos = 0
    while pos < len(pcm):
        n = min(9600, len(pcm) - pos)
        fin = pos + n >= len(pcm)
        on.is_first_chunk = False
        rows += on.ExtractFeats(pcm[pos:pos + n], fin)
        pos += n
    got = np.stack(rows)
    ref = fe.extract_feats(pcm, W["cmvn.mean"], W["cmvn.istd"])
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-6


def test_chunk_schedule_and_window_sizes(weights_mod):
    W = make(weights_mod)
    rng = np.random.default_rng(4)
    pcm = synth_pcm(1, 9600 * 4, rng)
    on = PO.ParaformerOnline(W)
    for k in range(3):
        on.Forward(pcm[k * 9600:(k + 1) * 9600], False)
    # every non-final window is 5|10|5 = 20 rows once the stream runs (first window: 10 zero rows + 10 new)
    assert [c["feats"].shape[0] for c in on.chunk_log] == [20, 20, 20]
    assert np.all(on.chunk_log[0]["feats"][:10] == 0)
    # the look-back/look-ahead overlap: last 10 rows of window k are the first 10 of window k+1
    assert np.array_equal(on.chunk_log[1]["feats"][-10:], on.chunk_log[2]["feats"][:10])
    # final call with a full chunk: first chunk padded to 20, then the last chunk (5 + k rows, no padding)
    on.Forward(pcm[3 * 9600:], True)
    sizes = [c["feats"].shape[0] for c in on.chunk_log[3:]]
    assert sizes[0] == 20 and len(sizes) == 2 and sizes[1] < 20
    # state is reset afterwards (:589-593)
    assert on.is_first_chunk and on.start_idx_cache_ == 0 and len(on.input_cache_) == 0 and not on.lfr_splice_cache_


def test_short_final_call_flushes_cache(weights_mod):
    """:532-540 — len < 960 && input_finished && !is_first_chunk: the 10 cached rows form the last window."""
    W = make(weights_mod)
    rng = np.random.default_rng(5)
    pcm = synth_pcm(2, 9600 * 2 + 500, rng)
    on = PO.ParaformerOnline(W)
    on.Forward(pcm[:9600], False)
    on.Forward(pcm[9600:19200], False)
    on.Forward(pcm[19200:], True)
    assert on.chunk_log[-1]["feats"].shape[0] == 10
    assert on.is_first_chunk


def test_cif_search_zeroes_lookback_and_lookahead(weights_mod):
    W = make(weights_mod)
    on = PO.ParaformerOnline(W)
    hidden = [np.full(512, float(i + 1), np.float32) for i in range(20)]
    alphas = np.full(20, 0.5, np.float32)
    frames = on.CifSearch(hidden, alphas)
    # only rows 5..14 carry weight: 10 * 0.5 = 5.0 -> 5 fires, each 0.5*h[i] + 0.5*h[i+1]
    assert len(frames) == 5
    assert np.allclose(frames[0], 0.5 * 6 + 0.5 * 7)
    assert on.alphas_cache_[0] == 0.0


def test_fsmn_cached_is_causal_and_carries_state():
    rng = np.random.default_rng(6)
    w = rng.standard_normal((8, 11)).astype(np.float32)
    x = rng.standard_normal((25, 8)).astype(np.float32)
    cache = np.zeros((10, 8), np.float32)
    whole, _ = PO.fsmn_cached(x, w, cache)
    a, c1 = PO.fsmn_cached(x[:7], w, cache)
    b, _ = PO.fsmn_cached(x[7:], w, c1)
    assert np.abs(np.concatenate([a, b]) - whole).max() < 1e-5


def test_online_vad_restatement_equals_offline_scores():
    """FsmnVadOnline restated (feature caches + network caches): fed in 600-ms steps it emits the same rows as the one-pass
    offline forward, row for row, except the final call (scored against zeroed caches, fsmn-vad-online.cpp:84-87)."""
    import importlib
    import __graft_entry__ as ge
    from conftest import synth_pcm
    from oracle import fsmn_vad as V
    from oracle import paraformer as P
    wt = importlib.import_module(ge.load_package().__name__ + ".weights")
    man, blob = wt.synth_vad_weights()
    W = P.Weights(man, blob)
    rng = np.random.default_rng(0)
    pcm = synth_pcm(0, 16000 * 2 + 777, rng)
    on = V.FsmnVadOnline(W)
    outs, wave_lens = [], []
    cuts = list(range(0, len(pcm), 9600)) + [len(pcm)]
    for j in range(len(cuts) - 1):
        p, w = on.Infer(pcm[cuts[j]:cuts[j + 1]], j == len(cuts) - 2)
        outs.append(p)
        wave_lens.append(len(w))
        assert len(w) >= (400 + 160 * (len(p) - 1) if len(p) else 0)              # the scorer's waveform covers at least these rows
    allp = np.concatenate(outs)
    off = V.FsmnVad(W).Forward(pcm, True)
    assert allp.shape == off.shape
    last = len(outs[-1])
    assert np.abs(allp[:-last] - off[:-last]).max() < 1e-6
    assert len(on.input_cache_) == 0 and len(on.reserve_waveforms_) == 0      # reset by the final call
