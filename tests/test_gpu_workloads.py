"""GPU: BASELINE.json configs exercised as WORKLOADS.

C4  `Hotword + timestamp Paraformer, batch=32`: ONE model carrying both the contextual bias decoder and the CifPredictorV3
    timestamp head, 32 ragged utterances, H = 16 hotwords — ids, log-probs, us_alphas / us_cif_peak and the
    character timestamps against the oracle.  (Parity unpinned: UPSTREAM architecture, synthetic weights.)
C3  `10 min stream, 600-ms chunks`: one connection fed 1000 chunks — ids identical to the streaming oracle call by call,
    window rows (x*sqrt(512) + position embedding at running index p up to 10^4, paraformer-online.cpp:240-268) checked
    against the oracle's at every call, decoder FSMN cache rotation exercised 1000 times."""
import numpy as np
import pytest

from conftest import synth_pcm
from oracle import paraformer as P
from oracle import paraformer_online as PO
from oracle import timestamp as TS

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_c4_hotword_and_timestamp_model_batch32(pkg, weights_mod):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=2, vocab=600, contextual=1, timestamp=1)
    man, blob = weights_mod.synth_weights(cfg, seed=404)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    W = P.Weights(man, blob)
    rng = np.random.default_rng(20251114)
    lens = [int(v) for v in rng.integers(16000 * 1, 16000 * 12, 32)]
    lens[5], lens[17] = 300, 16000 * 12 + 77            # one too short for a frame, one longest
    utts = [synth_pcm(i, n, rng) for i, n in enumerate(lens)]
    hot = [list(map(int, rng.integers(2, 600, int(rng.integers(2, 7))))) for _ in range(16)]      # H = 16 rows, lengths 2-6
    hw = model.CompileHotwordEmbedding(hot)
    assert hw.shape == (17, 512)
    got = model.forward_ids(utts, want_logp=True, hw_emb=hw, want_timestamps=True)
    n_checked = 0
    for b, u in enumerate(utts):
        ref = P.forward_pcm(u, W, hw_emb=hw)
        if ref["feats"].shape[0] == 0:
            assert got["n_frames"][b] == 0 and len(got["ids"][b]) == 0
            continue
        assert int(got["n_frames"][b]) == ref["enc"].shape[0]
        assert int(got["n_fires"][b]) == ref["emb"].shape[0] and int(got["token_num"][b]) == ref["token_num"]
        assert np.abs(got["logp"][b] - ref["logp"]).max() < 1e-3, b            # BASELINE.json tolerance
        assert list(got["ids"][b]) == list(ref["ids"]), b
        a_ref, p_ref = P.timestamp_head(ref["enc"], ref["token_num"], W)
        a, p = got["us_alphas"][b], got["us_peaks"][b]
        assert a.shape == a_ref.shape
        assert np.abs(a - a_ref).max() < 2e-6 and np.abs(p - p_ref).max() < 5e-5, b
        n_chars = max(0, ref["token_num"] - 1)
        if n_chars:
            assert TS.timestamp_onnx(a, p, n_chars) == TS.timestamp_onnx(a_ref, p_ref, n_chars), b
        n_checked += 1
    assert n_checked == 31
    # the hotwords are live in this combined model too
    other = model.forward_ids(utts[:4], want_logp=True, hw_emb=hw[::-1][:5].copy())
    assert np.abs(other["logp"][0] - got["logp"][0]).max() > 1e-3
    model.close()


def test_c3_ten_minute_stream_one_connection(pkg, weights_mod):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    cfg = weights_mod.small_config(enc_layers=1, dec_layers=2, vocab=311)
    man, blob = weights_mod.synth_weights(cfg, seed=505)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    W = P.Weights(man, blob)
    rng = np.random.default_rng(3)
    n_chunks = 1000
    pcm = synth_pcm(7, 9600 * n_chunks, rng)
    on = PO.ParaformerOnline(W)
    hip = pkg.ParaformerOnlineHip(model)
    worst_chunk, worst_late, total_ids, p_max = 0.0, 0.0, 0, 0
    for j in range(n_chunks):
        seg = pcm[9600 * j:9600 * (j + 1)]
        fin = j == n_chunks - 1
        before = len(on.chunk_log)
        p_before = on.start_idx_cache_
        ref_ids = on.Forward(seg, fin)
        got_ids = hip.Forward(seg, input_finished=fin)
        assert got_ids == ref_ids, (j, got_ids, ref_ids)
        total_ids += len(ref_ids)
        if len(on.chunk_log) > before:
            last = on.chunk_log[-1]
            chunk = hip.get_tensor("chunk", 128 * 560).reshape(-1, 560)
            assert chunk.shape == last["feats"].shape, j
            err = float(np.abs(chunk - last["feats"]).max())
            worst_chunk = max(worst_chunk, err)
            if p_before >= 9000:
                worst_late = max(worst_late, err)
            p_max = max(p_max, p_before)
            on.chunk_log[:-1] = []                      # keep memory flat
    assert p_max >= 9900                                # the running position index really got to ~10^4
    assert total_ids > 1500
    # rows are x*sqrt(512) (~22.6 x 2e-5 of fbank error) + PE; a position error of one frame, or of one fp32 ulp of the phase at
    # p = 10^4 (1e-3 rad), would exceed this
    assert worst_chunk < 5e-4 and worst_late < 5e-4, (worst_chunk, worst_late)
    print(f"stream window rows: max err {worst_chunk:.2e} (p >= 9000: {worst_late:.2e}), {total_ids} ids over {n_chunks} chunks")
    hip.close()
    model.close()


def test_c4_at_the_models_real_size(pkg, weights_mod):
    """BASELINE.json configs[3] with the deployed model's dimensions: Paraformer-large-sized random-init weights (50 / 16 layers,
    vocabulary 8404) carrying the contextual decoder and the timestamp head, a full batch of 32 ragged utterances (5-30 s) with
    H = 16 hotwords through one forward; three of them against the oracle — ids, log-probs within 1e-3 (north_star), the upsampled
    alphas / peaks and the character timestamps."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    cfg = dict(weights_mod.PARAFORMER_LARGE, contextual=1, timestamp=1)
    man, blob = weights_mod.synth_weights(cfg, seed=4040)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    W = P.Weights(man, blob)
    rng = np.random.default_rng(44)
    lens = [int(v) for v in rng.integers(16000 * 5, 16000 * 30, 32)]
    utts = [synth_pcm(100 + i, n, rng) for i, n in enumerate(lens)]
    hot = [list(map(int, rng.integers(2, 8404, int(rng.integers(2, 7))))) for _ in range(16)]
    hw = model.CompileHotwordEmbedding(hot)
    got = model.forward_ids(utts, want_logp=True, hw_emb=hw, want_timestamps=True)
    order = np.argsort(lens)
    for b in (int(order[0]), int(order[16]), int(order[-1])):          # shortest, median, longest
        ref = P.forward_pcm(utts[b], W, hw_emb=hw)
        assert int(got["n_frames"][b]) == ref["enc"].shape[0]
        assert int(got["token_num"][b]) == ref["token_num"]
        assert list(got["ids"][b]) == list(ref["ids"]), b
        assert np.abs(got["logp"][b] - ref["logp"]).max() < 1e-3, b
        a_ref, p_ref = P.timestamp_head(ref["enc"], ref["token_num"], W)
        a, p = got["us_alphas"][b], got["us_peaks"][b]
        assert a.shape == a_ref.shape
        assert np.abs(a - a_ref).max() < 1e-5 and np.abs(p - p_ref).max() < 2e-4, b
        n_chars = max(0, ref["token_num"] - 1)
        if n_chars:
            assert TS.timestamp_onnx(a, p, n_chars) == TS.timestamp_onnx(a_ref, p_ref, n_chars), b
    model.close()
