"""SURVEY §8 row a6 producer on the GPU: CifPredictorV3 upsampling head (ConvTranspose1d -> BLSTM -> alpha head ->
cif_wo_hidden), through the C ABI (pfhip_out.us_alphas / us_peaks), against the numpy restatement.  Parity unpinned
(UPSTREAM architecture, synthetic weights).  The recurrence is well conditioned (fp32 and fp64 restatements agree to 1e-8 on
the alphas), so the tolerances are tight — and the BLSTM output itself is compared, not only the alphas, which are a
1024-term average of it and hide a broken recurrence behind 1e-4."""
import numpy as np
import pytest

from conftest import synth_pcm
from oracle import paraformer as P
from oracle import timestamp as TS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(pkg, weights_mod):
    cfg = weights_mod.small_config(timestamp=1)
    man, blob = weights_mod.synth_weights(cfg, seed=77)
    h = pkg.ParaformerHip().InitAsr((man, blob))
    yield h, P.Weights(man, blob)
    h.close()


def test_timestamp_head_matches_oracle(model):
    h, W = model
    rng = np.random.default_rng(3)
    waves = [synth_pcm(i, n, rng) for i, n in enumerate([16000 * 4, 16000 * 2 + 333, 16000 * 7, 9000])]
    r = h.forward_ids(waves, want_timestamps=True)
    for b, w in enumerate(waves):
        ref = P.forward_pcm(w, W)
        assert int(r["token_num"][b]) == ref["token_num"]
        a_ref, p_ref = P.timestamp_head(ref["enc"], ref["token_num"], W)
        a, p = r["us_alphas"][b], r["us_peaks"][b]
        assert a.shape == a_ref.shape == (3 * ref["enc"].shape[0],)
        assert np.abs(a - a_ref).max() < 5e-7, b
        assert np.abs(p - p_ref).max() < 2e-5, b
        # what the consumer makes of it: same character timestamps (TimestampOnnx, util.cpp:838-963)
        n_chars = max(0, ref["token_num"] - 1)
        if n_chars:
            assert TS.timestamp_onnx(a, p, n_chars) == TS.timestamp_onnx(a_ref, p_ref, n_chars)


def test_blstm_output_matches_oracle(model):
    """The recurrent part alone: every h_t of both directions, ragged batch (frames of short utterances end early)."""
    h, W = model
    rng = np.random.default_rng(5)
    waves = [synth_pcm(i, n, rng) for i, n in enumerate([16000 * 5, 16000 * 1, 16000 * 3 + 77])]
    r = h.forward_ids(waves, want_timestamps=True)
    M, d = int(sum(r["n_frames"])), 512
    y = h.get_tensor("ts_y", 3 * M * 2 * d).reshape(3 * M, 2 * d)
    o = 0
    for b, w in enumerate(waves):
        enc = P.forward_pcm(w, W)["enc"]
        T = enc.shape[0]
        u = np.stack([enc @ W["pred.up.w"][:, :, j] + W["pred.up.b"] for j in range(3)], axis=1).reshape(3 * T, d).astype(np.float32)
        hf = P._lstm_dir(u, W["pred.blstm.w_ih"], W["pred.blstm.w_hh"], W["pred.blstm.b_ih"], W["pred.blstm.b_hh"], False)
        hb = P._lstm_dir(u, W["pred.blstm.w_ih_r"], W["pred.blstm.w_hh_r"], W["pred.blstm.b_ih_r"], W["pred.blstm.b_hh_r"], True)
        assert np.abs(y[o:o + 3 * T, :d] - hf).max() < 2e-5, b
        assert np.abs(y[o:o + 3 * T, d:] - hb).max() < 2e-5, b
        o += 3 * T


def test_full_batch_of_32_uses_both_row_tiles(model):
    """33 utterances: one launch with two MFMA row tiles + a second launch; same numbers as one at a time."""
    h, _ = model
    rng = np.random.default_rng(6)
    waves = [synth_pcm(i, 16000 + 700 * i, rng) for i in range(33)]
    both = h.forward_ids(waves, want_timestamps=True)
    for b in (0, 15, 16, 31, 32):
        one = h.forward_ids([waves[b]], want_timestamps=True)
        assert np.abs(one["us_alphas"][0] - both["us_alphas"][b]).max() < 1e-6
        assert np.abs(one["us_peaks"][0] - both["us_peaks"][b]).max() < 1e-5


def test_batch_composition_does_not_change_timestamps(model):
    h, _ = model
    rng = np.random.default_rng(4)
    waves = [synth_pcm(i, n, rng) for i, n in enumerate([16000 * 3, 16000 * 5 + 100, 16000 * 2])]
    both = h.forward_ids(waves, want_timestamps=True)
    for b, w in enumerate(waves):
        one = h.forward_ids([w], want_timestamps=True)
        assert np.abs(one["us_alphas"][0] - both["us_alphas"][b]).max() < 1e-6
        assert np.abs(one["us_peaks"][0] - both["us_peaks"][b]).max() < 1e-5


def test_plain_model_refuses_timestamps(pkg, weights_mod):
    man, blob = weights_mod.synth_weights(weights_mod.small_config(), seed=1)
    h = pkg.ParaformerHip().InitAsr((man, blob))
    with pytest.raises(pkg.PfhipError):
        h.forward_ids([np.zeros(16000, np.float32)], want_timestamps=True)
    h.close()


def test_blstm_barrier_timeout_falls_back_to_the_per_step_recurrence(model, pkg):
    """A step-barrier time-out of the persistent BLSTM kernel (the 32 blocks of a direction not co-resident on one XCD: other
    models share the GPU in the 2-pass server) does not fail the request: the recurrence is redone as one launch per step,
    with bit-identical results; the flag does not outlive the request (round-1 advisor finding)."""
    h, W = model
    rng = np.random.default_rng(11)
    waves = [synth_pcm(i, n, rng) for i, n in enumerate([16000 * 2, 16000 * 3])]
    good = h.forward_ids(waves, want_timestamps=True)
    before = h._lib.pfhip_debug_poke(h.handle, b"blstm_fallbacks", 0)
    assert h._lib.pfhip_debug_poke(h.handle, b"blstm_flag", 1) == 0
    via_fallback = h.forward_ids(waves, want_timestamps=True)
    assert h._lib.pfhip_debug_poke(h.handle, b"blstm_fallbacks", 0) == before + 1
    again = h.forward_ids(waves, want_timestamps=True)
    assert h._lib.pfhip_debug_poke(h.handle, b"blstm_fallbacks", 0) == before + 1
    for b in range(2):
        assert np.array_equal(via_fallback["us_alphas"][b], good["us_alphas"][b])
        assert np.array_equal(via_fallback["us_peaks"][b], good["us_peaks"][b])
        assert np.array_equal(again["us_alphas"][b], good["us_alphas"][b])


def test_per_step_recurrence_alone_matches_oracle(pkg, weights_mod):
    """PFHIP_BLSTM_STEPWISE=1 (read once per process, hence a child): the whole timestamp suite on the per-step form."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PFHIP_BLSTM_STEPWISE="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_timestamp.py", "-k",
                          "timestamp_head_matches_oracle or blstm_output_matches_oracle"], cwd=root, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
