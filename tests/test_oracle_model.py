"""CPU: known answers for the oracle's CIF / greedy / log-softmax pieces and its self-consistency."""
import numpy as np

from oracle import paraformer as P


def test_cif_constant_alpha_fires_every_fourth_frame():
    """alpha = 0.3: integrate reaches 1.2 at frame 3 (0-based) -> fire, carry 0.2; SURVEY §8c pin."""
    T, d = 13, 4
    hidden = np.arange(T * d, dtype=np.float32).reshape(T, d)
    alphas = np.full(T, 0.3, np.float32)
    emb, fires = P.cif(hidden, alphas, 1.0)
    # fires at i where cumulative sum crosses an integer: 0.3*4=1.2 (i=3), 0.3*7=2.1 (i=6), 3.0 (i=9), 3.9.. (i=13 none)
    assert emb.shape[0] == 3
    want0 = 0.3 * hidden[0] + 0.3 * hidden[1] + 0.3 * hidden[2] + np.float32(1 - np.float32(0.9)) * hidden[3]
    assert np.allclose(emb[0], want0, rtol=1e-5)
    assert np.all(fires[:3] < 1.0) and fires[3] >= 1.0


def test_cif_weights_sum_to_threshold():
    rng = np.random.default_rng(0)
    T = 200
    alphas = rng.uniform(0, 0.6, T).astype(np.float32)
    emb, _ = P.cif(np.ones((T, 3), np.float32), alphas, 1.0)
    assert emb.shape[0] == int(np.floor(alphas.sum(dtype=np.float64) + 1e-4)) or emb.shape[0] == int(np.floor(alphas.sum(dtype=np.float64)))
    assert np.allclose(emb, 1.0, atol=1e-5)      # each fired frame integrates exactly 1.0 of weight


def test_find_max_first_wins():
    """util.cpp:63-74: strict '>' scan."""
    row = np.array([0.5, 2.0, 2.0, -1.0], np.float32)
    assert P.find_max(row)[1] == 1
    assert P.greedy_search(np.stack([row, row[::-1]]), 2) == [1, 1]


def test_forward_small_model_shapes(weights_mod):
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300)
    man, blob = weights_mod.synth_weights(cfg)
    W = P.Weights(man, blob)
    rng = np.random.default_rng(1)
    w = (rng.standard_normal(16000) * 0.1).astype(np.float32)
    r = P.forward_pcm(w, W)
    assert r["feats"].shape == (17, 560) and r["enc"].shape == (17, 512)
    assert r["logp"].shape == (r["emb"].shape[0], 300)
    assert np.allclose(np.exp(r["logp"]).sum(-1), 1.0, atol=1e-4)
    assert len(r["ids"]) == min(r["token_num"], r["emb"].shape[0])
    # too short for one fbank window -> empty result (paraformer.cpp:477-480)
    r0 = P.forward_pcm(np.zeros(399, np.float32), W)
    assert r0["token_num"] == 0 and r0["ids"] == []
