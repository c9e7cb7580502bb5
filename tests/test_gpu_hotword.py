"""GPU: contextual (hotword) Paraformer — the hotword embedder (model_eb.onnx stand-in, paraformer.cpp:656-685) and the
bias decoder of the last layer — against the oracle (BASELINE config C4, contextual half)."""
import numpy as np
import pytest

from conftest import synth_pcm
from oracle import paraformer as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ctx(pkg, weights_mod):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=2, vocab=400, contextual=1)
    man, blob = weights_mod.synth_weights(cfg, seed=99)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    yield model, P.Weights(man, blob)
    model.close()


def test_hotword_embedding_matches_oracle(ctx):
    model, W = ctx
    rng = np.random.default_rng(1)
    hot = [list(rng.integers(2, 400, n)) for n in (1, 2, 3, 6, 10, 14)]        # 14 is truncated to 10 (:629)
    got = model.CompileHotwordEmbedding(hot)
    rows = [h[:10] + [0] * (10 - len(h[:10])) for h in hot] + [[1] + [0] * 9]
    lens = [len(h[:10]) for h in hot] + [1]
    ref = P.hotword_embed(rows, lens, W)
    assert got.shape == ref.shape == (7, 512)
    assert np.abs(got - ref).max() < 2e-5


def test_many_hotwords_use_the_tiled_gemm(ctx):
    model, W = ctx
    rng = np.random.default_rng(2)
    hot = [list(rng.integers(2, 400, int(rng.integers(1, 11)))) for _ in range(100)]
    got = model.CompileHotwordEmbedding(hot)
    rows = [h + [0] * (10 - len(h)) for h in hot] + [[1] + [0] * 9]
    lens = [len(h) for h in hot] + [1]
    assert np.abs(got - P.hotword_embed(rows, lens, W)).max() < 2e-5


def test_contextual_forward_matches_oracle_and_depends_on_hotwords(ctx, pkg):
    model, W = ctx
    rng = np.random.default_rng(3)
    utts = [synth_pcm(i, n, rng) for i, n in enumerate([48000, 80000, 32000])]
    hw = model.CompileHotwordEmbedding([list(rng.integers(2, 400, n)) for n in (2, 3, 4, 5, 2, 6)])
    got = model.forward_ids(utts, want_logp=True, hw_emb=hw)
    for b, u in enumerate(utts):
        ref = P.forward_pcm(u, W, hw_emb=hw)
        assert got["n_fires"][b] == ref["emb"].shape[0]
        assert np.abs(got["logp"][b] - ref["logp"]).max() < 1e-3
        assert list(got["ids"][b]) == list(ref["ids"])
    other = model.forward_ids(utts, want_logp=True, hw_emb=hw[::-1][:3].copy())
    assert np.abs(other["logp"][0] - got["logp"][0]).max() > 1e-3          # the bias path is live
    with pytest.raises(pkg.PfhipError, match="hw_emb is null"):
        model.forward_ids(utts)                                            # paraformer.cpp:516-520


def test_plain_model_ignores_hotwords(pkg, weights_mod):
    cfg = weights_mod.small_config(enc_layers=1, dec_layers=1, vocab=300)
    man, blob = weights_mod.synth_weights(cfg)
    m = pkg.ParaformerHip().InitAsr((man, blob))
    assert m.CompileHotwordEmbedding([[3, 4]]).shape == (1, 512) and not m.CompileHotwordEmbedding([[3, 4]]).any()
    rng = np.random.default_rng(4)
    u = synth_pcm(0, 32000, rng)
    a = m.forward_ids([u])
    b = m.forward_ids([u], hw_emb=np.ones((2, 512), np.float32))
    assert list(a["ids"][0]) == list(b["ids"][0])
    m.close()


def test_more_hotwords_than_audio_rows_on_a_fresh_handle(pkg, weights_mod):
    """300 hotwords against a 2-s utterance on a handle that has run nothing else: the hotword K/V projection is larger than
    every audio-side workspace (Mp = 128 rows), so it must live in a buffer of its own (round-1 advisor finding)."""
    cfg = weights_mod.small_config(enc_layers=1, dec_layers=1, vocab=400, contextual=1)
    man, blob = weights_mod.synth_weights(cfg, seed=7)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    W = P.Weights(man, blob)
    rng = np.random.default_rng(5)
    u = synth_pcm(3, 32000, rng)
    hw = model.CompileHotwordEmbedding([list(rng.integers(2, 400, int(rng.integers(1, 8)))) for _ in range(300)])
    assert hw.shape == (301, 512)
    got = model.forward_ids([u], want_logp=True, hw_emb=hw)
    ref = P.forward_pcm(u, W, hw_emb=hw)
    assert got["n_fires"][0] == ref["emb"].shape[0]
    assert np.abs(got["logp"][0] - ref["logp"]).max() < 1e-3
    assert list(got["ids"][0]) == list(ref["ids"])
    # a second, smaller hotword set on the same handle replaces the first
    hw2 = hw[:5].copy()
    got2 = model.forward_ids([u], want_logp=True, hw_emb=hw2)
    ref2 = P.forward_pcm(u, W, hw_emb=hw2)
    assert np.abs(got2["logp"][0] - ref2["logp"]).max() < 1e-3
    model.close()
