"""GPU: CT-Transformer forward (pfhip_punc_*) against the oracle, incl. the 5-of-6 argmax quirk
(ct-transformer.cpp:193-196) and the d_k = 32 instantiation of the attention kernel."""
import importlib

import numpy as np
import pytest

from oracle import ct_transformer as C
from oracle import paraformer as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def punc(pkg, weights_mod):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    cfg = dict(weights_mod.CT_TRANSFORMER, vocab=5000)          # small table; same widths as the real model
    man, blob = weights_mod.synth_punc_weights(cfg)
    h = pkg.CTTransformerHip().InitPunc((man, blob))
    yield h, P.Weights(man, blob)
    h.close()


@pytest.mark.parametrize("n", [1, 20, 33, 64, 65, 220])
def test_infer_matches_oracle(punc, n):
    h, W = punc
    rng = np.random.default_rng(n)
    ids = rng.integers(0, 5000, n).astype(np.int32)
    got_p, got_l = h.Infer(ids, want_logits=True)
    ref_l, ref_p = C.infer(ids, W)
    assert np.abs(got_l - ref_l).max() < 1e-3
    assert np.array_equal(got_p, ref_p)
    assert got_p.max() <= 4            # class 5 can never be chosen: Argmax over CANDIDATE_NUM-1 classes


def test_bad_token_id_is_an_error(punc, pkg):
    h, _ = punc
    with pytest.raises(pkg.PfhipError):
        h.Infer(np.asarray([1, 2, 999999], np.int32))


def test_attention_head_dim_32_op(pkg):
    ops = importlib.import_module("asr_2pass_amd.ops")
    rng = np.random.default_rng(5)
    H, dk = 8, 32
    lens = [1, 33, 100]
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    Q = rng.standard_normal((sum(lens), H * dk)).astype(np.float32)
    K = rng.standard_normal((sum(lens), H * dk)).astype(np.float32)
    V = rng.standard_normal((sum(lens), H * dk)).astype(np.float32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    ln = dev(np.asarray(lens, np.int32))
    O = ops.attention(dev(Q), dev(K), dev(V), dev(off), ln, dev(off), ln, H, dk ** -0.5, head_dim=32).cpu().numpy()
    for b, (o, L) in enumerate(zip(off, lens)):
        ref = P.mha(Q[o:o + L].astype(np.float64), K[o:o + L].astype(np.float64), V[o:o + L].astype(np.float64), H)
        assert np.abs(O[o:o + L] - ref).max() < 2e-5


@pytest.mark.parametrize("shift", [0, 5])
def test_online_variant_with_vad_mask(pkg, weights_mod, shift):
    """CTTransformerOnline::Infer: the VadMask prefix mask in every attention block + the shifted FSMN window."""
    cfg = dict(weights_mod.CT_TRANSFORMER, vocab=3000, sanm_shift=shift)
    man, blob = weights_mod.synth_punc_weights(cfg, seed=7 + shift)
    h = pkg.CTTransformerHip().InitPunc((man, blob))
    W = P.Weights(man, blob)
    rng = np.random.default_rng(40 + shift)
    for n, pos in [(30, 12), (30, 0), (30, 30), (75, 2), (140, 100)]:
        ids = rng.integers(0, 3000, n).astype(np.int32)
        gp, gl = h.Infer(ids, want_logits=True, nCacheSize=pos)
        rl, rp = C.forward_online(ids, W, pos)
        assert np.abs(gl - rl).max() < 1e-3, (n, pos)
        assert np.array_equal(gp, rp)
    # the mask matters: with a cut in the middle the early tokens' scores differ from the unmasked run
    ids = rng.integers(0, 3000, 60).astype(np.int32)
    _, a = h.Infer(ids, want_logits=True, nCacheSize=30)
    _, b = h.Infer(ids, want_logits=True, nCacheSize=0)
    assert np.abs(a[:20] - b[:20]).max() > 1e-3
    h.close()


def test_add_punc_mini_sentences(pkg, weights_mod):
    """AddPunc's id-level loop (20-token mini-sentences, carried tail, forced period beyond 200 carried tokens, final
    fix-up) against the restatement, with output biases that make sentence ends rare (long carries) or common."""
    for bias, seed in (((0.0, 3.0, 1.5, -3.0, -3.0, 0.0), 5), ((0.0, 2.0, 0.0, 1.0, 0.5, 0.0), 6), ((0.0, 4.0, 0.0, -6.0, -6.0, 0.0), 7)):
        cfg = dict(weights_mod.CT_TRANSFORMER, vocab=2000)
        man, blob = weights_mod.synth_punc_weights(cfg, seed=seed)
        W = P.Weights(man, blob)
        W["out.b"][:] = np.asarray(bias, np.float32)          # classes: unk, not-punc, comma, period, question, dun
        h = pkg.CTTransformerHip().InitPunc((man, blob))
        rng = np.random.default_rng(seed)
        for n in (1, 19, 20, 21, 75, 260, 431):
            ids = rng.integers(0, 2000, n).astype(np.int32)
            got = h.AddPuncIds(ids)
            ref = C.add_punc_ids(ids, W)
            assert list(got) == ref, (bias, n)
            assert len(ref) in (n, n + 1) and ref[-1] in (3, 4)
        h.close()


def test_batched_infer_equals_separate_calls(punc):
    """pfhip_punc_infer_batch packs sequences of different lengths (1 .. 220 tokens) into one pass: every sequence gets exactly
    the punctuation a call of its own gets — offline and with per-sequence realtime masks."""
    h, W = punc
    rng = np.random.default_rng(99)
    lens = [1, 20, 33, 7, 64, 129, 220, 2, 40]
    seqs = [rng.integers(0, 5000, n).astype(np.int32) for n in lens]
    got = h.InferBatch(seqs)
    for x, g in zip(seqs, got):
        assert np.array_equal(g, h.Infer(x))
    caches = [0, 3, 40, 2, 10, 128, 1, 5, 39]
    got = h.InferBatch(seqs, caches)
    for x, c, g in zip(seqs, caches, got):
        assert np.array_equal(g, h.Infer(x, nCacheSize=c))


def test_concurrent_infer_callers_are_merged(punc):
    import threading
    h, W = punc
    rng = np.random.default_rng(100)
    seqs = [rng.integers(0, 5000, 10 + 7 * i).astype(np.int32) for i in range(12)]
    want = [h.Infer(x) for x in seqs]
    want_on = [h.Infer(x, nCacheSize=4) for x in seqs]
    h.set_batching(2000, 16)
    got, got_on = [None] * 12, [None] * 12

    def run(i):
        got[i] = h.Infer(seqs[i])
        got_on[i] = h.Infer(seqs[i], nCacheSize=4)

    th = [threading.Thread(target=run, args=(i,)) for i in range(12)]
    [t.start() for t in th]
    [t.join() for t in th]
    h.set_batching(0, 1)
    for i in range(12):
        assert np.array_equal(got[i], want[i]) and np.array_equal(got_on[i], want_on[i])
