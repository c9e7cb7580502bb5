"""GPU: the guard of the fp16 two-plane domain (csrc/kernels.h LaunchCtx, pfhip.cpp fetch_locked).

The default large-batch kernels stage their operands as two fp16 planes (gemm_x3.hip / gemm_p3.hip / attention_x3.hip): fp32-grade
for values of ordinary magnitude, but |a| >= 65504 overflows and rows far below 1 lose relative precision.  The reference computes
in plain fp32 (onnxruntime/src/paraformer.cpp:496-541), so a model whose residual stream leaves that domain must still give the
reference's results: every forward carries a range flag (LayerNorm-folded rows with rms outside [2^-8, 2^11]; non-finite residual or
log-prob rows) and a flagged batch is redone once on the exact bf16 three-plane kernels inside the same handle, counted in
pfhip_debug_poke("range_fallbacks").  LayerNorm makes the network invariant to the scale of the residual stream, so the oracle
(fp32 numpy) is unaffected by the scalings below."""
import numpy as np
import pytest

from conftest import assert_ids_match, synth_pcm
from oracle import paraformer as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")


def view(man, blob, name):
    meta = man["tensors"][name]
    n = int(np.prod(meta["shape"]))
    return blob[meta["offset"] // 4: meta["offset"] // 4 + n].reshape(meta["shape"])


def utterances(n, seconds, seed):
    rng = np.random.default_rng(seed)
    return [synth_pcm(i, int(16000 * seconds) + 97 * i, rng) for i in range(n)]


def check(model, W, utts, picks, tol=1e-3):
    got = model.forward_ids(utts, want_logp=True)
    worst = 0.0
    for b in picks:
        ref = P.forward_pcm(utts[b], W)
        assert int(got["n_fires"][b]) == ref["emb"].shape[0], (b, got["n_fires"][b], ref["emb"].shape)
        assert np.isfinite(got["logp"][b]).all()
        worst = max(worst, float(np.abs(got["logp"][b] - ref["logp"]).max()))
        assert_ids_match(got["ids"][b], ref, tie_gap=tol)
    assert worst < tol, f"log-prob max abs err {worst}"
    return worst


@pytest.fixture(scope="module")
def base(weights_mod):
    cfg = weights_mod.small_config(enc_layers=3, dec_layers=1, vocab=257)
    return cfg, weights_mod.synth_weights(cfg, seed=61)


def test_default_workload_never_takes_the_fallback(pkg, base):
    """8 x 30 s (4000 rows: plane-image operands) and 6 x 22 s (the in-loop-split kernels): flag stays down, counter 0."""
    need_gpu()
    cfg, (man, blob) = base
    model = pkg.ParaformerHip().InitAsr((man, blob))
    assert model.debug_poke("always_exact") == 0 and 1 < model.debug_poke("static_bound") < 2000
    W = P.Weights(man, blob)
    check(model, W, utterances(8, 30, 1), picks=(0, 7))
    assert model.debug_poke("plane_forwards") == 1
    check(model, W, utterances(6, 22, 2), picks=(1,))
    assert model.debug_poke("range_fallbacks") == 0
    model.close()


@pytest.mark.parametrize("n_utts,seconds", [(8, 30), (6, 22), (2, 12)])
def test_residual_stream_beyond_fp16_range_is_redone_on_the_exact_kernels(pkg, weights_mod, base, n_utts, seconds):
    """Layer 0's two projections into the residual stream (attention output, FFN2) scaled by 2^18: the stream that layers 1.. read
    (LayerNorm folded into their GEMMs on large batches) holds values of 1e5-1e6, beyond fp16's 65504 — Inf / NaN on the
    two-plane kernels, and NaN does NOT reach the log-probs by itself (ReLU swallows it: fmaxf(NaN, 0) = 0), so the check sits on
    the residual rows.  With the guard: results within 1e-3 of the (scale-invariant) oracle, fallback counted; the 2 x 12 s batch
    runs the fp32-MFMA GEMMs but the split attention: no plane operand sees the stream, nothing to redo."""
    need_gpu()
    cfg, (man, blob0) = base
    blob = blob0.copy()
    for name in ("enc.0.ffn2.w", "enc.0.ffn2.b", "enc.0.out.w", "enc.0.out.b"):
        view(man, blob, name)[...] *= np.float32(2.0 ** 18)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    assert model.debug_poke("always_exact") == 0 and model.debug_poke("static_bound") < 2000
    utts = utterances(n_utts, seconds, 3)
    check(model, P.Weights(man, blob), utts, picks=(0, n_utts - 1))
    assert (model.debug_poke("range_fallbacks") >= 1) == (n_utts > 2)
    model.close()


def test_weights_that_can_overflow_the_planes_put_the_model_on_the_exact_kernels(pkg, base):
    """The value rows of layer 0's QKV projection scaled by 2^18: v — an operand of the split attention — reaches 1e5-1e6 whatever
    the input.  The load-time bound sqrt(K) ||w_n o gamma|| + |b_n + w_n . beta| sees that, and the handle runs the bf16 three-plane
    kernels for every forward (no re-run needed, none counted)."""
    need_gpu()
    cfg, (man, blob0) = base
    blob = blob0.copy()
    view(man, blob, "enc.0.qkv.w")[2 * 512:] *= np.float32(2.0 ** 18)
    view(man, blob, "enc.0.qkv.b")[2 * 512:] *= np.float32(2.0 ** 18)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    assert model.debug_poke("always_exact") == 1 and model.debug_poke("static_bound") > 65504
    check(model, P.Weights(man, blob), utterances(8, 30, 6), picks=(0, 4))
    check(model, P.Weights(man, blob), utterances(2, 9, 7), picks=(1,))
    assert model.debug_poke("range_fallbacks") == 0 and model.debug_poke("plane_forwards") == 0
    model.close()


def test_rows_far_below_one_are_redone_on_the_exact_kernels(pkg, base):
    """Everything layer 0 writes into the residual stream scaled by 2^-18: rows of rms ~1e-5, where the low fp16 plane is
    subnormal and the two planes keep ~8 bits.  LayerNorm rescales such rows to O(1) — in fp32 exactly, on the planes not."""
    need_gpu()
    cfg, (man, blob0) = base
    blob = blob0.copy()
    eps = np.float32(2.0 ** -18)
    for name in ("enc.0.ffn2.w", "enc.0.ffn2.b", "enc.0.out.w", "enc.0.out.b"):
        view(man, blob, name)[...] *= eps
    view(man, blob, "enc.0.qkv.w")[2 * 512:] *= eps
    view(man, blob, "enc.0.qkv.b")[2 * 512:] *= eps
    model = pkg.ParaformerHip().InitAsr((man, blob))
    check(model, P.Weights(man, blob), utterances(8, 30, 4), picks=(0, 5))
    assert model.debug_poke("range_fallbacks") >= 1
    model.close()


def test_a_raised_flag_redoes_the_batch_once_with_the_same_results(pkg, base):
    """The mechanism on its own (pfhip_debug_poke "range_flag"): the next forward finds its flag raised, is redone on the exact
    kernels — including under cross-request merging and on every execution context — and gives the oracle's results."""
    need_gpu()
    cfg, (man, blob) = base
    model = pkg.ParaformerHip().InitAsr((man, blob))
    W = P.Weights(man, blob)
    utts = utterances(8, 30, 5)
    assert model.debug_poke("range_flag", 1) == 0
    check(model, W, utts, picks=(3,))
    assert model.debug_poke("range_fallbacks") == 1 and model.debug_poke("plane_forwards") == 1       # the re-run is not a plane forward
    check(model, W, utts, picks=(3,))
    assert model.debug_poke("range_fallbacks") == 1                                                   # one forward only
    model.close()
