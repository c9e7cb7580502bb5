"""GPU: the encoder attention on K | V handed over as row-major fp16 planes (csrc/attention_p3.hip) and the QKV
projection's epilogue that writes them (gemm_p3.hip, OUT = 5), through the operator-level C ABI.

The planes are hi = fp16_rtz(x), lo = fp16_rn(x - hi) — exactly what attention_x3.hip computes for itself while it stages fp32 K / V — and
the new kernel keeps that kernel's MFMA order, so on planes split from the same fp32 K / V the two contexts must agree BIT FOR BIT;
against the fp64 restatement of the reference's MatMul-Softmax-MatMul (oracle P.mha; onnxruntime graph of paraformer.cpp:496-541) the
tolerance is attention_x3's 2e-5."""
import importlib

import numpy as np
import pytest

from oracle import paraformer as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

H, DK = 4, 128
D = H * DK


@pytest.fixture(scope="module")
def ops(pkg):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    return importlib.import_module("asr_2pass_amd.ops")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def offsets(lens):
    return np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)


def kv_planes(ops, K, V):
    """[R, 2 D] planes: K in columns 0 .. D - 1, V in D .. 2 D - 1."""
    out = ops.split_rows(dev(K), ldp=2 * D)
    return ops.split_rows(dev(V), ldp=2 * D, out=out, col=D)


def test_split_rows_is_the_two_plane_split(ops):
    """hi = round-toward-zero fp16, lo = round-to-nearest fp16 of the remainder: hi + lo within 2^-22 |x|, |hi| <= |x|."""
    rng = np.random.default_rng(1)
    X = (rng.standard_normal((77, 256)) * np.exp(rng.uniform(-6, 6, (77, 256)))).astype(np.float32)
    hi, lo = ops.split_rows(dev(X))
    hi, lo = hi.cpu().numpy(), lo.cpu().numpy()
    assert (np.abs(hi.astype(np.float32)) <= np.abs(X)).all()
    assert (np.abs(hi.astype(np.float64) + lo.astype(np.float64) - X) <= 2.0 ** -21 * np.abs(X) + 2.0 ** -25).all()
    # the high plane is the truncation: the next fp16 away from zero is beyond x
    up = np.nextafter(hi, np.where(X >= 0, np.float16(np.inf), np.float16(-np.inf)).astype(np.float16))
    nz = hi.astype(np.float32) != X
    assert (np.abs(up.astype(np.float32))[nz] > np.abs(X)[nz]).all()


@pytest.mark.parametrize("q_lens,kv_lens,spike", [
    ([1, 31, 32, 33, 100, 129, 500], None, False),            # self-attention, ragged: tails of every kind, two query blocks
    ([70, 300, 257, 500, 65, 256], None, False),
    ([3, 120, 1, 40, 300], [17, 500, 64, 33, 31], False),     # cross-attention shapes
    ([140], [200], True),                                     # the lazy-rescale branch
])
def test_context_equals_the_fp32_operand_kernel_bit_for_bit(ops, q_lens, kv_lens, spike):
    rng = np.random.default_rng(len(q_lens) * 7 + (3 if spike else 0))
    kv_lens = q_lens if kv_lens is None else kv_lens
    q_off, kv_off = offsets(q_lens), offsets(kv_lens)
    Q = rng.standard_normal((sum(q_lens), D)).astype(np.float32)
    K = rng.standard_normal((sum(kv_lens), D)).astype(np.float32)
    V = rng.standard_normal((sum(kv_lens), D)).astype(np.float32)
    if spike:
        K[kv_lens[0] - 2, :DK] = Q[1, :DK] * 4
    ql, kl = dev(np.asarray(q_lens, np.int32)), dev(np.asarray(kv_lens, np.int32))
    ref_kernel = ops.attention(dev(Q), dev(K), dev(V), dev(q_off), ql, dev(kv_off), kl, H, DK ** -0.5).cpu().numpy()
    got = ops.attention_kvplanes(dev(Q), kv_planes(ops, K, V), D, dev(q_off), ql, dev(kv_off), kl, H, DK ** -0.5).cpu().numpy()
    assert np.array_equal(got, ref_kernel), np.abs(got - ref_kernel).max()
    for b in range(len(q_lens)):
        q = Q[q_off[b]:q_off[b] + q_lens[b]].astype(np.float64)
        k = K[kv_off[b]:kv_off[b] + kv_lens[b]].astype(np.float64)
        v = V[kv_off[b]:kv_off[b] + kv_lens[b]].astype(np.float64)
        err = np.abs(got[q_off[b]:q_off[b] + q_lens[b]] - P.mha(q, k, v, H)).max()
        assert err < 2e-5, (b, err)


def test_memory_block_and_plane_image_output(ops):
    """The fused form the encoder uses: SAN-M memory of V (= hi + lo) added into the residual stream, context out as the plane images of
    gemm_p3.hip.  Memory: bit-identical to the stand-alone fsmn kernel on the values hi + lo; images: the fp32 context to 22 bits."""
    rng = np.random.default_rng(5)
    lens = [70, 300, 129, 500, 65]
    off = offsets(lens)
    M = sum(lens)
    Q = rng.standard_normal((M, D)).astype(np.float32)
    K = rng.standard_normal((M, D)).astype(np.float32)
    V = rng.standard_normal((M, D)).astype(np.float32)
    X = rng.standard_normal((M, D)).astype(np.float32)
    w = (rng.standard_normal((D, 11)) / 3).astype(np.float32)
    dl, doff = dev(np.asarray(lens, np.int32)), dev(off)
    kv = kv_planes(ops, K, V)
    Vp = (kv[0][:, D:].float() + kv[1][:, D:].float()).contiguous()          # the values the kernel sees
    assert np.abs(Vp.cpu().numpy() - V).max() <= 2.0 ** -21 * np.abs(V).max()
    O = ops.attention_kvplanes(dev(Q), kv, D, doff, dl, doff, dl, H, DK ** -0.5).cpu().numpy()
    for acc in (False, True):
        mem = dev(X.copy())
        hi, lo, rows = ops.attention_kvplanes(dev(Q), kv, D, doff, dl, doff, dl, H, DK ** -0.5, fsmn_w=dev(w), mem=mem, mem_accumulate=acc,
                                              want_planes=True)
        want = ops.fsmn(Vp, dev(w), doff, dl, res=dev(X) if acc else None).cpu().numpy()
        got_mem = mem.cpu().numpy()
        if acc:      # x + (v + conv) against (v + conv) + x: one rounding apart
            assert np.abs(got_mem - want).max() < 1e-5
        else:
            assert np.array_equal(got_mem, want)
        for o, L in zip(off, lens):
            ref = P.fsmn(V[o:o + L], w) + (X[o:o + L] if acc else 0)
            assert np.abs(got_mem[o:o + L] - ref).max() < 1e-5
        ctx = ops.planes_to_float(hi, lo, rows, D)
        assert np.abs(ctx[:M] - O).max() <= 2.0 ** -22 * np.abs(O).max() + 2.0 ** -24
        assert np.abs(ctx[M:]).max() == 0


@pytest.mark.parametrize("M,tile_rows,ln", [(1000, 128, True), (1000, 64, True), (300, 0, False), (4100, 0, True)])
def test_qkv_projection_leaves_q_as_fp32_and_kv_as_planes(ops, M, tile_rows, ln):
    """gemm_p3 OUT = 5 against the same GEMM with fp32 output: the Q columns are the same floats, the K | V planes are the split of the
    same floats (bit for bit), rows beyond M untouched."""
    rng = np.random.default_rng(M + tile_rows)
    K_, N = 512, 3 * D
    A = rng.standard_normal((M, K_)).astype(np.float32)
    W = (rng.standard_normal((N, K_)) / np.sqrt(K_)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ws = ops.best_w_scale(float(np.abs(W).max()))
    a_img, w_img = ops.split_planes(dev(A)), ops.split_planes(dev(W), scale=ws)
    kw = {}
    if ln:      # LayerNorm folded in: per-row (mean, M2) of four 128-column tiles, colsum of the weights
        Mp = (M + 127) // 128 * 128
        stats = np.zeros((Mp, 4, 2), np.float32)
        t = A.reshape(M, 4, 128).astype(np.float64)
        stats[:M, :, 0] = t.mean(2)
        stats[:M, :, 1] = ((t - t.mean(2, keepdims=True)) ** 2).sum(2)
        kw = dict(ln_stats=dev(stats), ln_tiles=4, ln_colsum=dev(W.astype(np.float64).sum(1).astype(np.float32)))
    C_ref, _ = ops.gemm_p3(a_img, w_img, M, N, K_, w_scale=ws, bias=dev(bias), tile_rows=tile_rows, **kw)
    C, (kvh, kvl) = ops.gemm_p3_qkv(a_img, w_img, M, N, K_, D, w_scale=ws, bias=dev(bias), tile_rows=tile_rows, **kw)
    C_ref = C_ref.cpu().numpy()
    assert np.array_equal(C.cpu().numpy()[:M], C_ref[:M, :D])
    want_h, want_l = ops.split_rows(dev(C_ref[:M, D:]))
    assert np.array_equal(kvh.cpu().numpy()[:M].view(np.uint16), want_h.cpu().numpy().view(np.uint16))
    assert np.array_equal(kvl.cpu().numpy()[:M].view(np.uint16), want_l.cpu().numpy().view(np.uint16))
    assert not kvh.cpu().numpy()[M:].any() and not kvl.cpu().numpy()[M:].any() and not C.cpu().numpy()[M:].any()
    if not ln:
        ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
        got = np.concatenate([C.cpu().numpy()[:M], kvh.cpu().numpy()[:M].astype(np.float64) + kvl.cpu().numpy()[:M].astype(np.float64)], 1)
        assert np.abs(got - ref).max() < 3e-5


def test_refuses_shapes_the_kernels_do_not_take(ops):
    rng = np.random.default_rng(2)
    Q = dev(rng.standard_normal((100, D)).astype(np.float32))
    kv = kv_planes(ops, rng.standard_normal((100, D)).astype(np.float32), rng.standard_normal((100, D)).astype(np.float32))
    off, ln = dev(np.zeros(1, np.int32)), dev(np.asarray([100], np.int32))
    with pytest.raises(RuntimeError):
        ops.attention_kvplanes(Q, kv, D - 8, off, ln, off, ln, H, 1.0)        # V overlapping K
    with pytest.raises(RuntimeError):
        ops.attention_kvplanes(Q, kv, 2 * D, off, ln, off, ln, H, 1.0)        # V beyond the row
    a_img, w_img = ops.split_planes(Q[:, :64].contiguous()), ops.split_planes(dev(rng.standard_normal((384, 64)).astype(np.float32)))
    with pytest.raises(RuntimeError):
        ops.gemm_p3_qkv(a_img, w_img, 100, 384, 64, 100)                     # q_cols not a whole tile
    with pytest.raises(RuntimeError):
        ops.gemm_p3_qkv(a_img, w_img, 100, 384, 64, 384)                     # nothing left for K | V


def test_model_forward_with_the_plane_hand_off(pkg, weights_mod, monkeypatch):
    """PFHIP_KV_PLANES=1 (opt-in): layers 1.. of the encoder run QKV -> row-major K | V planes -> attention_p3 on a batch that takes the
    plane path (8 x 30 s = 4000 rows).  Same token ids as the default fp32 hand-off, log-probs within 2e-5 of it (the attention is
    bit-identical; the memory block reads V as hi + lo = 22-23 bits) and within north_star's 1e-3 of the CPU restatement."""
    from conftest import assert_ids_match, synth_pcm
    cfg = weights_mod.small_config(enc_layers=3, dec_layers=1, vocab=257)
    man, blob = weights_mod.synth_weights(cfg, seed=61)
    rng = np.random.default_rng(1)
    utts = [synth_pcm(i, 480000 + 97 * i, rng) for i in range(8)]
    model = pkg.ParaformerHip().InitAsr((man, blob))
    monkeypatch.delenv("PFHIP_KV_PLANES", raising=False)
    base = model.forward_ids(utts, want_logp=True)
    assert model.debug_poke("plane_forwards") == 1 and model.debug_poke("kvplane_forwards") == 0
    monkeypatch.setenv("PFHIP_KV_PLANES", "1")
    got = model.forward_ids(utts, want_logp=True)
    assert model.debug_poke("plane_forwards") == 2 and model.debug_poke("kvplane_forwards") == 1
    for b in range(len(utts)):
        assert list(got["ids"][b]) == list(base["ids"][b])
        assert np.abs(got["logp"][b] - base["logp"][b]).max() < 2e-5
    W = P.Weights(man, blob)
    for b in (0, 7):
        ref = P.forward_pcm(utts[b], W)
        assert np.abs(got["logp"][b] - ref["logp"]).max() < 1e-3
        assert_ids_match(got["ids"][b], ref, tie_gap=1e-3)
    model.close()
