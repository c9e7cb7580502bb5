"""GPU: the chunk-streaming path (pfhip_stream_*) against the streaming oracle, call by call."""
import numpy as np
import pytest

from conftest import synth_pcm
from oracle import paraformer as P
from oracle import paraformer_online as PO

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def small(pkg, weights_mod):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    cfg = weights_mod.small_config(enc_layers=3, dec_layers=2, vocab=517)
    man, blob = weights_mod.synth_weights(cfg, seed=77)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    yield pkg, model, P.Weights(man, blob)
    model.close()


def run_both(pkg, model, W, pcm, steps):
    """steps: list of (n_samples, input_finished).  Returns per-call (oracle ids, hip ids) and checks tensors."""
    on = PO.ParaformerOnline(W)
    hip = pkg.ParaformerOnlineHip(model)
    hip.set_debug(True)
    pos = 0
    for n, fin in steps:
        seg = pcm[pos:pos + n]
        pos += n
        before = len(on.chunk_log)
        ref_ids = on.Forward(seg, fin)
        got_ids = hip.Forward(seg, input_finished=fin)
        assert got_ids == ref_ids, (n, fin, got_ids, ref_ids)
        if len(on.chunk_log) > before:
            last = on.chunk_log[-1]
            nrow = last["feats"].shape[0]
            chunk = hip.get_tensor("chunk", 128 * 560).reshape(-1, 560)
            assert chunk.shape[0] == nrow
            assert np.abs(chunk - last["feats"]).max() < 2e-3          # rows are x sqrt(512) ~ 22.6 * 2e-5
            enc = hip.get_tensor("enc", 128 * 512).reshape(-1, 512)
            assert np.abs(enc - last["enc"]).max() < 1e-3
            alphas = hip.get_tensor("alphas", 128)
            assert np.abs(alphas - last["alphas"]).max() < 1e-4
            if last["logp"] is not None:
                logp = hip.get_tensor("logp", 128 * W.cfg["vocab"]).reshape(-1, W.cfg["vocab"])
                assert logp.shape == last["logp"].shape
                assert np.abs(logp - last["logp"]).max() < 1e-3       # BASELINE.json tolerance
    hip.close()
    return on


def test_stream_600ms_steps_then_final_full_chunk(small):
    pkg, model, W = small
    rng = np.random.default_rng(11)
    pcm = synth_pcm(3, 9600 * 6, rng)
    on = run_both(pkg, model, W, pcm, [(9600, False)] * 5 + [(9600, True)])
    assert sum(len(c["ids"]) for c in on.chunk_log) > 0


def test_stream_short_final_flush_and_restart(small):
    pkg, model, W = small
    rng = np.random.default_rng(12)
    pcm = synth_pcm(4, 9600 * 3 + 700 + 9600 * 2, rng)
    # 3 chunks, a 700-sample final (flushes the look-back cache, :532-540), then a new utterance on the same stream
    run_both(pkg, model, W, pcm, [(9600, False)] * 3 + [(700, True), (9600, False), (9600, True)])


def test_stream_final_partial_chunk(small):
    pkg, model, W = small
    rng = np.random.default_rng(13)
    pcm = synth_pcm(5, 9600 * 2 + 4000, rng)
    run_both(pkg, model, W, pcm, [(9600, False), (9600, False), (4000, True)])       # nr + 5 <= 10 branch (:557-559)


def test_stream_irregular_steps(small):
    pkg, model, W = small
    rng = np.random.default_rng(14)
    pcm = synth_pcm(6, 60000, rng)
    run_both(pkg, model, W, pcm, [(8000, False), (12345, False), (9600, False), (16000, False), (14055, True)])


def test_two_streams_do_not_interfere(small):
    pkg, model, W = small
    rng = np.random.default_rng(15)
    a, b = synth_pcm(7, 9600 * 3, rng), synth_pcm(8, 9600 * 3, rng)
    ra, rb = PO.ParaformerOnline(W), PO.ParaformerOnline(W)
    ha, hb = pkg.ParaformerOnlineHip(model), pkg.ParaformerOnlineHip(model)
    for k in range(3):
        fin = k == 2
        sa, sb = a[k * 9600:(k + 1) * 9600], b[k * 9600:(k + 1) * 9600]
        assert ha.Forward(sa, input_finished=fin) == ra.Forward(sa, fin)
        assert hb.Forward(sb, input_finished=fin) == rb.Forward(sb, fin)
    ha.close()
    hb.close()


def test_graph_replay_equals_eager(small):
    """The hipGraph path (default) and the eager path (set_debug bit 1) must emit identical tokens, chunk by chunk,
    incl. graph re-use across equal window shapes and the final first/last-chunk split."""
    pkg, model, W = small
    rng = np.random.default_rng(21)
    pcm = synth_pcm(9, 9600 * 8 + 3000, rng)
    g = pkg.ParaformerOnlineHip(model)
    e = pkg.ParaformerOnlineHip(model)
    e.set_debug(2)
    ref = PO.ParaformerOnline(W)
    steps = [(9600, False)] * 8 + [(3000, True)]
    pos = 0
    total = 0
    for n, fin in steps:
        seg = pcm[pos:pos + n]
        pos += n
        a = g.Forward(seg, input_finished=fin)
        b = e.Forward(seg, input_finished=fin)
        r = ref.Forward(seg, fin)
        assert a == b == r
        total += len(a)
    assert total > 0
    g.close()
    e.close()


def test_batched_connections_equal_separate_calls(pkg, weights_mod):
    """pfhip_stream_forward_batch: N connections at different positions of their streams (different lengths, one ending
    early with a short flush, one with a final call that splits into two windows) give exactly the ids of N separate
    ParaformerOnline objects fed alone."""
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=2, vocab=500)
    man, blob = weights_mod.synth_weights(cfg, seed=21)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    rng = np.random.default_rng(8)
    lens = [9600 * 7, 9600 * 4 + 1234, 9600 * 9 + 5000, 9600 * 2 + 300, 9600 * 5]
    waves = [synth_pcm(i, n, rng) for i, n in enumerate(lens)]

    def feed_plan(n):
        """600-ms steps; the tail (whatever its length) goes with input_finished."""
        steps = [(k, min(k + 9600, n)) for k in range(0, n, 9600)]
        return steps

    # reference: each connection alone
    alone = []
    for w in waves:
        s = pkg.ParaformerOnlineHip(model)
        ids = []
        plan = feed_plan(len(w))
        for j, (a, b) in enumerate(plan):
            ids += s.Forward(w[a:b], input_finished=(j == len(plan) - 1))
        alone.append(ids)
        s.close()
    # batched: all connections advance together; finished ones drop out
    streams = [pkg.ParaformerOnlineHip(model) for _ in waves]
    plans = [feed_plan(len(w)) for w in waves]
    got = [[] for _ in waves]
    for j in range(max(len(p) for p in plans)):
        act = [i for i, p in enumerate(plans) if j < len(p)]
        res = pkg.ParaformerOnlineHip.forward_batch([streams[i] for i in act], [waves[i][plans[i][j][0]:plans[i][j][1]] for i in act],
                                                    [j == len(plans[i]) - 1 for i in act])
        for i, r in zip(act, res):
            got[i] += r
    assert got == alone
    assert sum(len(x) for x in alone) > 10
    # a stream must not appear twice; streams of different models must not mix
    with pytest.raises(pkg.PfhipError):
        pkg.ParaformerOnlineHip.forward_batch([streams[0], streams[0]], [waves[0][:9600]] * 2, [False, False])
    for s in streams:
        s.close()
    model.close()


def test_threads_per_connection_are_merged(pkg, weights_mod):
    """pfhip_set_stream_batching: one thread per connection, each calling the per-connection Forward (the 2-pass server's
    shape); concurrent calls are merged into batched forwards and every connection still gets exactly its own ids."""
    import threading
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=2, vocab=500)
    man, blob = weights_mod.synth_weights(cfg, seed=23)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    rng = np.random.default_rng(9)
    waves = [synth_pcm(i, 9600 * (4 + i % 3) + 111 * i, rng) for i in range(12)]

    def feed(stream, w, out):
        steps = list(range(0, len(w), 9600))
        for j, a in enumerate(steps):
            out += stream.Forward(w[a:a + 9600], input_finished=(j == len(steps) - 1))

    alone = []
    for w in waves:
        s = pkg.ParaformerOnlineHip(model)
        ids = []
        feed(s, w, ids)
        alone.append(ids)
        s.close()
    model.set_stream_batching(3000, 64)
    streams = [pkg.ParaformerOnlineHip(model) for _ in waves]
    got = [[] for _ in waves]
    ths = [threading.Thread(target=feed, args=(streams[i], waves[i], got[i])) for i in range(len(waves))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert got == alone
    model.set_stream_batching(0, 1)
    for s in streams:
        s.close()
    model.close()


def test_one_small_token_buffer_does_not_fail_the_other_connections(pkg, weights_mod):
    """Merged streaming calls carry one status per connection (round-1 advisor finding): a connection whose token buffer
    is too small gets PFHIP_ERR_CAPACITY and the count it needed; the connections merged with it get their ids."""
    import threading
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=2, vocab=500)
    man, blob = weights_mod.synth_weights(cfg, seed=23)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    rng = np.random.default_rng(9)
    waves = [synth_pcm(i, 9600 * 4, rng) for i in range(4)]
    alone = []
    for w in waves:
        s = pkg.ParaformerOnlineHip(model)
        alone.append([s.Forward(w[a:a + 9600], input_finished=(a + 9600 >= len(w))) for a in range(0, len(w), 9600)])
        s.close()
    step = next(j for j in range(4) if all(len(a[j]) > 0 for a in alone))      # a round in which every connection emits tokens
    # explicit batch call
    streams = [pkg.ParaformerOnlineHip(model) for _ in waves]
    for j in range(4):
        caps = [256, 0, 256, 256] if j == step else [256] * 4
        st, ids, nt = pkg.ParaformerOnlineHip.forward_batch(streams, [w[9600 * j:9600 * (j + 1)] for w in waves], [j == 3] * 4, caps)
        if j == step:
            assert st == 5                                   # PFHIP_ERR_CAPACITY
            assert nt[1] == len(alone[1][j]) and ids[1] == []
        else:
            assert st == 0 and ids[1] == alone[1][j]
        for i in (0, 2, 3):
            assert ids[i] == alone[i][j], (i, j)
    for s in streams:
        s.close()
    # one thread per connection, merged by the library
    model.set_stream_batching(20000, 4)
    streams = [pkg.ParaformerOnlineHip(model) for _ in waves]
    got = [[None] * 4 for _ in waves]
    errs = [None] * 4
    barrier = threading.Barrier(4)

    def feed(i):
        for j in range(4):
            barrier.wait()
            try:
                got[i][j] = streams[i].Forward(waves[i][9600 * j:9600 * (j + 1)], input_finished=(j == 3),
                                               cap=0 if (i == 1 and j == step) else 256)
            except pkg.PfhipError as e:
                errs[i] = (j, str(e))
    ths = [threading.Thread(target=feed, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert errs[0] is None and errs[2] is None and errs[3] is None
    assert errs[1] is not None and errs[1][0] == step and "token_ids too small" in errs[1][1]
    for i in (0, 2, 3):
        assert got[i] == alone[i]
    assert [g for j, g in enumerate(got[1]) if j != step] == [a for j, a in enumerate(alone[1]) if j != step]
    model.set_stream_batching(0, 1)
    for s in streams:
        s.close()
    model.close()


def test_full_size_stream_matches_oracle(pkg, weights_mod):
    """BASELINE.json configs[2] at the model's real size: Paraformer-large-sized random-init weights (50 encoder / 16 decoder layers,
    vocabulary 8404), one connection fed 600-ms chunks — the latency path with every window-sized launch shape of the deployed model
    (one-trip GEMVs over K = 512 / 2048, the 16-layer K/V projection, the vocabulary projection) — against the streaming oracle call by
    call: ids identical, window rows / encoder output / alphas / log-probs within the tolerances of the small-model tests above."""
    cfg = dict(weights_mod.PARAFORMER_LARGE)
    man, blob = weights_mod.synth_weights(cfg, seed=1234)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    W = P.Weights(man, blob)
    rng = np.random.default_rng(21)
    pcm = synth_pcm(9, 9600 * 5 + 3000, rng)
    on = run_both(pkg, model, W, pcm, [(9600, False)] * 5 + [(3000, True)])
    assert sum(len(c["ids"]) for c in on.chunk_log) > 0
    model.close()


def test_large_round_of_connections_with_strong_layernorm(pkg, weights_mod):
    """A round of 90 connections is 1800 rows: the batched encoder takes the LayerNorm-folded GEMMs (>= 1536 rows: 64-row tiles of the
    BF16-split kernels, row statistics from the producing epilogue), the (head, connection) window-attention launch that also writes
    the FSMN memory, and the decoder's per-connection cached FSMN.  gamma in [0.5, 1.5], beta in [-0.5, 0.5] for every LayerNorm.
    Three of the connections against the streaming oracle (ids identical, log-probs within 1e-3), all of them against the same
    connection fed alone (ids identical) over three rounds, the last one final."""
    cfg = weights_mod.small_config(enc_layers=3, dec_layers=2, vocab=600)
    man, blob = weights_mod.synth_weights(cfg, seed=31)
    rng = np.random.default_rng(6)
    for name, t in man["tensors"].items():
        if "norm" in name and (name.endswith(".g") or name.endswith(".b")):
            o, n = t["offset"] // 4, int(np.prod(t["shape"]))
            blob[o:o + n] = (rng.uniform(0.5, 1.5, n) if name.endswith(".g") else rng.uniform(-0.5, 0.5, n)).astype(np.float32)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    W = P.Weights(man, blob)
    B, rounds = 90, 3
    waves = [synth_pcm(200 + i, 9600 * rounds, rng) for i in range(B)]
    streams = [pkg.ParaformerOnlineHip(model) for _ in range(B)]
    got = [[] for _ in range(B)]
    for k in range(rounds):
        res = pkg.ParaformerOnlineHip.forward_batch(streams, [w[k * 9600:(k + 1) * 9600] for w in waves], [k == rounds - 1] * B)
        for i, r in enumerate(res):
            got[i] += r
    for s in streams:
        s.close()
    assert sum(len(g) for g in got) > B
    for i in range(B):
        s = pkg.ParaformerOnlineHip(model)
        ids = []
        for k in range(rounds):
            ids += s.Forward(waves[i][k * 9600:(k + 1) * 9600], input_finished=(k == rounds - 1))
        s.close()
        assert ids == got[i], i
    for i in (0, 44, 89):
        on = PO.ParaformerOnline(W)
        ids = []
        for k in range(rounds):
            ids += on.Forward(waves[i][k * 9600:(k + 1) * 9600], k == rounds - 1)
        assert ids == got[i], i
    model.close()
