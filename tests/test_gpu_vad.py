"""GPU: FSMN-VAD forward (pfhip_vad_*) against the oracle: whole file in one pass, slice-wise with carried caches
(the reference's 1-s slicing, audio.cpp:1183-1196), final-call cache semantics (fsmn-vad.cpp:129-134)."""
import numpy as np
import pytest

from conftest import synth_pcm
from oracle import fsmn_vad as V
from oracle import paraformer as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
TOL = 2e-5      # probabilities in [0,1]; fp32 GEMM order + GPU expf


@pytest.fixture(scope="module")
def vad(pkg, weights_mod):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    man, blob = weights_mod.synth_vad_weights()
    h = pkg.FsmnVadHip().InitVad((man, blob))
    yield h, P.Weights(man, blob)
    h.close()


def test_whole_file_one_pass(vad):
    h, W = vad
    rng = np.random.default_rng(1)
    pcm = synth_pcm(0, 16000 * 12 + 321, rng)
    ref = V.FsmnVad(W).Forward(pcm, True)
    h.InitCache()
    got = h.Forward(pcm, is_final=True)
    assert got.shape == ref.shape == (1200 - 1 + 1, 248)
    assert np.abs(got - ref).max() < TOL
    assert np.abs(got.sum(-1) - 1).max() < 1e-5


def test_slices_with_carried_caches_match_oracle(vad):
    h, W = vad
    rng = np.random.default_rng(2)
    pcm = synth_pcm(1, 16000 * 4, rng)
    o = V.FsmnVad(W)
    h.InitCache()
    for k, n in enumerate([16000, 16000, 400, 15600, 16000]):
        seg = pcm[sum([16000, 16000, 400, 15600, 16000][:k]):][:n]
        fin = k == 4
        ref = o.Forward(seg, fin)
        got = h.Forward(seg, is_final=fin)
        assert got.shape == ref.shape
        if ref.size:
            assert np.abs(got - ref).max() < TOL, k


def test_final_call_does_not_advance_caches(vad):
    h, W = vad
    rng = np.random.default_rng(3)
    a, b = synth_pcm(2, 8000, rng), synth_pcm(3, 8000, rng)
    h.InitCache()
    h.Forward(a, is_final=True)            # caches stay zero (fsmn-vad.cpp:129-134)
    got = h.Forward(b, is_final=True)
    ref = V.FsmnVad(W).Forward(b, True)
    assert np.abs(got - ref).max() < TOL
    assert h.Forward(np.zeros(399, np.float32)).shape == (0, 248)      # no full window -> no scores (:245-247)


def test_online_vad_matches_oracle_and_offline(pkg, weights_mod):
    """FsmnVadOnline (per-connection feature caches + network caches) fed in 600-ms and in ragged steps: scores and the
    waveform handed to the scorer equal the numpy restatement call by call; concatenated, the scores equal the one-pass
    offline scores except for the final call (which the reference scores against zeroed caches)."""
    man, blob = weights_mod.synth_vad_weights()
    W = P.Weights(man, blob)
    vad = pkg.FsmnVadHip().InitVad((man, blob))
    rng = np.random.default_rng(12)
    pcm = synth_pcm(3, 16000 * 4 + 1234, rng)
    for steps in ([9600] * 7, [16000, 300, 50, 4000, 23000, 9600, 12000, 3000]):
        cuts = np.cumsum([0] + steps)
        cuts = [c for c in cuts if c < len(pcm)] + [len(pcm)]
        on = pkg.FsmnVadOnlineHip(vad)
        ref = V.FsmnVadOnline(W)
        got_all = []
        for j in range(len(cuts) - 1):
            fin = j == len(cuts) - 2
            chunk = pcm[cuts[j]:cuts[j + 1]]
            sil, wv = on.InferScores(chunk, fin)
            rp, rw = ref.Infer(chunk, fin)
            assert sil.shape[0] == rp.shape[0], (j, sil.shape, rp.shape)
            assert np.array_equal(wv, rw), j
            if sil.size:
                assert np.abs(sil - rp[:, 0]).max() < 2e-5, j
            got_all.append(sil)
        allp = np.concatenate(got_all)
        off = vad.ForwardSil(pcm, is_final=True)
        assert allp.shape == off.shape
        last = len(got_all[-1])
        assert np.abs(allp[:len(allp) - last] - off[:len(off) - last]).max() < 2e-5
        on.close()
    vad.close()


def test_online_vad_batch_equals_separate_calls(pkg, weights_mod):
    """pfhip_vad_stream_infer_batch over connections at different positions (first call, mid-stream, tiny message with no
    rows, final call, empty final call) returns exactly what the same calls issued one by one return, round after round."""
    man, blob = weights_mod.synth_vad_weights()
    vad = pkg.FsmnVadHip().InitVad((man, blob))
    rng = np.random.default_rng(21)
    n = 6
    pcm = [synth_pcm(2 + i % 3, 16000 * 3 + 977 * i, rng) for i in range(n)]
    plans = [[9600] * 6, [16000, 300, 50, 4000, 23000], [3200] * 16, [200, 200, 9600, 9600, 30000], [48000], [9600] * 5 + [0]]
    cuts = []
    for i in range(n):
        c = np.minimum(np.cumsum([0] + plans[i]), len(pcm[i]))
        c = list(c)
        if i != 5 and c[-1] < len(pcm[i]):
            c.append(len(pcm[i]))
        if i == 5:
            c = list(np.minimum(np.cumsum([0] + plans[i][:-1]), len(pcm[i]))) + [len(pcm[i]), len(pcm[i])]
        cuts.append(c)
    a = [pkg.FsmnVadOnlineHip(vad) for _ in range(n)]
    b = [pkg.FsmnVadOnlineHip(vad) for _ in range(n)]
    rounds = max(len(c) - 1 for c in cuts)
    total_rows = 0
    for r in range(rounds):
        live = [i for i in range(n) if r < len(cuts[i]) - 1]
        chunks = [pcm[i][cuts[i][r]:cuts[i][r + 1]] for i in live]
        fins = [r == len(cuts[i]) - 2 for i in live]
        got = pkg.FsmnVadOnlineHip.InferScoresBatch([a[i] for i in live], chunks, fins)
        for k, i in enumerate(live):
            sil, wv = b[i].InferScores(chunks[k], fins[k])
            assert np.array_equal(got[k][1], wv), (r, i)
            assert got[k][0].shape == sil.shape, (r, i, got[k][0].shape, sil.shape)
            assert np.array_equal(got[k][0], sil), (r, i, np.abs(got[k][0] - sil).max())
            total_rows += sil.size
    assert total_rows > 1000
    with pytest.raises(RuntimeError):
        pkg.FsmnVadOnlineHip.InferScoresBatch([a[0], a[0]], [pcm[0][:1600]] * 2, [False, False])
    for x in a + b:
        x.close()
    vad.close()


def test_online_vad_threads_are_merged(pkg, weights_mod):
    """Eight connection threads calling Infer concurrently with merging on: every connection gets exactly the scores a
    lone connection gets for the same audio."""
    import threading
    man, blob = weights_mod.synth_vad_weights()
    vad = pkg.FsmnVadHip().InitVad((man, blob))
    rng = np.random.default_rng(5)
    pcm = [synth_pcm(2 + i % 2, 16000 * 2 + 500 * i, rng) for i in range(8)]

    def run(i, on, out):
        res = []
        for k in range(0, len(pcm[i]), 9600):
            res.append(on.InferScores(pcm[i][k:k + 9600], k + 9600 >= len(pcm[i]))[0])
        out[i] = np.concatenate(res)

    alone = {}
    for i in range(8):
        on = pkg.FsmnVadOnlineHip(vad)
        run(i, on, alone)
        on.close()
    vad.set_stream_batching(2000, 8)
    merged = {}
    ons = [pkg.FsmnVadOnlineHip(vad) for _ in range(8)]
    th = [threading.Thread(target=run, args=(i, ons[i], merged)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    vad.set_stream_batching(0, 1)
    for i in range(8):
        assert np.array_equal(alone[i], merged[i]), i
    for on in ons:
        on.close()
    vad.close()
