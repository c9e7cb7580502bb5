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
