"""GPU: model directories in the REFERENCE's layout (onnxruntime/include/com-define.h:52-88: model.onnx [+ decoder.onnx,
model_eb.onnx], am.mvn, config.yaml, tokens.json; written by tests/ref_layout.py as the PyTorch exporter leaves them) loaded in
C++ with the strings the reference itself passes, through the HIP forward, against the oracle on the ORIGINAL tensors
(SURVEY §8 rows b, f2, f3).

  * the C ABI: pfhip_create_from_files(<dir>/model.onnx | model.torchscript | model_quant.onnx, model_eb.onnx, am.mvn, config.yaml)
    -> ids identical, log-probs within 1e-3 of oracle.forward_pcm on the tensors the files were written from;
  * the C++ adapters behind the handle-API mirror: `offline_infer` makes OfflineStream::OfflineStream's calls
    (offline-stream.cpp:4-129: InitVad, InitHwCompiler BEFORE InitAsr, the five-argument InitAsr with model.torchscript under
    use_gpu), `tpass_infer` makes TpassStream::TpassStream's (tpass-stream.cpp:4-135: the NINE-argument InitAsr) and
    TpassOnlineStream's; their per-segment / per-call token ids equal the oracle's (offline) and the streaming oracle's (online).
The acoustic weights are synthetic (no real Paraformer file exists offline): what is pinned here is the file contract and the
loader, not the upstream layer names."""
import importlib
import json
import os
import subprocess

import numpy as np
import pytest
import torch

import ref_layout as RL
from conftest import assert_ids_match, synth_pcm
from oracle import audio_split as A
from oracle import paraformer as P
from oracle import paraformer_online as PO
from test_gpu_pipeline import make_file, shape_vad_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def conv(pkg):
    return importlib.import_module(pkg.__name__ + ".convert")


def need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")


@pytest.mark.parametrize("heads,name", [(dict(), "model.onnx"), (dict(), "model.torchscript"), (dict(contextual=1, timestamp=1), "model.onnx")])
def test_c_abi_loads_the_reference_files_and_matches_the_oracle(pkg, weights_mod, conv, tmp_path, heads, name):
    need_gpu()
    cfg = weights_mod.small_config(enc_layers=3, dec_layers=2, vocab=211, **heads)
    man, blob = weights_mod.synth_weights(cfg, seed=41)
    d = tmp_path / "asr"
    RL.write_asr_dir(str(d), conv, man, blob, cfg)
    hw = str(d / "model_eb.onnx") if heads else None
    model = pkg.ParaformerHip().InitAsr(str(d / name), str(d / "am.mvn"), str(d / "config.yaml"), str(d / "tokens.json"), hw_model=hw)
    assert model.cfg["vocab"] == 211 and model.cfg.get("contextual", 0) == heads.get("contextual", 0)
    rng = np.random.default_rng(5)
    utts = [synth_pcm(i, n, rng) for i, n in enumerate((16000 * 4, 16000 * 7 + 123, 16000 * 2 - 77))]
    W = P.Weights(man, blob)                                   # the ORIGINAL tensors, not what came back from the files
    hw_emb = None
    if heads:                                                  # model_eb.onnx's tensors: the embedder against the oracle's
        hot = [[5, 9], [17], [3, 4, 6, 8]]
        hw_emb = model.CompileHotwordEmbedding(hot)
        rows = [h + [0] * (10 - len(h)) for h in hot] + [[1] + [0] * 9]
        assert np.abs(hw_emb - P.hotword_embed(rows, [len(h) for h in hot] + [1], W)).max() < 2e-5
    got = model.forward_ids(utts, want_logp=True, hw_emb=hw_emb) if heads else model.forward_ids(utts, want_logp=True)
    for b, u in enumerate(utts):
        ref = P.forward_pcm(u, W, hw_emb=hw_emb) if heads else P.forward_pcm(u, W)
        assert int(got["n_fires"][b]) == ref["emb"].shape[0] and int(got["token_num"][b]) == ref["token_num"]
        err = float(np.abs(got["logp"][b] - ref["logp"]).max())
        assert err < 1e-3, f"utterance {b}: log-prob max abs err {err}"
        assert_ids_match(got["ids"][b], ref)
    model.close()
    assert os.path.exists(d / "model.pfhip.bin")               # the converted container was cached beside the source


def test_offline_stream_calls_on_a_reference_directory(pkg, weights_mod, conv, tmp_path):
    """`offline_infer` = FunOfflineInit (OfflineStream's calls on ParaformerHip / FsmnVadHip, use_gpu = true: the adapter is handed
    <dir>/model.torchscript) -> FunOfflineInferBuffer.  Directories hold ONNX files only.  Per VAD segment the ids equal the
    oracle's on the original tensors; text = Vocab::Vector2StringV2 of tokens.json; merged threads give the same."""
    need_gpu()
    rng = np.random.default_rng(13)
    pcm = make_file(rng)
    s16 = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2")
    f32 = (s16.astype(np.float32) / 32768.0).astype(np.float32)
    vman, vblob = shape_vad_weights(*weights_mod.synth_vad_weights())
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300)
    aman, ablob = weights_mod.synth_weights(cfg, seed=43)
    vocab = [chr(0x4E00 + i) for i in range(297)] + ["<s>", "</s>", "<unk>"]
    RL.write_asr_dir(str(tmp_path / "asr"), conv, aman, ablob, cfg, vocab_tokens=vocab)
    RL.write_vad_dir(str(tmp_path / "vad"), conv, vman, vblob)
    s16.tofile(tmp_path / "long.pcm")
    exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "offline_infer")
    out = subprocess.run([exe, str(tmp_path / "asr"), str(tmp_path / "vad"), str(tmp_path / "long.pcm"), "4", "3", "2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    segs = [l for l in out.stdout.splitlines() if l.startswith("seg ")]
    assert len(segs) == 5
    W = P.Weights(aman, ablob)
    text = ""
    for l in segs:
        head, _, tail = l.partition(":")
        s, e = (int(x) for x in head.split()[1:3])
        ref = P.forward_pcm(f32[s:e], W)
        ids = [int(x) for x in tail.split()]
        assert_ids_match(ids, ref)
        text += "".join(vocab[i] for i in ids if vocab[i] not in ("<s>", "</s>", "<unk>"))
    assert [l for l in out.stdout.splitlines() if l.startswith("text ")][0][5:] == text


def test_tpass_stream_calls_on_reference_directories(pkg, weights_mod, conv, tmp_path):
    """`tpass_infer` = FunTpassInit (TpassStream's calls: InitVad, the NINE-argument InitAsr with <offline>/model.onnx,
    <online>/model.onnx, <online>/decoder.onnx, <online>/am.mvn, <offline>/config.yaml, tokens, <online>/config.yaml) +
    FunTpassOnlineInit (ParaformerOnlineHip(asr_handle, chunk_size)) + FunTpassInferBuffer per 600-ms message.  Streaming ids per
    call equal the streaming ORACLE's on the chunks Audio::Split cuts; every closed segment's 2nd-pass ids equal the offline
    oracle's — both on the original tensors."""
    need_gpu()
    rng = np.random.default_rng(21)
    pcm = make_file(rng)[:16000 * 30]
    s16 = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2")
    vman, vblob = shape_vad_weights(*weights_mod.synth_vad_weights())
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300)
    aman, ablob = weights_mod.synth_weights(cfg, seed=31)
    oman, oblob = weights_mod.synth_weights(cfg, seed=32)
    RL.write_asr_dir(str(tmp_path / "asr"), conv, aman, ablob, cfg, vocab_tokens=[])
    RL.write_asr_dir(str(tmp_path / "online"), conv, oman, oblob, cfg, online=True, vocab_tokens=[])
    for k in ("asr", "online"):
        os.remove(tmp_path / k / "tokens.json")                # no vocabulary: the harness prints ids
    RL.write_vad_dir(str(tmp_path / "vad"), conv, vman, vblob)
    s16.tofile(tmp_path / "stream.pcm")
    exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "tpass_infer")
    out = subprocess.run([exe, str(tmp_path / "asr"), str(tmp_path / "online"), str(tmp_path / "vad"), str(tmp_path / "stream.pcm"), "9600", "2"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    got = [l.split(" | ") for l in out.stdout.splitlines() if l.startswith("call ")]
    # the same flow with the oracle's networks; the online VAD (device) only decides where segments close
    vad = pkg.FsmnVadHip().InitVad((vman, vblob))
    vad_on = pkg.FsmnVadOnlineHip(vad, 800, 60000, 0.9)
    Wa, Wo = P.Weights(aman, ablob), P.Weights(oman, oblob)
    online = PO.ParaformerOnline(Wo)
    audio = A.TpassAudio()
    f32 = (s16.astype(np.float32) / 32768.0).astype(np.float32)
    n_tpass = n_online_tokens = 0
    for j, off in enumerate(range(0, len(f32), 9600)):
        last = off + 9600 >= len(f32)
        audio.LoadPcmwavOnline(f32[off:off + 9600])
        audio.Split(lambda w, fin: vad_on.Infer(w, fin), 9600, last, A.ASR_TWO_PASS)
        want_online = []
        while True:
            fr = audio.FetchChunck()
            if fr is None:
                break
            want_online += [int(i) for i in online.Forward(fr["data"], fr["is_final"])]
        have_online = [int(x) for x in got[j][1][len("online "):].split()]
        assert have_online == want_online, (j, have_online, want_online)
        n_online_tokens += len(want_online)
        have_tpass = got[j][2][len("tpass "):]
        want = None
        while True:
            fr = audio.FetchTpass()
            if fr is None:
                break
            want = P.forward_pcm(np.asarray(fr["data"], np.float32), Wa)
            n_tpass += 1
        if want is None:
            assert have_tpass.strip() == ""
        else:
            assert_ids_match([int(x) for x in have_tpass.split()], want)
        if last:
            audio.ResetIndex()
    assert len(got) == j + 1 and n_tpass >= 3 and n_online_tokens > 20
    vad_on.close(); vad.close()
