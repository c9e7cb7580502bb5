"""GPU: the long-audio flow (VAD forward -> end-point detector -> length-sorted dynamic batches -> batched
Paraformer forward -> re-ordering; funasrruntime.cpp:208-337) against the same flow built from the oracles."""
import importlib

import numpy as np
import pytest

from conftest import synth_pcm
from oracle import e2e_vad as E
from oracle import fsmn_vad as V
from oracle import paraformer as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_file(rng):
    """~45 s: speech-like bursts separated by 1.2-s digital silence."""
    parts = []
    for i, sec in enumerate([4.0, 7.5, 2.2, 11.0, 5.3]):
        parts.append(synth_pcm(i, int(sec * 16000), rng))
        parts.append(np.zeros(int(1.2 * 16000), np.float32))
    return np.concatenate(parts)


def shape_vad_weights(man, blob):
    """Synthetic VAD weights do not separate speech from silence; make class 0 follow the frame energy so that the
    detector sees real segments: out2 bias favours silence, and the input layers pass log-mel energy through."""
    W = P.Weights(man, blob)
    for k in list(W.t):
        if k.endswith(".w") and not k.startswith("cmvn"):
            W.t[k][...] = 0
        if k.endswith(".b") and not k.startswith("cmvn"):
            W.t[k][...] = 0
    # energy path: mean log-mel of the centre frame -> unit 0 of every layer -> logit of class 0 (negative slope)
    W["in1.w"][0, 160:240] = 1.0 / 80
    W["in2.w"][0, 0] = 1.0
    for i in range(man["config"]["layers"]):
        W[f"blk.{i}.linear.w"][0, 0] = 1.0
        W[f"blk.{i}.affine.w"][0, 0] = 1.0
    W["out1.w"][0, 0] = 1.0
    W["out2.w"][0, 0] = -4.0
    W["out2.b"][0] = 8.0          # energy ~ (x - 8) * 0.3 after CMVN: silence (log eps) strongly negative -> relu 0
    return man, blob


def test_long_audio_flow_matches_oracle_flow(pkg, weights_mod):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    pipeline = importlib.import_module("asr_2pass_amd.pipeline")
    rng = np.random.default_rng(9)
    pcm = make_file(rng)
    vman, vblob = shape_vad_weights(*weights_mod.synth_vad_weights())
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300)
    aman, ablob = weights_mod.synth_weights(cfg)
    vad = pkg.FsmnVadHip().InitVad((vman, vblob))
    asr = pkg.ParaformerHip().InitAsr((aman, ablob))
    seg = pkg.E2EVadModelHost()
    ids, frames = pipeline.infer_buffer(pcm, asr, vad, seg, batch_size=4, vad_max_len=60000)
    # ---- oracle flow: 1-s slices through the oracle VAD + online detector, like Audio::CutSplit does ----
    VW, AW = P.Weights(vman, vblob), P.Weights(aman, ablob)
    sil_ref = V.FsmnVad(VW).Forward(pcm, True)[:, 0]
    sil_got = vad.ForwardSil(pcm, is_final=True)
    assert np.abs(sil_ref - sil_got).max() < 2e-5
    ref_segs = E.E2EVadModel()(sil_ref, pcm[:400 + 160 * (len(sil_ref) - 1)], True, False, 800, 60000, 0.9)
    assert [(s * 16, min(e * 16, len(pcm))) for s, e in ref_segs] == frames
    assert len(frames) == 5                                  # five bursts -> five segments
    for (s, e), got in zip(frames, ids):
        ref = P.forward_pcm(pcm[s:e], AW)
        assert list(got) == list(ref["ids"])
    vad.close(); asr.close(); seg.close()


def test_fetch_dynamic_rule(pkg):
    pipeline = importlib.import_module("asr_2pass_amd.pipeline")
    S = 16000
    # sorted ascending (CutSplit): 8 x 20 s, then one 70-s segment (>= max_sent)
    q = [(0, 20 * S)] * 8 + [(0, 70 * S)]
    b1 = pipeline.fetch_dynamic(q, 32)
    assert len(b1) == 8                       # 9th would be >= 60 s -> own batch
    b2 = pipeline.fetch_dynamic(q, 32)
    assert len(b2) == 1 and b2[0][1] == 70 * S and q == []
    # max_acc: max_len * count <= 300 s
    q = [(0, 50 * S)] * 10
    assert len(pipeline.fetch_dynamic(q, 32)) == 6
    # batch_size cap
    q = [(0, S)] * 10
    assert len(pipeline.fetch_dynamic(q, 4)) == 4


def test_cpp_handle_api_matches_python_flow(pkg, weights_mod, tmp_path):
    """The C++ mirror of FunOfflineInit / FunOfflineInferBuffer / FunASRGetResult (csrc/host/funasrruntime_hip.cpp, run
    through the `offline_infer` harness on a model directory and an s16 PCM file) gives the same segments and token ids
    as the Python flow above, for two batch sizes (the dynamic batcher only changes the grouping)."""
    import json
    import os
    import subprocess
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    pipeline = importlib.import_module("asr_2pass_amd.pipeline")
    rng = np.random.default_rng(11)
    pcm = make_file(rng)
    s16 = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2")
    pcm = (s16.astype(np.float32) / 32768.0).astype(np.float32)            # what LoadPcmwav makes of the file
    vman, vblob = shape_vad_weights(*weights_mod.synth_vad_weights())
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300, timestamp=1)
    aman, ablob = weights_mod.synth_weights(cfg)
    mdir, vdir = tmp_path / "asr", tmp_path / "vad"
    mdir.mkdir(); vdir.mkdir()
    weights_mod.save(str(mdir / "model.pfhip"), aman, ablob)
    weights_mod.save(str(vdir / "vad.pfhip"), vman, vblob)
    with open(mdir / "tokens.json", "w") as f:
        json.dump([f"<{i}>" for i in range(300)], f)
    s16.tofile(tmp_path / "long.pcm")
    exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "offline_infer")
    vad = pkg.FsmnVadHip().InitVad((vman, vblob))
    asr = pkg.ParaformerHip().InitAsr((aman, ablob))
    seg = pkg.E2EVadModelHost()
    ids, frames = pipeline.infer_buffer(pcm, asr, vad, seg, batch_size=4, vad_max_len=60000)
    for batch, threads in ((4, 1), (1, 1), (4, 6)):       # 6 threads share the handle: merged launches, same results
        out = subprocess.run([exe, str(mdir), str(vdir), str(tmp_path / "long.pcm"), str(batch), str(threads), "2"],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.splitlines()
        segs = [l for l in lines if l.startswith("seg ")]
        assert len(segs) == len(frames) == 5
        for l, (s, e), want in zip(segs, frames, ids):
            head, _, tail = l.partition(":")
            assert [int(x) for x in head.split()[1:3]] == [s, e]
            assert [int(x) for x in tail.split()] == list(want)
        text = [l for l in lines if l.startswith("text ")][0][5:]
        # time-stamp model + vocabulary: each segment's text comes out of PostProcess (Latin-like tokens -> single spaces)
        assert text == "".join(" ".join(f"<{i}>" for i in seg_ids) for seg_ids in ids)
        stamp = [l for l in lines if l.startswith("stamp ")][0][6:]
        pairs = json.loads(stamp) if stamp else []
        assert len(pairs) == sum(len(x) for x in ids)                                 # one [begin,end] ms pair per token
        assert all(b <= e for b, e in pairs) and pairs == sorted(pairs)
    vad.close(); asr.close(); seg.close()


def test_cpp_2pass_api_matches_python_flow(pkg, weights_mod, tmp_path):
    """FunTpassInit / FunTpassOnlineInit / FunTpassInferBuffer (C++ mirror, `tpass_infer` harness) on one connection fed in
    600-ms pieces: per call the streaming text and the 2nd-pass text of segments that closed equal the same flow assembled
    in Python from the oracle's Audio::Split restatement and the plug-ins (online VAD, streaming and offline Paraformer)."""
    import os
    import subprocess
    from oracle import audio_split as A
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    rng = np.random.default_rng(21)
    pcm = make_file(rng)[:16000 * 30]
    s16 = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2")
    vman, vblob = shape_vad_weights(*weights_mod.synth_vad_weights())
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300)
    aman, ablob = weights_mod.synth_weights(cfg, seed=31)
    oman, oblob = weights_mod.synth_weights(cfg, seed=32)
    dirs = {k: tmp_path / k for k in ("asr", "online", "vad")}
    for d in dirs.values():
        d.mkdir()
    weights_mod.save(str(dirs["asr"] / "model.pfhip"), aman, ablob)
    weights_mod.save(str(dirs["online"] / "model.pfhip"), oman, oblob)
    weights_mod.save(str(dirs["vad"] / "vad.pfhip"), vman, vblob)
    s16.tofile(tmp_path / "stream.pcm")
    exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "tpass_infer")
    out = subprocess.run([exe, str(dirs["asr"]), str(dirs["online"]), str(dirs["vad"]), str(tmp_path / "stream.pcm"), "9600", "2"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = [l.split(" | ") for l in out.stdout.splitlines() if l.startswith("call ")]
    # ---- the same flow in Python -------------------------------------------------------------------------------------
    vad = pkg.FsmnVadHip().InitVad((vman, vblob))
    asr = pkg.ParaformerHip().InitAsr((aman, ablob))
    asr_on_model = pkg.ParaformerHip().InitAsr((oman, oblob))
    vad_on = pkg.FsmnVadOnlineHip(vad, 800, 60000, 0.9)
    stream = pkg.ParaformerOnlineHip(asr_on_model)
    audio = A.TpassAudio()
    f32 = (s16.astype(np.float32) / 32768.0).astype(np.float32)
    n_calls = 0
    n_tpass = 0
    for j, off in enumerate(range(0, len(f32), 9600)):
        last = off + 9600 >= len(f32)
        audio.LoadPcmwavOnline(f32[off:off + 9600])
        audio.Split(lambda w, fin: vad_on.Infer(w, fin), 9600, last, A.ASR_TWO_PASS)
        online_txt = ""
        while True:
            fr = audio.FetchChunck()
            if fr is None:
                break
            piece = " ".join(str(i) for i in stream.Forward(fr["data"], input_finished=fr["is_final"]))
            online_txt += piece + (" " if piece and stream.last_path() == 2 else "")       # paraformer-online.cpp:585-587
        tpass_txt = ""
        while True:
            fr = audio.FetchTpass()
            if fr is None:
                break
            tpass_txt = " ".join(str(int(i)) for i in asr.forward_ids([fr["data"]])["ids"][0])
            n_tpass += 1
        if last:
            audio.ResetIndex()
        assert got[j][0] == f"call {j}"
        assert got[j][1] == "online " + online_txt, (j, got[j][1], online_txt)
        assert got[j][2] == "tpass " + tpass_txt, (j, got[j][2], tpass_txt)
        n_calls += 1
    assert n_calls == len(got) and n_tpass >= 3                 # several segments closed and were re-decoded offline
    for o in (vad_on, stream, vad, asr, asr_on_model):
        o.close()


def test_cpp_2pass_api_with_punctuation(pkg, weights_mod, tmp_path):
    """FunTpassInferBuffer with a PUNC_DIR: the 2nd-pass text of every closed segment goes through
    punc_online_handle->AddPunc(msg, punc_cache[1]) (funasrruntime.cpp:609-614) — the realtime class (cache carried from
    segment to segment, "。" appended on the final call) when the directory name contains "realtime", else CTTransformer.
    Expected texts: the un-punctuated run of the same harness pushed through the oracle's AddPunc restatements."""
    import json
    import os
    import subprocess
    from oracle import ct_transformer as C
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    rng = np.random.default_rng(23)
    pcm = make_file(rng)[:16000 * 30]
    s16 = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2")
    vman, vblob = shape_vad_weights(*weights_mod.synth_vad_weights())
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300)
    aman, ablob = weights_mod.synth_weights(cfg, seed=31)
    oman, oblob = weights_mod.synth_weights(cfg, seed=32)
    vocab = ["<blank>", "<s>", "</s>"] + [chr(0x4E00 + i) for i in range(296)] + ["<unk>"]
    pcfg = dict(weights_mod.CT_TRANSFORMER, vocab=len(vocab))
    pman, pblob = weights_mod.synth_punc_weights(pcfg)
    dirs = {k: tmp_path / k for k in ("asr", "online", "vad", "punc", "punc_realtime")}
    for d in dirs.values():
        d.mkdir()
    weights_mod.save(str(dirs["asr"] / "model.pfhip"), aman, ablob)
    weights_mod.save(str(dirs["online"] / "model.pfhip"), oman, oblob)
    weights_mod.save(str(dirs["vad"] / "vad.pfhip"), vman, vblob)
    for k in ("asr", "online", "punc", "punc_realtime"):
        with open(dirs[k] / "tokens.json", "w", encoding="utf-8") as f:
            json.dump(vocab, f, ensure_ascii=False)               # raw UTF-8, as FunASR writes it
    for k in ("punc", "punc_realtime"):
        weights_mod.save(str(dirs[k] / "punc.pfhip"), pman, pblob)
    s16.tofile(tmp_path / "stream.pcm")
    exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "tpass_infer")
    base = [exe, str(dirs["asr"]), str(dirs["online"]), str(dirs["vad"]), str(tmp_path / "stream.pcm"), "9600", "2"]

    def run(*extra):
        out = subprocess.run(base + list(extra), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        return [l.split(" | ") for l in out.stdout.splitlines() if l.startswith("call ")]

    plain = run()
    off = run(str(dirs["punc"]))
    real = run(str(dirs["punc_realtime"]))
    assert len(plain) == len(off) == len(real)
    W = P.Weights(pman, pblob)
    t2i = {t: i for i, t in enumerate(vocab)}
    infer_off = lambda ids: C.infer(np.asarray(ids, np.int32), W)[1]
    infer_on = lambda ids, n_cache: C.forward_online(np.asarray(ids, np.int32), W, n_cache)[1]
    cache, n_seg = [], 0
    for j, (p, o, r) in enumerate(zip(plain, off, real)):
        assert p[1] == o[1] == r[1]                               # streaming text is not punctuated in 2-pass mode
        text = p[2][len("tpass "):]
        if not text:
            assert o[2] == "tpass " and r[2] == "tpass "
            continue
        n_seg += 1
        last = j == len(plain) - 1
        assert o[2] == "tpass " + C.add_punc_text(text, infer_off, t2i), j
        want = C.add_punc_text_online(text, cache, infer_on, t2i) + ("。" if last else "")
        assert r[2] == "tpass " + want, (j, r[2], want)
    assert n_seg >= 3
