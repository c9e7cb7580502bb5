"""GPU: the C++ plug-in class `ParaformerHip` (csrc/host/paraformer_hip.cpp, stand-alone build) through the `decoder_handoff`
harness — the WFST hand-off of Paraformer::Forward (paraformer.cpp:563-579; per item of a batch as paraformer-torch.cpp:431-466):
with an LM configured and a FUNASR_DEC_HANDLE given, every utterance's log-prob rows go to Decoder::Search(rows, token_num,
vocab) and FinalizeDecode gives the text; the rows must be exactly the log-probs the C ABI returns for the same batch
(`pfhip_out.logp` == pfhip_get_tensor("logp")).  Without an LM the decoder handle is ignored (GreedySearch, :563/:572)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import synth_pcm

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def read_rows(path):
    raw = np.fromfile(path, dtype=np.uint8)
    out, o = [], 0
    while o < raw.size:
        n, v = np.frombuffer(raw[o:o + 8].tobytes(), np.int32)
        o += 8
        out.append(np.frombuffer(raw[o:o + 4 * n * v].tobytes(), np.float32).reshape(n, v).copy())
        o += 4 * n * v
    return out


@pytest.mark.parametrize("timestamp", [0, 1])
def test_logprob_rows_reach_the_decoder(pkg, weights_mod, tmp_path, timestamp):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "decoder_handoff")
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300, timestamp=timestamp)
    man, blob = weights_mod.synth_weights(cfg, seed=41)
    weights_mod.save(str(tmp_path / "model.pfhip"), man, blob)
    vocab = ["<blank>", "<s>", "</s>"] + [chr(0x4E00 + i) for i in range(296)] + ["<unk>"]
    with open(tmp_path / "tokens.json", "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    rng = np.random.default_rng(17)
    pcm = synth_pcm(2, 16000 * 14 + 300, rng)
    s16 = np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2")
    s16.tofile(tmp_path / "a.pcm")
    f32 = (s16.astype(np.float32) / 32768.0).astype(np.float32)
    n_utts, total = 4, len(f32)
    step = (total - 300) // (n_utts - 1)
    utts = [f32[k * step:k * step + (step - 1000 * k if k + 1 < n_utts else 300)] for k in range(n_utts)]
    model = pkg.ParaformerHip().InitAsr((man, blob))
    want = model.forward_ids(utts, want_logp=True, want_timestamps=bool(timestamp))
    model.close()
    assert want["n_frames"][-1] == 0 and min(want["token_num"][:-1]) > 0

    def run(with_lm, fin):
        out = subprocess.run([exe, str(tmp_path), str(tmp_path / "a.pcm"), str(n_utts), str(tmp_path / "rows.bin"), str(with_lm), str(fin)],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        return out.stdout.splitlines()

    # ---- LM configured, final call: Search + FinalizeDecode per utterance, decoder restarted between items ----------------
    lines = run(1, 1)
    rows = read_rows(tmp_path / "rows.bin")
    calls = [l for l in lines if l.split()[0] in ("search", "finalize", "start")]
    live = [b for b in range(n_utts) if want["n_frames"][b] > 0]
    assert len(rows) == len(live) == 3
    k = 0
    for j, b in enumerate(live):
        n = int(min(want["token_num"][b], want["n_fires"][b]))
        assert calls[k] == f"search {j} {n} 300"
        us = 3 * int(want["n_frames"][b]) if timestamp else 0
        assert calls[k + 1] == f"finalize {j} {timestamp} {us} {us}"
        assert calls[k + 2] == "start"
        k += 3
        assert rows[j].shape == (n, 300)
        assert np.array_equal(rows[j], want["logp"][b][:n]), b          # bit-exact: same kernels, same batch
    assert k == len(calls)
    res = {int(l.split()[1]): l.split(" ", 2)[2] if len(l.split(" ", 2)) > 2 else "" for l in lines if l.startswith("result ")}
    assert [res[b] for b in live] == ["F0", "F1", "F2"] and res[n_utts - 1] == ""
    # ---- not final: Search only (the reference keeps the partial hypothesis, :566-569) -------------------------------------
    lines = run(1, 0)
    assert [l.split()[0] for l in lines if l.split()[0] in ("search", "finalize", "start")] == ["search", "start"] * 3
    res = {int(l.split()[1]): l.split(" ", 2)[2] if len(l.split(" ", 2)) > 2 else "" for l in lines if l.startswith("result ")}
    assert [res[b] for b in live] == ["S0", "S1", "S2"]
    # ---- no LM: the handle is ignored, greedy text through Vector2StringV2 (or "text | stamps" for time-stamp models) ------
    lines = run(0, 1)
    assert not [l for l in lines if l.split()[0] in ("search", "finalize", "start")]
    ids = {int(l.split()[1]): [int(x) for x in l.split()[2:]] for l in lines if l.startswith("ids ")}
    res = {int(l.split()[1]): l.split(" ", 2)[2] if len(l.split(" ", 2)) > 2 else "" for l in lines if l.startswith("result ")}
    from oracle import vocab_text as VT
    v = VT.Vocab(vocab)
    for b in range(n_utts):
        assert ids[b] == list(want["ids"][b])
        text = v.vector2string_v2(ids[b], "zh-cn")
        if timestamp and ids[b]:
            # time-stamp mode: "<text> | <stamps>" (PostProcess, util.cpp:733-835; compared in detail by test_timestamp.py)
            assert res[b] == text or " | " in res[b]
        else:
            assert res[b] == text
