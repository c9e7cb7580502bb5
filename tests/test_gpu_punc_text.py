"""GPU: the text-in / text-out punctuation adapters (C++ `CTTransformerHip` / `CTTransformerOnlineHip`, run through the
`punc_infer` harness on a model directory) against the oracle's restatement of CTokenizer::Tokenize + AddPunc
(tokenizer.cpp:275-333, ct-transformer.cpp:39-155, ct-transformer-online.cpp:40-152).  Synthetic weights: parity unpinned
for the network, the text logic is pinned by the hand-derived answers in tests/test_punc_logic.py."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import ct_transformer as C
from oracle import paraformer as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def make_vocab():
    cjk = [chr(0x4E00 + i) for i in range(300)]
    eng = [f"w{i}" for i in range(290)] + ["hello", "world", "i'm", "a.i."]
    return ["<unk>"] + cjk + eng, cjk, eng


def random_text(rng, cjk, eng, n_words):
    out = []
    for _ in range(n_words):
        r = rng.random()
        if r < 0.55:
            out.append(cjk[rng.integers(len(cjk))])
        elif r < 0.9:
            w = eng[rng.integers(len(eng))]
            out.append((" " if out and out[-1][-1].isascii() else "") + (w.upper() if rng.random() < 0.2 else w))
        elif r < 0.95:
            out.append("龘")                                     # not in the vocabulary -> <unk>
        else:
            out.append("  ")                                     # repeated blanks
    return "".join(out).strip()


@pytest.fixture(scope="module")
def punc_dir(pkg, weights_mod, tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    vocab, cjk, eng = make_vocab()
    cfg = dict(weights_mod.CT_TRANSFORMER, vocab=len(vocab))
    man, blob = weights_mod.synth_punc_weights(cfg)
    man["config"]["punc_list"] = list(C.DEFAULT_PUNC_LIST)
    d = tmp_path_factory.mktemp("punc_realtime")
    weights_mod.save(str(d / "punc.pfhip"), man, blob)
    with open(d / "tokens.json", "w") as f:
        json.dump(vocab, f)                                      # ASCII-only file: \\uXXXX escapes
    exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "punc_infer")
    return d, exe, P.Weights(man, blob), {t: i for i, t in enumerate(vocab)}, cjk, eng


def run(exe, d, kind, lines, *extra):
    out = subprocess.run([exe, str(d), kind, *extra], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return out.stdout.splitlines()


def test_offline_add_punc_text(punc_dir):
    d, exe, W, t2i, cjk, eng = punc_dir
    rng = np.random.default_rng(77)
    lines = [random_text(rng, cjk, eng, n) for n in (1, 5, 20, 21, 47, 130, 260)] + ["hello world", "你"]
    got = run(exe, d, "offline", lines)
    infer = lambda ids: C.infer(np.asarray(ids, np.int32), W)[1]
    assert len(got) == len(lines)
    n_marks = 0
    for g, line in zip(got, lines):
        want = C.add_punc_text(line, infer, t2i)
        assert g == "out " + want, (line, g, want)
        n_marks += sum(want.count(m) for m in "，。？、")
    assert n_marks > 20
    got_en = run(exe, d, "offline", lines[:4], "en-bpe")
    for g, line in zip(got_en, lines[:4]):
        assert g == "out " + C.add_punc_text(line, infer, t2i, language="en-bpe")


def test_online_add_punc_text_with_cache(punc_dir):
    d, exe, W, t2i, cjk, eng = punc_dir
    rng = np.random.default_rng(78)
    lines = [random_text(rng, cjk, eng, n) for n in (6, 9, 3, 30, 12, 1, 55, 8, 14, 25)]
    lines.insert(5, "<reset>")
    got = run(exe, d, "online", lines)
    infer = lambda ids, n_cache: C.forward_online(np.asarray(ids, np.int32), W, n_cache)[1]
    cache, k, carried = [], 0, 0
    for line in lines:
        if line == "<reset>":
            cache = []
            continue
        want = C.add_punc_text_online(line, cache, infer, t2i)
        assert got[k] == "out " + want, (line, got[k], want)
        assert got[k + 1] == "cache " + "|".join(w.decode() for w in cache), (line, got[k + 1])
        carried += len(cache)
        k += 2
    assert k == len(got) and carried > 0
