"""CPU: token ids -> text (`Vocab::Vector2StringV2`, onnxruntime/src/vocab.cpp:164-305).

vocab.cpp cannot be compiled in place (glog / yaml-cpp) and the reference holds no fixture for it, so the pin
is a set of KNOWN ANSWERS traced by hand through the cited lines (each case says which branch it exercises); the oracle
(oracle/vocab_text.py, a line-by-line restatement) and the product's stand-alone implementation
(asr-2pass_amd/csrc/host/host_vocab.cpp, a restructured one, run through the `host_selftest vocabtext` harness) must both
give them, and agree with each other on random id sequences."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import vocab_text as VT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "asr-2pass_amd", "host_selftest")

TOKENS = ["<blank>", "<s>", "</s>", "你", "好", "hel@@", "lo", "a", "i", "world", "tv@@", "<unk>", "é", "@@x", "▁he", "llo", "▁i", "▁world", "un@@", "be@@", "liev@@", "able"]
ID = {t: i for i, t in enumerate(TOKENS)}


def ids(*toks):
    return [ID[t] for t in toks]


# (calls made on ONE vocabulary object, in order; each call = (tokens, expected text))
KNOWN = [
    # Chinese characters are appended bare (:243-246); <s> / </s> / <unk> are skipped (:181-182)
    [(("<s>", "你", "好", "</s>"), "你好")],
    # "hel@@" + "lo" glue to one word (:225-237); two Latin words get a space when either is longer than one letter (:261-280)
    [(("hel@@", "lo", "world"), "hello world")],
    # single letters stay glued to each other ("a" "i" -> "ai"), a longer word after them gets its space (:267-279)
    [(("a", "i", "world", "a"), "ai world a")],
    # a Latin word directly after a Chinese character: no space in a fresh object (first_word_need_space == false, :250-252)
    [(("你", "hel@@", "lo", "好"), "你hello好")],
    # "tv@@" in front of a Chinese character is closed with a trailing space (:205-214)
    [(("tv@@", "你"), "tv 你")],
    # ... also when it ends a glued word: "hel@@" "tv@@" 你 -> "heltv " + 你
    [(("hel@@", "tv@@", "你"), "heltv 你")],
    # a sub-word piece as the very last token loses its marker and gets no space (:215-224)
    [(("world", "hel@@"), "world hel")],
    # the marker is looked for ANYWHERE but the LAST two bytes are dropped (:201, :206): "@@x" -> "@" and keeps gluing
    [(("@@x", "lo"), "@lo")],
    # four pieces of one word
    [(("un@@", "be@@", "liev@@", "able", "你"), "unbelievable你")],
    # the object remembers that a call ended on a complete Latin word (:283-288): in the NEXT call every Latin word that
    # follows a Chinese character (or starts the text) gets a space in front (:176, :250-252) ...
    [(("你", "world"), "你world"), (("你", "world", "好", "lo"), "你 world好 lo"),
     # ... that second call ended on "lo" (Latin, complete) so the memory is still set; a call ending in Chinese clears it
     (("world", "你"), " world你"), (("你", "world"), "你world")],
    # a call ending on a sub-word piece does not set the memory (:215-224, :283-288)
    [(("你", "hel@@"), "你hel"), (("world",), "world")],
    # skipped specials at the end do not touch the memory: "world" "</s>" leaves what the PREVIOUS call left (:181-182)
    [(("world",), "world"), (("你", "</s>"), "你"), (("world", "</s>"), "world"), (("a",), "a")],
    # a non-Chinese 2-byte character counts as Latin of size 2 (:137-147, :261-280)
    [(("a", "é"), "a é")],
    # empty input
    [((), "")],
]
KNOWN_BPE = [
    # sentencepiece pieces (language "en-bpe", :183-198, :291-297): U+2581 opens a word, WordFormat capitalises "i"
    [(("<s>", "▁he", "llo", "▁i", "▁world", "</s>"), "hello I world")],
    [(("llo", "▁world"), "llo world")],
]


def test_last_call_memory_trace():
    """The fourth call of the memory case, by hand: after call 3 (`world 你`) the last token is Chinese -> memory false ->
    call 4 has no leading spaces.  (Guards the KNOWN table itself.)"""
    v = VT.Vocab(TOKENS)
    assert v.vector2string_v2(ids("你", "world")) == "你world" and v.last_is_complete_english_
    assert v.vector2string_v2(ids("你", "world", "好", "lo")) == "你 world好 lo" and v.last_is_complete_english_
    assert v.vector2string_v2(ids("world", "你")) == " world你" and not v.last_is_complete_english_


@pytest.mark.parametrize("calls", KNOWN)
def test_oracle_known_answers(calls):
    v = VT.Vocab(TOKENS)
    for toks, want in calls:
        assert v.vector2string_v2(ids(*toks), "zh-cn") == want, toks


@pytest.mark.parametrize("calls", KNOWN_BPE)
def test_oracle_known_answers_bpe(calls):
    v = VT.Vocab(TOKENS)
    for toks, want in calls:
        assert v.vector2string_v2(ids(*toks), "en-bpe") == want, toks


def run_harness(tmp_path, sequences, language):
    with open(tmp_path / "tokens.json", "w", encoding="utf-8") as f:
        json.dump(TOKENS, f, ensure_ascii=False)
    text = "".join(" ".join(str(i) for i in s) + "\n" for s in sequences)
    out = subprocess.run([EXE, "vocabtext", str(tmp_path / "tokens.json"), language], input=text, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return [bytes.fromhex(l).decode("utf-8") for l in out.stdout.split("\n")[:len(sequences)]]


needs_exe = pytest.mark.skipif(not os.path.exists(EXE), reason="host_selftest not built")


@needs_exe
@pytest.mark.parametrize("table,language", [(KNOWN, "zh-cn"), (KNOWN_BPE, "en-bpe")])
def test_product_known_answers(tmp_path, table, language):
    for calls in table:
        got = run_harness(tmp_path, [ids(*t) for t, _ in calls], language)
        assert got == [w for _, w in calls], calls


@needs_exe
@pytest.mark.parametrize("language", ["zh-cn", "en-bpe", ""])
def test_product_equals_oracle_on_random_sequences(tmp_path, language):
    rng = np.random.default_rng(5)
    seqs = [list(map(int, rng.integers(0, len(TOKENS), int(rng.integers(0, 12))))) for _ in range(400)]
    v = VT.Vocab(TOKENS)
    want = [v.vector2string_v2(s, language) for s in seqs]
    assert run_harness(tmp_path, seqs, language) == want
