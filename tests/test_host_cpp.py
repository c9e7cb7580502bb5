"""CPU: host-side C++ logic that has no C-ABI entry of its own, driven through the `host_selftest` harness
(csrc/host/host_selftest.cpp): the caller-merging queue under thread stress, the punctuation tokenizer
(CTokenizer::Tokenize restated, tokenizer.cpp:275-333) against the oracle's restatement, the JSON string-array reader."""
import json
import os
import subprocess

import pytest

from oracle import ct_transformer as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "asr-2pass_amd", "host_selftest")

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="host_selftest not built (python -c 'import __graft_entry__ as g; g.build()')")


@pytest.mark.parametrize("threads,rounds", [(64, 100), (3, 1500), (1, 50)])
def test_merge_queue_serves_every_request_once(threads, rounds):
    out = subprocess.run([EXE, "mergequeue", str(threads), str(rounds)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    f = dict(zip(out.stdout.split()[::2], out.stdout.split()[1::2]))
    assert int(f["served"]) == threads * rounds and int(f["errors"]) == 0 and int(f["leftover"]) == 0
    assert int(f["execs"]) <= threads * rounds
    if threads >= 64:
        assert int(f["max_batch"]) > 8                      # concurrent callers really were merged


def test_tokenizer_matches_oracle(tmp_path):
    vocab = ["<unk>", "你", "好", "吗", "hello", "world", "i'm", "a.i.", "é", "𠀀"]
    with open(tmp_path / "tokens.json", "w") as f:
        json.dump(vocab, f)                                 # \\uXXXX escapes incl. a surrogate pair
    with open(tmp_path / "man.json", "w", encoding="utf-8") as f:
        json.dump({"config": {"punc_list": ["<unk>", "_", "，", "。", "？", "、"]}}, f, ensure_ascii=False)
    lines = ["你好Hello  world吗x", "", "   ", "I'M a.i. é𠀀 ok", "hello,world 你 好", "ＡＢＣ１２３", "trailing space ", " leading"]
    out = subprocess.run([EXE, "tokenize", str(tmp_path / "tokens.json"), str(tmp_path / "man.json")], input="\n".join(lines) + "\n",
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    got = out.stdout.splitlines()
    t2i = {t: i for i, t in enumerate(vocab)}
    for line, g in zip(lines, got):
        words, ids = C.tokenize(line, t2i)
        n, w, i = g.split("|")
        assert int(n) == len(words), (line, g)
        assert w.split() == [x.decode() for x in words], (line, g)
        assert [int(x) for x in i.split()] == ids, (line, g)
    assert got[len(lines)] == "punc <unk> _ ， 。 ？ 、 ispunc 1 0"


def test_json_string_array_reader(tmp_path):
    items = ["plain", "quote\"inside", "back\\slash", "tab\tnl\n", "é", "你好", "𠀀", "", "</s>"]
    for ensure_ascii in (True, False):
        p = tmp_path / f"a{int(ensure_ascii)}.json"
        with open(p, "w", encoding="utf-8") as f:
            json.dump(items, f, ensure_ascii=ensure_ascii, indent=1)
        out = subprocess.run([EXE, "jsonstrings", str(p)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0
        assert [bytes.fromhex(l).decode("utf-8") for l in out.stdout.split("\n")[:len(items)]] == items
