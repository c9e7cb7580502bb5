"""CPU: hotword strings -> id matrix (the host half of Paraformer::CompileHotwordEmbedding, paraformer.cpp:600-651, with SegDict,
IsAllChineseCharactor and KeepChineseCharacterAndSplit), through `host_selftest hotwords`.  Known answers traced by hand through
the cited lines; the reference holds no fixture for this and its translation units need glog / onnxruntime headers."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "asr-2pass_amd", "host_selftest")
pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="host_selftest not built")

TOKENS = ["<blank>", "<s>", "</s>", "你", "好", "世", "界", "hel@@", "lo", "wor@@", "ld", "a", "<unk>"]
SEG = "hello\thel@@ lo\nworld\twor@@ ld\nbad\thel@@  lo\nlong\t" + " ".join(["a"] * 12) + "\nodd\tzz\nnotab only\n"


def run(lines, seg=True, tmp=None):
    with open(tmp / "tokens.json", "w", encoding="utf-8") as f:
        json.dump(TOKENS, f, ensure_ascii=False)
    with open(tmp / "seg_dict", "w", encoding="utf-8") as f:
        f.write(SEG)
    out = subprocess.run([EXE, "hotwords", str(tmp / "tokens.json"), str(tmp / "seg_dict") if seg else "-"], input="\n".join(lines) + "\n",
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    res = []
    for l in out.stdout.splitlines():
        n, lens, mat = l.split("|")
        ids = [int(x) for x in mat.split()]
        res.append(([int(x) for x in lens.split()], [ids[10 * i:10 * i + 10] for i in range(int(n))]))
    return res


BLANK = [1] + [0] * 9


def row(*ids):
    return list(ids) + [0] * (10 - len(ids))


def test_known_answers(tmp_path):
    got = run(["你好 世界", "hello 你好 world", "", "你好  世界", "你a好 好", "unknown 你", "bad 好", "long", "odd 世", "你好世界你好世界你好世界"], tmp=tmp_path)
    # two Chinese hotwords -> characters; the blank row [1, 0...] with length 1 closes the matrix (:648-651)
    assert got[0] == ([2, 2, 1], [row(3, 4), row(5, 6), BLANK])
    # Latin words go through the segmentation dictionary (:617-622)
    assert got[1] == ([2, 2, 2, 1], [row(7, 8), row(3, 4), row(9, 10), BLANK])
    # empty string: only the blank row (:604, :648)
    assert got[2] == ([1], [BLANK])
    # a double space yields an empty hotword, which has no units and is skipped (:624-626)
    assert got[3] == ([2, 2, 1], [row(3, 4), row(5, 6), BLANK])
    # mixed script is not "all Chinese" -> dictionary lookup -> not found -> skipped
    assert got[4] == ([1, 1], [row(4), BLANK])
    assert got[5] == ([1, 1], [row(3), BLANK])
    # a dictionary entry with an empty piece ("hel@@  lo"): the empty unit has no id -> the hotword is dropped as OOV (:631-640)
    assert got[6] == ([1, 1], [row(4), BLANK])
    # more than 10 units: truncated to 10 (:628-629)
    assert got[7] == ([10, 1], [[11] * 10, BLANK])
    # pieces that are not in the token list -> OOV -> dropped
    assert got[8] == ([1, 1], [row(5), BLANK])
    assert got[9] == ([10, 1], [[3, 4, 5, 6, 3, 4, 5, 6, 3, 4], BLANK])


def test_without_a_dictionary_latin_hotwords_are_skipped(tmp_path):
    got = run(["hello 你好"], seg=False, tmp=tmp_path)
    assert got[0] == ([2, 1], [row(3, 4), BLANK])
