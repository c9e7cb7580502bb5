"""CTTransformer::AddPunc mini-sentence bookkeeping (ct-transformer.cpp:39-155) restated on ids: hand-derived known answers
with scripted Infer results (no network involved)."""
from oracle import ct_transformer as C

NOT, COMMA, PERIOD, QUESTION, DUN = 1, 2, 3, 4, 5


def test_short_text_gets_final_period():
    assert C.add_punc_ids([7, 8, 9], None, infer_fn=lambda x: [NOT] * len(x)) == [NOT, NOT, NOT, PERIOD]
    assert C.add_punc_ids([7, 8, 9], None, infer_fn=lambda x: [NOT, NOT, COMMA]) == [NOT, NOT, PERIOD]      # trailing comma -> period
    assert C.add_punc_ids([7, 8, 9], None, infer_fn=lambda x: [NOT, NOT, DUN]) == [NOT, NOT, PERIOD]
    assert C.add_punc_ids([7, 8, 9], None, infer_fn=lambda x: [NOT, NOT, QUESTION]) == [NOT, NOT, QUESTION]
    assert C.add_punc_ids([], None, infer_fn=lambda x: []) == []


def test_tail_after_sentence_end_is_carried_into_the_next_call():
    calls = []

    def infer(x):
        calls.append(list(x))
        p = [NOT] * len(x)
        if len(calls) == 1:
            p[11] = PERIOD            # first mini-sentence (20 tokens): sentence ends at index 11 -> 8 tokens are carried
        return p

    ids = list(range(100, 130))       # 30 tokens -> two mini-sentences
    out = C.add_punc_ids(ids, None, infer_fn=infer)
    assert calls[0] == ids[:20]
    assert calls[1] == ids[12:20] + ids[20:30]                 # RemainIDs + next 10
    assert out == [NOT] * 11 + [PERIOD] + [NOT] * 18 + [PERIOD]
    assert len(out) == 31


def test_sentence_end_search_skips_first_and_last_position():
    # a period at index 0 or at the last index of a non-final mini-sentence is not a cut point (loop runs size-2 .. 1)
    def infer(x):
        p = [NOT] * len(x)
        p[0] = PERIOD
        p[-1] = PERIOD
        return p

    calls = []
    out = C.add_punc_ids(list(range(40)), None, infer_fn=lambda x: (calls.append(len(x)), infer(x))[1])
    assert calls == [20, 40]          # nothing was cut after the first call: all 20 tokens carried
    assert out[-1] == PERIOD and len(out) == 40


def test_forced_period_at_last_comma_beyond_200_carried_tokens():
    def infer(x):
        p = [NOT] * len(x)
        if len(x) > 3:
            p[len(x) - 3] = COMMA
        return p

    calls = []
    out = C.add_punc_ids(list(range(300)), None, infer_fn=lambda x: (calls.append(len(x)), infer(x))[1])
    # carried text grows 20, 40, ... until it exceeds 200: then the last comma becomes a period and the rest is carried
    assert calls[:11] == [20 * k for k in range(1, 12)]
    assert calls[11] == 2 + 20        # after the cut at index 217 of 220 two tokens remain
    assert out.count(PERIOD) >= 2


# ---- text level: CTokenizer::Tokenize + string assembly, hand-derived known answers ---------------------------------------------
T2I = {"<unk>": 0, "你": 1, "好": 2, "hello": 3, "world": 4, "吗": 5}


def test_tokenize_splits_ascii_runs_and_utf8_characters():
    words, ids = C.tokenize("你好Hello  world吗x", T2I)
    assert [w.decode() for w in words] == ["你", "好", "Hello", "world", "吗", "x"]
    assert ids == [1, 2, 3, 4, 5, 0]                    # looked up lower-cased; the words keep their case; unknown -> <unk>
    assert C.tokenize("", T2I) == ([], [])
    assert C.tokenize("   ", T2I) == ([], [])


def test_offline_text_assembly():
    # punctuation after 好 and a forced final period; ASCII neighbours are separated by one blank
    def infer(ids):
        return [COMMA if i == 2 else NOT for i in ids]
    assert C.add_punc_text("你好hello world", infer, T2I) == "你好，hello world。"
    assert C.add_punc_text("你好", infer, T2I) == "你好。"                      # trailing comma becomes the period
    assert C.add_punc_text("hello world", lambda ids: [NOT, QUESTION], T2I) == "hello world？"
    assert C.add_punc_text("hello world", lambda ids: [NOT, QUESTION], T2I, language="en-bpe") == "hello world?"
    assert C.add_punc_text("", infer, T2I) == ""


def test_offline_no_blank_at_a_mini_sentence_start():
    # 21 ASCII words, a period after word 10: the second Infer gets words 11..20; word 11 starts that mini-sentence and gets
    # no blank in front of it (the k > 0 test is per mini-sentence, ct-transformer.cpp:98-103)
    text = " ".join(f"w{i}" for i in range(21))
    state = {"n": 0}

    def infer(ids):
        state["n"] += 1
        p = [NOT] * len(ids)
        if state["n"] == 1:
            p[10] = PERIOD
        return p

    out = C.add_punc_text(text, infer, {"<unk>": 0})
    assert out == " ".join(f"w{i}" for i in range(11)) + "。" + "w11 " + " ".join(f"w{i}" for i in range(12, 21)) + "。"


def test_online_cache_roundtrip():
    # first call: "你好吗" with ？ after 吗 (last position: not a cut point) -> everything is cached, the trailing mark is held back
    cache = []
    out = C.add_punc_text_online("你好吗", cache, lambda ids, n: [NOT, NOT, QUESTION], T2I)
    assert out == "你好吗" and [w.decode() for w in cache] == ["你", "好", "吗"]
    # second call: cached words are skipped in the output, but the mark of the LAST cached word is emitted (nSkipNum rule)
    seen = []

    def infer(ids, n_cache):
        seen.append((list(ids), n_cache))
        return [NOT, NOT, QUESTION, NOT, PERIOD, NOT]

    out = C.add_punc_text_online("hello world你", cache, infer, T2I)
    assert seen == [([1, 2, 5, 3, 4, 1], 3)]
    assert out == "？hello world。你"
    assert [w.decode() for w in cache] == ["你"]          # words after the last sentence end
    # ASCII cache tail + ASCII input: a blank is inserted at the junction (ct-transformer-online.cpp:48-50)
    cache = [b"hello"]
    out = C.add_punc_text_online("world", cache, lambda ids, n: [NOT, NOT], T2I)
    assert out == "world" and cache == [b"hello ", b"world"]
