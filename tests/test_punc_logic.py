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
