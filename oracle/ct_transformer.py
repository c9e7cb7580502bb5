"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (numpy fp32) restatement of the CT-Transformer punctuation forward:
  CTTransformer::Infer   onnxruntime/src/ct-transformer.cpp:162-204  — ids i32 [1,N] (+ length) -> f32 [1,N,6];
                         punctuation id = Argmax over the FIRST CANDIDATE_NUM-1 = 5 classes (:193-196,
                         com-define.h:128, commonfunc.h:106-108: std::max_element -> first maximum wins).
The graph is PARITY UNPINNED (onnxruntime + ModelScope file absent).  UPSTREAM architecture: Embedding(vocab, 256)
-> x*sqrt(256) + sinusoidal PE(256) -> SAN-M encoder (4 blocks, 8 heads of 32, FFN 1024, FSMN kernel 11; every
block has in_size == size so the attention residual applies to all) -> LayerNorm -> Linear(256 -> 6).
"""
from __future__ import annotations

import math

import numpy as np

from . import frontend as fe
from . import paraformer as P

F32 = np.float32


def forward(ids, W):
    cfg = W.cfg
    d = cfg["d_model"]
    x = W["embed.w"][np.asarray(ids, np.int64)].astype(F32)
    x = (x * F32(math.sqrt(d)) + fe.pos_emb(len(ids), d)).astype(F32)
    for i in range(cfg["layers"]):
        x = P.encoder_layer(x, W, f"enc.{i}.", cfg["n_head"])
    x = P.layer_norm(x, W["enc.after_norm.g"], W["enc.after_norm.b"])
    return P.linear(x, W["out.w"], W["out.b"])


def infer(ids, W):
    """CTTransformer::Infer: (logits [N,6], punctuation ids [N])."""
    logits = forward(ids, W)
    punc = np.argmax(logits[:, :W.cfg["n_punc"] - 1], axis=-1).astype(np.int32)
    return logits, punc


def vad_mask(n, vad_pos):
    """CTTransformerOnline::VadMask (ct-transformer-online.cpp:225-240)."""
    m = np.ones((n, n), F32)
    if vad_pos <= 0 or vad_pos >= n:
        return m
    m[:vad_pos - 1, vad_pos:] = 0.0
    return m


def fsmn_shift(v, w, shift):
    """Depthwise conv with left padding (k-1)/2 + shift and right padding (k-1)/2 - shift (UPSTREAM sanm_shfit)."""
    T, d = v.shape
    k = w.shape[1]
    lp = (k - 1) // 2 + shift
    vp = np.zeros((T + k - 1, d), F32)
    vp[lp:lp + T] = v
    out = v.astype(F32).copy()
    for j in range(k):
        out += vp[j:j + T] * w[:, j][None, :]
    return out.astype(F32)


def mha_masked(q, k, v, n_head, mask):
    Lq, d = q.shape
    dk = d // n_head
    out = np.empty((Lq, d), F32)
    for h in range(n_head):
        sl = slice(h * dk, (h + 1) * dk)
        s = ((q[:, sl] * F32(dk ** -0.5)).astype(F32) @ k[:, sl].T).astype(F32)
        s = np.where(mask > 0, s, F32(-np.inf))
        out[:, sl] = P.softmax_rows(s) @ v[:, sl]
    return out


def forward_online(ids, W, vad_pos):
    cfg = W.cfg
    d = cfg["d_model"]
    n = len(ids)
    mask = vad_mask(n, vad_pos)            # one mask for both mask inputs (ct-transformer-online.cpp:182-197)
    x = W["embed.w"][np.asarray(ids, np.int64)].astype(F32)
    x = (x * F32(math.sqrt(d)) + fe.pos_emb(n, d)).astype(F32)
    for i in range(cfg["layers"]):
        pfx = f"enc.{i}."
        y = P.layer_norm(x, W[pfx + "norm1.g"], W[pfx + "norm1.b"])
        qkv = P.linear(y, W[pfx + "qkv.w"], W[pfx + "qkv.b"])
        q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        mem = fsmn_shift(v, W[pfx + "fsmn.w"], cfg.get("sanm_shift", 0))
        att = P.linear(mha_masked(q, k, v, cfg["n_head"], mask), W[pfx + "out.w"], W[pfx + "out.b"]) + mem
        x = (x + att).astype(F32)
        y = P.layer_norm(x, W[pfx + "norm2.g"], W[pfx + "norm2.b"])
        h = np.maximum(P.linear(y, W[pfx + "ffn1.w"], W[pfx + "ffn1.b"]), F32(0))
        x = (x + P.linear(h, W[pfx + "ffn2.w"], W[pfx + "ffn2.b"])).astype(F32)
    x = P.layer_norm(x, W["enc.after_norm.g"], W["enc.after_norm.b"])
    logits = P.linear(x, W["out.w"], W["out.b"])
    return logits, np.argmax(logits[:, :cfg["n_punc"] - 1], axis=-1).astype(np.int32)


def add_punc_ids(ids, W, infer_fn=None):
    """CTTransformer::AddPunc (ct-transformer.cpp:39-155) on token ids: returns the punctuation id list NewPuncOut."""
    TOKEN_LEN, CACHE_POP_TRIGGER_LIMIT = 20, 200
    COMMA, PERIOD, QUESTION, DUN = 2, 3, 4, 5
    infer_fn = infer_fn or (lambda x: list(infer(x, W)[1]))
    ids = [int(v) for v in ids]
    n = len(ids)
    total = -(-n // TOKEN_LEN)
    remain, new_punc = [], []
    for i in range(0, n, TOKEN_LEN):
        inp = remain + ids[i:i + TOKEN_LEN]
        punc = [int(v) for v in infer_fn(inp)]
        if i // TOKEN_LEN < total - 1:
            sent_end, last_comma = -1, -1
            for k in range(len(punc) - 2, 0, -1):
                if punc[k] in (PERIOD, QUESTION):
                    sent_end = k
                    break
                if last_comma < 0 and punc[k] == COMMA:
                    last_comma = k
            if sent_end < 0 and len(inp) > CACHE_POP_TRIGGER_LIMIT and last_comma > 0:
                sent_end = last_comma
                punc[sent_end] = PERIOD
            remain = inp[sent_end + 1:]
            punc = punc[:sent_end + 1]
        new_punc += punc
    if new_punc:
        if new_punc[-1] in (COMMA, DUN):
            new_punc[-1] = PERIOD
        elif new_punc[-1] not in (PERIOD, QUESTION):
            new_punc.append(PERIOD)
    return new_punc


# ---- text level: CTokenizer + the string assembly of AddPunc ------------------------------------------------------------------
TOKEN_LEN, CACHE_POP_TRIGGER_LIMIT = 20, 200                  # com-define.h:126,136
NOTPUNC_INDEX, COMMA_INDEX, PERIOD_INDEX, QUESTION_INDEX, DUN_INDEX = 1, 2, 3, 4, 5
DEFAULT_PUNC_LIST = ["<unk>", "_", "，", "。", "？", "、"]


def _is_ascii_byte(b):
    return not (b & 0x80)


def split_chinese(bs):
    """CTokenizer::SplitChineseString (tokenizer.cpp:230-246): one UTF-8 sequence per entry (length = leading one bits)."""
    out, i = [], 0
    while i < len(bs):
        ln = 1
        for j in range(6):
            if not (bs[i] & (0x80 >> j)):
                break
            ln = j + 1
        out.append(bs[i:i + ln])
        i += ln
    return out


def tokenize(text, token2id):
    """CTokenizer::Tokenize (tokenizer.cpp:275-333, seg_jieba off) on bytes: split on ' ', then runs of ASCII bytes are one
    word each and every non-ASCII UTF-8 sequence is its own word.  ids: String2Ids (:188-200) looks up the LOWER-CASED
    word, <unk> otherwise; the words themselves keep their case."""
    bs = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    words = []
    if bs:
        for item in (bs + b" ").split(b" ")[:-1]:                  # StrSplit (:255-272)
            eng, chn = b"", b""
            for ch in item:
                if _is_ascii_byte(ch):
                    if chn:
                        words += split_chinese(chn)
                        chn = b""
                    eng += bytes([ch])
                else:
                    if eng:
                        words.append(eng)
                        eng = b""
                    chn += bytes([ch])
            if chn:
                words += split_chinese(chn)
            if eng:
                words.append(eng)
    unk = token2id["<unk>"]
    ids = [token2id.get(w.lower().decode("utf-8", "replace"), unk) for w in words]
    return words, ids


def _cut_at_sentence_end(punc, inp_words, inp_ids):
    """The not-the-last-mini-sentence branch shared by both AddPunc variants (ct-transformer.cpp:67-93)."""
    sent_end, last_comma = -1, -1
    for k in range(len(punc) - 2, 0, -1):
        if punc[k] in (PERIOD_INDEX, QUESTION_INDEX):
            sent_end = k
            break
        if last_comma < 0 and punc[k] == COMMA_INDEX:
            last_comma = k
    if sent_end < 0 and len(inp_words) > CACHE_POP_TRIGGER_LIMIT and last_comma > 0:
        sent_end = last_comma
        punc[sent_end] = PERIOD_INDEX
    return sent_end


def add_punc_text(text, infer_fn, token2id, punc_list=None, language="zh-cn"):
    """CTTransformer::AddPunc(const char*, language) (ct-transformer.cpp:39-155).  infer_fn(ids) -> punctuation ids."""
    pl = [p.encode("utf-8") for p in (punc_list or DEFAULT_PUNC_LIST)]
    words, ids = tokenize(text, token2id)
    n = len(ids)
    total = -(-n // TOKEN_LEN)
    rem_w, rem_i, new_punc, new_str = [], [], [], []
    out = []
    for i in range(0, n, TOKEN_LEN):
        inp_w = rem_w + words[i:i + TOKEN_LEN]
        inp_i = rem_i + ids[i:i + TOKEN_LEN]
        punc = [int(v) for v in infer_fn(inp_i)]
        cur = i // TOKEN_LEN
        if cur < total - 1:
            se = _cut_at_sentence_end(punc, inp_w, inp_i)
            rem_w, rem_i = inp_w[se + 1:], inp_i[se + 1:]
            inp_w, punc = inp_w[:se + 1], punc[:se + 1]
        new_punc += punc
        for k in range(len(inp_w)):
            # a space between two ASCII words — tested on the PREVIOUS word after it may already have got its own space, and
            # never for the first word of a mini-sentence (:98-103)
            if k > 0 and _is_ascii_byte(inp_w[k - 1][0]) and _is_ascii_byte(inp_w[k][0]):
                inp_w[k] = b" " + inp_w[k]
            new_str.append(inp_w[k])
            if punc[k] != NOTPUNC_INDEX:
                new_str.append(pl[punc[k]])
        out = list(new_str)
        if cur == total - 1:
            if new_str[-1] in (pl[COMMA_INDEX], pl[DUN_INDEX]):
                out = new_str[:-1] + [pl[PERIOD_INDEX]]
            elif new_str[-1] not in (pl[PERIOD_INDEX], pl[QUESTION_INDEX]):
                out = new_str + [pl[PERIOD_INDEX]]
    res = b"".join(out)
    if language == "en-bpe":
        for zh, en in (("，", b","), ("。", b"."), ("、", b","), ("？", b"?")):
            res = res.replace(zh.encode("utf-8"), en)
    return res.decode("utf-8", "replace")


def add_punc_text_online(text, cache, infer_fn, token2id, punc_list=None):
    """CTTransformerOnline::AddPunc(const char*, arr_cache, language) (ct-transformer-online.cpp:40-152).
    infer_fn(ids, cache_size) -> punctuation ids; cache (a list of byte strings) is updated in place."""
    pl = [p.encode("utf-8") for p in (punc_list or DEFAULT_PUNC_LIST)]
    tb = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    full = b"".join(cache)
    if full and _is_ascii_byte(full[-1]) and tb and _is_ascii_byte(tb[0]):
        full += b" "
    full += tb
    words, ids = tokenize(full, token2id)
    n = len(ids)
    total = -(-n // TOKEN_LEN)
    n_cache = len(cache)
    rem_w, rem_i = [], []
    punc_all, words_all = [], []
    for i in range(0, n, TOKEN_LEN):
        inp_w = rem_w + words[i:i + TOKEN_LEN]
        inp_i = rem_i + ids[i:i + TOKEN_LEN]
        punc = [int(v) for v in infer_fn(inp_i, n_cache)]
        if i // TOKEN_LEN < total - 1:
            se = _cut_at_sentence_end(punc, inp_w, inp_i)
            rem_w, rem_i = inp_w[se + 1:], inp_i[se + 1:]
            inp_w, punc = inp_w[:se + 1], punc[:se + 1]
        punc_all += punc
        words_all += inp_w
    out, skip = [], 0
    for i in range(len(words_all)):
        if _is_ascii_byte(words_all[i][0]) and i + 1 < len(words_all) and _is_ascii_byte(words_all[i + 1][0]):
            words_all[i] = words_all[i] + b" "
        if skip < n_cache:
            skip += 1
        else:
            out.append(words_all[i])
        if skip >= n_cache:
            # (:123-130) — also true for the LAST cached word, whose punctuation is therefore emitted again
            if pl[punc_all[i]] != b"_":                            # compared as strings against NOTPUNC here
                out.append(pl[punc_all[i]])
    se = -1
    for i in range(len(punc_all) - 2, 0, -1):
        if punc_all[i] in (PERIOD_INDEX, QUESTION_INDEX):
            se = i
            break
    cache[:] = words_all[se + 1:]
    if out and out[-1] in pl:                                      # IsPunc: any entry of punc_list, "<unk>" and "_" included
        out = out[:-1]
    return b"".join(out).decode("utf-8", "replace")
