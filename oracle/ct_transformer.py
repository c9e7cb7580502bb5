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
