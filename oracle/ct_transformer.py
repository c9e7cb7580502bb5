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
