"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (numpy fp32) restatement of the FSMN-VAD forward:
  FsmnVad::FbankKaldi      onnxruntime/src/fsmn-vad.cpp:137-152   (same knf chain as the ASR front end)
  FsmnVad::LfrCmvn         onnxruntime/src/fsmn-vad.cpp:198-238   (LFR m=5, n=1: com-define.h:103-109)
  FsmnVad::Forward         onnxruntime/src/fsmn-vad.cpp:72-135    (graph: UPSTREAM; caches [1,128,19,1] in-tree :98)
The graph itself is PARITY UNPINNED (onnxruntime + ModelScope file absent).  UPSTREAM architecture (FunASR FSMN):
in_linear1 400->140, in_linear2 140->250, ReLU, 4 x {linear 250->128 (no bias); memory block: out = p + causal
depthwise conv over the 20 most recent frames (left order 20 => 19-frame cache); affine 128->250; ReLU},
out_linear1 250->140, out_linear2 140->248, softmax.  Pinned in-tree: input 400 = 5 x 80, 4 caches of 128 x 19,
class 0 = silence (e2e-vad.h:103), caches only copied back when not final (fsmn-vad.cpp:129-134).
"""
from __future__ import annotations

import numpy as np

from . import frontend as fe
from . import paraformer as P

F32 = np.float32


def lfr_cmvn(fb, means, istd, lfr_m=5, lfr_n=1):
    """FsmnVad::LfrCmvn — textually the same routine as Paraformer::LfrCmvn."""
    return fe.lfr_cmvn(fb, means, istd, lfr_m, lfr_n)


def memory_block(p, w, cache):
    """p [T,128], w [128, lorder], cache [lorder-1, 128] -> (p + conv, new cache)."""
    lorder = w.shape[1]
    xcat = np.concatenate([cache, p], axis=0).astype(F32)
    T = p.shape[0]
    out = p.astype(F32).copy()
    for j in range(lorder):
        out += xcat[j:j + T] * w[:, j][None, :]
    return out.astype(F32), xcat[-(lorder - 1):].copy()


def forward(feats, W, caches):
    cfg = W.cfg
    x = P.linear(feats, W["in1.w"], W["in1.b"])
    x = np.maximum(P.linear(x, W["in2.w"], W["in2.b"]), F32(0))
    new_caches = []
    for i in range(cfg["layers"]):
        p = P.linear(x, W[f"blk.{i}.linear.w"])
        f, c = memory_block(p, W[f"blk.{i}.fsmn.w"], caches[i])
        new_caches.append(c)
        x = np.maximum(P.linear(f, W[f"blk.{i}.affine.w"], W[f"blk.{i}.affine.b"]), F32(0))
    x = P.linear(x, W["out1.w"], W["out1.b"])
    x = P.linear(x, W["out2.w"], W["out2.b"])
    m = x.max(axis=-1, keepdims=True)
    e = np.exp((x - m).astype(F32)).astype(F32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32), new_caches


class FsmnVad:
    """FsmnVad::Infer up to the score matrix (fsmn-vad.cpp:240-250), cache handling of :129-134."""

    def __init__(self, W):
        self.W = W
        self.InitCache()

    def InitCache(self):
        cfg = self.W.cfg
        self.in_cache_ = [np.zeros((cfg["lorder"] - 1, cfg["proj"]), F32) for _ in range(cfg["layers"])]

    def Forward(self, waves, is_final):
        cfg = self.W.cfg
        fb = fe.fbank(waves)
        if fb.shape[0] == 0:
            return np.zeros((0, cfg["n_out"]), F32)
        feats = lfr_cmvn(fb, self.W["cmvn.mean"], self.W["cmvn.istd"], cfg["lfr_m"], cfg["lfr_n"])
        probs, caches = forward(feats, self.W, self.in_cache_)
        if not is_final:
            self.in_cache_ = caches
        return probs
