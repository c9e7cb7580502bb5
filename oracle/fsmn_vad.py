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


class FsmnVadOnline:
    """FsmnVadOnline::Infer up to the score matrix + the waveform handed to the scorer (fsmn-vad-online.cpp:11-151):
    FbankKaldi with input_cache_ (:11-38), ExtractFeats with lfr_splice_cache_ / reserve_waveforms_ (:40-88), OnlineLfrCmvn
    (:90-133).  Infer returns (probs [n, n_out], waves) — `waves` is what vad_scorer receives for its dB computation (:148)."""

    def __init__(self, W):
        self.W = W
        cfg = W.cfg
        self.lfr_m, self.lfr_n = cfg["lfr_m"], cfg["lfr_n"]
        self.fl, self.fs = 400, 160                      # 25 ms / 10 ms at 16 kHz (fsmn-vad-online.h:89-90)
        self.InitCache()
        self.ResetCache()

    def InitCache(self):
        cfg = self.W.cfg
        self.in_cache_ = [np.zeros((cfg["lorder"] - 1, cfg["proj"]), F32) for _ in range(cfg["layers"])]

    def ResetCache(self):
        self.reserve_waveforms_ = np.zeros(0, F32)
        self.input_cache_ = np.zeros(0, F32)
        self.lfr_splice_cache_ = np.zeros((0, 80), F32)

    def _fbank(self, waves):
        """FbankKaldi (:11-38): returns (frames, waves trimmed to the samples the frames cover)."""
        waves = np.concatenate([self.input_cache_, np.asarray(waves, F32)]).astype(F32)
        n = len(waves)
        frame_number = (n - self.fl) // self.fs + 1 if n >= self.fl else 0
        self.input_cache_ = waves[frame_number * self.fs:].copy()
        if frame_number == 0:
            return np.zeros((0, 80), F32), waves
        waves = waves[:(frame_number - 1) * self.fs + self.fl]
        return fe.fbank(waves), waves

    def _online_lfr_cmvn(self, feats, input_finished):
        m, n = self.lfr_m, self.lfr_n
        T = feats.shape[0]
        T_lrf = int(np.ceil((T - (m - 1) // 2) / float(n)))
        splice = T_lrf
        out = []
        for i in range(T_lrf):
            if m <= T - i * n:
                out.append(feats[i * n:i * n + m].reshape(-1))
            elif input_finished:
                pad = m - (T - i * n)
                out.append(np.concatenate([feats[i * n:].reshape(-1)] + [feats[-1]] * pad))
            else:
                splice = i
                break
        splice = min(T - 1, splice * n)
        self.lfr_splice_cache_ = feats[splice:].copy()
        rows = np.stack(out).astype(F32) if out else np.zeros((0, 80 * m), F32)
        rows = ((rows + self.W["cmvn.mean"][None, :]).astype(F32) * self.W["cmvn.istd"][None, :]).astype(F32)
        return rows, splice

    def ExtractFeats(self, waves, input_finished):
        m = self.lfr_m
        feats, waves = self._fbank(waves)
        rows = np.zeros((0, 80 * m), F32)
        if feats.shape[0] > 0:
            if len(self.reserve_waveforms_):
                waves = np.concatenate([self.reserve_waveforms_, waves]).astype(F32)
            if self.lfr_splice_cache_.shape[0] == 0:
                self.lfr_splice_cache_ = np.repeat(feats[:1], (m - 1) // 2, axis=0)
            if feats.shape[0] + self.lfr_splice_cache_.shape[0] >= m:
                feats = np.concatenate([self.lfr_splice_cache_, feats]).astype(F32)
                frame_from_waves = (len(waves) - self.fl) // self.fs + 1
                minus_frame = (m - 1) // 2 if len(self.reserve_waveforms_) == 0 else 0
                rows, splice = self._online_lfr_cmvn(feats, input_finished)
                reserve_frame_idx = abs(splice - minus_frame)
                self.reserve_waveforms_ = waves[reserve_frame_idx * self.fs:frame_from_waves * self.fs].copy()
                waves = waves[:(frame_from_waves - 1) * self.fs + self.fl]
            else:
                self.reserve_waveforms_ = waves[self.fl - self.fs:].copy()
                self.lfr_splice_cache_ = np.concatenate([self.lfr_splice_cache_, feats]).astype(F32)
                # the reference leaves the raw 80-dim frames in vad_feats here and runs the network on them (a latent
                # bug, only reachable with < 4 frames in a call); no rows are produced in this restatement
        else:
            if input_finished:
                if len(self.reserve_waveforms_):
                    waves = self.reserve_waveforms_
                feats = self.lfr_splice_cache_
                if feats.shape[0] > 0:
                    rows, _ = self._online_lfr_cmvn(feats, input_finished)
        if input_finished:
            self.InitCache_pending = True          # Reset() + ResetCache() happen after the forward below uses in_cache_
            self.ResetCache()
        return rows, waves

    def Infer(self, waves, input_finished):
        self.InitCache_pending = False
        caches_before = self.in_cache_
        rows, waves = self.ExtractFeats(waves, input_finished)
        if self.InitCache_pending:
            # the reference calls Reset() (fresh in_cache_) inside ExtractFeats, BEFORE Forward runs on this call's rows
            # (:84-87 precede :143): the last call of a stream is scored against zeroed caches
            self.InitCache()
            caches_before = self.in_cache_
        if rows.shape[0] == 0:
            return np.zeros((0, self.W.cfg["n_out"]), F32), waves
        probs, caches = forward(rows, self.W, caches_before)
        if not input_finished:
            self.in_cache_ = caches
        return probs, waves
