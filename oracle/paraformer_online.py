"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (numpy fp32) restatement of the chunk-streaming Paraformer of the reference, statement by statement:

  a8   ParaformerOnline::FbankKaldi / ExtractFeats / OnlineLfrCmvn   onnxruntime/src/paraformer-online.cpp:119-238
  a9   scaling + GetPosEmb                                          :549-555, 240-268
  a10  AddOverlapChunk / InitCache                                  :397-413, 347-384
  a11  ForwardChunk encoder Run                                     :426-466      (graph: UPSTREAM)
  a12  CifSearch                                                    :270-345
  a13  ForwardChunk decoder Run + cache rotation, OnlineGreedySearch :472-515, paraformer.cpp:362-371
       Forward (chunk scheduling, first/last chunk split, resets)    :525-601

The host-side control flow (sample caches, LFR splice cache, overlap window, CIF carry, resets) is PINNED by the
reference text it transcribes.  The two `Run` graphs are PARITY UNPINNED (onnxruntime + ModelScope weights absent,
SURVEY §8c); they are restated from the UPSTREAM architecture: the streaming encoder = the SAN-M encoder stack + the
predictor's alpha head applied to the chunk as given (scaling and PE are done by the caller, :549-555); the streaming
decoder = the offline decoder with the FSMN's symmetric padding replaced by a (kernel-1)-frame left cache
(`[1, 512, 10]` per layer, :374) and cross-attention over the current chunk's encoder output.
"""
from __future__ import annotations

import math

import numpy as np

from . import frontend as fe
from . import paraformer as P

F32 = np.float32


def compute_frame_num(sample_length, frame_sample_length, frame_shift_sample_length):
    """paraformer-online.h:25-31."""
    frame_num = int((sample_length - frame_sample_length) / frame_shift_sample_length + 1)
    if frame_num >= 1 and sample_length >= frame_sample_length:
        return frame_num
    return 0


def encoder_chunk(x, W):
    """Streaming encoder session (paraformer-online.cpp:448): [n,560] -> enc [n,512], alphas [n]."""
    cfg = W.cfg
    for i in range(cfg["enc_layers"]):
        x = P.encoder_layer(x, W, f"enc.{i}.", cfg["n_head"])
    enc = P.layer_norm(x, W["enc.after_norm.g"], W["enc.after_norm.b"])
    _, alphas = P.predictor_alphas(enc, W)
    return enc, alphas[:-1]


def fsmn_cached(t2, w, cache):
    """Causal FSMN with a (k-1)-frame left cache: xcat = [cache; t2]; out[n] = sum_j w[:,j]*xcat[n+j] + t2[n];
    new cache = last k-1 rows of xcat.  cache: [k-1, d] (time-major view of the reference's [1,d,k-1])."""
    k = w.shape[1]
    xcat = np.concatenate([cache, t2], axis=0).astype(F32)
    N = t2.shape[0]
    out = t2.astype(F32).copy()
    for j in range(k):
        out += xcat[j:j + N] * w[:, j][None, :]
    return out.astype(F32), xcat[-(k - 1):].copy()


def decoder_chunk(emb, enc, caches, W):
    """Streaming decoder session (paraformer-online.cpp:500): returns log-probs [N,V] and the new caches."""
    cfg = W.cfg
    x = emb
    new_caches = []
    for i in range(cfg["dec_layers"]):
        pfx = f"dec.{i}."
        residual = x
        t = P.decoder_ffn(P.layer_norm(x, W[pfx + "norm1.g"], W[pfx + "norm1.b"]), W, pfx)
        t2 = P.layer_norm(t, W[pfx + "norm2.g"], W[pfx + "norm2.b"])
        f, c = fsmn_cached(t2, W[pfx + "fsmn.w"], caches[i])
        new_caches.append(c)
        x = (residual + f).astype(F32)
        residual = x
        y = P.layer_norm(x, W[pfx + "norm3.g"], W[pfx + "norm3.b"])
        q = P.linear(y, W[pfx + "q.w"], W[pfx + "q.b"])
        kv = P.linear(enc, W[pfx + "kv.w"], W[pfx + "kv.b"])
        d = q.shape[1]
        ctx = P.mha(q, kv[:, :d], kv[:, d:], cfg["n_head"])
        x = (residual + P.linear(ctx, W[pfx + "out.w"], W[pfx + "out.b"])).astype(F32)
    x = P.decoder_ffn(P.layer_norm(x, W["dec3.norm1.g"], W["dec3.norm1.b"]), W, "dec3.")
    x = P.layer_norm(x, W["dec.after_norm.g"], W["dec.after_norm.b"])
    logits = P.linear(x, W["dec.out.w"], W["dec.out.b"])
    m = logits.max(axis=-1, keepdims=True)
    z = (logits - m).astype(F32)
    lse = np.log(np.exp(z).sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    return (z - lse).astype(F32), new_caches


class ParaformerOnline:
    """State and methods named after the reference class (paraformer-online.h)."""

    def __init__(self, W, chunk_size=(5, 10, 5)):
        self.W = W
        cfg = W.cfg
        self.chunk_size = list(chunk_size)
        self.lfr_m, self.lfr_n = cfg["lfr_m"], cfg["lfr_n"]
        self.feat_dims = cfg["n_mels"] * cfg["lfr_m"]
        self.encoder_size = cfg["d_model"]
        self.fsmn_lorder = cfg["kernel"] - 1
        self.cif_threshold = F32(cfg["cif_threshold"])
        self.tail_alphas = F32(cfg["tail_threshold"])
        self.sqrt_factor = F32(math.sqrt(self.encoder_size))          # :108
        self.frame_sample_length_ = 16000 // 1000 * 25                 # :114
        self.frame_shift_sample_length_ = 16000 // 1000 * 10           # :115
        self.means, self.istd = W["cmvn.mean"], W["cmvn.istd"]
        self.input_cache_ = np.zeros(0, F32)
        self.reserve_waveforms_ = np.zeros(0, F32)
        self.lfr_splice_cache_ = []
        self.chunk_log = []          # (chunk_feats, enc, alphas, emb, logp) per ForwardChunk, for the tests
        self.InitCache()

    # :347-384
    def InitCache(self):
        self.start_idx_cache_ = 0
        self.is_first_chunk = True
        self.is_last_chunk = False
        self.hidden_cache_ = [np.zeros(self.encoder_size, F32)]
        self.alphas_cache_ = [F32(0)]
        self.feats_cache_ = [np.zeros(self.feat_dims, F32) for _ in range(self.chunk_size[0] + self.chunk_size[2])]
        self.decoder_caches = [np.zeros((self.fsmn_lorder, self.encoder_size), F32) for _ in range(self.W.cfg["dec_layers"])]

    def Reset(self):
        self.InitCache()

    # :391-395
    def ResetCache(self):
        self.reserve_waveforms_ = np.zeros(0, F32)
        self.input_cache_ = np.zeros(0, F32)
        self.lfr_splice_cache_ = []

    # :119-145 — returns (wav_feats list of [80] frames, waves after the erase)
    def FbankKaldi(self, waves):
        waves = np.concatenate([self.input_cache_, waves]).astype(F32)
        frame_number = compute_frame_num(len(waves), self.frame_sample_length_, self.frame_shift_sample_length_)
        self.input_cache_ = waves[frame_number * self.frame_shift_sample_length_:].copy()
        if frame_number == 0:
            return [], waves
        waves = waves[:(frame_number - 1) * self.frame_shift_sample_length_ + self.frame_sample_length_]
        fb = fe.fbank(waves)
        return [fb[i] for i in range(fb.shape[0])], waves

    # :196-238
    def OnlineLfrCmvn(self, wav_feats, input_finished):
        out_feats = []
        T = len(wav_feats)
        lfr_m, lfr_n = self.lfr_m, self.lfr_n
        T_lrf = int(math.ceil((T - (lfr_m - 1) // 2) / float(lfr_n)))
        lfr_splice_frame_idxs = T_lrf
        for i in range(T_lrf):
            if lfr_m <= T - i * lfr_n:
                out_feats.append(np.concatenate(wav_feats[i * lfr_n:i * lfr_n + lfr_m]))
            else:
                if input_finished:
                    num_padding = lfr_m - (T - i * lfr_n)
                    p = list(wav_feats[i * lfr_n:]) + [wav_feats[-1]] * num_padding
                    out_feats.append(np.concatenate(p))
                else:
                    lfr_splice_frame_idxs = i
                    break
        lfr_splice_frame_idxs = min(T - 1, lfr_splice_frame_idxs * lfr_n)
        self.lfr_splice_cache_ = list(wav_feats[lfr_splice_frame_idxs:])
        out = [((f + self.means).astype(F32) * self.istd).astype(F32) for f in out_feats]
        return out, lfr_splice_frame_idxs

    # :147-194
    def ExtractFeats(self, waves, input_finished):
        wav_feats, waves = self.FbankKaldi(waves)
        lfr_m = self.lfr_m
        fs, fl = self.frame_shift_sample_length_, self.frame_sample_length_
        if len(wav_feats) > 0:
            if len(self.reserve_waveforms_):
                waves = np.concatenate([self.reserve_waveforms_, waves])
            if not self.lfr_splice_cache_:
                self.lfr_splice_cache_ = [wav_feats[0] for _ in range((lfr_m - 1) // 2)]
            if len(wav_feats) + len(self.lfr_splice_cache_) >= lfr_m:
                wav_feats = self.lfr_splice_cache_ + wav_feats
                frame_from_waves = (len(waves) - fl) // fs + 1
                minus_frame = (lfr_m - 1) // 2 if len(self.reserve_waveforms_) == 0 else 0
                wav_feats, lfr_splice_frame_idxs = self.OnlineLfrCmvn(wav_feats, input_finished)
                reserve_frame_idx = abs(lfr_splice_frame_idxs - minus_frame)
                self.reserve_waveforms_ = waves[reserve_frame_idx * fs:frame_from_waves * fs].copy()
            else:
                self.reserve_waveforms_ = waves[fl - fs:].copy()
                self.lfr_splice_cache_ = self.lfr_splice_cache_ + wav_feats
                wav_feats = []          # NOTE: the reference leaves the raw 80-dim frames in wav_feats here
                # (:172-177) and Forward then feeds them on; that only happens for < 4 frames of audio in a
                # non-final call, which the 2-pass server never produces (it sends 9600-sample chunks).
        else:
            if input_finished:
                wav_feats = list(self.lfr_splice_cache_)
                if len(wav_feats) > 0:
                    wav_feats, _ = self.OnlineLfrCmvn(wav_feats, input_finished)
        if input_finished:
            self.ResetCache()
        return wav_feats

    # :240-268
    def GetPosEmb(self, wav_feats):
        n = len(wav_feats)
        pe = fe.pos_emb(n, self.feat_dims, start=self.start_idx_cache_)
        self.start_idx_cache_ += n
        return [(wav_feats[i] + pe[i]).astype(F32) for i in range(n)]

    # :397-413
    def AddOverlapChunk(self, wav_feats, input_finished):
        wav_feats = list(self.feats_cache_) + list(wav_feats)
        if input_finished:
            self.feats_cache_ = wav_feats[-self.chunk_size[0]:]
            if not self.is_last_chunk:
                padding_length = sum(self.chunk_size) - len(wav_feats)
                wav_feats = wav_feats + [np.zeros(self.feat_dims, F32) for _ in range(max(padding_length, 0))]
        else:
            self.feats_cache_ = wav_feats[-(self.chunk_size[0] + self.chunk_size[2]):]
        return wav_feats

    # :270-345
    def CifSearch(self, hidden, alphas):
        hidden = [h for h in hidden]
        alphas = np.asarray(alphas, F32).copy()
        alphas[:self.chunk_size[0]] = 0.0
        chunk_size_suf = self.chunk_size[0] + self.chunk_size[1]
        alphas[chunk_size_suf:] = 0.0
        alphas = list(alphas)
        if len(self.hidden_cache_) > 0:
            hidden = list(self.hidden_cache_) + hidden
            alphas = list(self.alphas_cache_) + alphas
            self.hidden_cache_, self.alphas_cache_ = [], []
        if self.is_last_chunk:
            hidden.append(np.zeros(self.encoder_size, F32))
            alphas.append(self.tail_alphas)
        thr = self.cif_threshold
        integrate = F32(0.0)
        frames = np.zeros(self.encoder_size, F32)
        list_frame = []
        for i in range(len(alphas)):
            alpha = F32(alphas[i])
            if F32(alpha + integrate) < thr:
                integrate = F32(integrate + alpha)
                frames = (frames + alpha * hidden[i]).astype(F32)
            else:
                frames = (frames + F32(thr - integrate) * hidden[i]).astype(F32)
                list_frame.append(frames.copy())
                integrate = F32(integrate + alpha)
                integrate = F32(integrate - thr)
                frames = (integrate * hidden[i]).astype(F32)
        self.alphas_cache_ = [integrate]
        if integrate > 0.0:
            self.hidden_cache_ = [(frames / integrate).astype(F32)]
        else:
            self.hidden_cache_ = [frames.copy()]
        return list_frame

    # :415-523
    def ForwardChunk(self, chunk_feats):
        x = np.stack(chunk_feats).astype(F32)
        enc, alphas = encoder_chunk(x, self.W)
        list_frame = self.CifSearch([enc[i] for i in range(enc.shape[0])], alphas)
        ids, logp, emb = [], None, None
        if len(list_frame) > 0:
            emb = np.stack(list_frame).astype(F32)
            logp, self.decoder_caches = decoder_chunk(emb, enc, self.decoder_caches, self.W)
            ids = P.greedy_search(logp, emb.shape[0])        # OnlineGreedySearch paraformer.cpp:362-371
        self.chunk_log.append(dict(feats=x, enc=enc, alphas=alphas, emb=emb, logp=logp, ids=list(ids)))
        return ids

    # :525-601 — returns the list of token ids emitted by this call (the reference returns their text)
    def Forward(self, din, input_finished):
        waves = np.asarray(din, F32)
        if len(waves) < 16 * 60 and input_finished and not self.is_first_chunk:
            self.is_last_chunk = True
            result = self.ForwardChunk(list(self.feats_cache_))
            self.ResetCache()
            self.Reset()
            return result
        if self.is_first_chunk:
            self.is_first_chunk = False
        wav_feats = self.ExtractFeats(waves, input_finished)
        if len(wav_feats) == 0:
            return []
        wav_feats = [(f * self.sqrt_factor).astype(F32) for f in wav_feats]
        wav_feats = self.GetPosEmb(wav_feats)
        if input_finished:
            if len(wav_feats) + self.chunk_size[2] <= self.chunk_size[1]:
                self.is_last_chunk = True
                wav_feats = self.AddOverlapChunk(wav_feats, input_finished)
            else:
                first_chunk = self.AddOverlapChunk(list(wav_feats), input_finished)
                r1 = self.ForwardChunk(first_chunk)
                self.is_last_chunk = True
                k = len(wav_feats) + self.chunk_size[2] - self.chunk_size[1]
                last_chunk = self.AddOverlapChunk(list(wav_feats[-k:]), input_finished)
                r2 = self.ForwardChunk(last_chunk)
                self.ResetCache()
                self.Reset()
                return r1 + r2
        else:
            wav_feats = self.AddOverlapChunk(wav_feats, input_finished)
        result = self.ForwardChunk(wav_feats)
        if input_finished:
            self.ResetCache()
            self.Reset()
        return result
