"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (numpy fp32, BLAS-backed matmul) restatement of the Paraformer-large offline graph that the
reference runs as one opaque `m_session_->Run` (onnxruntime/src/paraformer.cpp:541) plus the C++
pieces around it:

  a4  the ONNX graph (UPSTREAM FunASR export) — batch 1, as the reference
      hard-wires (paraformer.cpp:470-473, 496-507)
  a5  Paraformer::GreedySearch / FindMax      paraformer.cpp:386-408, util.cpp:63-74
  a12 CIF integrate-and-fire recurrence       paraformer-online.cpp:306-327 (the reference's own C++
      statement of the scan; the offline graph uses the same recurrence inside the ONNX file)

PARITY UNPINNED for a4: the arithmetic of this span lives in onnxruntime 1.14.0 (absent:
.MISSING_LARGE_BLOBS:7-8) applied to ModelScope weight files (absent), and the reference holds no
fixture at this boundary.  What IS pinned in-tree and honoured here: tensor ranks /
dtypes / order at the Run boundary (paraformer.cpp:496-562), d_model 512 / 16 decoder blocks / FSMN
order 10 / CIF threshold 1.0 and tail 0.45 (paraformer.h:112-121), log-prob output consumed by
argmax with first-max-wins (util.cpp:63-74).
"""
from __future__ import annotations

import math

import numpy as np

from . import frontend

F32 = np.float32
LN_EPS = F32(1e-12)          # UPSTREAM ESPnet-style LayerNorm eps


def layer_norm(x, g, b):
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = (x - mu).astype(F32)
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + LN_EPS) * g + b).astype(F32)


def linear(x, w, b=None):
    """torch Linear: w is [out, in]."""
    y = x.astype(F32) @ w.T.astype(F32)
    if b is not None:
        y = y + b
    return y.astype(F32)


def fsmn(v, w):
    """Depthwise conv over time, kernel k, symmetric zero padding (k-1)/2, no bias, plus identity.
    w: [d, k].  v: [T, d]."""
    T, d = v.shape
    k = w.shape[1]
    lp = (k - 1) // 2
    vp = np.zeros((T + k - 1, d), F32)
    vp[lp:lp + T] = v
    out = v.astype(F32).copy()
    for j in range(k):
        out += vp[j:j + T] * w[:, j][None, :]
    return out.astype(F32)


def softmax_rows(s):
    m = s.max(axis=-1, keepdims=True)
    e = np.exp((s - m).astype(F32)).astype(F32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


def mha(q, k, v, n_head):
    """q [Lq, d], k/v [Lk, d]; returns context [Lq, d] (before the output projection)."""
    Lq, d = q.shape
    dk = d // n_head
    scale = F32(dk ** -0.5)
    out = np.empty((Lq, d), F32)
    for h in range(n_head):
        sl = slice(h * dk, (h + 1) * dk)
        s = ((q[:, sl] * scale).astype(F32) @ k[:, sl].T).astype(F32)
        p = softmax_rows(s)
        out[:, sl] = p @ v[:, sl]
    return out


class Weights:
    """name -> float32 ndarray view over the flat blob described by the manifest."""

    def __init__(self, manifest: dict, blob: np.ndarray):
        self.cfg = manifest["config"]
        self.t = {}
        for name, meta in manifest["tensors"].items():
            n = int(np.prod(meta["shape"])) if meta["shape"] else 1
            off = meta["offset"] // 4
            self.t[name] = blob[off:off + n].reshape(meta["shape"])

    def __getitem__(self, k):
        return self.t[k]

    def has(self, k):
        return k in self.t


def encoder_layer(x, W, pfx, n_head):
    in_size = x.shape[1]
    y = layer_norm(x, W[pfx + "norm1.g"], W[pfx + "norm1.b"])
    qkv = linear(y, W[pfx + "qkv.w"], W[pfx + "qkv.b"])
    d = qkv.shape[1] // 3
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    mem = fsmn(v, W[pfx + "fsmn.w"])
    ctx = mha(q, k, v, n_head)
    att = linear(ctx, W[pfx + "out.w"], W[pfx + "out.b"]) + mem
    x = (x + att).astype(F32) if in_size == d else att.astype(F32)
    y = layer_norm(x, W[pfx + "norm2.g"], W[pfx + "norm2.b"])
    h = np.maximum(linear(y, W[pfx + "ffn1.w"], W[pfx + "ffn1.b"]), F32(0))
    x = (x + linear(h, W[pfx + "ffn2.w"], W[pfx + "ffn2.b"])).astype(F32)
    return x


def embed(feats, d_model):
    """x * sqrt(d_model) + PE(depth = feature dim), positions from 1 (paraformer-online.cpp:549-555,
    240-268 state the same two steps for the streaming model)."""
    T, D = feats.shape
    return (feats * F32(math.sqrt(d_model)) + frontend.pos_emb(T, D)).astype(F32)


def encoder(feats, W):
    cfg = W.cfg
    x = embed(feats, cfg["d_model"])
    for i in range(cfg["enc_layers"]):
        x = encoder_layer(x, W, f"enc.{i}.", cfg["n_head"])
    return layer_norm(x, W["enc.after_norm.g"], W["enc.after_norm.b"])


def predictor_alphas(enc, W):
    """CifPredictorV2 (UPSTREAM): conv1d k=3 pad 1 -> (+residual if cfg) -> ReLU -> linear -> sigmoid ->
    relu(a*smooth - noise); then the tail: one extra slot with tail_threshold and a zero hidden frame."""
    cfg = W.cfg
    T, d = enc.shape
    cw = W["pred.conv.w"]            # [d_out, d_in, 3]
    hp = np.zeros((T + 2, d), F32)
    hp[1:T + 1] = enc
    mem = np.zeros((T, d), F32)
    for j in range(3):
        mem += hp[j:j + T] @ cw[:, :, j].T
    mem = (mem + W["pred.conv.b"]).astype(F32)
    if cfg.get("pred_residual", 0):
        mem = (mem + enc).astype(F32)
    o = np.maximum(mem, F32(0))
    logit = (o @ W["pred.out.w"].reshape(-1) + W["pred.out.b"].reshape(-1)[0]).astype(F32)
    alphas = (F32(1) / (F32(1) + np.exp(-logit))).astype(F32)
    alphas = np.maximum(alphas * F32(cfg.get("smooth_factor", 1.0)) - F32(cfg.get("noise_threshold", 0.0)), F32(0)).astype(F32)
    alphas = np.concatenate([alphas, np.asarray([cfg["tail_threshold"]], F32)]).astype(F32)
    hidden = np.concatenate([enc, np.zeros((1, d), F32)], axis=0)
    return hidden, alphas


def cif(hidden, alphas, threshold=1.0):
    """The scalar recurrence exactly as ParaformerOnline::CifSearch states it
    (paraformer-online.cpp:301-327), without the streaming cache handling."""
    thr = F32(threshold)
    integrate = F32(0.0)
    frames = np.zeros(hidden.shape[1], F32)
    out = []
    fires = []
    for i in range(alphas.shape[0]):
        alpha = F32(alphas[i])
        if F32(alpha + integrate) < thr:
            integrate = F32(integrate + alpha)
            fires.append(integrate)
            frames = (frames + alpha * hidden[i]).astype(F32)
        else:
            frames = (frames + F32(thr - integrate) * hidden[i]).astype(F32)
            out.append(frames.copy())
            integrate = F32(integrate + alpha)
            fires.append(integrate)
            integrate = F32(integrate - thr)
            frames = (integrate * hidden[i]).astype(F32)
    emb = np.stack(out).astype(F32) if out else np.zeros((0, hidden.shape[1]), F32)
    return emb, np.asarray(fires, F32)


def decoder_ffn(x, W, pfx):
    h = np.maximum(linear(x, W[pfx + "ffn1.w"], W[pfx + "ffn1.b"]), F32(0))
    h = layer_norm(h, W[pfx + "ffn_norm.g"], W[pfx + "ffn_norm.b"])
    return linear(h, W[pfx + "ffn2.w"])


def decoder_layer(x, mem, W, pfx, n_head):
    residual = x
    t = decoder_ffn(layer_norm(x, W[pfx + "norm1.g"], W[pfx + "norm1.b"]), W, pfx)
    t2 = layer_norm(t, W[pfx + "norm2.g"], W[pfx + "norm2.b"])
    x = (residual + fsmn(t2, W[pfx + "fsmn.w"])).astype(F32)
    residual = x
    y = layer_norm(x, W[pfx + "norm3.g"], W[pfx + "norm3.b"])
    q = linear(y, W[pfx + "q.w"], W[pfx + "q.b"])
    kv = linear(mem, W[pfx + "kv.w"], W[pfx + "kv.b"])
    d = q.shape[1]
    ctx = mha(q, kv[:, :d], kv[:, d:], n_head)
    x = (residual + linear(ctx, W[pfx + "out.w"], W[pfx + "out.b"])).astype(F32)
    return x


def hotword_embed(hotword_matrix, lengths, W):
    """model_eb.onnx + row selection (paraformer.cpp:656-685).  UPSTREAM: Embedding(vocab, d) -> LSTM(d, d), torch gate
    order i,f,g,o; output [10, H, d], row len_j-1 of hotword j is its embedding."""
    ids = np.asarray(hotword_matrix, np.int64)
    H, L = ids.shape
    d = W.cfg["d_model"]
    x = W["bias.embed.w"][ids].astype(F32)                     # [H, L, d]
    w_ih, w_hh = W["bias.lstm.w_ih"], W["bias.lstm.w_hh"]
    b_ih, b_hh = W["bias.lstm.b_ih"], W["bias.lstm.b_hh"]
    h = np.zeros((H, d), F32)
    c = np.zeros((H, d), F32)
    out = np.zeros((H, d), F32)
    sig = lambda z: (F32(1) / (F32(1) + np.exp(-z))).astype(F32)
    for t in range(L):
        g = (x[:, t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh).astype(F32)
        i_, f_, g_, o_ = g[:, :d], g[:, d:2 * d], g[:, 2 * d:3 * d], g[:, 3 * d:]
        c = (sig(f_) * c + sig(i_) * np.tanh(g_)).astype(F32)
        h = (sig(o_) * np.tanh(c)).astype(F32)
        pick = np.asarray(lengths) - 1 == t
        out[pick] = h[pick]
    return out


def _lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of torch.nn.LSTM (gates i,f,g,o) over x [L, d]; returns h for every step [L, H]."""
    L = x.shape[0]
    H = w_hh.shape[1]
    gx = (x @ w_ih.T + (b_ih + b_hh)).astype(F32)
    h = np.zeros(H, F32)
    c = np.zeros(H, F32)
    out = np.zeros((L, H), F32)
    sig = lambda z: (F32(1) / (F32(1) + np.exp(-z))).astype(F32)
    for t in (range(L - 1, -1, -1) if reverse else range(L)):
        g = (gx[t] + w_hh @ h).astype(F32)
        i_, f_, g_, o_ = g[:H], g[H:2 * H], g[2 * H:3 * H], g[3 * H:]
        c = (sig(f_) * c + sig(i_) * np.tanh(g_)).astype(F32)
        h = (sig(o_) * np.tanh(c)).astype(F32)
        out[t] = h
    return out


def cif_wo_hidden(alphas, threshold):
    """UPSTREAM cif_wo_hidden: the integrate value after each frame's alpha is added (before a fire is subtracted)."""
    integrate = F32(0)
    out = np.zeros(len(alphas), F32)
    for t, a in enumerate(alphas):
        integrate = F32(integrate + F32(a))
        out[t] = integrate
        if integrate >= F32(threshold):
            integrate = F32(integrate - F32(threshold))
    return out


def timestamp_head(enc, token_num, W):
    """CifPredictorV3.get_upsample_timestmap as exported for the timestamp model (SURVEY §8 row a6 producer; UPSTREAM,
    parity unpinned): enc [T, d] -> (us_alphas [3T], us_cif_peak [3T]), the two extra outputs Paraformer::Forward hands to
    TimestampOnnx (paraformer.cpp:545-562).  ConvTranspose1d(k = stride = 3) -> BLSTM -> Linear -> sigmoid ->
    relu(a * smooth_factor2 - noise_threshold2), rescaled so that it sums to token_num, then cif_wo_hidden(threshold - 1e-4)."""
    cfg = W.cfg
    T, d = enc.shape
    up = int(cfg.get("upsample_times", 3))
    wt = W["pred.up.w"]                                         # [ci, co, k]
    u = np.zeros((T, up, d), F32)
    for j in range(up):
        u[:, j, :] = (enc @ wt[:, :, j] + W["pred.up.b"]).astype(F32)      # out[3t + j] = x[t] @ W[:, :, j] + b
    u = u.reshape(T * up, d)
    hf = _lstm_dir(u, W["pred.blstm.w_ih"], W["pred.blstm.w_hh"], W["pred.blstm.b_ih"], W["pred.blstm.b_hh"], False)
    hb = _lstm_dir(u, W["pred.blstm.w_ih_r"], W["pred.blstm.w_hh_r"], W["pred.blstm.b_ih_r"], W["pred.blstm.b_hh_r"], True)
    y = np.concatenate([hf, hb], axis=1)
    z = (y @ W["pred.out2.w"].reshape(-1) + W["pred.out2.b"][0]).astype(F32)
    a2 = (F32(1) / (F32(1) + np.exp(-z))).astype(F32)
    a2 = np.maximum(a2 * F32(cfg.get("smooth_factor2", 0.25)) - F32(cfg.get("noise_threshold2", 0.01)), F32(0)).astype(F32)
    total = F32(np.cumsum(a2, dtype=F32)[-1])
    us_alphas = (a2 * (F32(token_num) / total)).astype(F32)
    return us_alphas, cif_wo_hidden(us_alphas, F32(cfg["cif_threshold"]) - F32(1e-4))


def contextual_last_layer(x, mem, hw_emb, W, pfx, n_head):
    """UPSTREAM ContextualDecoderLayer + ContextualBiasDecoder + bias_output: the last decoder
    layer's cross-attention output is concatenated with a cross-attention over the hotword embeddings and projected
    2d -> d, then added to the layer's self-attention (FSMN) output."""
    residual = x
    t = decoder_ffn(layer_norm(x, W[pfx + "norm1.g"], W[pfx + "norm1.b"]), W, pfx)
    t2 = layer_norm(t, W[pfx + "norm2.g"], W[pfx + "norm2.b"])
    x_self = (residual + fsmn(t2, W[pfx + "fsmn.w"])).astype(F32)
    d = x.shape[1]
    y = layer_norm(x_self, W[pfx + "norm3.g"], W[pfx + "norm3.b"])
    q = linear(y, W[pfx + "q.w"], W[pfx + "q.b"])
    kv = linear(mem, W[pfx + "kv.w"], W[pfx + "kv.b"])
    x_src = linear(mha(q, kv[:, :d], kv[:, d:], n_head), W[pfx + "out.w"], W[pfx + "out.b"])
    yb = layer_norm(x_self, W["bias.dec.norm3.g"], W["bias.dec.norm3.b"])
    qb = linear(yb, W["bias.dec.q.w"], W["bias.dec.q.b"])
    kvb = linear(hw_emb, W["bias.dec.kv.w"], W["bias.dec.kv.b"])
    cx = linear(mha(qb, kvb[:, :d], kvb[:, d:], n_head), W["bias.dec.out.w"], W["bias.dec.out.b"])
    merged = linear(np.concatenate([x_src, cx], axis=1), W["bias.out.w"])
    return (x_self + merged).astype(F32)


def decoder(emb, mem, W, hw_emb=None):
    cfg = W.cfg
    x = emb
    for i in range(cfg["dec_layers"]):
        if cfg.get("contextual", 0) and i == cfg["dec_layers"] - 1:
            x = contextual_last_layer(x, mem, hw_emb, W, f"dec.{i}.", cfg["n_head"])
        else:
            x = decoder_layer(x, mem, W, f"dec.{i}.", cfg["n_head"])
    # decoders3: FFN-only layer, NO residual (UPSTREAM DecoderLayerSANM with self_attn=src_attn=None)
    x = decoder_ffn(layer_norm(x, W["dec3.norm1.g"], W["dec3.norm1.b"]), W, "dec3.")
    x = layer_norm(x, W["dec.after_norm.g"], W["dec.after_norm.b"])
    logits = linear(x, W["dec.out.w"], W["dec.out.b"])
    m = logits.max(axis=-1, keepdims=True)
    z = (logits - m).astype(F32)
    lse = np.log(np.exp(z).sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    return (z - lse).astype(F32)


def find_max(row):
    """util.cpp:63-74 — strict '>' scan: first maximum wins; returns (val, idx)."""
    idx = int(np.argmax(row))
    return row[idx], idx


def greedy_search(logp, n_len):
    """paraformer.cpp:386-395 — argmax of the first n_len rows."""
    return [find_max(logp[i])[1] for i in range(min(n_len, logp.shape[0]))]


def forward_feats(feats, W, stages=None, hw_emb=None):
    """feats [T, 560] -> dict(logp [L, V], token_num, alphas [T+1], enc [T, d], emb [L, d], ids)."""
    enc = encoder(feats, W)
    hidden, alphas = predictor_alphas(enc, W)
    # token_num = floor(sum alphas); summed left-to-right in fp32 (the order the HIP scan uses; the
    # ONNX ReduceSum order is unknowable here and only matters within 1e-5 of an integer)
    token_num = int(math.floor(float(np.cumsum(alphas, dtype=F32)[-1])))
    emb, fires = cif(hidden, alphas, W.cfg["cif_threshold"])
    res = dict(enc=enc, alphas=alphas, token_num=token_num, emb=emb, fires=fires)
    if emb.shape[0] > 0:
        logp = decoder(emb, enc, W, hw_emb)
    else:
        logp = np.zeros((0, W.cfg["vocab"]), F32)
    res["logp"] = logp
    res["ids"] = greedy_search(logp, token_num)
    return res


def forward_pcm(waves, W, hw_emb=None):
    """Model::Forward for one utterance: pcm [-1,1) float32 -> result dict (paraformer.cpp:463-589)."""
    feats = frontend.extract_feats(waves, W["cmvn.mean"], W["cmvn.istd"])
    if feats.shape[0] == 0:
        return dict(feats=feats, logp=np.zeros((0, W.cfg["vocab"]), F32), token_num=0, ids=[])
    r = forward_feats(feats, W, hw_emb=hw_emb)
    r["feats"] = feats
    return r
