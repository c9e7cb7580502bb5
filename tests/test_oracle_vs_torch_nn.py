"""CPU (build container; torch is importable here): every NN primitive of the oracle (oracle/paraformer.py, paraformer_online.py,
fsmn_vad.py, ct_transformer.py) against `torch.nn` modules assembled the way UPSTREAM FunASR assembles its layers —
nn.LayerNorm(eps=1e-12), nn.Linear, nn.Conv1d(groups=d, bias=False) behind nn.ConstantPad1d, nn.Conv1d(k=3, padding=1),
nn.ConvTranspose1d(k=stride=3), nn.LSTM (uni- and bidirectional, gate order i,f,g,o), nn.Embedding, torch.softmax.

The oracle was written from memory of those layers (VERDICT r3 "What's weak" 1: a shared misreading — LSTM gate order,
ConvTranspose1d weight layout, LayerNorm eps placement, FSMN padding side — would pass the HIP-vs-oracle tests on both sides).
Here the tensors go into the torch modules through the UPSTREAM state_dict names of convert.py's name map (so the torch LAYOUTS
that the file reader promises are checked as well: Linear [out, in], depthwise Conv1d [d, 1, k], ConvTranspose1d [in, out, k],
LSTM weight_ih_l0 / weight_hh_l0[_reverse]) and must reproduce the numpy restatement to fp32 round-off.

What this does NOT pin: that FunASR wires these modules in this order (that needs the real export; SURVEY §8c "parity
unpinned") — DESIGN §3 lists the wiring choices that remain hypotheses."""
import importlib
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
nn = torch.nn

from oracle import ct_transformer as C
from oracle import frontend as FE
from oracle import fsmn_vad as V
from oracle import paraformer as P
from oracle import paraformer_online as PO

TOL = 2e-5


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def close(a, b, tol=TOL):
    a = a.detach().numpy() if hasattr(a, "detach") else np.asarray(a)
    err = float(np.abs(a - np.asarray(b)).max()) if a.size else 0.0
    scale = max(1.0, float(np.abs(np.asarray(b)).max())) if a.size else 1.0
    assert a.shape == np.asarray(b).shape and err <= tol * scale, (a.shape, np.asarray(b).shape, err)


@pytest.fixture(scope="module")
def mods(pkg):
    return importlib.import_module(pkg.__name__ + ".convert"), importlib.import_module(pkg.__name__ + ".weights")


@pytest.fixture(scope="module")
def asr(mods):
    conv, wt = mods
    cfg = wt.small_config(enc_layers=2, dec_layers=2, vocab=113, contextual=1, timestamp=1, smooth_factor2=0.25, noise_threshold2=0.01)
    man, blob = wt.synth_weights(cfg, seed=77)
    W = P.Weights(man, blob)
    names = conv.paraformer_name_map(cfg)
    state = {}
    for k, key in names.items():
        a = W[k]
        if k.endswith("fsmn.w"):
            a = a[:, None, :]                    # depthwise Conv1d(d, d, k, groups=d): [d, 1, k]
        if k == "bias.out.w":
            a = a[:, :, None]                    # Conv1d(2d, d, 1): [d, 2d, 1]
        state[key] = t(a)
    return cfg, W, state


def load(module, state, prefix):
    sd = {k[len(prefix) + 1:]: v for k, v in state.items() if k.startswith(prefix + ".")}
    missing = set(module.state_dict()) - set(sd)
    assert not missing, (prefix, missing)
    module.load_state_dict({k: sd[k] for k in module.state_dict()})
    return module.eval()


# ---- UPSTREAM-style modules out of torch.nn (funasr/models/sanm/attention.py, encoder.py, decoder.py as recalled) -------------------
class AttSANM(nn.Module):
    def __init__(self, n_head, in_feat, n_feat, kernel, shift=0):
        super().__init__()
        self.h, self.d_k = n_head, n_feat // n_head
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.linear_q_k_v = nn.Linear(in_feat, n_feat * 3)
        self.fsmn_block = nn.Conv1d(n_feat, n_feat, kernel, stride=1, padding=0, groups=n_feat, bias=False)
        left = (kernel - 1) // 2 + shift
        self.pad_fn = nn.ConstantPad1d((left, kernel - 1 - left), 0.0)

    def forward(self, x, mask=None):           # x [1, T, in]
        q, k, v = torch.split(self.linear_q_k_v(x), self.h * self.d_k, dim=-1)
        b, T, _ = q.shape
        heads = lambda z: z.reshape(b, T, self.h, self.d_k).transpose(1, 2)
        mem = self.fsmn_block(self.pad_fn(v.transpose(1, 2))).transpose(1, 2) + v
        scores = torch.matmul(heads(q) * self.d_k ** (-0.5), heads(k).transpose(-2, -1))
        if mask is not None:
            scores = scores.masked_fill(mask[None, None] == 0, float("-inf"))
        ctx = torch.matmul(torch.softmax(scores, dim=-1), heads(v)).transpose(1, 2).reshape(b, T, self.h * self.d_k)
        return self.linear_out(ctx) + mem


class FFN(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.w_1, self.w_2 = nn.Linear(d, hidden), nn.Linear(hidden, d)

    def forward(self, x):
        return self.w_2(torch.relu(self.w_1(x)))


class EncLayer(nn.Module):
    def __init__(self, in_size, d, n_head, ffn, kernel, shift=0):
        super().__init__()
        self.in_size, self.size = in_size, d
        self.self_attn, self.feed_forward = AttSANM(n_head, in_size, d, kernel, shift), FFN(d, ffn)
        self.norm1, self.norm2 = nn.LayerNorm(in_size, eps=1e-12), nn.LayerNorm(d, eps=1e-12)

    def forward(self, x, mask=None):
        a = self.self_attn(self.norm1(x), mask)
        x = x + a if self.in_size == self.size else a
        return x + self.feed_forward(self.norm2(x))


class DecFFN(nn.Module):
    def __init__(self, d, hidden):
        super().__init__()
        self.w_1, self.w_2, self.norm = nn.Linear(d, hidden), nn.Linear(hidden, d, bias=False), nn.LayerNorm(hidden, eps=1e-12)

    def forward(self, x):
        return self.w_2(self.norm(torch.relu(self.w_1(x))))


class DecFsmn(nn.Module):                       # MultiHeadedAttentionSANMDecoder
    def __init__(self, d, kernel, shift=0):
        super().__init__()
        self.fsmn_block = nn.Conv1d(d, d, kernel, stride=1, padding=0, groups=d, bias=False)
        left = (kernel - 1) // 2 + shift
        self.pad_fn = nn.ConstantPad1d((left, kernel - 1 - left), 0.0)

    def forward(self, x, cache=None):           # cache [1, d, k-1]: the streaming decoder concatenates instead of padding
        z = x.transpose(1, 2)
        z = self.pad_fn(z) if cache is None else torch.cat((cache, z), dim=2)
        return self.fsmn_block(z).transpose(1, 2) + x, z[:, :, -(self.fsmn_block.kernel_size[0] - 1):]


class CrossAtt(nn.Module):                      # MultiHeadedAttentionCrossAtt
    def __init__(self, n_head, d):
        super().__init__()
        self.h, self.d_k = n_head, d // n_head
        self.linear_q, self.linear_k_v, self.linear_out = nn.Linear(d, d), nn.Linear(d, 2 * d), nn.Linear(d, d)

    def forward(self, x, memory):
        b = x.shape[0]
        k, v = torch.split(self.linear_k_v(memory), self.h * self.d_k, dim=-1)
        heads = lambda z: z.reshape(b, -1, self.h, self.d_k).transpose(1, 2)
        scores = torch.matmul(heads(self.linear_q(x)) * self.d_k ** (-0.5), heads(k).transpose(-2, -1))
        ctx = torch.matmul(torch.softmax(scores, dim=-1), heads(v)).transpose(1, 2).reshape(b, -1, self.h * self.d_k)
        return self.linear_out(ctx)


class DecLayer(nn.Module):                      # DecoderLayerSANM
    def __init__(self, d, n_head, ffn, kernel, shift=0, attn=True):
        super().__init__()
        self.feed_forward, self.norm1 = DecFFN(d, ffn), nn.LayerNorm(d, eps=1e-12)
        self.self_attn = DecFsmn(d, kernel, shift) if attn else None
        self.src_attn = CrossAtt(n_head, d) if attn else None
        if attn:
            self.norm2, self.norm3 = nn.LayerNorm(d, eps=1e-12), nn.LayerNorm(d, eps=1e-12)

    def forward(self, tgt, memory, cache=None):
        residual = tgt
        x = self.feed_forward(self.norm1(tgt))
        new_cache = None
        if self.self_attn is not None:
            y, new_cache = self.self_attn(self.norm2(x), cache)
            x = residual + y
            x = x + self.src_attn(self.norm3(x), memory)
        return x, new_cache


# ---- the primitives ---------------------------------------------------------------------------------------------------------
def test_layer_norm_eps_inside_the_root(asr):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((9, 512)).astype(np.float32) * 1e-7          # tiny rows: eps placement would show
    g, b = rng.standard_normal(512).astype(np.float32), rng.standard_normal(512).astype(np.float32)
    ln = nn.LayerNorm(512, eps=1e-12)
    ln.load_state_dict({"weight": t(g), "bias": t(b)})
    close(ln(t(x)), P.layer_norm(x, g, b), 1e-4)
    x = rng.standard_normal((9, 512)).astype(np.float32) * 30
    close(ln(t(x)), P.layer_norm(x, g, b))


def test_encoder_layers_first_and_inner(asr):
    cfg, W, state = asr
    rng = np.random.default_rng(1)
    x0 = rng.standard_normal((1, 37, 560)).astype(np.float32)
    l0 = load(EncLayer(560, 512, cfg["n_head"], cfg["ffn"], cfg["kernel"]), state, "encoder.encoders0.0")
    y0 = l0(t(x0))
    close(y0[0], P.encoder_layer(x0[0], W, "enc.0.", cfg["n_head"]))           # in_size != size: no attention residual
    l1 = load(EncLayer(512, 512, cfg["n_head"], cfg["ffn"], cfg["kernel"]), state, "encoder.encoders.0")
    close(l1(y0)[0], P.encoder_layer(y0[0].detach().numpy(), W, "enc.1.", cfg["n_head"]))
    an = load(nn.LayerNorm(512, eps=1e-12), state, "encoder.after_norm")
    close(an(y0)[0], P.layer_norm(y0[0].detach().numpy(), W["enc.after_norm.g"], W["enc.after_norm.b"]))


def test_sinusoidal_position_encoding_and_scale():
    """SinusoidalPositionEncoder (UPSTREAM): positions from 1, depth = feature dim, sin half | cos half,
    log_timescale_increment = log(10000) / (depth / 2 - 1); the reference states the same in C++ (paraformer-online.cpp:240-268)."""
    T, D = 23, 560
    pos = torch.arange(1, T + 1, dtype=torch.float32)[None]
    inc = math.log(10000.0) / (D / 2 - 1)
    inv = torch.exp(torch.arange(D // 2, dtype=torch.float32) * (-inc))
    scaled = pos[:, :, None] * inv[None, None, :]
    enc = torch.cat([torch.sin(scaled), torch.cos(scaled)], dim=2)[0]
    close(enc, FE.pos_emb(T, D), 2e-5)
    x = np.random.default_rng(2).standard_normal((T, D)).astype(np.float32)
    close(t(x) * 512 ** 0.5 + enc, P.embed(x, 512), 2e-5)


def test_cif_predictor_conv_and_alphas(asr):
    cfg, W, state = asr
    rng = np.random.default_rng(3)
    enc = rng.standard_normal((41, 512)).astype(np.float32)
    conv = load(nn.Conv1d(512, 512, 3, padding=0), state, "predictor.cif_conv1d")         # behind ConstantPad1d((1, 1))
    out = load(nn.Linear(512, 1), state, "predictor.cif_output")
    q = nn.ConstantPad1d((1, 1), 0.0)(t(enc).T[None])
    o = torch.relu(conv(q)).transpose(1, 2)
    alphas = torch.relu(torch.sigmoid(out(o))[0, :, 0] * cfg["smooth_factor"] - cfg["noise_threshold"])
    hidden, want = P.predictor_alphas(enc, W)
    close(alphas, want[:-1])
    assert want[-1] == np.float32(cfg["tail_threshold"]) and hidden.shape == (42, 512) and not hidden[-1].any()


def test_decoder_layers_and_head(asr):
    cfg, W, state = asr
    rng = np.random.default_rng(4)
    emb = rng.standard_normal((1, 11, 512)).astype(np.float32)
    mem = rng.standard_normal((1, 29, 512)).astype(np.float32)
    l0 = load(DecLayer(512, cfg["n_head"], cfg["dec_ffn"], cfg["kernel"]), state, "decoder.decoders.0")
    y, _ = l0(t(emb), t(mem))
    close(y[0], P.decoder_layer(emb[0], mem[0], W, "dec.0.", cfg["n_head"]))
    l3 = load(DecLayer(512, cfg["n_head"], cfg["dec_ffn"], cfg["kernel"], attn=False), state, "decoder.decoders3.0")
    an = load(nn.LayerNorm(512, eps=1e-12), state, "decoder.after_norm")
    ol = load(nn.Linear(512, cfg["vocab"]), state, "decoder.output_layer")
    z, _ = l3(y, t(mem))                                                       # FFN only, no residual
    logp = torch.log_softmax(ol(an(z)), dim=-1)[0]
    x = y[0].detach().numpy()
    x = P.decoder_ffn(P.layer_norm(x, W["dec3.norm1.g"], W["dec3.norm1.b"]), W, "dec3.")
    x = P.layer_norm(x, W["dec.after_norm.g"], W["dec.after_norm.b"])
    logits = P.linear(x, W["dec.out.w"], W["dec.out.b"])
    m = logits.max(-1, keepdims=True)
    close(logp, logits - m - np.log(np.exp(logits - m).sum(-1, keepdims=True)), 5e-5)


def test_streaming_decoder_fsmn_cache_is_the_causal_conv(asr):
    """The online decoder's FSMN (kernel 11, sanm_shfit 5: left padding 10, right 0) fed chunk by chunk with a [1, d, 10] cache =
    one pass over the whole sequence; oracle.paraformer_online.fsmn_cached states the same recurrence."""
    cfg, W, state = asr
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 17, 512)).astype(np.float32)
    m = load(DecFsmn(512, cfg["kernel"], shift=5), {"a.fsmn_block.weight": state["decoder.decoders.0.self_attn.fsmn_block.weight"]}, "a")
    whole, _ = m(t(x))
    cache_t, cache_o = torch.zeros(1, 512, 10), np.zeros((10, 512), np.float32)
    for a, b in ((0, 6), (6, 7), (7, 17)):
        yt, cache_t = m(t(x[:, a:b]), cache_t)
        yo, cache_o = PO.fsmn_cached(x[0, a:b], W["dec.0.fsmn.w"], cache_o)
        close(yt[0], yo)
        close(yt[0], whole[0, a:b].detach().numpy())
        close(cache_t[0].T, cache_o)


def test_lstm_gate_order_hotword_embedder(asr):
    cfg, W, state = asr
    emb = load(nn.Embedding(cfg["vocab"], 512), state, "bias_embed")
    lstm = load(nn.LSTM(512, 512, 1, batch_first=True), {"a." + k.split(".", 1)[1]: v for k, v in state.items() if k.startswith("bias_encoder.")}, "a")
    ids = np.array([[5, 9, 11, 0, 0, 0, 0, 0, 0, 0], [17, 0, 0, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0]], np.int64)
    lens = np.array([3, 1, 1])
    out, _ = lstm(emb(torch.from_numpy(ids)))
    picked = torch.stack([out[j, lens[j] - 1] for j in range(3)])
    close(picked, P.hotword_embed(ids, lens, W))


def test_timestamp_head_conv_transpose_and_blstm(asr):
    cfg, W, state = asr
    rng = np.random.default_rng(6)
    enc = rng.standard_normal((19, 512)).astype(np.float32)
    up = load(nn.ConvTranspose1d(512, 512, 3, 3), state, "predictor.upsample_cnn")
    blstm = load(nn.LSTM(512, 512, 1, bias=True, batch_first=True, bidirectional=True), state, "predictor.blstm")
    out2 = load(nn.Linear(1024, 1), state, "predictor.cif_output2")
    o2 = up(t(enc).T[None]).transpose(1, 2)                                    # [1, 3T, d]
    y, _ = blstm(o2)
    a2 = torch.relu(torch.sigmoid(out2(y))[0, :, 0] * cfg["smooth_factor2"] - cfg["noise_threshold2"])
    token_num = 7
    us = a2 * (token_num / a2.sum())
    want_a, want_p = P.timestamp_head(enc, token_num, W)
    close(us, want_a, 1e-4)
    # cif_wo_hidden (UPSTREAM): integrate += alpha; record; fire subtracts the threshold
    integ, peaks = 0.0, []
    for a in us.detach().numpy():
        integ = np.float32(integ + a)
        peaks.append(integ)
        if integ >= np.float32(cfg["cif_threshold"]) - np.float32(1e-4):
            integ = np.float32(integ - (np.float32(cfg["cif_threshold"]) - np.float32(1e-4)))
    close(np.asarray(peaks, np.float32), want_p, 1e-4)


def test_contextual_bias_decoder_layer(asr):
    cfg, W, state = asr
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 9, 512)).astype(np.float32)
    mem = rng.standard_normal((1, 21, 512)).astype(np.float32)
    hw = rng.standard_normal((1, 4, 512)).astype(np.float32)
    last = load(DecLayer(512, cfg["n_head"], cfg["dec_ffn"], cfg["kernel"]), state, "decoder.last_decoder")
    norm3b = load(nn.LayerNorm(512, eps=1e-12), state, "decoder.bias_decoder.norm3")
    catt = load(CrossAtt(cfg["n_head"], 512), state, "decoder.bias_decoder.src_attn")
    bout = load(nn.Conv1d(1024, 512, 1, bias=False), state, "decoder.bias_output")
    # ContextualDecoderLayer: x_self = residual + fsmn(norm2(ffn(norm1))); x_src = src_attn(norm3(x_self), memory)
    tgt = t(x)
    xf = last.feed_forward(last.norm1(tgt))
    x_self = tgt + last.self_attn(last.norm2(xf))[0]
    x_src = last.src_attn(last.norm3(x_self), t(mem))
    cx = catt(norm3b(x_self), t(hw))
    merged = bout(torch.cat([x_src, cx], dim=2).transpose(1, 2)).transpose(1, 2)
    close((x_self + merged)[0], P.contextual_last_layer(x[0], mem[0], hw[0], W, f"dec.{cfg['dec_layers'] - 1}.", cfg["n_head"]))


def test_fsmn_vad_memory_block_is_a_causal_conv2d(mods):
    """FSMN-VAD (UPSTREAM funasr/models/fsmn_vad_streaming/encoder.py as recalled): LinearTransform (no bias) -> FSMNBlock:
    conv_left = Conv2d(proj, proj, [lorder, 1], groups=proj, bias=False) over the cache-prefixed sequence, out = x + conv -> AffineTransform
    -> ReLU."""
    conv, wt = mods
    man, blob = wt.synth_vad_weights(seed=78)
    W = P.Weights(man, blob)
    cfg = man["config"]
    rng = np.random.default_rng(8)
    feats = rng.standard_normal((31, cfg["input_dim"])).astype(np.float32)
    names = conv.vad_name_map(cfg)
    st = {key: t(W[k][:, None, :, None] if k.endswith("fsmn.w") else W[k]) for k, key in names.items()}
    lin = lambda key, i, o, bias=True: load(nn.Linear(i, o, bias=bias), st, key)
    x = lin("encoder.in_linear1.linear", 400, 140)(t(feats))
    x = torch.relu(lin("encoder.in_linear2.linear", 140, 250)(x))
    caches_t = []
    for i in range(cfg["layers"]):
        p = lin(f"encoder.fsmn.{i}.linear.linear", 250, 128, bias=False)(x)
        c2 = load(nn.Conv2d(128, 128, [cfg["lorder"], 1], dilation=[1, 1], stride=[1, 1], groups=128, bias=False), st, f"encoder.fsmn.{i}.fsmn_block.conv_left")
        z = p.T[None, :, :, None]                                              # [1, proj, T, 1]
        zc = torch.cat((torch.zeros(1, 128, cfg["lorder"] - 1, 1), z), dim=2)  # the cache in front
        caches_t.append(zc[:, :, -(cfg["lorder"] - 1):, :])
        f = (z + c2(zc))[0, :, :, 0].T
        x = torch.relu(lin(f"encoder.fsmn.{i}.affine.linear", 128, 250)(f))
    x = lin("encoder.out_linear2.linear", 140, 248)(lin("encoder.out_linear1.linear", 250, 140)(x))
    probs, caches_o = V.forward(feats, W, [np.zeros((cfg["lorder"] - 1, 128), np.float32) for _ in range(cfg["layers"])])
    close(torch.softmax(x, dim=-1), probs)
    for ct, co in zip(caches_t, caches_o):
        close(ct[0, :, :, 0].T, co)


@pytest.mark.parametrize("shift", [0, 5])
def test_ct_transformer_blocks_with_vad_mask_and_shifted_fsmn(mods, shift):
    conv, wt = mods
    cfg = dict(wt.CT_TRANSFORMER, vocab=97, sanm_shift=shift)
    man, blob = wt.synth_punc_weights(cfg, seed=79)
    W = P.Weights(man, blob)
    st = {key: t(W[k][:, None, :] if k.endswith("fsmn.w") else W[k]) for k, key in conv.punc_name_map(cfg).items()}
    ids = np.array([3, 9, 11, 50, 7, 2, 96, 4, 4, 13, 8], np.int64)
    emb = load(nn.Embedding(97, 256), st, "embed")
    x = emb(torch.from_numpy(ids))[None] * 256 ** 0.5 + t(FE.pos_emb(len(ids), 256))
    mask = t(C.vad_mask(len(ids), 6)) if shift else None
    for i in range(cfg["layers"]):
        src = "encoder.encoders0.0" if i == 0 else f"encoder.encoders.{i - 1}"
        x = load(EncLayer(256, 256, 8, 1024, 11, shift), st, src)(x, mask)
    x = load(nn.LayerNorm(256, eps=1e-12), st, "encoder.after_norm")(x)
    logits = load(nn.Linear(256, 6), st, "decoder")(x)[0]
    want = C.forward_online(ids, W, 6)[0] if shift else C.forward(ids, W)
    close(logits, want, 5e-5)
