"""Weight container for the HIP Paraformer path + a seeded synthetic-weight generator.

Container = one flat little-endian float32 blob + a JSON manifest
    {"config": {...}, "tensors": {name: {"shape": [...], "offset": <bytes>}}}
The file contract it stands in for is the reference's model directory
(`model.onnx`, `am.mvn`, `config.yaml`; onnxruntime/include/com-define.h:52-73) — the real files are
downloaded from ModelScope at server start (websocket/bin/funasr-wss-server.cpp:203-320) and are not
available offline, so benchmarks and tests use random-init weights of the same architecture
(SURVEY.md §8d "Synthetic inputs", appendix A for the tensor shapes).

Linear weights are stored torch-style `[out, in]` (K contiguous), which is also the layout the MFMA
GEMM kernels stream as their B operand.
"""
from __future__ import annotations

import json
import math

import numpy as np

ALIGN = 256  # bytes

# Paraformer-large offline (in-tree corroboration paraformer.h:112-121)
PARAFORMER_LARGE = dict(
    d_model=512, n_head=4, ffn=2048, enc_layers=50, dec_layers=16, dec_ffn=2048, kernel=11,
    vocab=8404, n_mels=80, lfr_m=7, lfr_n=6, cif_threshold=1.0, tail_threshold=0.45,
    smooth_factor=1.0, noise_threshold=0.0, pred_residual=0,
)


def small_config(**over):
    """Same widths as Paraformer-large (kernels are specialised for d_model 512 / d_k 128) but few
    layers and a small ragged vocabulary, so the CPU oracle finishes in seconds."""
    cfg = dict(PARAFORMER_LARGE)
    cfg.update(enc_layers=3, dec_layers=2, vocab=1003)
    cfg.update(over)
    return cfg


def tensor_specs(cfg):
    """Ordered (name, shape, init) list. init: ('normal', std) | ('const', v)."""
    d, f, fd, V, k = cfg["d_model"], cfg["ffn"], cfg["dec_ffn"], cfg["vocab"], cfg["kernel"]
    feat = cfg["n_mels"] * cfg["lfr_m"]
    specs = []

    def lin(name, out_f, in_f, bias=True):
        specs.append((name + ".w", [out_f, in_f], ("normal", 1.0 / math.sqrt(in_f))))
        if bias:
            specs.append((name + ".b", [out_f], ("normal", 0.02)))

    def ln(name, n):
        specs.append((name + ".g", [n], ("const", 1.0)))
        specs.append((name + ".b", [n], ("const", 0.0)))

    specs.append(("cmvn.mean", [feat], ("const", -8.0)))
    specs.append(("cmvn.istd", [feat], ("const", 0.3)))
    for i in range(cfg["enc_layers"]):
        p = f"enc.{i}."
        in_f = feat if i == 0 else d
        ln(p + "norm1", in_f)
        lin(p + "qkv", 3 * d, in_f)
        specs.append((p + "fsmn.w", [d, k], ("normal", 1.0 / math.sqrt(k))))
        lin(p + "out", d, d)
        ln(p + "norm2", d)
        lin(p + "ffn1", f, d)
        lin(p + "ffn2", d, f)
    ln("enc.after_norm", d)
    specs.append(("pred.conv.w", [d, d, 3], ("normal", 1.0 / math.sqrt(3 * d))))
    specs.append(("pred.conv.b", [d], ("normal", 0.02)))
    # alphas ~ sigmoid(N(bias, ~0.6)): mean ~0.25 -> ~4 tokens per second of audio
    specs.append(("pred.out.w", [1, d], ("normal", 1.0 / math.sqrt(d))))
    specs.append(("pred.out.b", [1], ("const", -1.25)))
    for i in range(cfg["dec_layers"]):
        p = f"dec.{i}."
        ln(p + "norm1", d)
        lin(p + "ffn1", fd, d)
        ln(p + "ffn_norm", fd)
        lin(p + "ffn2", d, fd, bias=False)
        ln(p + "norm2", d)
        specs.append((p + "fsmn.w", [d, k], ("normal", 1.0 / math.sqrt(k))))
        ln(p + "norm3", d)
        lin(p + "q", d, d)
        lin(p + "kv", 2 * d, d)
        lin(p + "out", d, d)
    ln("dec3.norm1", d)
    lin("dec3.ffn1", fd, d)
    ln("dec3.ffn_norm", fd)
    lin("dec3.ffn2", d, fd, bias=False)
    ln("dec.after_norm", d)
    lin("dec.out", V, d)
    if cfg.get("timestamp", 0):
        # CifPredictorV3 timestamp head (UPSTREAM bicif_paraformer, upsample_type cnn_blstm): ConvTranspose1d(d, d, k=3,
        # stride=3) -> BLSTM(d, d) -> Linear(2d, 1); torch layouts ([in, out, k]; gates i,f,g,o; *_r = reverse direction)
        specs.append(("pred.up.w", [d, d, 3], ("normal", 1.0 / math.sqrt(d))))
        specs.append(("pred.up.b", [d], ("normal", 0.02)))
        for sfx in ("", "_r"):
            specs.append((f"pred.blstm.w_ih{sfx}", [4 * d, d], ("normal", 1.0 / math.sqrt(d))))
            specs.append((f"pred.blstm.w_hh{sfx}", [4 * d, d], ("normal", 1.0 / math.sqrt(d))))
            specs.append((f"pred.blstm.b_ih{sfx}", [4 * d], ("normal", 0.1)))
            specs.append((f"pred.blstm.b_hh{sfx}", [4 * d], ("normal", 0.1)))
        specs.append(("pred.out2.w", [1, 2 * d], ("normal", 1.0 / math.sqrt(2 * d))))
        specs.append(("pred.out2.b", [1], ("const", 0.3)))
    if cfg.get("contextual", 0):
        # hotword embedder (model_eb.onnx: Embedding + 1-layer LSTM) and the bias decoder of the last layer
        specs.append(("bias.embed.w", [V, d], ("normal", 1.0)))
        specs.append(("bias.lstm.w_ih", [4 * d, d], ("normal", 1.0 / math.sqrt(d))))
        specs.append(("bias.lstm.w_hh", [4 * d, d], ("normal", 1.0 / math.sqrt(d))))
        specs.append(("bias.lstm.b_ih", [4 * d], ("normal", 0.1)))
        specs.append(("bias.lstm.b_hh", [4 * d], ("normal", 0.1)))
        ln("bias.dec.norm3", d)
        lin("bias.dec.q", d, d)
        lin("bias.dec.kv", 2 * d, d)
        lin("bias.dec.out", d, d)
        specs.append(("bias.out.w", [d, 2 * d], ("normal", 1.0 / math.sqrt(2 * d))))
    return specs


def build_manifest(cfg):
    tensors = {}
    off = 0
    for name, shape, _ in tensor_specs(cfg):
        n = int(np.prod(shape))
        tensors[name] = {"shape": shape, "offset": off}
        off += (n * 4 + ALIGN - 1) // ALIGN * ALIGN
    return {"config": cfg, "tensors": tensors, "total_bytes": off}


def synth_weights(cfg, seed=1234):
    """Returns (manifest dict, float32 blob ndarray).  Seed 1234, N(0, 1/fan_in), LayerNorm gamma 1
    beta 0 with a small seeded perturbation so that gamma/beta code paths are exercised."""
    man = build_manifest(cfg)
    blob = np.zeros(man["total_bytes"] // 4, np.float32)
    rng = np.random.default_rng(seed)
    for name, shape, init in tensor_specs(cfg):
        n = int(np.prod(shape))
        o = man["tensors"][name]["offset"] // 4
        if init[0] == "normal":
            blob[o:o + n] = rng.standard_normal(n, dtype=np.float32) * np.float32(init[1])
        else:
            blob[o:o + n] = np.float32(init[1])
            if name.endswith(".g") or (name.endswith(".b") and "norm" in name):
                blob[o:o + n] += rng.standard_normal(n, dtype=np.float32) * np.float32(0.05)
    return man, blob


def save(path_prefix, man, blob):
    blob.tofile(path_prefix + ".bin")
    with open(path_prefix + ".json", "w") as f:
        json.dump(man, f)


def load(path_prefix):
    with open(path_prefix + ".json") as f:
        man = json.load(f)
    blob = np.fromfile(path_prefix + ".bin", dtype=np.float32)
    return man, blob


# ---- FSMN-VAD (UPSTREAM FunASR FSMN: 400-140-250-[128|250]x4-140-248) -----------------
FSMN_VAD = dict(model="fsmn_vad", n_mels=80, lfr_m=5, lfr_n=1, input_dim=400, affine=140, linear=250, proj=128,
                lorder=20, layers=4, out_affine=140, n_out=248)


def vad_tensor_specs(cfg):
    specs = [("cmvn.mean", [cfg["input_dim"]], ("const", -8.0)), ("cmvn.istd", [cfg["input_dim"]], ("const", 0.3))]

    def lin(name, out_f, in_f, bias=True):
        specs.append((name + ".w", [out_f, in_f], ("normal", 1.0 / math.sqrt(in_f))))
        if bias:
            specs.append((name + ".b", [out_f], ("normal", 0.05)))

    lin("in1", cfg["affine"], cfg["input_dim"])
    lin("in2", cfg["linear"], cfg["affine"])
    for i in range(cfg["layers"]):
        lin(f"blk.{i}.linear", cfg["proj"], cfg["linear"], bias=False)
        specs.append((f"blk.{i}.fsmn.w", [cfg["proj"], cfg["lorder"]], ("normal", 1.0 / math.sqrt(cfg["lorder"]))))
        lin(f"blk.{i}.affine", cfg["linear"], cfg["proj"])
    lin("out1", cfg["out_affine"], cfg["linear"])
    lin("out2", cfg["n_out"], cfg["out_affine"])
    return specs


def synth_generic(cfg, specs, seed):
    tensors, off = {}, 0
    for name, shape, _ in specs:
        tensors[name] = {"shape": shape, "offset": off}
        off += (int(np.prod(shape)) * 4 + ALIGN - 1) // ALIGN * ALIGN
    man = {"config": cfg, "tensors": tensors, "total_bytes": off}
    blob = np.zeros(off // 4, np.float32)
    rng = np.random.default_rng(seed)
    for name, shape, init in specs:
        n = int(np.prod(shape))
        o = tensors[name]["offset"] // 4
        if init[0] == "normal":
            blob[o:o + n] = rng.standard_normal(n, dtype=np.float32) * np.float32(init[1])
        else:
            blob[o:o + n] = np.float32(init[1])
    return man, blob


def synth_vad_weights(cfg=None, seed=4321):
    cfg = dict(FSMN_VAD) if cfg is None else cfg
    return synth_generic(cfg, vad_tensor_specs(cfg), seed)


def energy_vad_weights(man, blob):
    """Random VAD weights do not tell speech from silence.  For flows that need real segments (bench.py's long-audio block, the
    pipeline tests) shape them so that class 0 follows the frame energy: every weight zeroed except one path — mean log-mel of the
    centre frame -> unit 0 of every layer -> a negative slope on the silence logit.  Returns (manifest, blob), edited in place."""
    def view(name):
        meta = man["tensors"][name]
        n = int(np.prod(meta["shape"]))
        o = meta["offset"] // 4
        return blob[o:o + n].reshape(meta["shape"])
    for k in man["tensors"]:
        if (k.endswith(".w") or k.endswith(".b")) and not k.startswith("cmvn"):
            view(k)[...] = 0
    view("in1.w")[0, 160:240] = 1.0 / 80
    view("in2.w")[0, 0] = 1.0
    for i in range(man["config"]["layers"]):
        view(f"blk.{i}.linear.w")[0, 0] = 1.0
        view(f"blk.{i}.affine.w")[0, 0] = 1.0
    view("out1.w")[0, 0] = 1.0
    view("out2.w")[0, 0] = -4.0
    view("out2.b")[0] = 8.0          # energy ~ (x - 8) * 0.3 after CMVN: silence (log eps) strongly negative -> relu 0
    return man, blob


# ---- CT-Transformer punctuation (UPSTREAM: vocab 272727, 256-d SAN-M x4, 8 heads, FFN 1024) ----
CT_TRANSFORMER = dict(model="ct_transformer", vocab=272727, d_model=256, n_head=8, ffn=1024, layers=4, kernel=11, n_punc=6)


def punc_tensor_specs(cfg):
    d, f, k = cfg["d_model"], cfg["ffn"], cfg["kernel"]
    specs = [("embed.w", [cfg["vocab"], d], ("normal", 1.0))]

    def lin(name, out_f, in_f):
        specs.append((name + ".w", [out_f, in_f], ("normal", 1.0 / math.sqrt(in_f))))
        specs.append((name + ".b", [out_f], ("normal", 0.02)))

    def ln(name, n):
        specs.append((name + ".g", [n], ("const", 1.0)))
        specs.append((name + ".b", [n], ("const", 0.0)))

    for i in range(cfg["layers"]):
        p = f"enc.{i}."
        ln(p + "norm1", d)
        lin(p + "qkv", 3 * d, d)
        specs.append((p + "fsmn.w", [d, k], ("normal", 1.0 / math.sqrt(k))))
        lin(p + "out", d, d)
        ln(p + "norm2", d)
        lin(p + "ffn1", f, d)
        lin(p + "ffn2", d, f)
    ln("enc.after_norm", d)
    lin("out", cfg["n_punc"], d)
    return specs


def synth_punc_weights(cfg=None, seed=2468):
    cfg = dict(CT_TRANSFORMER) if cfg is None else cfg
    return synth_generic(cfg, punc_tensor_specs(cfg), seed)
