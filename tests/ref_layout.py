"""Test helper: writes model directories laid out as the reference's server reads them (onnxruntime/include/com-define.h:52-88;
opened at offline-stream.cpp:17-19,60-89,111-117 and tpass-stream.cpp:17-19,52-77,104-109): `model.onnx` [+ `decoder.onnx`,
`model_eb.onnx`], `am.mvn`, `config.yaml`, `tokens.json` — from a synthetic weight container, in the form the PyTorch exporter
leaves (anonymous transposed MatMul initializers, ONNX LSTM gate order, named LayerNorm / Conv parameters).  Data only: no
reference file is copied."""
import json
import os

import numpy as np

import onnx_writer as OW

NOT_LINEAR = ("fsmn_block", "cif_conv1d", "upsample_cnn", "bias_embed", "bias_output", "blstm", "bias_encoder", "norm", "embed", "conv_left")


def view(man, blob, name):
    meta = man["tensors"][name]
    n = int(np.prod(meta["shape"]))
    return blob[meta["offset"] // 4: meta["offset"] // 4 + n].reshape(meta["shape"])


def upstream_state(name_map, man, blob, expand):
    """container -> {upstream state_dict key: array in torch layout}; `expand(container name, array)` restores singleton dims."""
    return {key: expand(name, view(man, blob, name).copy()) for name, key in name_map.items()}


def write_onnx(path, state, keys=None, wrapper_model_component=False, quantize=False):
    """One ONNX file holding the listed keys of `state`.  wrapper_model_component: name modules `encoder.model.encoders0...` as
    FunASR's export wrappers may; quantize: Linear weights as onnxruntime's quantize_dynamic leaves them."""
    g = OW.GraphBuilder()
    keys = list(state if keys is None else keys)
    modules = []
    for k in keys:
        mod = k.rsplit(".", 1)[0]
        if mod not in modules and "weight_" not in k and "bias_ih" not in k and "bias_hh" not in k:
            modules.append(mod)

    def wrap(mod):
        if not wrapper_model_component:
            return mod
        head, _, rest = mod.partition(".")
        return f"{head}.model.{rest}" if rest and head in ("encoder", "decoder") else mod

    for i, mod in enumerate(modules):
        w, b = state.get(mod + ".weight"), state.get(mod + ".bias")
        if any(t in mod for t in NOT_LINEAR) or w is None or w.ndim != 2:
            arrays = [(sfx, a) for sfx, a in (("weight", w), ("bias", b)) if a is not None]
            g.named("LayerNormalization" if "norm" in mod else "Gather" if "embed" in mod else "Conv", wrap(mod), arrays,
                    form="typed" if i % 3 == 0 else "raw")
        elif quantize:
            quantized_linear(g, wrap(mod), w, b)
        else:
            g.linear(wrap(mod), w, b, form="typed" if i % 4 == 1 else "raw")
    lstm = lambda p, sfx="": tuple(state[f"{p}.{n}_l0{sfx}"] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
    for p in sorted({k.rsplit(".", 1)[0] for k in keys if k.endswith("weight_ih_l0")}):
        g.lstm(p, *lstm(p), reverse=lstm(p, "_reverse") if f"{p}.weight_ih_l0_reverse" in state else None)
    with open(path, "wb") as f:
        f.write(g.build())


def quantized_linear(g, module, w, b):
    """MatMul weight as quantize_dynamic leaves it: <W>_quantized int8 [in, out] + per-tensor <W>_scale / <W>_zero_point consumed by
    MatMulInteger -> Cast -> Mul -> Add(bias)."""
    g.n += 1
    base = f"onnx::MatMul_{g.n + 5000}"
    scale = np.float32(max(np.abs(w).max(), 1e-8) / 127)
    q = np.clip(np.round(w.T / scale), -127, 127).astype(np.int8)
    g.inits += [OW.tensor(base + "_quantized", q), OW.tensor(base + "_scale", scale.reshape(())),
                OW.tensor(base + "_zero_point", np.zeros((), np.int8))]
    path = "/" + module.replace(".", "/")
    xq, xs, xz, acc, accf, y0 = (g.fresh() for _ in range(6))
    g.nodes += [OW.node("DynamicQuantizeLinear", [g.cur], [xq, xs, xz], path + "/MatMul_quant_dql"),
                OW.node("MatMulInteger", [xq, base + "_quantized", xz, base + "_zero_point"], [acc], path + "/MatMul_quant"),
                OW.node("Cast", [acc], [accf], path + "/cast"), OW.node("Mul", [accf, xs], [y0], path + "/mul")]
    g.cur = y0
    if b is not None:
        g.inits.append(OW.tensor(module + ".bias", b))
        o = g.fresh()
        g.nodes.append(OW.node("Add", [g.cur, module + ".bias"], [o], path + "/Add"))
        g.cur = o
    return np.float32(scale) * q.T.astype(np.float32)          # what the reader must give back ([out, in])


def write_mvn(path, mean, istd):
    row = lambda v: " ".join(repr(float(x)) for x in v)
    d = len(mean)
    with open(path, "w") as f:
        f.write(f"<Nnet>\n<Splice> {d} {d}\n[ 0 ]\n<AddShift> {d} {d}\n<LearnRateCoef> 0 [ {row(mean)} ]\n"
                f"<Rescale> {d} {d}\n<LearnRateCoef> 0 [ {row(istd)} ]\n</Nnet>\n")


def asr_config_yaml(cfg, lang=None):
    y = (f"# written by tests/ref_layout.py\nencoder: SANMEncoder\nencoder_conf:\n    output_size: {cfg['d_model']}    # d_model\n    attention_heads: {cfg['n_head']}\n"
         f"    linear_units: {cfg['ffn']}\n    num_blocks: {cfg['enc_layers']}\n    kernel_size: {cfg['kernel']}\n    sanm_shfit: 0\n"
         f"decoder_conf:\n    att_layer_num: {cfg['dec_layers']}\n    num_blocks: {cfg['dec_layers']}\n    linear_units: {cfg['dec_ffn']}\n    kernel_size: {cfg['kernel']}\n"
         f"predictor_conf:\n    threshold: {cfg.get('cif_threshold', 1.0)}\n    tail_threshold: {cfg.get('tail_threshold', 0.45)}\n    smooth_factor: 1.0\n    noise_threshold: 0\n"
         "frontend: WavFrontend\nfrontend_conf:\n    fs: 16000\n    window: hamming\n    n_mels: 80\n    frame_length: 25\n    frame_shift: 10\n    lfr_m: 7\n    lfr_n: 6\n"
         "specaug_conf: {apply_time_warp: false, freq_mask_width_range: [0, 30]}\ntokenizer_conf:\n  unk_symbol: '<unk>'\n")
    if lang:
        y += f"lang: {lang}\n"
    return y


def asr_expand(name, a):
    if name.endswith("fsmn.w"):
        return a[:, None, :]                  # depthwise Conv1d [d, 1, k]
    if name == "bias.out.w":
        return a[:, :, None]                  # Conv1d(2d, d, 1)
    return a


def write_asr_dir(dst, conv, man, blob, cfg, online=False, vocab_tokens=None, **onnx_kw):
    """Offline: model.onnx [+ model_eb.onnx].  online=True: model.onnx = encoder + predictor, decoder.onnx = decoder (ENCODER_NAME /
    DECODER_NAME, com-define.h:77-80)."""
    os.makedirs(dst, exist_ok=True)
    state = upstream_state(conv.paraformer_name_map(cfg), man, blob, asr_expand)
    eb = [k for k in state if k.startswith(("bias_embed", "bias_encoder"))]
    rest = [k for k in state if k not in eb]
    if online:
        write_onnx(os.path.join(dst, "model.onnx"), state, [k for k in rest if not k.startswith("decoder.")], **onnx_kw)
        write_onnx(os.path.join(dst, "decoder.onnx"), state, [k for k in rest if k.startswith("decoder.")], **onnx_kw)
    else:
        write_onnx(os.path.join(dst, "model.onnx"), state, rest, **onnx_kw)
    if eb:
        write_onnx(os.path.join(dst, "model_eb.onnx"), state, eb)
        with open(os.path.join(dst, "seg_dict"), "w") as f:
            f.write("hello hel@@ lo\n")
    write_mvn(os.path.join(dst, "am.mvn"), view(man, blob, "cmvn.mean"), view(man, blob, "cmvn.istd"))
    with open(os.path.join(dst, "config.yaml"), "w") as f:
        f.write(asr_config_yaml(cfg))
    with open(os.path.join(dst, "tokens.json"), "w") as f:
        json.dump(vocab_tokens if vocab_tokens is not None else [f"<{i}>" for i in range(cfg["vocab"])], f)
    return state


def write_vad_dir(dst, conv, man, blob):
    os.makedirs(dst, exist_ok=True)
    cfg = man["config"]
    state = upstream_state(conv.vad_name_map(cfg), man, blob, lambda n, a: a[:, None, :, None] if n.endswith("fsmn.w") else a)
    write_onnx(os.path.join(dst, "model.onnx"), state)
    write_mvn(os.path.join(dst, "am.mvn"), view(man, blob, "cmvn.mean"), view(man, blob, "cmvn.istd"))
    with open(os.path.join(dst, "config.yaml"), "w") as f:
        f.write("frontend_conf:\n  fs: 16000\n  window: hamming\n  n_mels: 80\n  frame_length: 25\n  frame_shift: 10\n  dither: 0.0\n  lfr_m: 5\n  lfr_n: 1\n"
                "model_conf:\n  max_end_silence_time: 800\n  max_single_segment_time: 60000\n  speech_noise_thres: 0.9\n"
                f"encoder_conf:\n  fsmn_layers: {cfg['layers']}\n")
    return state


def write_punc_dir(dst, conv, man, blob, tokens, punc_list=("<unk>", "_", "，", "。", "？", "、")):
    os.makedirs(dst, exist_ok=True)
    cfg = man["config"]
    state = upstream_state(conv.punc_name_map(cfg), man, blob, lambda n, a: a[:, None, :] if n.endswith("fsmn.w") else a)
    write_onnx(os.path.join(dst, "model.onnx"), state)
    with open(os.path.join(dst, "config.yaml"), "w", encoding="utf-8") as f:
        f.write(f"model: CTTransformer\nmodel_conf:\n    ignore_id: 0\n    punc_list:\n" + "".join(f"    - {json.dumps(p, ensure_ascii=False) if p in ('_',) else p}\n" for p in punc_list) +
                f"encoder_conf:\n    output_size: {cfg['d_model']}\n    attention_heads: {cfg['n_head']}\n    linear_units: {cfg['ffn']}\n    num_blocks: {cfg['layers']}\n"
                f"    kernel_size: {cfg['kernel']}\n    sanm_shfit: {cfg.get('sanm_shift', 0)}\n")
    with open(os.path.join(dst, "tokens.json"), "w", encoding="utf-8") as f:
        json.dump(list(tokens), f, ensure_ascii=False)
    return state
