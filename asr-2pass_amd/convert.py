"""Real-weight loader: FunASR checkpoints -> the pfhip weight container.

The reference server loads `model.onnx` + `am.mvn` + `config.yaml` from a ModelScope directory
(onnxruntime/src/paraformer.cpp:21-53, websocket/bin/funasr-wss-server.cpp:203-320).  The same directories ship the
PyTorch checkpoint the ONNX file was exported from (`model.pt` / `model.pb`).  This module maps either to the container of
`weights.py`:

  * `state_from_onnx_dir(dir)`: the reference's OWN file contract (onnxruntime/include/com-define.h:52-88) — `model.onnx`
    (or `model_quant.onnx`, dequantised) [+ `decoder.onnx` for the online model, + `model_eb.onnx` for the hotword embedder],
    read by the dependency-free protobuf reader `onnx_reader.py`; anonymous MatMul weights are named after their layer and
    transposed back to torch layout, LSTM gate blocks re-ordered (ONNX i,o,f,c -> torch i,f,g,o);
  * `load_state_dict(model.pt)`: the checkpoint, `torch.load(..., weights_only=True)` only.

Tensor names follow UPSTREAM FunASR (funasr/models/{sanm,paraformer,contextual_paraformer,bicif_paraformer,
fsmn_vad_streaming,ct_transformer}), written down from memory: none of those files is available offline, so the mapping is
checked for self-consistency only (tests/test_convert.py round-trips synthetic weights through the upstream naming, and
through a synthetic ONNX file for the directory path; the ONNX READER itself is tested on the real .onnx files the reference
ships under utils/DNSMOS).  First contact with real Paraformer files is the milestone that turns "parity unpinned" into
token-for-token evidence.
"""
from __future__ import annotations

import numpy as np

from . import weights as Wt


def parse_am_mvn(text, scale=1.0):
    """kaldi-nnet `am.mvn` -> (shift, rescale), the two vectors `LoadCmvn` keeps (paraformer.cpp:325-360): the row after
    `<AddShift>` / `<Rescale>` starts `<LearnRateCoef> 0 [` and ends `]`; values in between."""
    shift, rescale = None, None
    lines = text.splitlines()
    for i, line in enumerate(lines[:-1]):
        head = line.split()
        nxt = lines[i + 1].split()
        if not head or not nxt or nxt[0] != "<LearnRateCoef>":
            continue
        vals = [float(v) for v in nxt[3:len(nxt) - 1]]
        if head[0] == "<AddShift>":
            shift = np.asarray(vals, np.float32)
        elif head[0] == "<Rescale>":
            rescale = np.asarray(vals, np.float32) * np.float32(scale)
    if shift is None or rescale is None:
        raise ValueError("am.mvn: <AddShift>/<Rescale> rows not found")
    return shift, rescale


def config_from_yaml(y):
    """config.yaml (dict, yaml.safe_load) -> container config; defaults = Paraformer-large."""
    cfg = dict(Wt.PARAFORMER_LARGE)
    enc, dec, pred, fe = (y.get(k, {}) or {} for k in ("encoder_conf", "decoder_conf", "predictor_conf", "frontend_conf"))
    cfg.update(d_model=int(enc.get("output_size", cfg["d_model"])), n_head=int(enc.get("attention_heads", cfg["n_head"])),
               ffn=int(enc.get("linear_units", cfg["ffn"])), enc_layers=int(enc.get("num_blocks", cfg["enc_layers"])),
               kernel=int(enc.get("kernel_size", cfg["kernel"])),
               dec_layers=int(dec.get("att_layer_num", dec.get("num_blocks", cfg["dec_layers"]))),
               dec_ffn=int(dec.get("linear_units", cfg["dec_ffn"])),
               cif_threshold=float(pred.get("threshold", cfg["cif_threshold"])),
               tail_threshold=float(pred.get("tail_threshold", cfg["tail_threshold"])),
               smooth_factor=float(pred.get("smooth_factor", cfg["smooth_factor"])),
               noise_threshold=float(pred.get("noise_threshold", cfg["noise_threshold"])),
               n_mels=int(fe.get("n_mels", cfg["n_mels"])), lfr_m=int(fe.get("lfr_m", cfg["lfr_m"])),
               lfr_n=int(fe.get("lfr_n", cfg["lfr_n"])))
    return cfg


def paraformer_name_map(cfg):
    """container tensor name -> UPSTREAM state_dict key (weight containers use .w/.b/.g, torch uses .weight/.bias)."""
    m = {}

    def lin(dst, src, bias=True):
        m[dst + ".w"] = src + ".weight"
        if bias:
            m[dst + ".b"] = src + ".bias"

    def ln(dst, src):
        m[dst + ".g"] = src + ".weight"
        m[dst + ".b"] = src + ".bias"

    for i in range(cfg["enc_layers"]):
        src = "encoder.encoders0.0" if i == 0 else f"encoder.encoders.{i - 1}"
        p = f"enc.{i}."
        ln(p + "norm1", src + ".norm1")
        lin(p + "qkv", src + ".self_attn.linear_q_k_v")
        m[p + "fsmn.w"] = src + ".self_attn.fsmn_block.weight"          # [d, 1, k] depthwise Conv1d
        lin(p + "out", src + ".self_attn.linear_out")
        ln(p + "norm2", src + ".norm2")
        lin(p + "ffn1", src + ".feed_forward.w_1")
        lin(p + "ffn2", src + ".feed_forward.w_2")
    ln("enc.after_norm", "encoder.after_norm")
    m["pred.conv.w"] = "predictor.cif_conv1d.weight"
    m["pred.conv.b"] = "predictor.cif_conv1d.bias"
    lin("pred.out", "predictor.cif_output")
    for i in range(cfg["dec_layers"]):
        src = f"decoder.decoders.{i}"
        p = f"dec.{i}."
        ln(p + "norm1", src + ".norm1")
        lin(p + "ffn1", src + ".feed_forward.w_1")
        ln(p + "ffn_norm", src + ".feed_forward.norm")
        lin(p + "ffn2", src + ".feed_forward.w_2", bias=False)
        ln(p + "norm2", src + ".norm2")
        m[p + "fsmn.w"] = src + ".self_attn.fsmn_block.weight"
        ln(p + "norm3", src + ".norm3")
        lin(p + "q", src + ".src_attn.linear_q")
        lin(p + "kv", src + ".src_attn.linear_k_v")
        lin(p + "out", src + ".src_attn.linear_out")
    ln("dec3.norm1", "decoder.decoders3.0.norm1")
    lin("dec3.ffn1", "decoder.decoders3.0.feed_forward.w_1")
    ln("dec3.ffn_norm", "decoder.decoders3.0.feed_forward.norm")
    lin("dec3.ffn2", "decoder.decoders3.0.feed_forward.w_2", bias=False)
    ln("dec.after_norm", "decoder.after_norm")
    lin("dec.out", "decoder.output_layer")
    if cfg.get("timestamp", 0):           # CifPredictorV3 (bicif_paraformer): ConvTranspose1d -> BLSTM -> Linear(2d, 1)
        m["pred.up.w"], m["pred.up.b"] = "predictor.upsample_cnn.weight", "predictor.upsample_cnn.bias"
        for sfx, tsfx in (("", ""), ("_r", "_reverse")):
            m[f"pred.blstm.w_ih{sfx}"] = f"predictor.blstm.weight_ih_l0{tsfx}"
            m[f"pred.blstm.w_hh{sfx}"] = f"predictor.blstm.weight_hh_l0{tsfx}"
            m[f"pred.blstm.b_ih{sfx}"] = f"predictor.blstm.bias_ih_l0{tsfx}"
            m[f"pred.blstm.b_hh{sfx}"] = f"predictor.blstm.bias_hh_l0{tsfx}"
        lin("pred.out2", "predictor.cif_output2")
    if cfg.get("contextual", 0):          # ContextualParaformer: the last attention layer is `last_decoder`, + bias decoder
        last = cfg["dec_layers"] - 1
        for k in list(m):
            if k.startswith(f"dec.{last}."):
                m[k] = m[k].replace(f"decoder.decoders.{last}.", "decoder.last_decoder.")
        m["bias.embed.w"] = "bias_embed.weight"
        for a, b in (("w_ih", "weight_ih_l0"), ("w_hh", "weight_hh_l0"), ("b_ih", "bias_ih_l0"), ("b_hh", "bias_hh_l0")):
            m["bias.lstm." + a] = "bias_encoder." + b
        ln("bias.dec.norm3", "decoder.bias_decoder.norm3")
        lin("bias.dec.q", "decoder.bias_decoder.src_attn.linear_q")
        lin("bias.dec.kv", "decoder.bias_decoder.src_attn.linear_k_v")
        lin("bias.dec.out", "decoder.bias_decoder.src_attn.linear_out")
        m["bias.out.w"] = "decoder.bias_output.weight"            # Conv1d(2d, d, 1, bias=False): [d, 2d, 1]
    return m


def vad_name_map(cfg):
    m = {}
    for dst, src in (("in1", "encoder.in_linear1.linear"), ("in2", "encoder.in_linear2.linear"),
                     ("out1", "encoder.out_linear1.linear"), ("out2", "encoder.out_linear2.linear")):
        m[dst + ".w"], m[dst + ".b"] = src + ".weight", src + ".bias"
    for i in range(cfg["layers"]):
        m[f"blk.{i}.linear.w"] = f"encoder.fsmn.{i}.linear.linear.weight"
        m[f"blk.{i}.fsmn.w"] = f"encoder.fsmn.{i}.fsmn_block.conv_left.weight"      # [proj, 1, lorder, 1]
        m[f"blk.{i}.affine.w"] = f"encoder.fsmn.{i}.affine.linear.weight"
        m[f"blk.{i}.affine.b"] = f"encoder.fsmn.{i}.affine.linear.bias"
    return m


def punc_name_map(cfg):
    m = {"embed.w": "embed.weight", "out.w": "decoder.weight", "out.b": "decoder.bias",
         "enc.after_norm.g": "encoder.after_norm.weight", "enc.after_norm.b": "encoder.after_norm.bias"}
    for i in range(cfg["layers"]):
        src = "encoder.encoders0.0" if i == 0 else f"encoder.encoders.{i - 1}"
        p = f"enc.{i}."
        for dst, s2 in (("norm1", ".norm1"), ("norm2", ".norm2")):
            m[p + dst + ".g"], m[p + dst + ".b"] = src + s2 + ".weight", src + s2 + ".bias"
        for dst, s2 in (("qkv", ".self_attn.linear_q_k_v"), ("out", ".self_attn.linear_out"), ("ffn1", ".feed_forward.w_1"),
                        ("ffn2", ".feed_forward.w_2")):
            m[p + dst + ".w"], m[p + dst + ".b"] = src + s2 + ".weight", src + s2 + ".bias"
        m[p + "fsmn.w"] = src + ".self_attn.fsmn_block.weight"
    return m


def _without_model_components(key):
    """FunASR's export wrappers keep the original module under `.model` (encoder.model.encoders0.0...): the key with every
    `model` path component dropped (the C++ reader does the same, csrc/model_files.cpp)."""
    return ".".join(p for p in key.split(".") if p != "model")


def _fill(specs, name_map, state, extra):
    """specs: [(name, shape, init)]; returns (manifest tensors, blob); a tensor is taken from `extra`, else from the
    state_dict through name_map, reshaped when only singleton dims differ ([d,1,k] -> [d,k])."""
    tensors, off = {}, 0
    for name, shape, _ in specs:
        tensors[name] = {"shape": list(shape), "offset": off}
        off += (int(np.prod(shape)) * 4 + Wt.ALIGN - 1) // Wt.ALIGN * Wt.ALIGN
    blob = np.zeros(off // 4, np.float32)
    missing = []
    state = dict(state)
    for k in list(state):
        state.setdefault(_without_model_components(k), state[k])
    used = set()
    for name, shape, _ in specs:
        if name in extra:
            arr = np.asarray(extra[name], np.float32)
        else:
            key = name_map.get(name)
            if key is None or key not in state:
                missing.append(f"{name} <- {key}")
                continue
            t = state[key]
            used.add(key)
            arr = t.detach().cpu().float().numpy() if hasattr(t, "detach") else np.asarray(t, np.float32)
        if list(arr.shape) != list(shape):
            if arr.size != int(np.prod(shape)) or [d for d in arr.shape if d != 1] != [d for d in shape if d != 1]:
                raise ValueError(f"{name}: checkpoint shape {list(arr.shape)} does not fit {list(shape)}")
            arr = arr.reshape(shape)
        o = tensors[name]["offset"] // 4
        blob[o:o + arr.size] = arr.reshape(-1)
    if missing:
        near = [k for k in state if k not in used and _without_model_components(k) not in used][:8]
        raise KeyError("checkpoint lacks: " + "; ".join(missing[:8]) + (" ..." if len(missing) > 8 else "") +
                       (f"  (unmatched keys in the checkpoint: {', '.join(near)})" if near else ""))
    return tensors, blob, off


def convert_paraformer(state, cfg, shift, rescale):
    cfg = dict(cfg)
    cfg["vocab"] = int(state[paraformer_name_map(cfg)["dec.out.w"]].shape[0])
    tensors, blob, total = _fill(Wt.tensor_specs(cfg), paraformer_name_map(cfg), state,
                                 {"cmvn.mean": shift, "cmvn.istd": rescale})
    return {"config": cfg, "tensors": tensors, "total_bytes": total}, blob


def convert_vad(state, shift, rescale, cfg=None):
    cfg = dict(Wt.FSMN_VAD) if cfg is None else dict(cfg)
    tensors, blob, total = _fill(Wt.vad_tensor_specs(cfg), vad_name_map(cfg), state, {"cmvn.mean": shift, "cmvn.istd": rescale})
    return {"config": cfg, "tensors": tensors, "total_bytes": total}, blob


def convert_punc(state, cfg=None):
    cfg = dict(Wt.CT_TRANSFORMER) if cfg is None else dict(cfg)
    nm = punc_name_map(cfg)
    cfg["vocab"] = int(state[nm["embed.w"]].shape[0])
    cfg["n_punc"] = int(state[nm["out.w"]].shape[0])
    tensors, blob, total = _fill(Wt.punc_tensor_specs(cfg), nm, state, {})
    return {"config": cfg, "tensors": tensors, "total_bytes": total}, blob


def state_from_onnx_dir(src, quantized=False):
    """{torch-style key: ndarray} from a model directory laid out as the reference reads it (com-define.h:52-88;
    paraformer.cpp:39-46 offline, :92-121 online encoder + decoder, :178-186 hotword embedder): model.onnx or model_quant.onnx,
    plus decoder.onnx / decoder_quant.onnx and model_eb.onnx / model_eb_quant.onnx when present.  Later files win on a clash."""
    import os
    from . import onnx_reader as R
    sfx = "_quant.onnx" if quantized else ".onnx"
    main = os.path.join(src, "model" + sfx)
    if not os.path.exists(main):
        raise FileNotFoundError(main)
    state, files = {}, []
    for stem in ("model", "decoder", "model_eb"):
        p = os.path.join(src, stem + sfx)
        if not os.path.exists(p):
            continue
        m = R.read_model(p)
        if m.external:
            raise ValueError(f"{p}: {len(m.external)} initializers live in external data files (not supported)")
        bad = R.check_closed(m)
        if bad:
            raise ValueError(f"{p}: graph is not closed over its initializers, e.g. {bad[:3]}")
        state.update(R.torch_style_state(m))
        files.append(p)
    return state, files


def detect_heads(state):
    """contextual / timestamp flags from the tensors that are there (config.yaml names the model class, the weights decide)."""
    return dict(contextual=int("decoder.bias_output.weight" in state or "bias_embed.weight" in state),
                timestamp=int("predictor.upsample_cnn.weight" in state))


def convert_model_dir(kind, src, prefer="auto", quantized=False):
    """kind in {asr, vad, punc}; src = a model directory as the reference's server receives it (--model-dir / --vad-dir /
    --punc-dir: model.onnx [+ decoder.onnx, model_eb.onnx], am.mvn / vad.mvn, config.yaml / vad.yaml / punc.yaml, tokens.json).
    prefer: "onnx", "pt" or "auto" (ONNX when present).  Returns (manifest, blob, source files)."""
    import os
    import yaml
    has_onnx = os.path.exists(os.path.join(src, "model_quant.onnx" if quantized else "model.onnx"))
    if prefer == "onnx" or (prefer == "auto" and has_onnx):
        state, files = state_from_onnx_dir(src, quantized)
    else:
        pt = next((os.path.join(src, n) for n in ("model.pt", "model.pb") if os.path.exists(os.path.join(src, n))), None)
        if pt is None:
            raise FileNotFoundError(f"{src}: neither model.onnx nor model.pt")
        state, files = load_state_dict(pt), [pt]
    if kind == "punc":
        man, blob = convert_punc(state)
        return man, blob, files
    mvn = next((os.path.join(src, n) for n in (("vad.mvn", "am.mvn") if kind == "vad" else ("am.mvn",)) if os.path.exists(os.path.join(src, n))), None)
    if mvn is None:
        raise FileNotFoundError(f"{src}: am.mvn")
    shift, rescale = parse_am_mvn(open(mvn).read())
    if kind == "vad":
        man, blob = convert_vad(state, shift, rescale)
        return man, blob, files + [mvn]
    ypath = os.path.join(src, "config.yaml")
    with open(ypath) as f:
        cfg = config_from_yaml(yaml.safe_load(f))
    cfg.update(detect_heads(state))
    man, blob = convert_paraformer(state, cfg, shift, rescale)
    return man, blob, files + [mvn, ypath]


def load_state_dict(path):
    """model.pt / model.pb of a FunASR model directory; tensors only (weights_only=True)."""
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)
    for k in ("state_dict", "model", "model_state_dict"):
        if isinstance(obj, dict) and k in obj and isinstance(obj[k], dict):
            obj = obj[k]
    return obj
