"""Real-weight loader: FunASR checkpoints -> the pfhip weight container.

The reference server loads `model.onnx` + `am.mvn` + `config.yaml` from a ModelScope directory
(onnxruntime/src/paraformer.cpp:21-53, websocket/bin/funasr-wss-server.cpp:203-320).  The same directories ship the
PyTorch checkpoint the ONNX file was exported from (`model.pt` / `model.pb`); this module maps ITS state_dict to the
container of `weights.py`.  Tensor names follow UPSTREAM FunASR (funasr/models/{sanm,paraformer,fsmn_vad_streaming,
ct_transformer}); none of those files is available offline, so the mapping is checked only for self-consistency
(tests/test_convert.py round-trips synthetic weights through the upstream naming) — first contact with real files
is the f2 milestone that turns "parity unpinned" into token-for-token evidence.

Checkpoints are read with `torch.load(..., weights_only=True)` only.
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


def _fill(specs, name_map, state, extra):
    """specs: [(name, shape, init)]; returns (manifest tensors, blob); a tensor is taken from `extra`, else from the
    state_dict through name_map, reshaped when only singleton dims differ ([d,1,k] -> [d,k])."""
    tensors, off = {}, 0
    for name, shape, _ in specs:
        tensors[name] = {"shape": list(shape), "offset": off}
        off += (int(np.prod(shape)) * 4 + Wt.ALIGN - 1) // Wt.ALIGN * Wt.ALIGN
    blob = np.zeros(off // 4, np.float32)
    missing = []
    for name, shape, _ in specs:
        if name in extra:
            arr = np.asarray(extra[name], np.float32)
        else:
            key = name_map.get(name)
            if key is None or key not in state:
                missing.append(f"{name} <- {key}")
                continue
            t = state[key]
            arr = t.detach().cpu().float().numpy() if hasattr(t, "detach") else np.asarray(t, np.float32)
        if list(arr.shape) != list(shape):
            if arr.size != int(np.prod(shape)) or [d for d in arr.shape if d != 1] != [d for d in shape if d != 1]:
                raise ValueError(f"{name}: checkpoint shape {list(arr.shape)} does not fit {list(shape)}")
            arr = arr.reshape(shape)
        o = tensors[name]["offset"] // 4
        blob[o:o + arr.size] = arr.reshape(-1)
    if missing:
        raise KeyError("checkpoint lacks: " + "; ".join(missing[:8]) + (" ..." if len(missing) > 8 else ""))
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


def load_state_dict(path):
    """model.pt / model.pb of a FunASR model directory; tensors only (weights_only=True)."""
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)
    for k in ("state_dict", "model", "model_state_dict"):
        if isinstance(obj, dict) and k in obj and isinstance(obj[k], dict):
            obj = obj[k]
    return obj
