"""CPU: the dependency-free ONNX weight reader (asr-2pass_amd/onnx_reader.py).

(1) On REAL bytes: the five .onnx files the reference ships (utils/DNSMOS, utils/pDNSMOS — keras2onnx exports, unrelated to the
    ASR path but genuine ONNX files with raw_data AND float_data tensors, int64 shape constants and Conv/MatMul/Add graphs):
    tensor counts, shapes, byte totals, a graph closed over its initializers.  No code path of the files is executed.
(2) Wire-format edge cases against a writer kept in tests/onnx_writer.py.
(3) The converter's directory path — model.onnx [+ model_eb.onnx] + am.mvn + config.yaml + tokens.json as
    onnxruntime/include/com-define.h:52-88 lays a model directory out — round-tripped through a synthetic ONNX file written the
    way the PyTorch exporter leaves one (anonymous transposed MatMul weights, ONNX LSTM gate order).
Real Paraformer files are not available offline: the UPSTREAM layer names stay unverified ("parity unpinned")."""
import importlib
import json
import os

import numpy as np
import pytest

import onnx_writer as OW

REF = "/root/reference/utils"
REAL = {   # file -> (initializers, initializer bytes, nodes, a dense weight and its [in, out] shape)
    "DNSMOS/bak_ovr.onnx": (20, 736908, 29, "mos_estimator_v1/dense_2/MatMul/ReadVariableOp/resource:0", (64, 3)),
    "DNSMOS/model_v8.onnx": (16, 219780, 25, "mos_estimator_small_1/dense_5/MatMul/ReadVariableOp/resource:0", (64, 1)),
    "DNSMOS/sig.onnx": (20, 736388, 29, "mos_estimator_sig_v1/dense_4/MatMul/ReadVariableOp/resource:0", (128, 64)),
    "DNSMOS/sig_bak_ovr.onnx": (35, 1149256, 48, "mos_estimator_logpow/dense_1/MatMul/ReadVariableOp/resource:0", (128, 64)),
    "pDNSMOS/sig_bak_ovr.onnx": (35, 1149256, 48, "mos_estimator_logpow/dense_3/MatMul/ReadVariableOp/resource:0", (64, 3)),
}


@pytest.fixture(scope="module")
def mods(pkg):
    return (importlib.import_module(pkg.__name__ + ".onnx_reader"), importlib.import_module(pkg.__name__ + ".convert"),
            importlib.import_module(pkg.__name__ + ".weights"))


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree (real .onnx files) is not on this machine")
@pytest.mark.parametrize("rel", sorted(REAL))
def test_reads_the_reference_s_real_onnx_files(mods, rel):
    R = mods[0]
    path = os.path.join(REF, rel)
    n_init, n_bytes, n_nodes, dense, dense_shape = REAL[rel]
    m = R.read_model(path)
    assert m.ir_version == 7 and m.producer == "keras2onnx" and m.opsets == {"": 12}
    assert len(m.initializers) == n_init and len(m.nodes) == n_nodes and not m.external
    assert sum(a.nbytes for a in m.initializers.values()) == n_bytes
    assert 0.97 * os.path.getsize(path) < n_bytes < os.path.getsize(path)          # the file IS its weights
    assert R.check_closed(m) == []                        # every node input resolves: field numbers and strings were read right
    assert m.initializers[dense].shape == dense_shape and m.initializers[dense].dtype == np.float32
    assert [i[0] for i in m.inputs] == ["input_1"] and [o[0] for o in m.outputs] == ["Identity:0"] and m.inputs[0][2][0] == "N"
    for k, a in m.initializers.items():
        assert a.size == int(np.prod(a.shape)) and (a.dtype.kind != "f" or np.isfinite(a).all()), k
    convs = [nd for nd in m.nodes if nd.op_type == "Conv"]
    for nd in convs:                                      # attributes: int lists; weights: [out, in, k...] matching them
        w = m.initializers[nd.inputs[1]]
        ks = list(nd.attrs["kernel_shape"])
        assert w.ndim == 2 + len(ks) and list(w.shape[2:]) == ks and nd.attrs["group"] == 1, (nd.name, w.shape, nd.attrs)
    # the graph walk names each dense weight after the bias that follows its MatMul? keras names carry no ".bias": the weight
    # keeps its own name, transposed to [out, in]
    st = R.torch_style_state(m)
    assert st[dense + ".weight"].shape == dense_shape[::-1]
    assert np.array_equal(st[dense + ".weight"], m.initializers[dense].T)
    s = R.summary(m)
    assert s["ops"]["MatMul"] == 3 and s["initializer_bytes"] == n_bytes


def test_wire_format_edge_cases(mods):
    R = mods[0]
    rng = np.random.default_rng(0)
    a = rng.standard_normal((3, 5)).astype(np.float32)
    big = np.asarray([-1, 2 ** 40, -2 ** 62, 0, 300], np.int64)
    inits = [OW.tensor("raw", a, "raw"), OW.tensor("typed", a, "typed"), OW.tensor("unpacked_dims", a, "unpacked"),
             OW.tensor("i64", big, "typed"), OW.tensor("i64raw", big, "raw"), OW.tensor("scalar", np.float32(2.5).reshape(()), "raw"),
             OW.tensor("empty", np.zeros((0, 4), np.float32), "raw"), OW.tensor("h", a.astype(np.float16), "raw"),
             OW.tensor("u8", np.arange(7, dtype=np.uint8), "raw")]
    blob = OW.model([OW.node("Identity", ["x"], ["y"], "/id")], inits, [OW.value_info("x", dims=("N", 3, 5))], [OW.value_info("y")])
    m = R.read_model(blob)
    for k in ("raw", "typed", "unpacked_dims"):
        assert np.array_equal(m.initializers[k], a), k
    assert m.initializers["i64"].tolist() == big.tolist() == m.initializers["i64raw"].tolist()
    assert m.initializers["scalar"].shape == () and float(m.initializers["scalar"]) == 2.5
    assert m.initializers["empty"].shape == (0, 4) and m.initializers["h"].dtype == np.float16 and m.initializers["u8"].tolist() == list(range(7))
    assert m.inputs == [("x", 1, ["N", 3, 5])] and R.check_closed(m) == []
    with pytest.raises(R.OnnxFormatError):
        R.read_model(blob[:len(blob) // 2])                               # truncated file
    with pytest.raises(R.OnnxFormatError):
        R.read_model(b"\x00\x01\x02")                                     # not protobuf
    bad = OW.ld(1, OW.varint(4) + OW.varint(4)) + OW.key(2, 0) + OW.varint(1) + OW.ld(9, b"\0" * 60) + OW.s(8, "short")      # 16 floats promised, 15 stored
    with pytest.raises(R.OnnxFormatError):
        R.read_model(OW.model([], [bad], [], []))
    m2 = R.read_model(OW.model([OW.node("Relu", ["nowhere"], ["y"], "/r")], [], [], [OW.value_info("y")]))
    assert R.check_closed(m2) == ["Relu:/r:nowhere"]


def upstream_state(conv, man, blob, cfg):
    """container -> {upstream key: array in torch layout}"""
    state = {}
    for name, key in conv.paraformer_name_map(cfg).items():
        meta = man["tensors"][name]
        arr = blob[meta["offset"] // 4: meta["offset"] // 4 + int(np.prod(meta["shape"]))].reshape(meta["shape"]).copy()
        if name.endswith("fsmn.w"):
            arr = arr[:, None, :]                  # depthwise Conv1d [d, 1, k]
        if name == "bias.out.w":
            arr = arr[:, :, None]                  # Conv1d(2d, d, 1)
        state[key] = arr
    return state


NOT_LINEAR = ("fsmn_block", "cif_conv1d", "upsample_cnn", "bias_embed", "bias_output", "blstm", "bias_encoder", "norm")


def write_model_dir(tmp, conv, man, blob, cfg, mvn_name="am.mvn"):
    state = upstream_state(conv, man, blob, cfg)
    g, eb = OW.GraphBuilder(), OW.GraphBuilder()
    modules = []
    for k in state:
        mod = k.rsplit(".", 1)[0]
        if mod not in modules and "lstm" not in k and "weight_" not in k and "bias_ih" not in k and "bias_hh" not in k:
            modules.append(mod)
    for i, mod in enumerate(modules):
        w, b = state.get(mod + ".weight"), state.get(mod + ".bias")
        tgt = eb if mod.startswith("bias_embed") else g
        if any(t in mod for t in NOT_LINEAR) or w is None or w.ndim != 2:
            arrays = [(sfx, a) for sfx, a in (("weight", w), ("bias", b)) if a is not None]
            tgt.named("LayerNormalization" if "norm" in mod else "Gather" if "embed" in mod else "Conv", mod, arrays,
                      form="typed" if i % 3 == 0 else "raw")
        else:
            tgt.linear(mod, w, b, form="typed" if i % 4 == 1 else "raw")
    lstm = lambda p, sfx="": tuple(state[f"{p}.{n}_l0{sfx}"] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
    if cfg.get("timestamp"):
        g.lstm("predictor.blstm", *lstm("predictor.blstm"), reverse=lstm("predictor.blstm", "_reverse"))
    if cfg.get("contextual"):
        eb.lstm("bias_encoder", *lstm("bias_encoder"))
        with open(tmp / "model_eb.onnx", "wb") as f:
            f.write(eb.build())
    with open(tmp / "model.onnx", "wb") as f:
        f.write(g.build())
    t = man["tensors"]
    get = lambda n: blob[t[n]["offset"] // 4: t[n]["offset"] // 4 + int(np.prod(t[n]["shape"]))]
    row = lambda v: " ".join(repr(float(x)) for x in v)
    with open(tmp / mvn_name, "w") as f:
        f.write(f"<Nnet>\n<Splice> 560 560\n[ 0 ]\n<AddShift> 560 560\n<LearnRateCoef> 0 [ {row(get('cmvn.mean'))} ]\n"
                f"<Rescale> 560 560\n<LearnRateCoef> 0 [ {row(get('cmvn.istd'))} ]\n</Nnet>\n")
    with open(tmp / "config.yaml", "w") as f:
        f.write(f"encoder_conf:\n  output_size: {cfg['d_model']}\n  attention_heads: {cfg['n_head']}\n  linear_units: {cfg['ffn']}\n"
                f"  num_blocks: {cfg['enc_layers']}\n  kernel_size: {cfg['kernel']}\ndecoder_conf:\n  att_layer_num: {cfg['dec_layers']}\n"
                f"  num_blocks: {cfg['dec_layers']}\n  linear_units: {cfg['dec_ffn']}\npredictor_conf:\n  threshold: 1.0\n  tail_threshold: 0.45\n"
                "frontend_conf:\n  n_mels: 80\n  lfr_m: 7\n  lfr_n: 6\n")
    with open(tmp / "tokens.json", "w") as f:
        json.dump([f"<{i}>" for i in range(cfg["vocab"])], f)


@pytest.mark.parametrize("heads", [dict(), dict(contextual=1, timestamp=1)])
def test_model_directory_with_onnx_files_round_trips(mods, tmp_path, heads):
    R, conv, wt = mods
    cfg = wt.small_config(enc_layers=2, dec_layers=2, vocab=97, **heads)
    man, blob = wt.synth_weights(cfg, seed=11)
    write_model_dir(tmp_path, conv, man, blob, cfg)
    m = R.read_model(str(tmp_path / "model.onnx"))
    assert R.check_closed(m) == [] and any(k.startswith("onnx::MatMul_") for k in m.initializers)
    man2, blob2, files = conv.convert_model_dir("asr", str(tmp_path))
    assert [os.path.basename(f) for f in files][0] == "model.onnx" and ("model_eb.onnx" in [os.path.basename(f) for f in files]) == bool(heads)
    assert man2["tensors"] == man["tensors"]
    for k in ("vocab", "contextual", "timestamp", "enc_layers", "dec_layers", "d_model"):
        assert man2["config"].get(k, 0) == cfg.get(k, 0), k
    assert np.array_equal(blob2, blob)                    # transposes, LSTM gate order, conv singleton dims: all undone
    # the CLI writes the container (+ tokens.json beside it) that FunOfflineInit opens
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    cli = importlib.import_module("convert_funasr")
    out = tmp_path / "out"
    out.mkdir()
    assert cli.main(["asr", str(tmp_path), str(out / "model.pfhip")]) == 0
    man3, blob3 = wt.load(str(out / "model.pfhip"))
    assert man3["tensors"] == man["tensors"] and np.array_equal(blob3, blob) and os.path.exists(out / "tokens.json")
    # a layer that lost its bias name: resolved from the MatMul node's own name; a missing tensor is an error, not a zero
    os.remove(tmp_path / "model.onnx")
    with pytest.raises(FileNotFoundError):
        conv.convert_model_dir("asr", str(tmp_path), prefer="onnx")


def test_quantized_initializers_are_folded_back(mods):
    R = mods[0]
    rng = np.random.default_rng(3)
    w = rng.standard_normal((8, 6)).astype(np.float32)                   # torch [out, in]
    scale = np.float32(np.abs(w).max() / 127)
    q = np.clip(np.round(w.T / scale), -127, 127).astype(np.int8)        # stored [in, out]
    inits = [OW.tensor("onnx::MatMul_7_quantized", q), OW.tensor("onnx::MatMul_7_scale", scale.reshape(())),
             OW.tensor("onnx::MatMul_7_zero_point", np.zeros((), np.int8)), OW.tensor("enc.l.bias", np.zeros(8, np.float32))]
    nodes = [OW.node("DynamicQuantizeLinear", ["x"], ["xq", "xs", "xz"], "/enc/l/MatMul_quant_dql"),
             OW.node("MatMulInteger", ["xq", "onnx::MatMul_7_quantized", "xz", "onnx::MatMul_7_zero_point"], ["acc"], "/enc/l/MatMul_quant"),
             OW.node("Cast", ["acc"], ["accf"], "/enc/l/cast"), OW.node("Mul", ["accf", "xs"], ["y0"], "/enc/l/mul"),
             OW.node("Add", ["y0", "enc.l.bias"], ["y"], "/enc/l/Add")]
    m = R.read_model(OW.model(nodes, inits, [OW.value_info("x")], [OW.value_info("y")]))
    assert R.check_closed(m) == []
    st = R.torch_style_state(m)
    assert set(st) == {"enc.l.weight", "enc.l.bias"}
    assert np.abs(st["enc.l.weight"] - w).max() <= scale / 2 + 1e-7
