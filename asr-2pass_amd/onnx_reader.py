"""Dependency-free reader of the weights inside an ONNX file.

The reference deploys `model.onnx` / `model_quant.onnx` (+ `model_eb.onnx`, `decoder.onnx`) next to `am.mvn`, `config.yaml` and
`tokens.json` (onnxruntime/include/com-define.h:52-88; opened at onnxruntime/src/paraformer.cpp:39-46, 178-241, 325-360) and
hands the file to onnxruntime.  This build has no onnxruntime and no `onnx` package: the file is protobuf, and the part needed
here — `ModelProto.graph.initializer[]` plus enough of `graph.node[]` to name the anonymous ones — is a few length-delimited
fields.  The reader below walks the wire format directly (varints, 64-/32-bit scalars, length-delimited records; field numbers
from the public onnx.proto3 schema, quoted next to each use).  Nothing in the file is executed: bytes in, numpy arrays out.

    model = read_model(path)            # OnnxModel(initializers: {name: ndarray}, nodes: [OnnxNode], inputs, outputs, ...)
    state = torch_style_state(model)    # {"encoder.encoders.3.feed_forward.w_1.weight": [out, in] ndarray, ...}

`torch_style_state` undoes what the PyTorch exporter does to `nn.Linear`: the weight becomes an anonymous, TRANSPOSED
initializer (`onnx::MatMul_4711`, [in, out]) feeding a MatMul whose output goes into an Add with the still-named bias
(`....linear_q_k_v.bias`).  The walk names each such initializer after its bias sibling, or, for bias-free layers, after the
MatMul node's own name (`/decoder/decoders.0/feed_forward/w_2/MatMul` -> `decoder.decoders.0.feed_forward.w_2.weight`), and
transposes it back to torch's [out, in].  onnxruntime's dynamic quantisation (`model_quant.onnx`: `X_quantized` int8 /
`X_scale` / `X_zero_point` consumed by MatMulInteger / DynamicQuantizeMatMul) is folded back to float32 the same way.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

# TensorProto.DataType (onnx.proto3 `enum DataType`)
_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 4: np.uint16, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_,
           10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64}
DTYPE_NAMES = {1: "FLOAT", 2: "UINT8", 3: "INT8", 4: "UINT16", 5: "INT16", 6: "INT32", 7: "INT64", 8: "STRING", 9: "BOOL", 10: "FLOAT16",
               11: "DOUBLE", 12: "UINT32", 13: "UINT64", 14: "COMPLEX64", 15: "COMPLEX128", 16: "BFLOAT16"}


class OnnxFormatError(ValueError):
    pass


# ---- protobuf wire format -----------------------------------------------------------------------------------------
def _varint(buf, pos: int) -> Tuple[int, int]:
    result, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise OnnxFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise OnnxFormatError("varint longer than 10 bytes")


def _fields(buf) -> Iterator[Tuple[int, int, object]]:
    """Yields (field number, wire type, value) of one message: value = int for varint / fixed, memoryview for length-delimited."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if fno == 0:
            raise OnnxFormatError("field number 0")
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            if pos + 8 > end:
                raise OnnxFormatError("truncated fixed64")
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > end:
                raise OnnxFormatError("length-delimited field runs past its message")
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            if pos + 4 > end:
                raise OnnxFormatError("truncated fixed32")
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise OnnxFormatError(f"unsupported wire type {wt} (groups are not used by ONNX)")
        yield fno, wt, v


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_varints(v) -> List[int]:
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed64(x))
    return out


def _ints(wt, v) -> List[int]:            # a repeated int64 field arrives packed (wire type 2) or one varint per key
    return _packed_varints(v) if wt == 2 else [_signed64(v)]


# ---- the ONNX messages that carry weights ---------------------------------------------------------------------------
@dataclass
class OnnxNode:
    op_type: str = ""
    name: str = ""
    inputs: List[str] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)
    attrs: Dict[str, object] = field(default_factory=dict)        # ints, floats, strings, int lists, tensors (ndarray)


@dataclass
class OnnxModel:
    ir_version: int = 0
    producer: str = ""
    opsets: Dict[str, int] = field(default_factory=dict)
    graph_name: str = ""
    initializers: Dict[str, np.ndarray] = field(default_factory=dict)
    init_dtype: Dict[str, int] = field(default_factory=dict)       # TensorProto.DataType per initializer
    nodes: List[OnnxNode] = field(default_factory=list)
    inputs: List[Tuple[str, int, list]] = field(default_factory=list)     # (name, elem_type, dims: int | str | None)
    outputs: List[Tuple[str, int, list]] = field(default_factory=list)
    external: List[str] = field(default_factory=list)              # initializers whose bytes live in another file (not loaded)


def _tensor(buf) -> Tuple[str, Optional[np.ndarray], int, bool]:
    """TensorProto: 1 dims, 2 data_type, 4 float_data, 5 int32_data, 6 string_data, 7 int64_data, 8 name, 9 raw_data,
    10 double_data, 11 uint64_data, 13 external_data, 14 data_location."""
    dims: List[int] = []
    dtype, name, raw = 0, "", None
    f32: List[bytes] = []
    typed: List[int] = []
    f64: List[bytes] = []
    external = False
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _ints(wt, v)
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode("utf-8")
        elif fno == 9:
            raw = v
        elif fno == 4:
            f32.append(bytes(v) if wt == 2 else struct.pack("<I", v))
        elif fno in (5, 7, 11):
            typed += _ints(wt, v)
        elif fno == 10:
            f64.append(bytes(v) if wt == 2 else struct.pack("<Q", v))
        elif fno == 13 or (fno == 14 and v == 1):
            external = True
    if any(d < 0 for d in dims):
        raise OnnxFormatError(f"tensor {name}: negative dimension")
    n = int(np.prod(dims, dtype=np.int64)) if dims else 1
    if external:
        return name, None, dtype, True
    if dtype not in _DTYPES:
        if dtype == 16 and raw is not None:                   # bfloat16: widen to float32
            if len(raw) != n * 2:
                raise OnnxFormatError(f"tensor {name}: raw_data holds {len(raw)} bytes, dims {dims} need {n * 2}")
            u = np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16
            return name, u.view(np.float32).reshape(dims), dtype, False
        raise OnnxFormatError(f"tensor {name}: unsupported data type {DTYPE_NAMES.get(dtype, dtype)}")
    np_dt = np.dtype(_DTYPES[dtype])
    if raw is not None:
        if len(raw) != n * np_dt.itemsize:
            raise OnnxFormatError(f"tensor {name}: raw_data holds {len(raw)} bytes, dims {dims} need {n * np_dt.itemsize}")
        arr = np.frombuffer(raw, dtype=np_dt.newbyteorder("<")).astype(np_dt, copy=False)
    elif f32:
        arr = np.frombuffer(b"".join(f32), dtype="<f4")
    elif f64:
        arr = np.frombuffer(b"".join(f64), dtype="<f8")
    elif typed:
        try:
            if dtype == 10:                                   # float16 travels as its bit pattern in int32_data
                arr = (np.asarray(typed, np.int64) & 0xFFFF).astype(np.uint16).view(np.float16)
            else:
                arr = np.asarray(typed, np.int64 if np_dt.kind in "iub" else np.float64).astype(np_dt)
        except (OverflowError, ValueError) as e:
            raise OnnxFormatError(f"tensor {name}: {e}") from None
    else:
        arr = np.zeros(0, np_dt)
    if arr.size != n:
        raise OnnxFormatError(f"tensor {name}: {arr.size} values for dims {dims}")
    try:
        return name, arr.reshape(dims), dtype, False
    except ValueError as e:
        raise OnnxFormatError(f"tensor {name}: {e}") from None


def _attribute(buf):
    """AttributeProto: 1 name, 2 f, 3 i, 4 s, 5 t, 7 floats, 8 ints, 9 strings, 20 type (sub-graphs 6 / 11 are skipped)."""
    name, val = "", None
    floats: List[float] = []
    ints: List[int] = []
    strs: List[str] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode("utf-8")
        elif fno == 2:
            val = struct.unpack("<f", struct.pack("<I", v))[0]
        elif fno == 3:
            val = _signed64(v)
        elif fno == 4:
            val = bytes(v).decode("utf-8", "replace")
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            floats += list(np.frombuffer(bytes(v), "<f4")) if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]]
        elif fno == 8:
            ints += _ints(wt, v)
        elif fno == 9:
            strs.append(bytes(v).decode("utf-8", "replace"))
    if val is None:
        val = ints or floats or strs or None
    return name, val


def _node(buf) -> OnnxNode:
    """NodeProto: 1 input, 2 output, 3 name, 4 op_type, 5 attribute, 7 domain."""
    nd = OnnxNode()
    for fno, _, v in _fields(buf):
        if fno == 1:
            nd.inputs.append(bytes(v).decode("utf-8"))
        elif fno == 2:
            nd.outputs.append(bytes(v).decode("utf-8"))
        elif fno == 3:
            nd.name = bytes(v).decode("utf-8")
        elif fno == 4:
            nd.op_type = bytes(v).decode("utf-8")
        elif fno == 5:
            k, val = _attribute(v)
            nd.attrs[k] = val
    return nd


def _value_info(buf) -> Tuple[str, int, list]:
    """ValueInfoProto: 1 name, 2 type -> TypeProto.tensor_type (1) -> {1 elem_type, 2 shape -> dim (1) -> {1 dim_value, 2 dim_param}}."""
    name, elem, dims = "", 0, []
    for fno, _, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode("utf-8")
        elif fno == 2:
            for f2, _, v2 in _fields(v):
                if f2 != 1:
                    continue
                for f3, _, v3 in _fields(v2):
                    if f3 == 1:
                        elem = v3
                    elif f3 == 2:
                        for f4, _, v4 in _fields(v3):
                            if f4 != 1:
                                continue
                            d = None
                            for f5, _, v5 in _fields(v4):
                                if f5 == 1:
                                    d = _signed64(v5)
                                elif f5 == 2:
                                    d = bytes(v5).decode("utf-8")
                            dims.append(d)
    return name, elem, dims


def read_model(path_or_bytes) -> OnnxModel:
    """ModelProto: 1 ir_version, 2 producer_name, 7 graph, 8 opset_import {1 domain, 2 version};
    GraphProto: 1 node, 2 name, 5 initializer, 11 input, 12 output."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = memoryview(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            data = memoryview(f.read())
    m = OnnxModel()
    graph = None
    for fno, wt, v in _fields(data):
        if fno == 1 and wt == 0:
            m.ir_version = v
        elif fno == 2 and wt == 2:
            m.producer = bytes(v).decode("utf-8", "replace")
        elif fno == 7 and wt == 2:
            graph = v
        elif fno == 8 and wt == 2:
            dom, ver = "", 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    dom = bytes(v2).decode("utf-8")
                elif f2 == 2:
                    ver = v2
            m.opsets[dom] = ver
    if graph is None:
        raise OnnxFormatError("no graph in the ModelProto (not an ONNX file?)")
    for fno, wt, v in _fields(graph):
        if wt != 2:
            continue
        if fno == 1:
            m.nodes.append(_node(v))
        elif fno == 2:
            m.graph_name = bytes(v).decode("utf-8", "replace")
        elif fno == 5:
            name, arr, dt, ext = _tensor(v)
            if ext:
                m.external.append(name)
            else:
                m.initializers[name] = arr
            m.init_dtype[name] = dt
        elif fno == 11:
            m.inputs.append(_value_info(v))
        elif fno == 12:
            m.outputs.append(_value_info(v))
    return m


def summary(m: OnnxModel) -> dict:
    by_dtype: Dict[str, int] = {}
    for k, a in m.initializers.items():
        by_dtype[DTYPE_NAMES.get(m.init_dtype[k], "?")] = by_dtype.get(DTYPE_NAMES.get(m.init_dtype[k], "?"), 0) + a.nbytes
    ops: Dict[str, int] = {}
    for nd in m.nodes:
        ops[nd.op_type] = ops.get(nd.op_type, 0) + 1
    return {"ir_version": m.ir_version, "producer": m.producer, "opsets": m.opsets, "initializers": len(m.initializers),
            "initializer_bytes": int(sum(a.nbytes for a in m.initializers.values())), "bytes_by_dtype": by_dtype, "nodes": len(m.nodes),
            "ops": ops, "inputs": [i[0] for i in m.inputs], "outputs": [o[0] for o in m.outputs]}


def check_closed(m: OnnxModel) -> List[str]:
    """Every node input must be a graph input, an initializer or an earlier node's output: returns the names that are not —
    empty for a well-formed file READ CORRECTLY (a mis-parsed field number shows up here at once)."""
    known = {i[0] for i in m.inputs} | set(m.initializers) | set(m.external) | {""}
    bad = []
    for nd in m.nodes:
        for x in nd.inputs:
            if x not in known:
                bad.append(f"{nd.op_type}:{nd.name}:{x}")
        known.update(nd.outputs)
    bad += [f"output:{o[0]}" for o in m.outputs if o[0] not in known]
    return bad


# ---- anonymous initializers -> torch-style names ------------------------------------------------------------------------
def _module_of_node(name: str) -> Optional[str]:
    """'/encoder/encoders.3/feed_forward/w_1/MatMul' -> 'encoder.encoders.3.feed_forward.w_1' (the torch exporter names nodes
    after the module path); None when the node name carries no path."""
    if not name.startswith("/"):
        return None
    parts = [p for p in name.split("/") if p]
    if len(parts) < 2:
        return None
    return ".".join(parts[:-1])


def _is_anonymous(name: str) -> bool:
    return name.startswith("onnx::") or name.isdigit() or name.startswith("_v_") or "::" in name


def dequantize_initializers(m: OnnxModel) -> Dict[str, np.ndarray]:
    """onnxruntime quantize_dynamic leaves `<W>_quantized` (int8/uint8), `<W>_scale`, `<W>_zero_point` for every MatMul weight:
    W = (W_q - zero_point) * scale (per tensor, or per output column when scale is a vector).  Returns {<W>: float32}."""
    out = {}
    for k, q in m.initializers.items():
        if not k.endswith("_quantized"):
            continue
        base = k[:-len("_quantized")]
        sc, zp = m.initializers.get(base + "_scale"), m.initializers.get(base + "_zero_point")
        if sc is None or zp is None:
            continue
        out[base] = ((q.astype(np.int32) - zp.astype(np.int32)).astype(np.float32) * sc.astype(np.float32)).astype(np.float32)
    return out


def torch_style_state(m: OnnxModel) -> Dict[str, np.ndarray]:
    """{torch state_dict key: float32 ndarray in torch layout} from the initializers of an exported model (see module docstring).
    Named initializers pass through; MatMul / Gemm weights are renamed after their layer and transposed back to [out, in]."""
    init = dict(m.initializers)
    init.update(dequantize_initializers(m))
    consumers: Dict[str, List[OnnxNode]] = {}
    for nd in m.nodes:
        for x in nd.inputs:
            consumers.setdefault(x, []).append(nd)
    state: Dict[str, np.ndarray] = {}
    used = set()

    def put(key, arr):
        if key in state and not np.array_equal(state[key], arr):
            raise OnnxFormatError(f"two different tensors resolve to {key}")
        state[key] = np.ascontiguousarray(arr, dtype=np.float32) if arr.dtype.kind == "f" else arr

    for nd in m.nodes:
        if nd.op_type in ("MatMul", "MatMulInteger", "DynamicQuantizeMatMul"):
            wname = nd.inputs[1] if len(nd.inputs) > 1 else ""
            base = wname[:-len("_quantized")] if wname.endswith("_quantized") else wname
            if base not in init or init[base].ndim != 2:
                continue
            w = init[base]
            module = None
            # bias sibling: MatMul -> [Cast / Mul (quantised paths)] -> Add(named bias)
            frontier, hops = list(nd.outputs), 0
            while frontier and hops < 4 and module is None:
                nxt = []
                for o in frontier:
                    for c in consumers.get(o, []):
                        if c.op_type == "Add":
                            for x in c.inputs:
                                if x in m.initializers and x.endswith(".bias") and m.initializers[x].shape == (w.shape[1],):
                                    module = x[:-len(".bias")]
                        elif c.op_type in ("Cast", "Mul"):
                            nxt += c.outputs
                frontier, hops = nxt, hops + 1
            if module is None:
                module = _module_of_node(nd.name[:-len("_quant")] if nd.name.endswith("_quant") else nd.name)
            if module is None and not _is_anonymous(base):
                module = base[:-len(".weight")] if base.endswith(".weight") else base
            if module is None:
                continue
            put(module + ".weight", w.T)
            used.update({wname, base, base + "_scale", base + "_zero_point"})
        elif nd.op_type == "Gemm" and len(nd.inputs) > 1 and nd.inputs[1] in init:
            w = init[nd.inputs[1]]
            if not _is_anonymous(nd.inputs[1]):
                continue                                        # named: passes through below in its stored layout
            module = None
            if len(nd.inputs) > 2 and nd.inputs[2].endswith(".bias"):
                module = nd.inputs[2][:-len(".bias")]
            module = module or _module_of_node(nd.name)
            if module is None:
                continue
            put(module + ".weight", w if nd.attrs.get("transB", 0) else w.T)
            used.add(nd.inputs[1])
        elif nd.op_type == "LSTM" and len(nd.inputs) > 2 and nd.inputs[1] in init and nd.inputs[2] in init:
            # ONNX LSTM: W [dirs, 4h, in], R [dirs, 4h, h], B [dirs, 8h] = Wb | Rb, gate order i, o, f, c; torch: i, f, g(c), o
            module = _module_of_node(nd.name)
            if module is None:
                for x in (nd.inputs[1], nd.inputs[2]):
                    if not _is_anonymous(x):
                        module = x.rsplit(".", 1)[0]
            if module is None:
                continue
            Wm, Rm = init[nd.inputs[1]], init[nd.inputs[2]]
            Bm = init.get(nd.inputs[3]) if len(nd.inputs) > 3 and nd.inputs[3] else None
            h = Rm.shape[2]

            def torch_gates(a):           # rows (i, o, f, c) -> (i, f, c, o)
                i_, o_, f_, c_ = (a[g * h:(g + 1) * h] for g in range(4))
                return np.concatenate([i_, f_, c_, o_], 0)
            for dnum in range(Wm.shape[0]):
                sfx = "_reverse" if dnum == 1 else ""
                put(f"{module}.weight_ih_l0{sfx}", torch_gates(Wm[dnum]))
                put(f"{module}.weight_hh_l0{sfx}", torch_gates(Rm[dnum]))
                if Bm is not None:
                    put(f"{module}.bias_ih_l0{sfx}", torch_gates(Bm[dnum][:4 * h]))
                    put(f"{module}.bias_hh_l0{sfx}", torch_gates(Bm[dnum][4 * h:]))
            used.update(nd.inputs[1:4])
    for k, a in init.items():
        if k in used or _is_anonymous(k) or k.endswith(("_quantized", "_scale", "_zero_point")):
            continue
        if a.dtype.kind == "f" and k not in state:
            put(k, a)
    return state
