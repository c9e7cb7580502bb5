"""Test helper: a minimal ONNX (protobuf) WRITER, so that the converter's directory path can be driven through real ONNX bytes
without the `onnx` package.  Field numbers from the public onnx.proto3 schema; only what a weight file needs."""
import struct

import numpy as np

_DT = {np.dtype("float32"): 1, np.dtype("uint8"): 2, np.dtype("int8"): 3, np.dtype("int32"): 6, np.dtype("int64"): 7,
       np.dtype("float16"): 10, np.dtype("float64"): 11}


def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def key(fno, wt):
    return varint(fno << 3 | wt)


def ld(fno, payload):            # length-delimited field
    return key(fno, 2) + varint(len(payload)) + payload


def s(fno, text):
    return ld(fno, text.encode("utf-8"))


def tensor(name, arr, form="raw"):
    """TensorProto; form: 'raw' (raw_data), 'typed' (float_data / int64_data / int32_data packed), 'unpacked' (one key per dim)."""
    arr = np.asarray(arr, order="C")            # (ascontiguousarray would turn a 0-d scalar into shape (1,))
    out = b""
    if form == "unpacked":
        for d in arr.shape:
            out += key(1, 0) + varint(d)
    else:
        out += ld(1, b"".join(varint(d) for d in arr.shape))
    out += key(2, 0) + varint(_DT[arr.dtype])
    if form == "raw" or arr.dtype not in (np.dtype("float32"), np.dtype("int64"), np.dtype("int32")):
        out += ld(9, arr.astype(arr.dtype.newbyteorder("<")).tobytes())
    elif arr.dtype == np.dtype("float32"):
        out += ld(4, arr.astype("<f4").tobytes())
    else:
        out += ld(7 if arr.dtype == np.dtype("int64") else 5, b"".join(varint(int(x)) for x in arr.reshape(-1)))
    out += s(8, name)
    return out


def attr_int(name, v):
    return s(1, name) + key(3, 0) + varint(v) + key(20, 0) + varint(2)


def node(op, inputs, outputs, name="", attrs=()):
    out = b"".join(s(1, x) for x in inputs) + b"".join(s(2, x) for x in outputs)
    if name:
        out += s(3, name)
    out += s(4, op)
    for a in attrs:
        out += ld(5, a)
    return out


def value_info(name, elem=1, dims=("N", 10)):
    shape = b""
    for d in dims:
        shape += ld(1, s(2, d) if isinstance(d, str) else key(1, 0) + varint(d))
    ttype = key(1, 0) + varint(elem) + ld(2, shape)
    return s(1, name) + ld(2, ld(1, ttype))


def model(nodes, initializers, inputs, outputs, producer="pfhip-test", opset=13):
    g = b"".join(ld(1, n) for n in nodes) + s(2, "g") + b"".join(ld(5, t) for t in initializers)
    g += b"".join(ld(11, i) for i in inputs) + b"".join(ld(12, o) for o in outputs)
    return key(1, 0) + varint(7) + s(2, producer) + ld(7, g) + ld(8, s(1, "") + key(2, 0) + varint(opset))


class GraphBuilder:
    """Builds a chain the way the PyTorch exporter leaves one: named LayerNorm / conv / bias initializers, anonymous transposed
    MatMul weights (`onnx::MatMul_<n>`), node names after the module path."""

    def __init__(self):
        self.nodes, self.inits, self.n, self.cur = [], [], 100, "x"

    def fresh(self):
        self.n += 1
        return f"/t{self.n}"

    def linear(self, module, w, b=None, anonymous=True, form="raw"):
        path = "/" + module.replace(".", "/")
        wname = f"onnx::MatMul_{self.n + 1000}" if anonymous else module + ".weight"
        self.inits.append(tensor(wname, np.ascontiguousarray(w.T), form))            # exporter stores [in, out]
        o = self.fresh()
        self.nodes.append(node("MatMul", [self.cur, wname], [o], path + "/MatMul"))
        self.cur = o
        if b is not None:
            self.inits.append(tensor(module + ".bias", b, form))
            o = self.fresh()
            self.nodes.append(node("Add", [module + ".bias", self.cur], [o], path + "/Add"))
            self.cur = o

    def named(self, op, module, arrays, form="raw"):
        """An op that keeps its parameters' names: LayerNormalization(weight, bias), Conv(weight[, bias]), Gather(weight)."""
        names = []
        for suffix, a in arrays:
            self.inits.append(tensor(f"{module}.{suffix}", a, form))
            names.append(f"{module}.{suffix}")
        o = self.fresh()
        self.nodes.append(node(op, [self.cur] + names, [o], "/" + module.replace(".", "/") + "/" + op))
        self.cur = o

    def lstm(self, module, w_ih, w_hh, b_ih, b_hh, reverse=None):
        """torch gate order (i, f, g, o) -> ONNX LSTM W / R / B with gates (i, o, f, c); `reverse` = the second direction."""
        def onnx_gates(a):
            h = a.shape[0] // 4
            i_, f_, c_, o_ = (a[g * h:(g + 1) * h] for g in range(4))
            return np.concatenate([i_, o_, f_, c_], 0)
        dirs = [(w_ih, w_hh, b_ih, b_hh)] + ([reverse] if reverse else [])
        W = np.stack([onnx_gates(d[0]) for d in dirs])
        R = np.stack([onnx_gates(d[1]) for d in dirs])
        B = np.stack([np.concatenate([onnx_gates(d[2]), onnx_gates(d[3])]) for d in dirs])
        names = [f"onnx::LSTM_{self.n + 2000 + k}" for k in range(3)]
        for nm, a in zip(names, (W, R, B)):
            self.inits.append(tensor(nm, a))
        o = self.fresh()
        self.nodes.append(node("LSTM", [self.cur] + names, [o], "/" + module.replace(".", "/") + "/LSTM",
                               [attr_int("hidden_size", w_hh.shape[1])]))
        self.cur = o

    def build(self):
        return model(self.nodes, self.inits, [value_info("x")], [value_info(self.cur)])
