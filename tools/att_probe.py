"""Dev tool (GPU): times the d_k = 128 attention launch on the model's two shapes (encoder self-attention 32 x 500 x 500,
cross-attention 32 x ~219 x 500) through the ops ABI; used with PFHIP_LIB=<variant> for A/B and timing-only ablation builds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
ops = importlib.import_module("asr_2pass_amd.ops")
torch.manual_seed(0)
B, T, H, D = 32, 500, 4, 128
lib, P, S = ops._lib(), ops._p, ops._stream
for name, Lq in (("self", T), ("cross", 219)):
    qkv = torch.randn(B * T, 3 * H * D, device="cuda")
    q = torch.randn(B * Lq, H * D, device="cuda") if Lq != T else qkv
    O = torch.empty(B * Lq, H * D, device="cuda")
    q_off = torch.arange(B, dtype=torch.int32, device="cuda") * Lq
    q_len = torch.full((B,), Lq, dtype=torch.int32, device="cuda")
    k_off = torch.arange(B, dtype=torch.int32, device="cuda") * T
    k_len = torch.full((B,), T, dtype=torch.int32, device="cuda")
    K, V = qkv[:, H * D:], qkv[:, 2 * H * D:]
    ts = []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            rc = lib.pfhip_op_attention_hd(P(q), q.stride(0), P(K), K.stride(0), P(V), V.stride(0), P(O), O.stride(0), P(q_off), P(q_len),
                                           P(k_off), P(k_len), B, H, Lq, D ** -0.5, D, S())
            assert rc == 0
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = float(np.median(ts[2:])) * 1e-3
    fl = 4.0 * B * Lq * T * H * D
    print(f"{name:6s} Lq={Lq:4d}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF", flush=True)

# the same two shapes with K | V as row-major planes (attention_p3.hip), without and with the fused memory block + plane-image output
for name, Lq in (("self", T), ("cross", 219)):
    kvsrc = torch.randn(B * T, 2 * H * D, device="cuda")
    kv = ops.split_rows(kvsrc)
    q = torch.randn(B * Lq, H * D, device="cuda")
    q_off = torch.arange(B, dtype=torch.int32, device="cuda") * Lq
    q_len = torch.full((B,), Lq, dtype=torch.int32, device="cuda")
    k_off = torch.arange(B, dtype=torch.int32, device="cuda") * T
    k_len = torch.full((B,), T, dtype=torch.int32, device="cuda")
    O = torch.empty(B * Lq, H * D, device="cuda")
    w = torch.randn(H * D, 11, device="cuda") / 3
    mem = torch.zeros(B * T, H * D, device="cuda")
    rows = ops.round_up(B * Lq, 128)
    nb = int(lib.pfhip_op_plane_image_bytes(rows, H * D))
    ph, pl = torch.zeros(nb, dtype=torch.uint8, device="cuda"), torch.zeros(nb, dtype=torch.uint8, device="cuda")
    ops.attention_kvplanes(q, kv, H * D, q_off, q_len, k_off, k_len, H, D ** -0.5)      # sets argtypes
    for fused in ((False, True) if Lq == T else (False,)):
        ts = []
        for rep in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                rc = lib.pfhip_op_attention_kvplanes(P(q), q.stride(0), P(kv[0]), P(kv[1]), 2 * H * D, H * D, B * T, None if fused else P(O),
                                                     0 if fused else O.stride(0), P(ph) if fused else None, P(pl) if fused else None, rows,
                                                     P(q_off), P(q_len), P(k_off), P(k_len), B, H, Lq, B * Lq, D ** -0.5,
                                                     P(w) if fused else None, P(mem) if fused else None, H * D if fused else 0, 1 if fused else 0, S())
                assert rc == 0
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        t = float(np.median(ts[2:])) * 1e-3
        fl = 4.0 * B * Lq * T * H * D
        print(f"kv-planes {name:6s} Lq={Lq:4d}{' + memory block, image out' if fused else '':28s}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF", flush=True)
