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
