"""Times the BF16-split GEMM on the model's launch shapes through the operator ABI (bias, residual and ReLU as in the model;
no row statistics / folded LayerNorm: those need the model-level launcher).  With PFHIP_LIB pointing at a timing-only build
(epilogue without its stores, or without the epilogue) the difference to the real library is what that phase costs:
    python3 tools/epilogue_probe.py            # in-tree library
    PFHIP_LIB=build/ab/libpfhip_noepi.so python3 tools/epilogue_probe.py
To defeat back-to-back cache reuse each shape cycles over 4 operand / result sets (the model's activations are ~130 MB apart)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
ops = importlib.import_module("asr_2pass_amd.ops")
torch.manual_seed(0)
SHAPES = [  # name, M, N, K, kind, residual, relu
    ("enc qkv", 16000, 1536, 512, 5, False, False), ("enc out-proj", 16000, 512, 512, 5, True, False),
    ("enc ffn1", 16000, 2048, 512, 5, False, True), ("enc ffn2", 16000, 512, 2048, 4, True, False),
    ("dec kv", 16000, 1024, 512, 5, False, False), ("dec ffn1", 7015, 2048, 512, 5, False, True),
    ("dec ffn2", 7015, 512, 2048, 5, True, False), ("dec q/out", 7015, 512, 512, 5, True, False),
]
print("library:", os.environ.get("PFHIP_LIB", "in-tree"))
for name, M, N, K, kind, res, relu in SHAPES:
    Mp = ((M + 255) // 256) * 256
    sets = []
    for _ in range(4):
        W = torch.randn(N, K, device="cuda") / K ** 0.5; A = torch.randn(Mp, K, device="cuda"); C = torch.empty(Mp, N, device="cuda")
        R = torch.randn(Mp, N, device="cuda") if res else None
        b = torch.randn(N, device="cuda")
        sets.append((A, W, C, R, b))
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            A, W, C, R, b = sets[i & 3]
            ops.gemm_f32(A, W, bias=b, R1=R, relu=relu, out=C, M=M, N=N, guard=True, kind=kind)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 12)
    t = float(np.median(ts[2:])) * 1e-3
    print(f"{name:14s} M={M:5d} N={N:5d} K={K:5d} kind {kind}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:6.1f} TF", flush=True)
