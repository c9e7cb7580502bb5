"""Dev tool (GPU): the encoder's QKV projection at 16000 rows with its two epilogues — fp32 Q | K | V (OUT = 1) against fp32 Q + row-major
K | V planes (gemm_p3 OUT = 5, the producer side of attention_p3.hip), same session, four operand sets cycled."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
ops = importlib.import_module("asr_2pass_amd.ops")
torch.manual_seed(0)
M, N, K, D = 16000, 1536, 512, 512
Mp = (M + 127) // 128 * 128
sets = []
for _ in range(4):
    A = torch.randn(Mp, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    ws = ops.best_w_scale(float(W.abs().max()))
    sets.append((ops.split_planes(A, rows=Mp), ops.split_planes(W, scale=ws), ws, torch.randn(N, device="cuda"),
                 torch.zeros(Mp, 4, 2, device="cuda") + torch.tensor([0.0, 128.0], device="cuda"), W.sum(1).contiguous()))
lib, P, S = ops._lib(), ops._p, ops._stream
C = torch.zeros(Mp, N, device="cuda"); Cq = torch.zeros(Mp, D, device="cuda")
kvh = torch.zeros(Mp, N - D, dtype=torch.float16, device="cuda"); kvl = torch.zeros_like(kvh)
ops.gemm_p3_qkv(sets[0][0], sets[0][1], M, N, K, D, w_scale=sets[0][2], bias=sets[0][3], ln_stats=sets[0][4], ln_tiles=4, ln_colsum=sets[0][5])
for name in ("fp32 Q|K|V (OUT=1)", "fp32 Q + K|V planes (OUT=5)", "fp32 Q|K|V (OUT=1)", "fp32 Q + K|V planes (OUT=5)"):
    ts = []
    for r in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            Ai, Wi, ws, b, st, cs = sets[i & 3]
            if "OUT=1" in name:
                ops.gemm_p3(Ai, Wi, M, N, K, w_scale=ws, bias=b, want_c=True, want_planes=False, ln_stats=st, ln_tiles=4, ln_colsum=cs, out=C)
            else:
                rc = lib.pfhip_op_gemm_p3_qkv(P(Ai[0]), P(Ai[1]), Ai[2], P(Wi[0]), P(Wi[1]), Wi[2], float(ws), P(Cq), D, P(kvh), P(kvl), N - D, D, P(b),
                                              M, N, K, P(st), 4, P(cs), 0, S())
                assert rc == 0
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 12)
    print(f"{name:32s} {float(np.median(ts[2:])) * 1e3:7.1f} us", flush=True)
