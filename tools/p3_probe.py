"""Times gemm_p3 (plane-image operands) on the encoder's four launch shapes in their model configuration, cycling over four
operand / result sets per shape so that back-to-back launches do not hit in cache.  PFHIP_LIB selects a variant build
(tools/x3_variant.sh <name> "<-D flags>" gemm_p3.hip):
    python3 tools/p3_probe.py [rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
ops = importlib.import_module("asr_2pass_amd.ops")
torch.manual_seed(0)
M = int(os.environ.get("P3_ROWS", "16000"))
SHAPES = [  # name, N, K, LN fold, fp32 out, planes out, residual, relu, stats out
    ("qkv", 1536, 512, True, True, False, False, False, False),
    ("out-proj", 512, 512, False, True, True, True, False, True),
    ("ffn1", 2048, 512, True, False, True, False, True, False),
    ("ffn2", 512, 2048, False, True, True, True, False, True),
]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
print("library:", os.environ.get("PFHIP_LIB", "in-tree"), "rows", M)
Mp = ((M + 127) // 128) * 128
for name, N, K, ln, want_c, want_p, res, relu, st_out in SHAPES:
    sets = []
    for _ in range(4):
        A = torch.randn(Mp, K, device="cuda")
        W = torch.randn(N, K, device="cuda") / K ** 0.5
        ws = ops.best_w_scale(float(W.abs().max()))
        Ai = ops.split_planes(A, rows=Mp)
        Wi = ops.split_planes(W, scale=ws)
        b = torch.randn(N, device="cuda")
        R = torch.randn(Mp, N, device="cuda") if res else None
        stats = torch.zeros(Mp, 4, 2, device="cuda"); stats[:, :, 1] = 128.0
        colsum = W.sum(1).contiguous()
        C, P = ops.gemm_p3(Ai, Wi, M, N, K, w_scale=ws, bias=b, R1=R, relu=relu, want_c=want_c, want_planes=want_p,
                           ln_stats=stats if ln else None, ln_tiles=4 if ln else 0, ln_colsum=colsum if ln else None,
                           stats_out=torch.zeros(Mp, N // 128, 2, device="cuda") if st_out else None)
        so = torch.zeros(Mp, N // 128, 2, device="cuda") if st_out else None
        sets.append((Ai, Wi, ws, b, R, stats, colsum, C, P, so))
    ts = []
    for r in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            Ai, Wi, ws, b, R, stats, colsum, C, P, so = sets[i & 3]
            ops.gemm_p3(Ai, Wi, M, N, K, w_scale=ws, bias=b, R1=R, relu=relu, want_c=want_c, want_planes=want_p,
                        ln_stats=stats if ln else None, ln_tiles=4 if ln else 0, ln_colsum=colsum if ln else None, stats_out=so,
                        out=C, out_planes=P)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 12)
    t = float(np.median(ts[2:])) * 1e-3
    print(f"{name:9s} M={M:5d} N={N:5d} K={K:5d}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:6.1f} TF", flush=True)
