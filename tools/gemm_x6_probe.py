"""Dev tool (GPU): fp32 MFMA GEMM (kind 1) vs the bf16x6 split kernel (kind 4) on the model's shapes — error against an fp64
reference and time, the two variants interleaved across repetitions."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
ops = importlib.import_module("asr_2pass_amd.ops")
torch.manual_seed(0)
kinds = [int(x) for x in os.environ.get("KINDS", "1,4").split(",")]
shapes = [(16000, 1536, 512), (16000, 512, 512), (16000, 2048, 512), (16000, 512, 2048), (3840, 8404, 512), (3840, 1024, 512), (1000, 512, 512)]
if os.environ.get("SHAPES") == "enc4":
    shapes = [(16000, 1536, 512), (16000, 512, 512), (16000, 2048, 512), (16000, 512, 2048)]
if os.environ.get("SHAPES") == "dec":
    shapes = [(M, N, K) for M in (2048, 4000, 7015, 12000) for (N, K) in ((512, 512), (1024, 512), (2048, 512), (512, 2048), (1536, 512))] + [(7015, 8404, 512)]
for (M, N, K) in shapes:
    Np = ((N + 127) // 128) * 128
    Mp = ((M + 255) // 256) * 256
    W = torch.zeros(Np, K, device="cuda"); W[:N] = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    A = torch.randn(Mp, K, device="cuda")
    R = torch.randn(Mp, Np, device="cuda")
    C = {k: torch.empty(Mp, Np, device="cuda") for k in kinds}
    ref = (A[:M].double() @ W[:N].double().T + b.double() + R[:M, :N].double()).clamp_min(0)
    ts = {k: [] for k in kinds}
    ws = ops.best_w_scale(float(W.abs().max()))
    for r in range(10):
        for k in kinds:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_f32(A, W, bias=b, R1=R, relu=True, out=C[k], M=M, N=N, guard=True, kind=k, w_scale=(ws if k >= 8 else None))
            e1.record(); torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 5)
    if os.environ.get("P3") == "1" and N % 128 == 0:          # the pre-split-operand kernel (gemm_p3.hip) on the same shape
        a_img = ops.split_planes(A[:M].contiguous())
        w_img = ops.split_planes(W[:N].contiguous(), scale=ws)
        tp = []
        for r in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                Cp, _ = ops.gemm_p3(a_img, w_img, M, N, K, w_scale=ws, bias=b, R1=R, relu=True, want_c=True)
            e1.record(); torch.cuda.synchronize()
            tp.append(e0.elapsed_time(e1) / 5)
        tpp, tpb = [], []
        _, Pbuf = ops.gemm_p3(a_img, w_img, M, N, K, w_scale=ws, bias=b, relu=True, want_c=False, want_planes=True)
        Cbuf = torch.empty_like(Cp)
        for r in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_p3(a_img, w_img, M, N, K, w_scale=ws, bias=b, relu=True, want_c=False, want_planes=True, out_planes=Pbuf)
            e1.record(); torch.cuda.synchronize()
            tpp.append(e0.elapsed_time(e1) / 5)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm_p3(a_img, w_img, M, N, K, w_scale=ws, bias=b, R1=R, relu=True, want_c=True, want_planes=True, out=Cbuf, out_planes=Pbuf)
            e1.record(); torch.cuda.synchronize()
            tpb.append(e0.elapsed_time(e1) / 5)
        t = float(np.median(tp[2:])) * 1e-3
        t2 = float(np.median(tpp[2:])) * 1e-3
        t3 = float(np.median(tpb[2:])) * 1e-3
        err = float((Cp[:M, :N].double() - ref).abs().max())
        print(f"M={M:5d} N={N:5d} K={K:5d}:  p3 fp32-out: {t * 1e6:8.1f} us {2.0 * M * N * K / t / 1e12:6.1f} TF err {err:.1e}   planes-out: {t2 * 1e6:8.1f} us {2.0 * M * N * K / t2 / 1e12:6.1f} TF   both: {t3 * 1e6:8.1f} us", flush=True)
    line = f"M={M:5d} N={N:5d} K={K:5d}:"
    for k in kinds:
        t = float(np.median(ts[k][2:])) * 1e-3
        err = float((C[k][:M, :N].double() - ref).abs().max())
        rms = float(((C[k][:M, :N].double() - ref) ** 2).mean().sqrt())
        line += f"  kind {k}: {t * 1e6:8.1f} us {2.0 * M * N * K / t / 1e12:6.1f} TF err {err:.1e} rms {rms:.1e}"
    print(line, flush=True)
