"""Dev tool (GPU): times the fp32 MFMA GEMM on the bench shapes, interleaved rounds in one process."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
ops = importlib.import_module("asr_2pass_amd.ops")

shapes = [(16000, 2048, 512), (16000, 512, 2048), (16000, 1536, 512), (16000, 512, 512), (16000, 1024, 512),
          (7040, 512, 512), (7040, 2048, 512), (7040, 8448, 512)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
bufs = {}
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    R = torch.randn(M, N, device="cuda")
    bufs[(M, N, K)] = (A, W, b, C, R)
for sh in shapes:                      # warm-up + correctness spot check
    A, W, b, C, R = bufs[sh]
    ops.gemm_f32(A, W, bias=b, R1=R, out=C, guard=False)
    ref = A[:256] @ W.T + b + R[:256]
    print(sh, "max err", float((C[:256] - ref).abs().max()))
times = {sh: [] for sh in shapes}
for r in range(rounds):
    for sh in shapes:
        A, W, b, C, R = bufs[sh]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_f32(A, W, bias=b, R1=R, out=C, guard=False)
        e1.record()
        torch.cuda.synchronize()
        times[sh].append(e0.elapsed_time(e1))
for sh in shapes:
    M, N, K = sh
    t = np.asarray(times[sh][2:])
    fl = 2.0 * M * N * K
    print(f"{sh}: median {np.median(t)*1e3:8.1f} us  min {t.min()*1e3:8.1f} us  -> {fl/np.median(t)/1e9:7.1f} TF (median) {fl/t.min()/1e9:7.1f} TF (best)")
