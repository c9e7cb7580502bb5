import sys, os
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
ops = importlib.import_module("asr_2pass_amd.ops")
torch.manual_seed(0)
for (M, N, K) in [(500,512,512),(500,1536,512),(500,2048,512),(500,512,2048),(1000,512,512),(1000,1536,512),(1000,2048,512),(1000,512,2048),(2000,512,512),(2000,1536,512),(2000,2048,512),(2000,512,2048),(2560,512,512),(2560,2048,512)]:
    Np = ((N + 127) // 128) * 128; Mp = ((M + 255) // 256) * 256
    W = torch.randn(Np, K, device="cuda") / K ** 0.5; A = torch.randn(Mp, K, device="cuda"); C = torch.empty(Mp, Np, device="cuda")
    line = f"M={M:5d} N={N:5d} K={K:5d} tiles128={((M+127)//128)*(Np//128):4d}:"
    for k in (0, 5, 7):
        ts = []
        for r in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_f32(A, W, out=C, M=M, N=N, guard=True, kind=k)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        t = float(np.median(ts[2:])) * 1e-3
        line += f"  kind {k}: {t*1e6:7.1f} us {2.0*M*N*K/t/1e12:6.1f} TF"
    print(line, flush=True)
