import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
ops = importlib.import_module("asr_2pass_amd.ops")
torch.manual_seed(0)
for (M, N, K) in [(7015,512,512),(7015,512,2048),(7015,2048,512),(7015,1024,512),(5000,512,512),(5000,512,2048),(4000,512,512),(4000,512,2048),(3000,512,512),(3000,512,2048)]:
    Np = ((N + 127) // 128) * 128; Mp = ((M + 255) // 256) * 256
    sets = []
    for _ in range(4):
        W = torch.randn(Np, K, device="cuda") / K ** 0.5; A = torch.randn(Mp, K, device="cuda"); C = torch.empty(Mp, Np, device="cuda"); R = torch.randn(Mp, Np, device="cuda")
        sets.append((A, W, C, R))
    line = f"M={M:5d} N={N:5d} K={K:5d} tiles128={((M+127)//128)*(Np//128):4d}:"
    for k in (0, 4, 5, 7):
        ts = []
        for r in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                A, W, C, R = sets[i & 3]
                ops.gemm_f32(A, W, R1=R, out=C, M=M, N=N, guard=True, kind=k)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 12)
        t = float(np.median(ts[2:])) * 1e-3
        line += f"  kind {k}: {t*1e6:6.1f} us {2.0*M*N*K/t/1e12:5.1f} TF"
    print(line, flush=True)
