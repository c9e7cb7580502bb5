"""Dev tool (GPU): tiled vs weight-streaming GEMM kernel over M (the crossover that sets kSkinnyMaxM in gemm.hip)."""
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
for (N, K) in [(1536, 512), (512, 512), (2048, 512), (512, 2048), (8448, 512)]:
    W = torch.randn(((N + 127) // 128) * 128, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    for M in (640, 1024, 2048, 3072, 4000, 5000, 7040, 8000, 12000, 16000):
        Mp = ((M + 127) // 128) * 128
        A = torch.randn(Mp, K, device="cuda")
        C = torch.empty(Mp, ((N + 127) // 128) * 128, device="cuda")
        res = {}
        errs = {}
        for kind in (1, 2, 3):
            ts = []
            for r in range(12):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.gemm_f32(A, W, bias=b, out=C, M=M, N=N, guard=True, kind=kind)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            res[kind] = float(np.median(ts[2:])) * 1e3
            errs[kind] = float((C[:M, :N] - (A[:M] @ W[:N].T + b)).abs().max())
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        print(f"N={N:5d} K={K:5d} M={M:5d} tiles {tiles:5d}: tiled {res[1]:8.1f} us  streaming {res[2]:8.1f} us  64-row {res[3]:8.1f} us  "
              f"best {min(res, key=res.get)}  (err {max(errs.values()):.1e})", flush=True)
