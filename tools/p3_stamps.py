"""Dev tool (GPU): timeline of gemm_p3_128_kernel from in-kernel s_memtime stamps (build: tools/x3_variant.sh p3st "-DPFHIP_P3_STAMPS=1"
gemm_p3.hip; run with PFHIP_LIB=build/libpfhip_p3st.so).  Two stamped workgroups: tile (0, 0) — in the first round of workgroups — and a
mid-grid tile.  Per wave: prologue, per-K-step DMA issue / MFMAs + fragment reads / wait + barrier, epilogue.
    python3 tools/p3_stamps.py [qkv|out]"""
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
which = sys.argv[1] if len(sys.argv) > 1 else "qkv"
M, K = 16000, 512
N, ln, res, planes, st = (1536, True, False, False, False) if which == "qkv" else (512, False, True, True, True)
Mp = (M + 127) // 128 * 128
A = torch.randn(Mp, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
ws = ops.best_w_scale(float(W.abs().max()))
Ai, Wi = ops.split_planes(A, rows=Mp), ops.split_planes(W, scale=ws)
b = torch.randn(N, device="cuda"); R = torch.randn(Mp, N, device="cuda") if res else None
stats = torch.zeros(Mp, 4, 2, device="cuda"); stats[:, :, 1] = 128.0
colsum = W.sum(1).contiguous()
so = torch.zeros(Mp, N // 128, 2, device="cuda") if st else None
for _ in range(20):
    C, P = ops.gemm_p3(Ai, Wi, M, N, K, w_scale=ws, bias=b, R1=R, want_c=True, want_planes=planes, ln_stats=stats if ln else None,
                       ln_tiles=4 if ln else 0, ln_colsum=colsum if ln else None, stats_out=so, tile_rows=128)
torch.cuda.synchronize()
Cn = C.cpu().numpy()
nk = K // 16
for tm in (0, (M // 128) // 2):
    print(f"tile ({tm}, 0):")
    for w in range(4):
        st_ = np.concatenate([Cn[128 * tm + 32 * w + j, :128].view(np.uint64) for j in range(3)])[:4 + 4 * nk + 8].astype(np.int64)
        pre = st_[:5]; steps = st_[4:4 + 4 * nk + 1]; post = st_[4 + 4 * nk:]
        d = np.diff(steps).reshape(nk, 4)
        print(f"  wave {w}: entry->DMA0 {pre[1]-pre[0]}, 4 stages issued {pre[2]-pre[1]}, landed+barrier {pre[3]-pre[2]}, first fragments {pre[4]-pre[3]}; "
              f"loop {steps[-1]-steps[0]} = {nk} x (stamp {d[2:,0].mean():.0f}, DMA issue {d[2:,1].mean():.0f}, MFMAs+reads {d[2:,2].mean():.0f}, wait+barrier {d[2:,3].mean():.0f}); "
              f"drain+barrier {post[2]-post[0]}, C tile to LDS {post[3]-post[2]}, LDS reads {post[4]-post[3]}, fp32 pass {post[5]-post[4]}, rest {post[6]-post[5]}; total {post[6]-pre[0]} cycles")
