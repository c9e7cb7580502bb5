"""Dev tool (GPU): the one-window fused LN+GEMM (stream_fused.hip) in isolation — error against fp64 and time per launch with
COLD weights (a ring of weight copies larger than the Infinity Cache, as in a streaming chunk that walks 0.88 GB per pass)."""
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
M = int(os.environ.get("M", "20"))
for (N, K, ln) in ((1536, 512, True), (512, 512, False), (2048, 512, True), (512, 2048, False), (512, 2048, True), (1024, 512, False), (8404, 512, True)):
    Np = ((N + 127) // 128) * 128
    copies = max(2, int(600e6 // (Np * K * 4)))
    Ws = torch.randn(copies, Np, K, device="cuda") / K ** 0.5
    X = torch.randn(32, K, device="cuda")
    g = torch.rand(K, device="cuda") + 0.5
    b = torch.randn(K, device="cuda") * 0.1
    bias = torch.randn(Np, device="cuda")
    R = torch.randn(32, Np, device="cuda")
    out = torch.zeros(32, Np, device="cuda")
    xd = X[:M].double()
    if ln:
        xd = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-12) * g.double() + b.double()
    ref = (xd @ Ws[0, :N].double().T + bias[:N].double() + R[:M, :N].double()).clamp_min(0)
    ops.fused_ln_gemm(X, Ws[0], M, N, g=g if ln else None, b=b if ln else None, bias=bias, R1=R, relu=True, out=out)
    err = float((out[:M, :N].double() - ref).abs().max())
    ts = []
    import ctypes
    lib = pkg.load_lib()
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.pfhip_dev_fused_ln_gemm_bench.argtypes = [vp, ci, ci, vp, vp, vp, ci, ctypes.c_size_t, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 200
        e0.record()
        lib.pfhip_dev_fused_ln_gemm_bench(X.data_ptr(), K, K, g.data_ptr() if ln else None, b.data_ptr() if ln else None, Ws.data_ptr(), K,
                                          Np * K, copies, out.data_ptr(), Np, bias.data_ptr(), R.data_ptr(), Np, M, N, K, 1, n,
                                          torch.cuda.current_stream().cuda_stream)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    t = float(np.median(ts[1:]))
    line = f"M={M} N={N:5d} K={K:5d} LN={int(ln)}: {t:7.2f} us per launch  ({Np * K * 4 / t / 1e3:7.1f} GB/s of weights)  err {err:.1e}"
    # one-trip form (folded weights when LN)
    if N <= 4096 and M <= 20:
        if ln:
            Wf = torch.empty_like(Ws)
            for cp in reversed(range(copies)):      # ends with copy 0: bf / cs belong to the copy the error check uses
                Wf[cp], bf, cs = ops.fold_layernorm(Ws[cp], bias, g, b)
        else:
            Wf, bf, cs = Ws, bias, None
        out2 = torch.zeros(32, Np, device="cuda")
        ops.fused_gemv_1trip(X, Wf[0], M, N, bias=bf, ln_colsum=cs, R1=R, relu=True, out=out2)
        err2 = float((out2[:M, :N].double() - ref).abs().max())
        lib.pfhip_dev_fused_gemv_1trip_bench.argtypes = [vp, ci, vp, ci, ctypes.c_size_t, ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.pfhip_dev_fused_gemv_1trip_bench(X.data_ptr(), K, Wf.data_ptr(), K, Np * K, copies, out2.data_ptr(), Np, bf.data_ptr(),
                                                      cs.data_ptr() if ln else None, R.data_ptr(), Np, M, N, K, 1, n,
                                                      torch.cuda.current_stream().cuda_stream)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0, rc
            ts.append(e0.elapsed_time(e1) / n * 1e3)
        t2 = float(np.median(ts[1:]))
        line += f"   | one trip: {t2:7.2f} us  err {err2:.1e}"
    print(line, flush=True)
