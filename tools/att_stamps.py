"""Dev tool (GPU): phase timeline of attention_p3.hip from in-kernel s_memtime stamps (build: tools/x3_variant.sh attpst
"-DPFHIP_ATTP_STAMPS=1" attention_p3.hip; run with PFHIP_LIB=build/libpfhip_attpst.so).  Prints, per wave of workgroup (0, 0, 0), the
average cycles of S / softmax / PV / barrier wait / advance per tile."""
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
B, T, H, D = 32, 500, 4, 128
kv = ops.split_rows(torch.randn(B * T, 2 * H * D, device="cuda"))
q = torch.randn(B * T, H * D, device="cuda")
off = torch.arange(B, dtype=torch.int32, device="cuda") * T
ln = torch.full((B,), T, dtype=torch.int32, device="cuda")
for _ in range(20):
    O = ops.attention_kvplanes(q, kv, H * D, off, ln, off, ln, H, D ** -0.5)
torch.cuda.synchronize()
raw = O.cpu().numpy().view(np.uint64).reshape(-1)[:8 * 128].reshape(8, 128)
names = ["S", "softmax", "PV", "barrier", "advance"]
names_b = ["softmax", "PV", "S(next)", "barrier", "-"]
shift = os.environ.get("ATT_STAMPS_SHIFT") == "1"
for w in range(8):
    st = raw[w][:81].astype(np.int64)
    d = np.diff(st).reshape(16, 5)
    print(f"wave {w}: total {int(st[80] - st[0])} ticks; per tile " + "  ".join(f"{n} {d[2:-1, i].mean():7.0f}" for i, n in enumerate(names_b if shift and w >= 4 else names)) +
          f"   tile {d[2:].sum(1).mean():7.0f}")
    o = raw[w][100:107].astype(np.int64)
    rt = raw[w][110:112].astype(np.int64)
    print(f"        entry to after the output transpose: {o[5] - o[0]} cycles = {(rt[1] - rt[0]) / 100.0:.2f} us -> {(o[5] - o[0]) / ((rt[1] - rt[0]) * 10.0):.3f} GHz")
    print(f"        outside the loop: setup {o[1] - o[0]}, memory block + Q planes {o[2] - o[1]}, to barrier #-1 {o[3] - o[2]}, loop {o[4] - o[3]} (entry to end {o[4] - o[0]})")
print("first stamps of the eight waves (relative):", [int(raw[w][0] - raw[0][0]) for w in range(8)])
