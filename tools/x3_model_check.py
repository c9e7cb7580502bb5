"""Dev tool (GPU): the model-level effect of the GEMM form — max |enc - oracle| and max |logp - oracle| on the strong-LayerNorm
small model (tests/test_gpu_forward.py::test_layernorm_folded_...) for the form the environment selects (PFHIP_GEMM_X3=0/1).
    N_UTTS=5 python tools/x3_model_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from conftest import synth_pcm
from oracle import paraformer as P
import importlib
pkg = ge.load_package()
wt = importlib.import_module("asr_2pass_amd.weights")
n_utts = int(os.environ.get("N_UTTS", "5"))
cfg = wt.small_config(enc_layers=4, dec_layers=2, vocab=700)
man, blob = wt.synth_weights(cfg, seed=77)
rng = np.random.default_rng(5)
if os.environ.get("STRONG", "1") == "1":
    for name, t in man["tensors"].items():
        if "norm" in name and (name.endswith(".g") or name.endswith(".b")):
            o, n = t["offset"] // 4, int(np.prod(t["shape"]))
            blob[o:o + n] = (rng.uniform(0.5, 1.5, n) if name.endswith(".g") else rng.uniform(-0.5, 0.5, n)).astype(np.float32)
model = pkg.ParaformerHip().InitAsr((man, blob))
W = P.Weights(man, blob)
utts = [synth_pcm(i, 480000 - 1234 * (i % 9), rng) for i in range(n_utts)]
got = model.forward_ids(utts, want_logp=True)
enc = model.get_tensor("enc", int(sum(got["n_frames"])) * 512).reshape(-1, 512)
o = 0
for i, u in enumerate(utts):
    T = int(got["n_frames"][i])
    if i in (0, n_utts - 1):
        ref = P.forward_pcm(u, W)
        e_enc = np.abs(enc[o:o + T] - ref["enc"]).max()
        L = min(len(got["logp"][i]), len(ref["logp"]))
        e_lp = np.abs(got["logp"][i][:L] - ref["logp"][:L])
        same = [int(a) == int(b) for a, b in zip(got["ids"][i], ref["ids"])]
        top2 = np.sort(ref["logp"], axis=1)[:, -2:]
        gap = (top2[:, 1] - top2[:, 0])
        print(f"X3={os.environ.get('PFHIP_GEMM_X3', '1')} utt {i}: rows {T} enc err {e_enc:.2e}  logp err max {e_lp.max():.2e} (row {int(e_lp.max(1).argmax())})  "
              f"ids equal {all(same)} ({sum(same)}/{len(same)})  smallest top-2 gap of the oracle {gap.min():.2e} at row {int(gap.argmin())}", flush=True)
    o += T
