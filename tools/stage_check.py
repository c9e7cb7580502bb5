"""Dev tool (GPU): per-stage comparison of the HIP path against the oracle on a small config."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
weights = importlib.import_module("asr_2pass_amd.weights")
from oracle import paraformer as P, frontend as fe

cfg = weights.small_config()
man, blob = weights.synth_weights(cfg)
W = P.Weights(man, blob)
rng = np.random.default_rng(20251114)
utts = []
for i, n in enumerate((16000 * 2, 16000 * 5 + 123, 399, 16000 * 3)):
    t = np.arange(n) / 16000.0
    f0 = 110 * 2 ** (i / 12)
    pcm = np.clip(np.round(8000 * (0.6 * np.sin(2 * np.pi * f0 * t) + 0.4 * rng.standard_normal(n))), -32768, 32767)
    utts.append((pcm / 32768.0).astype(np.float32))
m = pkg.ParaformerHip().InitAsr((man, blob))
feats = m.extract_feats(utts)
for b, u in enumerate(utts):
    ref = fe.extract_feats(u, W["cmvn.mean"], W["cmvn.istd"])
    print("feats", b, feats[b].shape, ref.shape, np.abs(feats[b] - ref).max() if ref.size else 0.0)
t0 = time.time()
got = m.forward_ids(utts, want_logp=True)
print("forward", time.time() - t0, got["token_num"], got["n_fires"], got["n_frames"])
M = int(got["n_frames"].sum()); ML = int(got["n_fires"].sum())
enc = m.get_tensor("enc", M * 512).reshape(M, 512)
alphas = m.get_tensor("alphas", M)
emb = m.get_tensor("emb", max(ML, 1) * 512).reshape(-1, 512)
ro = 0; to = 0
for b, u in enumerate(utts):
    r = P.forward_pcm(u, W)
    T = r["feats"].shape[0]
    if T == 0:
        print(b, "empty", got["token_num"][b]); continue
    L = r["emb"].shape[0]
    print(b, "T", T, "enc", np.abs(enc[ro:ro + T] - r["enc"]).max(), "alphas", np.abs(alphas[ro:ro + T] - r["alphas"][:T]).max(),
          "fires", got["n_fires"][b], L, "token_num", got["token_num"][b], r["token_num"])
    if L == got["n_fires"][b]:
        print("   emb", np.abs(emb[to:to + L] - r["emb"]).max(), "logp", np.abs(got["logp"][b] - r["logp"]).max(),
              "ids equal", list(got["ids"][b]) == list(r["ids"]))
    ro += T; to += int(got["n_fires"][b])
