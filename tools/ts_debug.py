"""Dev: stage-by-stage comparison of the timestamp head with the oracle (one utterance)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from conftest import synth_pcm
from oracle import paraformer as P
import importlib
pkg = ge.load_package(); wt = importlib.import_module(pkg.__name__ + ".weights")
cfg = wt.small_config(timestamp=1); man, blob = wt.synth_weights(cfg, seed=77); W = P.Weights(man, blob)
h = pkg.ParaformerHip().InitAsr((man, blob))
rng = np.random.default_rng(3)
waves = [synth_pcm(i, n, rng) for i, n in enumerate([16000 * 4, 16000 * 2 + 333, 16000 * 7, 9000])][: int(sys.argv[1]) if len(sys.argv) > 1 else 4]
r = h.forward_ids(waves, want_timestamps=True)
M = int(sum(r["n_frames"])); d = 512
up = h.get_tensor("ts_up", 3 * M * d).reshape(3 * M, d)
gx = h.get_tensor("ts_gx", 3 * M * 8 * d).reshape(3 * M, 8 * d)
y = h.get_tensor("ts_y", 3 * M * 2 * d).reshape(3 * M, 2 * d)
o = 0
for b, w in enumerate(waves):
    ref = P.forward_pcm(w, W); enc = ref["enc"]; T = enc.shape[0]
    wtw = W["pred.up.w"]; u = np.zeros((T, 3, d), np.float32)
    for j in range(3): u[:, j, :] = enc @ wtw[:, :, j] + W["pred.up.b"]
    u = u.reshape(3 * T, d)
    print(b, "T", T, "up diff", np.abs(up[o:o + 3 * T] - u).max())
    gxr = np.concatenate([u @ W["pred.blstm.w_ih"].T + W["pred.blstm.b_ih"] + W["pred.blstm.b_hh"],
                          u @ W["pred.blstm.w_ih_r"].T + W["pred.blstm.b_ih_r"] + W["pred.blstm.b_hh_r"]], 1)
    print("   gx diff", np.abs(gx[o:o + 3 * T] - gxr).max())
    hf = P._lstm_dir(u, W["pred.blstm.w_ih"], W["pred.blstm.w_hh"], W["pred.blstm.b_ih"], W["pred.blstm.b_hh"], False)
    hb = P._lstm_dir(u, W["pred.blstm.w_ih_r"], W["pred.blstm.w_hh_r"], W["pred.blstm.b_ih_r"], W["pred.blstm.b_hh_r"], True)
    df = np.abs(y[o:o + 3 * T, :d] - hf); db = np.abs(y[o:o + 3 * T, d:] - hb)
    print("   y fwd diff max", df.max(), "at frame", int(df.max(1).argmax()), "unit", int(df.max(0).argmax()), " per-frame max (first 8):", df.max(1)[:8])
    print("   y bwd diff max", db.max(), "at frame", int(db.max(1).argmax()), "unit", int(db.max(0).argmax()), " per-frame max (last 8):", db.max(1)[-8:])
    bad = np.where(df.max(0) > 1e-5)[0]; print("   fwd units > 1e-5:", bad[:40], len(bad))
    a_ref, p_ref = P.timestamp_head(enc, ref["token_num"], W)
    print("   alphas diff", np.abs(r["us_alphas"][b] - a_ref).max())
    o += 3 * T
