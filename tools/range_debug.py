"""Dev probe for the range guard: HIP vs oracle on a model whose layer-0 output is scaled by 2^k."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
wt = importlib.import_module("asr_2pass_amd.weights")
from oracle import paraformer as P
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import synth_pcm

k = float(sys.argv[1]) if len(sys.argv) > 1 else 18
cfg = wt.small_config(enc_layers=3, dec_layers=1, vocab=257)
man, blob = wt.synth_weights(cfg, seed=61)
def view(name):
    meta = man["tensors"][name]; n = int(np.prod(meta["shape"]))
    return blob[meta["offset"] // 4: meta["offset"] // 4 + n].reshape(meta["shape"])
sc = np.float32(2.0 ** k)
for name in ("enc.0.ffn2.w", "enc.0.ffn2.b", "enc.0.out.w", "enc.0.out.b"):
    view(name)[...] *= sc
if len(sys.argv) > 2:
    view("enc.0.qkv.w")[2 * 512:] *= sc
    view("enc.0.qkv.b")[2 * 512:] *= sc
W = P.Weights(man, blob)
rng = np.random.default_rng(3)
utts = [synth_pcm(i, 16000 * 30 + 97 * i, rng) for i in range(8)]
ref = P.forward_pcm(utts[0], W)
print("oracle fires", ref["emb"].shape[0], "token_num", ref["token_num"], "enc rms", float(np.sqrt((ref["enc"] ** 2).mean())))
model = pkg.ParaformerHip().InitAsr((man, blob))
for n in (1, 2, 8):
    got = model.forward_ids(utts[:n], want_logp=True)
    err = float(np.abs(got["logp"][0][:min(got["n_fires"][0], ref["logp"].shape[0])] - ref["logp"][:min(got["n_fires"][0], ref["logp"].shape[0])]).max()) if got["n_fires"][0] else -1
    enc = model.get_tensor("enc", 4096 * 512)[:500 * 512].reshape(500, 512)
    print("batch", n, "fires", int(got["n_fires"][0]), "fallbacks", model.debug_poke("range_fallbacks"), "planes", model.debug_poke("plane_forwards"), "bound", model.debug_poke("static_bound"), "always", model.debug_poke("always_exact"), "logp err", err,
          "enc err", float(np.abs(enc - ref["enc"][:500]).max()) if enc is not None else None)
