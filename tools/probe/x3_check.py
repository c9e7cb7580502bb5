import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
weights = importlib.import_module("asr_2pass_amd.weights")
from conftest import synth_pcm
from oracle import paraformer as P
cfg = dict(weights.PARAFORMER_LARGE)
man, blob = weights.synth_weights(cfg, seed=1234)
model = pkg.ParaformerHip().InitAsr((man, blob))
rng = np.random.default_rng(20251114)
utts = [synth_pcm(i, 480000, rng) for i in range(32)]
got = model.forward_ids(utts, want_logp=True)
W = P.Weights(man, blob)
for i in (0, 17):
    ref = P.forward_pcm(utts[i], W)
    print("utt", i, "ids equal", list(got["ids"][i]) == list(ref["ids"]), "max |dlogp|", float(np.abs(got["logp"][i] - ref["logp"]).max()), "tokens", len(ref["ids"]), flush=True)
