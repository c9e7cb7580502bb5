"""Dev tool (GPU): N timestamp forwards of a ragged 32-utterance batch (contextual + timestamp Paraformer-large shapes) — the
persistent BLSTM's sentinel-ring exchange must never time out into the per-step form (blstm_fallbacks stays 0) and every run must
reproduce the first one bit for bit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from conftest import synth_pcm
pkg = ge.load_package()
import importlib
wt = importlib.import_module(pkg.__name__ + ".weights")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(3)
waves = [synth_pcm(i, int(16000 * (8 + 22 * rng.random())), rng) for i in range(32)]
cfg = dict(wt.PARAFORMER_LARGE, contextual=1, timestamp=1)
man, blob = wt.synth_weights(cfg, seed=1234)
h = pkg.ParaformerHip().InitAsr((man, blob))
hw = h.CompileHotwordEmbedding([list(rng.integers(2, 8000, int(rng.integers(2, 8)))) for _ in range(16)])
first = None
t0 = time.perf_counter()
for k in range(N):
    r = h.forward_ids(waves, hw_emb=hw, want_timestamps=True)
    if first is None:
        first = r
    else:
        for a, b in zip(first["us_alphas"], r["us_alphas"]):
            assert np.array_equal(a, b), k
        for a, b in zip(first["ids"], r["ids"]):
            assert list(a) == list(b), k
    if k % 10 == 9:
        print(f"{k + 1} forwards, {(time.perf_counter() - t0) / (k + 1) * 1e3:.1f} ms each, fallbacks "
              f"{h._lib.pfhip_debug_poke(h.handle, b'blstm_fallbacks', 0)}", flush=True)
assert h._lib.pfhip_debug_poke(h.handle, b"blstm_fallbacks", 0) == 0
print("ok")
h.close()
