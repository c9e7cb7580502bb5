"""Dev tool (GPU): offline forward latency over batch shapes (host buffers in, ids out), Paraformer-large-sized random weights.
C1 of BASELINE.json is the single 5-s utterance; the stock server batches at most ~10 segments of one request."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from conftest import synth_pcm
pkg = ge.load_package()
import importlib
wt = importlib.import_module(pkg.__name__ + ".weights")
man, blob = wt.synth_weights(dict(wt.PARAFORMER_LARGE), seed=1234)
h = pkg.ParaformerHip().InitAsr((man, blob))
rng = np.random.default_rng(0)
for B, secs in [(1, 5), (1, 30), (2, 30), (4, 30), (8, 30), (16, 30), (32, 30), (64, 30), (10, 12)]:
    waves = [synth_pcm(i, int(16000 * secs), rng) for i in range(B)]
    h.forward_ids(waves)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        h.forward_ids(waves)
    dt = (time.perf_counter() - t0) / n
    print(f"batch {B:3d} x {secs:4.0f} s: {dt * 1e3:8.2f} ms  -> {B * secs / dt:8.0f} audio-s/s", flush=True)
h.close()
