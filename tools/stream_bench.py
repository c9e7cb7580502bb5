"""Dev tool (GPU): BASELINE config C3 — one 10-min stream fed in 600-ms (9600-sample) steps through
pfhip_stream_forward with the Paraformer-large-sized online model (random-init)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
weights = importlib.import_module("asr_2pass_amd.weights")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pcm

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = dict(weights.PARAFORMER_LARGE)
man, blob = weights.synth_weights(cfg, seed=1234)
model = pkg.ParaformerHip().InitAsr((man, blob))
rng = np.random.default_rng(20251114)
pcm = synth_pcm(0, 9600 * chunks, rng)
s = pkg.ParaformerOnlineHip(model)
for k in range(3):
    s.Forward(pcm[k * 9600:(k + 1) * 9600], input_finished=False)
s.Reset()
lat = []
ntok = 0
t0 = time.perf_counter()
for k in range(chunks):
    t1 = time.perf_counter()
    ids = s.Forward(pcm[k * 9600:(k + 1) * 9600], input_finished=(k == chunks - 1))
    lat.append(time.perf_counter() - t1)
    ntok += len(ids)
dt = time.perf_counter() - t0
lat = np.asarray(lat) * 1e3
print(f"chunks {chunks} tokens {ntok} total {dt:.3f} s  audio {chunks*0.6:.1f} s  xRT {chunks*0.6/dt:.1f}  "
      f"per-chunk ms: median {np.median(lat):.2f} p95 {np.percentile(lat,95):.2f} max {lat.max():.2f}")
