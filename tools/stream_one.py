"""Dev tool (GPU): ONE connection fed 600-ms chunks through pfhip_stream_forward with the Paraformer-large-sized model
(random-init) — the latency path of BASELINE config C3.  Prints ms per chunk; run under rocprofv3 --kernel-trace --stats for
the per-kernel picture (profiles/r02/stream_one_*)."""
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

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 100
man, blob = weights.synth_weights(dict(weights.PARAFORMER_LARGE), seed=1234)
model = pkg.ParaformerHip().InitAsr((man, blob))
rng = np.random.default_rng(20251114)
pcm = synth_pcm(0, 9600 * (chunks + 5), rng)
s = pkg.ParaformerOnlineHip(model)
for k in range(5):
    s.Forward(pcm[k * 9600:(k + 1) * 9600], input_finished=False)
lat, ntok = [], 0
for k in range(5, chunks + 5):
    t1 = time.perf_counter()
    ntok += len(s.Forward(pcm[k * 9600:(k + 1) * 9600], input_finished=False))
    lat.append(time.perf_counter() - t1)
lat = np.asarray(lat) * 1e3
print(f"chunks {chunks} tokens {ntok} per-chunk ms: median {np.median(lat):.3f} p95 {np.percentile(lat, 95):.3f} min {lat.min():.3f}")
