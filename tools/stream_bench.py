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

# ---- many connections advancing together: pfhip_stream_forward_batch packs their windows into one forward ----------
s.close()
for B in (8, 32, 64, 128, 256):
    rounds = 30
    streams = [pkg.ParaformerOnlineHip(model) for _ in range(B)]
    waves = [synth_pcm(i, 9600 * rounds, rng) for i in range(B)]
    for k in range(2):
        pkg.ParaformerOnlineHip.forward_batch(streams, [w[k * 9600:(k + 1) * 9600] for w in waves], [False] * B)
    t0 = time.perf_counter()
    ntok = 0
    for k in range(2, rounds):
        res = pkg.ParaformerOnlineHip.forward_batch(streams, [w[k * 9600:(k + 1) * 9600] for w in waves], [k == rounds - 1] * B)
        ntok += sum(len(r) for r in res)
    dt = time.perf_counter() - t0
    n = rounds - 2
    print(f"{B:4d} connections: {dt / n * 1e3:7.2f} ms per 600-ms round  ({dt / n / B * 1e3:6.3f} ms per connection-chunk)  "
          f"aggregate xRT {B * n * 0.6 / dt:8.0f}  tokens {ntok}", flush=True)
    for x in streams:
        x.close()
