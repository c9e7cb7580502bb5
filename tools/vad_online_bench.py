"""Dev tool (GPU): cost of one online-VAD call (600 ms of audio) per connection."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from conftest import synth_pcm
pkg = ge.load_package()
import importlib
wt = importlib.import_module(pkg.__name__ + ".weights")
man, blob = wt.synth_vad_weights()
vad = pkg.FsmnVadHip().InitVad((man, blob))
rng = np.random.default_rng(0)
pcm = synth_pcm(0, 9600 * 60, rng)
on = pkg.FsmnVadOnlineHip(vad)
for k in range(5):
    on.Infer(pcm[k * 9600:(k + 1) * 9600], False)
t0 = time.perf_counter()
n = 50
for k in range(5, 5 + n):
    on.Infer(pcm[k * 9600:(k + 1) * 9600], False)
dt = (time.perf_counter() - t0) / n
print(f"online VAD: {dt * 1e3:.3f} ms per 600-ms call per connection ({0.6 / dt:.0f} x real time)")
on.close()
for nconn in [int(x) for x in os.environ.get("NCONN", "16,128,512").split(",")]:
    ss = [pkg.FsmnVadOnlineHip(vad) for _ in range(nconn)]
    offs = rng.integers(0, 20, nconn)
    def round_(k):
        return pkg.FsmnVadOnlineHip.InferScoresBatch(ss, [pcm[(o + k) * 9600:(o + k + 1) * 9600] for o in offs], [False] * nconn)
    for k in range(3):
        round_(k)
    lib = ss[0]._lib
    real = lib.pfhip_vad_stream_infer_batch
    spent = [0.0]
    def timed(*a):
        t = time.perf_counter(); r = real(*a); spent[0] += time.perf_counter() - t; return r
    lib.__dict__["pfhip_vad_stream_infer_batch"] = timed
    t0 = time.perf_counter()
    for k in range(3, 23):
        round_(k)
    dt = (time.perf_counter() - t0) / 20
    lib.__dict__["pfhip_vad_stream_infer_batch"] = real
    print(f"  inside the C call: {spent[0] / 20 * 1e3:.3f} ms per round")
    print(f"online VAD, {nconn} connections per batched call: {dt * 1e3:.3f} ms per round "
          f"({dt * 1e6 / nconn:.1f} us per connection, {0.6 * nconn / dt:.0f} x real time)")
    for x in ss:
        x.close()
vad.close()
