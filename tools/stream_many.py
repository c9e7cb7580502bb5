"""Dev tool (GPU): B connections advancing together through pfhip_stream_forward_batch (BASELINE config C3, the throughput side),
Paraformer-large-sized random-init model — the workload to put under rocprofv3:
    python3 tools/stream_many.py [B=128] [rounds=14]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 14
warm = 2
man, blob = weights.synth_weights(dict(weights.PARAFORMER_LARGE), seed=1234)
NR = int(os.environ.get("REPLICAS", "1"))      # > 1: one handle fronting NR replicas on device 0; the batch call spans them concurrently
model = pkg.ParaformerHip().InitAsr((man, blob), devices=[0] * NR) if NR > 1 else pkg.ParaformerHip().InitAsr((man, blob))
rng = np.random.default_rng(20251114)
streams = [pkg.ParaformerOnlineHip(model) for _ in range(B)]
waves = [synth_pcm(1000 + i, 9600 * (rounds + warm), rng) for i in range(B)]
tok = 0
for k in range(warm + rounds):
    if k == warm:
        t0 = time.perf_counter()
    res = pkg.ParaformerOnlineHip.forward_batch(streams, [w[k * 9600:(k + 1) * 9600] for w in waves], [False] * B)
    if k >= warm:
        tok += sum(len(r) for r in res)
dt = (time.perf_counter() - t0) / rounds
print(f"replicas {NR} B {B} rounds {rounds} tokens {tok}: {1e3 * dt:.2f} ms per round = {B * 0.6 / dt:.0f} x real time")
