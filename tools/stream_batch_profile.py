"""Dev tool (GPU): one batched streaming configuration, for rocprofv3 (kernel time vs wall time per 600-ms round)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from conftest import synth_pcm
pkg = ge.load_package()
import importlib
wt = importlib.import_module(pkg.__name__ + ".weights")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
man, blob = wt.synth_weights(dict(wt.PARAFORMER_LARGE), seed=1234)
model = pkg.ParaformerHip().InitAsr((man, blob))
rng = np.random.default_rng(1)
streams = [pkg.ParaformerOnlineHip(model) for _ in range(B)]
waves = [synth_pcm(i, 9600 * rounds, rng) for i in range(B)]
for k in range(2):
    pkg.ParaformerOnlineHip.forward_batch(streams, [w[k * 9600:(k + 1) * 9600] for w in waves], [False] * B)
t0 = time.perf_counter()
for k in range(2, rounds):
    pkg.ParaformerOnlineHip.forward_batch(streams, [w[k * 9600:(k + 1) * 9600] for w in waves], [False] * B)
dt = (time.perf_counter() - t0) / (rounds - 2)
print(f"{B} connections: {dt * 1e3:.2f} ms wall per round")
