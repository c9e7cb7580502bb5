"""Dev tool (GPU): does running two half-batches concurrently (two model handles, two streams, two host threads) beat one
full batch?  32 x 30 s as 1 x 32, 2 x 16 and 4 x 8 concurrent forwards."""
import sys, os, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
wt = importlib.import_module(pkg.__name__ + ".weights")
from conftest import synth_pcm
cfg = dict(wt.PARAFORMER_LARGE)
man, blob = wt.synth_weights(cfg, seed=1234)
rng = np.random.default_rng(20251114)
utts = [synth_pcm(i, 30 * 16000, rng) for i in range(32)]
for n_ctx in (1, 2, 4):
    models = [pkg.ParaformerHip().InitAsr((man, blob)) for _ in range(n_ctx)]
    per = 32 // n_ctx
    def work(i, reps):
        for _ in range(reps):
            models[i].forward_ids(utts[i * per:(i + 1) * per])
    def run(reps):
        th = [threading.Thread(target=work, args=(i, reps)) for i in range(n_ctx)]
        t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        return (time.perf_counter() - t0) / reps
    run(2)
    dt = min(run(5), run(5))
    print(f"{n_ctx} concurrent forwards of {per} utterances: {dt * 1e3:.2f} ms per 32 x 30 s -> {960 / dt:.0f} audio-s/s", flush=True)
    for m in models:
        m.close()
