"""Dev tool (GPU): N timestamp forwards of a ragged 32-utterance batch (contextual + timestamp Paraformer-large shapes) — the
persistent BLSTM's sentinel-ring exchange must never time out into the per-step form (blstm_fallbacks stays 0) and every run must
reproduce the first one bit for bit.  With contexts > 1 the forwards come from that many threads on as many execution contexts of the
one handle (pfhip_set_inflight): their persistent recurrences queue on one stream of the device.
    python tools/blstm_soak.py [forwards=60] [contexts=1]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from conftest import synth_pcm
pkg = ge.load_package()
import importlib
wt = importlib.import_module(pkg.__name__ + ".weights")
import threading
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
CTX = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(3)
waves = [synth_pcm(i, int(16000 * (8 + 22 * rng.random())), rng) for i in range(32)]
cfg = dict(wt.PARAFORMER_LARGE, contextual=1, timestamp=1)
man, blob = wt.synth_weights(cfg, seed=1234)
h = pkg.ParaformerHip().InitAsr((man, blob))
hw = h.CompileHotwordEmbedding([list(rng.integers(2, 8000, int(rng.integers(2, 8)))) for _ in range(16)])
if CTX > 1:
    h.set_inflight(CTX)
first = h.forward_ids(waves, hw_emb=hw, want_timestamps=True)
done = [0]
lock = threading.Lock()
t0 = time.perf_counter()


def run(n):
    for _ in range(n):
        r = h.forward_ids(waves, hw_emb=hw, want_timestamps=True)
        for a, b in zip(first["us_alphas"], r["us_alphas"]):
            assert np.array_equal(a, b)
        for a, b in zip(first["us_peaks"], r["us_peaks"]):
            assert np.array_equal(a, b)
        for a, b in zip(first["ids"], r["ids"]):
            assert list(a) == list(b)
        with lock:
            done[0] += 1
            if done[0] % 10 == 0:
                print(f"{done[0]} forwards, {(time.perf_counter() - t0) / done[0] * 1e3:.1f} ms each, fallbacks "
                      f"{h._lib.pfhip_debug_poke(h.handle, b'blstm_fallbacks', 0)}", flush=True)


threads = [threading.Thread(target=run, args=(N // CTX,)) for _ in range(CTX)]
for t in threads:
    t.start()
for t in threads:
    t.join()
dt = time.perf_counter() - t0
fallbacks = sum(c for c in [h._lib.pfhip_debug_poke(h.handle, b"blstm_fallbacks", 0)])
print(f"contexts {CTX}: {done[0]} forwards in {dt:.2f} s = {dt / max(1, done[0]) * 1e3:.1f} ms per 32-utterance batch; fallbacks {fallbacks}")
assert done[0] == (N // CTX) * CTX
assert fallbacks == 0
print("ok")
h.close()
