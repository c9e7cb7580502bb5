"""BASELINE config C4 on one GPU: hotword (contextual) + timestamp Paraformer-large, batch 32 x 30 s, random-init weights.
Prints ms per batch for: plain forward, + hotwords, + timestamp head (the persistent BLSTM)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from conftest import synth_pcm  # noqa: E402


def main():
    pkg = ge.load_package()
    import importlib
    wt = importlib.import_module(pkg.__name__ + ".weights")
    B, secs = 32, 30
    rng = np.random.default_rng(0)
    waves = [synth_pcm(i, 16000 * secs, rng) for i in range(B)]
    for name, over in (("plain", {}), ("contextual+timestamp", dict(contextual=1, timestamp=1))):
        cfg = dict(wt.PARAFORMER_LARGE, **over)
        man, blob = wt.synth_weights(cfg, seed=1234)
        h = pkg.ParaformerHip().InitAsr((man, blob))
        hw = None
        if over:
            hot = [list(rng.integers(2, 8000, int(rng.integers(2, 8)))) for _ in range(50)]
            hw = h.CompileHotwordEmbedding(hot)
        for ts in ((False, True) if over else (False,)):
            h.forward_ids(waves, hw_emb=hw, want_timestamps=ts)
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                r = h.forward_ids(waves, hw_emb=hw, want_timestamps=ts)
            dt = (time.perf_counter() - t0) / n
            print(f"{name:22s} timestamps={ts!s:5s}: {dt * 1e3:7.1f} ms per batch (host buffers in, ids out)  -> {B * secs / dt:8.0f} audio-s/s"
                  + (f"  us_len={len(r['us_alphas'][0])}" if ts else ""), flush=True)
        h.close()


if __name__ == "__main__":
    main()
