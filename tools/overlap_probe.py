"""Dev tool (GPU): does running the batch as K concurrent sub-batches (K model replicas, K streams, K host
threads) beat one batch of 32 on one stream?"""
import sys, os, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
weights = importlib.import_module("asr_2pass_amd.weights")
import bench

cfg = dict(weights.PARAFORMER_LARGE)
man, blob = weights.synth_weights(cfg, seed=1234)
n = 30 * 16000
rng = np.random.default_rng(bench.SEED_PCM)
utts = [bench.synth_pcm(i, n, rng) for i in range(32)]
d_pcm = torch.from_numpy(np.concatenate(utts)).cuda()
max_tokens = n // 960 + 2
for K in (1, 2, 4):
    models = [pkg.ParaformerHip().InitAsr((man, blob)) for _ in range(K)]
    per = 32 // K
    streams = [torch.cuda.Stream() for _ in range(K)]
    def work(k):
        so = (np.arange(per, dtype=np.int64) + k * per) * n
        ns = np.full(per, n, np.int32)
        models[k].enqueue_device(d_pcm.data_ptr(), so, ns, streams[k].cuda_stream)
        return models[k].fetch(per, max_tokens)
    def step():
        ths = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        for t in ths: t.start()
        for t in ths: t.join()
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"K={K}: {dt*1e3:.2f} ms/step  {960/dt:.0f} audio-s/s", flush=True)
    for m in models: m.close()
