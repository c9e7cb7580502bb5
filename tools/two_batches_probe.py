"""Dev probe (GPU): do two FULL 32 x 30 s batches in flight at once (two model replicas on one device, one host thread and one
stream each) beat the same number of batches run back to back?  One launch of the headline workload already fills the chip, so
the only thing to win is one batch's bandwidth-bound phases (epilogues, FSMN prologue, what is left of LayerNorm) under the
other's matrix-core phases.   python3 tools/two_batches_probe.py [steps=20]"""
import sys, os, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
weights = importlib.import_module("asr_2pass_amd.weights")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_pcm

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, n = 32, 30 * 16000
man, blob = weights.synth_weights(dict(weights.PARAFORMER_LARGE), seed=1234)
NR = int(os.environ.get('REPLICAS', '2'))
models = [pkg.ParaformerHip().InitAsr((man, blob)) for _ in range(NR)]
rng = np.random.default_rng(20251114)
utts = [synth_pcm(i, n, rng) for i in range(B)]
d_pcm = torch.from_numpy(np.concatenate(utts)).cuda()
sample_off = np.arange(B, dtype=np.int64) * n
n_samples = np.full(B, n, np.int32)
max_tokens = n // 960 + 2
streams = [torch.cuda.Stream() for _ in range(NR)]


def run(i, k):
    for _ in range(k):
        models[i].enqueue_device(d_pcm.data_ptr(), sample_off, n_samples, streams[i].cuda_stream)
        models[i].fetch(B, max_tokens)


for i in range(NR):
    run(i, 2)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(0, steps)
    torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(i, steps // NR)) for i in range(NR)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t_par = time.perf_counter() - t0
    print(f"{steps} batches back to back: {1e3 * t_seq / steps:.2f} ms per batch; {NR} in flight: {1e3 * t_par / (steps // NR * NR):.2f} ms per batch "
          f"({B * 30 * steps / t_seq:.0f} vs {B * 30 * (steps // NR * NR) / t_par:.0f} audio-s/s)", flush=True)

# ---- the same through the product's router: ONE handle fronting NR replicas on this device (pfhip_create_group, what
# PFHIP_DEVICES=0,0,0 builds), host float** buffers in (the reference's Model::Forward boundary), NR caller threads
for m_ in models:
    m_.close()
grp = pkg.ParaformerHip().InitAsr((man, blob), devices=[0] * NR)
one = pkg.ParaformerHip().InitAsr((man, blob))


def run_host(model, k):
    for _ in range(k):
        model.forward_ids(utts, max_tokens=max_tokens)


run_host(grp, NR); run_host(one, 2)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_host(one, steps)
    torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=run_host, args=(grp, steps // NR)) for i in range(NR)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t_par = time.perf_counter() - t0
    print(f"host buffers, one handle: back to back {1e3 * t_seq / steps:.2f} ms per batch; group of {NR} replicas, {NR} caller threads: "
          f"{1e3 * t_par / (steps // NR * NR):.2f} ms per batch   replica calls {grp.group_stats()['calls']}", flush=True)
