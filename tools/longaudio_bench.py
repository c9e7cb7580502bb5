"""Dev tool (GPU): BASELINE config C5 on ONE GPU — long files (default 8 x 600 s with seeded 0.5-s gaps every 7-25 s)
through FSMN-VAD (one pass per file) -> end-point detector -> length-sorted dynamic batches -> Paraformer-large.
Per-stage wall times and audio-seconds per second for one decoder worker; the 8-GPU case shards files over replicas."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
weights = importlib.import_module("asr_2pass_amd.weights")
pipeline = importlib.import_module("asr_2pass_amd.pipeline")
from conftest import synth_pcm
from test_gpu_pipeline import shape_vad_weights

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seconds = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rng = np.random.default_rng(20251114)

def make_file(i):
    parts, t = [], 0.0
    while t < seconds:
        dur = float(rng.uniform(7, 25))
        dur = min(dur, seconds - t)
        parts.append(synth_pcm(i, int(dur * 16000), rng))
        parts.append(np.zeros(16000, np.float32))      # 1.0-s gaps: longer than the 800-ms end-silence threshold
        t += dur + 1.0
    return np.concatenate(parts)[:seconds * 16000]

files = [make_file(i) for i in range(n_files)]
vman, vblob = shape_vad_weights(*weights.synth_vad_weights())
aman, ablob = weights.synth_weights(dict(weights.PARAFORMER_LARGE), seed=1234)
vad = pkg.FsmnVadHip().InitVad((vman, vblob))
asr = pkg.ParaformerHip().InitAsr((aman, ablob))
seg = pkg.E2EVadModelHost()
pipeline.infer_buffer(files[0][:16000 * 60], asr, vad, seg, batch_size=32)      # warm-up
import threading
workers = int(sys.argv[3]) if len(sys.argv) > 3 else n_files
MAXU = int(os.environ.get('MAXU', '96'))
asr.set_batching(int(os.environ.get('WAIT_US', '3000')) if workers > 1 else 0, MAXU)
stats = dict(vad=0.0, seg=0.0, asr=0.0, nseg=0, ntok=0)
lock = threading.Lock()
nxt = [0]

def worker():
    while True:
        with lock:
            i = nxt[0]; nxt[0] += 1
        if i >= n_files:
            return
        f = files[i]
        a = time.perf_counter()
        sil = vad.ForwardSil(f, is_final=True)
        b = time.perf_counter()
        segs = seg_objs[threading.get_ident()](sil, f[:400 + 160 * (len(sil) - 1)], True, False, 800, 60000, 0.9)
        c = time.perf_counter()
        frames = [(s * 16, min(e * 16, len(f))) for s, e in segs]
        order = sorted(range(len(frames)), key=lambda k: (frames[k][1] - frames[k][0], k))
        queue = [frames[k] for k in order]
        ntok = 0
        while queue:
            batch = pipeline.fetch_dynamic(queue, int(os.environ.get('BATCH', '32')))
            r = asr.forward_ids([f[s:e] for s, e in batch])
            ntok += sum(len(x) for x in r["ids"])
        d = time.perf_counter()
        with lock:
            stats["vad"] += b - a; stats["seg"] += c - b; stats["asr"] += d - c; stats["nseg"] += len(frames); stats["ntok"] += ntok

class SegMap(dict):
    def __missing__(self, k):
        self[k] = pkg.E2EVadModelHost()
        return self[k]
seg_objs = SegMap()
t0 = time.perf_counter()
ths = [threading.Thread(target=worker) for _ in range(workers)]
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0
audio = n_files * seconds
print(f"files {n_files} x {seconds}s  workers {workers}  segments {stats['nseg']}  tokens {stats['ntok']}  total {dt:.2f}s  xRT {audio/dt:.0f}  "
      f"(summed over workers) vad {stats['vad']:.3f}s  endpoint(host) {stats['seg']:.3f}s  asr {stats['asr']:.3f}s")

# ---- the same flow through the C++ handle-API mirror (FunOfflineInit / FunOfflineInferBuffer), one file repeated --------
if os.environ.get("CPP", "1") != "0":
    import subprocess, tempfile
    vad.close(); asr.close()
    with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None) as td:
        os.makedirs(os.path.join(td, "asr")); os.makedirs(os.path.join(td, "vad"))
        weights.save(os.path.join(td, "asr", "model.pfhip"), aman, ablob)
        weights.save(os.path.join(td, "vad", "vad.pfhip"), vman, vblob)
        np.clip(np.round(files[0] * 32768.0), -32768, 32767).astype("<i2").tofile(os.path.join(td, "long.pcm"))
        exe = os.path.join(ROOT, "asr-2pass_amd", "offline_infer")
        for th, rep in ((1, 4), (8, 2)):
            out = subprocess.run([exe, os.path.join(td, "asr"), os.path.join(td, "vad"), os.path.join(td, "long.pcm"), "32", str(th), str(rep)],
                                 capture_output=True, text=True, timeout=600)
            print(f"C++ FunOfflineInferBuffer, {th} thread(s):", out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-300:])
