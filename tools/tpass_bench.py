"""Dev tool (GPU): BASELINE config C3 end to end through the C++ 2-pass handle API — N concurrent connections share one
TpassStream (Paraformer-large-sized offline + online models, FSMN-VAD, CT-Transformer; random-init weights, the VAD shaped
so that it follows the frame energy) and stream a synthetic file (speech-like bursts separated by 1.2-s silences) in 600-ms
pieces through `tpass_bench`.  Usage: python tools/tpass_bench.py [seconds=60] [connections=16,64,128] [punc=1]"""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
wt = importlib.import_module(pkg.__name__ + ".weights")
from conftest import synth_pcm
from test_gpu_pipeline import shape_vad_weights

seconds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
conns = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,64,128").split(",")]
use_punc = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
rng = np.random.default_rng(5)
parts, total, i = [], 0, 0
while total < seconds * 16000:
    sec = [4.0, 7.5, 2.2, 11.0, 5.3][i % 5]
    parts += [synth_pcm(i, int(sec * 16000), rng), np.zeros(int(1.2 * 16000), np.float32)]
    total += len(parts[-2]) + len(parts[-1]); i += 1
pcm = np.concatenate(parts)[:seconds * 16000]
d = tempfile.mkdtemp(prefix="tpass_bench_")
np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2").tofile(os.path.join(d, "stream.pcm"))
cfg = dict(wt.PARAFORMER_LARGE)
vocab = ["<blank>", "<s>", "</s>"] + [chr(0x4E00 + k) for k in range(cfg["vocab"] - 4)] + ["<unk>"]
t0 = time.time()
for name, seed in (("asr", 31), ("online", 32)):
    os.mkdir(os.path.join(d, name))
    man, blob = wt.synth_weights(cfg, seed=seed)
    wt.save(os.path.join(d, name, "model.pfhip"), man, blob)
    with open(os.path.join(d, name, "tokens.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
os.mkdir(os.path.join(d, "vad"))
vman, vblob = shape_vad_weights(*wt.synth_vad_weights())
wt.save(os.path.join(d, "vad", "vad.pfhip"), vman, vblob)
pdir = "-"
if use_punc:
    pdir = os.path.join(d, "punc_realtime")
    os.mkdir(pdir)
    pcfg = dict(wt.CT_TRANSFORMER)
    pman, pblob = wt.synth_punc_weights(pcfg)
    wt.save(os.path.join(pdir, "punc.pfhip"), pman, pblob)
    pv = ["<unk>"] + vocab[3:-1] + [f"t{k}" for k in range(pcfg["vocab"] - len(vocab) + 3)]
    with open(os.path.join(pdir, "tokens.json"), "w", encoding="utf-8") as f:
        json.dump(pv[:pcfg["vocab"]], f, ensure_ascii=False)
print(f"model directories written in {time.time() - t0:.1f} s ({d})", flush=True)
exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "tpass_bench")
# TPASS_VARIANTS: JSON list of {"punc": 0|1, "mode": 1|2, "env": {"PFHIP_STREAM_WAIT_US": "500", ...}} — default: one run as given
variants = json.loads(os.environ.get("TPASS_VARIANTS", "[{}]"))
for var in variants:
    env = dict(os.environ, **{k: str(v) for k, v in var.get("env", {}).items()})
    p = pdir if var.get("punc", 1 if use_punc else 0) and use_punc else "-"
    for n in conns:
        out = subprocess.run([exe, os.path.join(d, "asr"), os.path.join(d, "online"), os.path.join(d, "vad"), p,
                              os.path.join(d, "stream.pcm"), str(n), str(var.get("mode", 2))], capture_output=True, text=True,
                             timeout=900, env=env)
        print(json.dumps(var), out.stdout.strip() or out.stderr[-2000:], flush=True)
        if out.returncode != 0:
            print("rc", out.returncode, out.stderr[-2000:], flush=True)
import shutil
shutil.rmtree(d, ignore_errors=True)
