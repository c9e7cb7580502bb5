"""Dev tool (GPU): builds the small model directory of tests/test_gpu_contexts.py and runs the serve_threads harness on it."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
import importlib
weights_mod = importlib.import_module("asr_2pass_amd.weights")
ts = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = weights_mod.small_config(enc_layers=6, dec_layers=3, vocab=300, timestamp=ts)
man, blob = weights_mod.synth_weights(cfg, seed=21)
d = tempfile.mkdtemp()
weights_mod.save(os.path.join(d, "model.pfhip"), man, blob)
json.dump([f"<{i}>" for i in range(300)], open(os.path.join(d, "tokens.json"), "w"))
exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "serve_threads")
env = {k: v for k, v in os.environ.items() if not k.startswith("PFHIP_") or k in sys.argv[2:]}
out = subprocess.run([exe, d, "-", "16", "96", "2", "9", "1"], capture_output=True, text=True, env=env)
print(out.stdout); print(out.stderr)
