"""Dev tool (GPU): the full-size batch (32 x 30 s, Paraformer-large shapes) through the plane-image encoder path and through the
fp32-operand path (PFHIP_PLANES_MIN_ROWS out of reach), log-probs of both against the CPU restatement and against each other."""
import os, sys, subprocess, json
import numpy as np
UTTS = (3, 31, 0, 7, 12, 17, 20, 25)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import __graft_entry__ as ge
    pkg = ge.load_package()
    import importlib
    weights_mod = importlib.import_module("asr_2pass_amd.weights")
    from conftest import synth_pcm
    cfg = dict(weights_mod.PARAFORMER_LARGE)
    man, blob = weights_mod.synth_weights(cfg)
    model = pkg.ParaformerHip().InitAsr((man, blob))
    rng = np.random.default_rng(20251114)
    utts = [synth_pcm(i, 480000, rng) for i in range(32)]
    got = model.forward_ids(utts, want_logp=True)
    np.savez(sys.argv[1], **{f"logp{i}": got["logp"][i] for i in UTTS}, **{f"ids{i}": np.asarray(got["ids"][i]) for i in range(32)})
    if len(sys.argv) > 2:
        from oracle import paraformer as P
        W = P.Weights(man, blob)
        np.savez(sys.argv[2], **{f"logp{i}": P.forward_pcm(utts[i], W)["logp"] for i in UTTS})
    sys.exit(0)
subprocess.check_call([sys.executable, __file__, "/tmp/pl1.npz", "/tmp/ref.npz"])
subprocess.check_call([sys.executable, __file__, "/tmp/pl0.npz"], env=dict(os.environ, PFHIP_PLANES_MIN_ROWS="100000000"))
a, b, r = np.load("/tmp/pl1.npz"), np.load("/tmp/pl0.npz"), np.load("/tmp/ref.npz")
for i in UTTS:
    n = min(len(a[f"logp{i}"]), len(r[f"logp{i}"]))
    print(f"utt {i}: planes vs oracle {np.abs(a[f'logp{i}'][:n] - r[f'logp{i}'][:n]).max():.3e}  fp32-operand path vs oracle "
          f"{np.abs(b[f'logp{i}'][:n] - r[f'logp{i}'][:n]).max():.3e}  planes vs fp32-operand path {np.abs(a[f'logp{i}'][:n] - b[f'logp{i}'][:n]).max():.3e}")
print("ids equal on all 32:", all(np.array_equal(a[f"ids{i}"], b[f"ids{i}"]) for i in range(32)))
