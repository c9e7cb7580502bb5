"""GPU: the in-process multi-GPU router (pfhip_create_group / PFHIP_DEVICES; SURVEY §8e "replicas only") —
ONE handle, one replica per listed device; offline calls go to the least-loaded replica, a new connection is pinned to the
replica with the fewest open streams.  The 1-GPU test box lists device 0 twice (two replicas on one GPU): routing is disjoint,
results are identical whichever replica serves a call, streams of one explicit batch may sit on different replicas."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import synth_pcm

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def models(pkg, weights_mod):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: the HIP path has no CPU fallback")
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=2, vocab=400)
    man, blob = weights_mod.synth_weights(cfg, seed=61)
    single = pkg.ParaformerHip().InitAsr((man, blob))
    group = pkg.ParaformerHip().InitAsr((man, blob), devices=[0, 0, 0])
    yield single, group
    single.close()
    group.close()


def test_offline_calls_are_spread_and_results_identical(models):
    single, group = models
    assert group._lib.pfhip_group_size(group.handle) == 3 and single._lib.pfhip_group_size(single.handle) == 1
    rng = np.random.default_rng(5)
    batches = [[synth_pcm(10 * k + i, int(rng.integers(16000, 16000 * 6)), rng) for i in range(1 + k % 3)] for k in range(9)]
    want = [single.forward_ids(b, want_logp=True) for b in batches]
    before = group.group_stats()
    # sequential calls: nothing in flight, ties go round-robin -> 3 calls per replica
    for b, w in zip(batches, want):
        got = group.forward_ids(b, want_logp=True)
        for i in range(len(b)):
            assert list(got["ids"][i]) == list(w["ids"][i])
            assert np.array_equal(got["logp"][i], w["logp"][i])            # same kernels, same batch: bit-identical on any replica
    after = group.group_stats()
    calls = [a - b for a, b in zip(after["calls"], before["calls"])]
    utts = [a - b for a, b in zip(after["utterances"], before["utterances"])]
    assert calls == [3, 3, 3] and sum(utts) == sum(len(b) for b in batches)
    assert after["devices"] == [0, 0, 0]
    # concurrent callers (the server's decoder threads): every replica takes part, every caller gets its own result
    res = [None] * len(batches)

    def work(k):
        res[k] = group.forward_ids(batches[k])
    for _ in range(3):
        ths = [threading.Thread(target=work, args=(k,)) for k in range(len(batches))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for k, w in enumerate(want):
            assert [list(x) for x in res[k]["ids"]] == [list(x) for x in w["ids"]]
    final = group.group_stats()
    assert all(c > 3 for c in [a - b for a, b in zip(final["calls"], before["calls"])])


def test_streams_are_pinned_and_batches_may_span_replicas(models, pkg):
    single, group = models
    rng = np.random.default_rng(6)
    waves = [synth_pcm(i, 9600 * 5 + 500 * i, rng) for i in range(6)]

    def feed_alone(model, w):
        s = pkg.ParaformerOnlineHip(model)
        steps = list(range(0, len(w), 9600))
        out = [s.Forward(w[a:a + 9600], input_finished=(j == len(steps) - 1)) for j, a in enumerate(steps)]
        s.close()
        return out
    want = [feed_alone(single, w) for w in waves]
    streams = [pkg.ParaformerOnlineHip(group) for _ in waves]
    assert group.group_stats()["open_streams"] == [2, 2, 2]                  # least-loaded placement
    got = [[] for _ in waves]
    for j in range(max(len(w) for w in want)):
        act = [i for i in range(len(waves)) if j < len(want[i])]
        res = pkg.ParaformerOnlineHip.forward_batch([streams[i] for i in act], [waves[i][9600 * j:9600 * (j + 1)] for i in act],
                                                    [j == len(want[i]) - 1 for i in act])
        for i, r in zip(act, res):
            got[i].append(r)
    assert got == want
    for s in streams[:3]:
        s.close()
    assert sorted(group.group_stats()["open_streams"]) == [1, 1, 1]
    for s in streams[3:]:
        s.close()


def test_env_var_builds_the_group_for_an_unchanged_caller(weights_mod, tmp_path):
    """PFHIP_DEVICES is read by pfhip_create*: a caller that knows nothing about groups (here: a fresh interpreter using
    the plain InitAsr path) gets one."""
    code = (
        "import sys, numpy as np, importlib\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "import __graft_entry__ as ge\n"
        "pkg = ge.load_package(); weights = importlib.import_module('asr_2pass_amd.weights')\n"
        "from conftest import synth_pcm\n"
        "man, blob = weights.synth_weights(weights.small_config(enc_layers=1, dec_layers=1, vocab=300), seed=3)\n"
        "m = pkg.ParaformerHip().InitAsr((man, blob))\n"
        "rng = np.random.default_rng(1)\n"
        "for k in range(4): m.forward_ids([synth_pcm(k, 32000, rng)])\n"
        "print('GROUP', pkg.load_lib().pfhip_group_size(m.handle), m.group_stats()['calls'])\n")
    env = dict(os.environ, PFHIP_DEVICES="0,0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GROUP 2 [2, 2]" in out.stdout


def test_c5_long_audio_workers_over_a_replica_group(pkg, weights_mod):
    """BASELINE configs[4] in miniature: 8 concurrent decoder workers, each transcribing a VAD-segmented long file through ONE
    shared handle that fronts several replicas (here two on device 0; on an 8-GPU node PFHIP_DEVICES=0,...,7): every worker's
    segments and ids equal the single-model, single-thread flow; both replicas served calls."""
    import importlib
    from test_gpu_pipeline import make_file, shape_vad_weights
    pipeline = importlib.import_module("asr_2pass_amd.pipeline")
    vman, vblob = shape_vad_weights(*weights_mod.synth_vad_weights())
    cfg = weights_mod.small_config(enc_layers=2, dec_layers=1, vocab=300)
    aman, ablob = weights_mod.synth_weights(cfg)
    files = [make_file(np.random.default_rng(100 + i)) for i in range(8)]
    vad = pkg.FsmnVadHip().InitVad((vman, vblob))
    single = pkg.ParaformerHip().InitAsr((aman, ablob))
    want = []
    for f in files:
        seg = pkg.E2EVadModelHost()
        want.append(pipeline.infer_buffer(f, single, vad, seg, batch_size=4, vad_max_len=60000))
        seg.close()
    single.close()
    group = pkg.ParaformerHip().InitAsr((aman, ablob), devices=[0, 0])
    group.set_batching(2000, 32)                   # the server's shape: decoder threads share the handle, calls are merged per replica
    got = [None] * 8
    vads = [pkg.FsmnVadHip().InitVad((vman, vblob)) for _ in range(8)]

    def worker(i):
        seg = pkg.E2EVadModelHost()
        got[i] = pipeline.infer_buffer(files[i], group, vads[i], seg, batch_size=4, vad_max_len=60000)
        seg.close()
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(8):
        assert got[i][1] == want[i][1], i                                   # segments
        assert [list(x) for x in got[i][0]] == [list(x) for x in want[i][0]], i      # ids per segment
    st = group.group_stats()
    assert min(st["calls"]) > 0 and sum(st["utterances"]) == sum(len(w[1]) for w in want)
    group.close(); vad.close()
    for v in vads:
        v.close()
