"""CPU: weight container, synthetic generator, bench sharding helpers; gloo world_size-2 rehearsal of the
multi-GPU timing protocol (replicas only, no data-path collective)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_manifest_layout(weights_mod):
    cfg = weights_mod.small_config()
    man = weights_mod.build_manifest(cfg)
    offs = sorted((v["offset"], int(np.prod(v["shape"])) * 4) for v in man["tensors"].values())
    for (o0, n0), (o1, _) in zip(offs, offs[1:]):
        assert o0 % 256 == 0 and o0 + n0 <= o1
    assert man["tensors"]["enc.0.qkv.w"]["shape"] == [1536, 560]
    assert man["tensors"]["enc.1.qkv.w"]["shape"] == [1536, 512]
    assert man["tensors"]["dec.out.w"]["shape"] == [cfg["vocab"], 512]
    json.dumps(man)


def test_full_size_parameter_count(weights_mod):
    man = weights_mod.build_manifest(dict(weights_mod.PARAFORMER_LARGE))
    n = sum(int(np.prod(v["shape"])) for v in man["tensors"].values())
    assert 200e6 < n < 240e6          # ~220 M parameters


def test_synth_is_seeded(weights_mod):
    cfg = weights_mod.small_config(enc_layers=1, dec_layers=0)
    _, a = weights_mod.synth_weights(cfg, seed=3)
    _, b = weights_mod.synth_weights(cfg, seed=3)
    _, c = weights_mod.synth_weights(cfg, seed=4)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_save_load_roundtrip(weights_mod, tmp_path):
    cfg = weights_mod.small_config(enc_layers=1, dec_layers=0, vocab=130)
    man, blob = weights_mod.synth_weights(cfg)
    weights_mod.save(str(tmp_path / "m"), man, blob)
    man2, blob2 = weights_mod.load(str(tmp_path / "m"))
    assert man2["config"] == man["config"] and np.array_equal(blob, blob2)


def test_two_rank_gloo_timing_protocol(tmp_path):
    """bench.py's N>1 protocol on CPU: barrier, each rank times its own replica, MAX over ranks,
    rank 0 reports units of ALL ranks / max time."""
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, time, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import torch, torch.distributed as dist\n"
        "import bench\n"
        "dist.init_process_group('gloo')\n"
        "rank, world = dist.get_rank(), dist.get_world_size()\n"
        "ids = bench.shard_utterance_ids(rank, world, 4)\n"
        "dt = 0.05 * (rank + 1)\n"
        "tmax = bench.max_over_ranks(dt, dist, torch.device('cpu'))\n"
        "allids = [None] * world\n"
        "dist.all_gather_object(allids, ids)\n"
        "if rank == 0: print(json.dumps({'tmax': tmax, 'ids': allids}))\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert abs(r["tmax"] - 0.10) < 1e-9
    flat = sorted(i for ids in r["ids"] for i in ids)
    assert flat == list(range(8))          # disjoint shards, every utterance exactly once
