#!/usr/bin/env python3
"""bench.py — audio-seconds per wall-second of the Paraformer-large offline forward on MI355X.

Workload (BASELINE.json configs[1]): batch = 32 x 30 s synthetic 16 kHz utterances per GPU, weights
random-init with the Paraformer-large architecture (no network for the ModelScope files).  One "step"
= one pass of the hot path (PCM resident in HBM -> fbank/LFR/CMVN -> 50-layer SAN-M encoder ->
CIF predictor -> 16-layer decoder -> log-softmax/argmax -> token ids back on the host) over one batch.

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver through torch.distributed.run, one rank per GPU; utterance batches
are sharded over ranks as independent replicas (no data-path collective), the only
communication is the timing barrier / max-over-ranks.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 32
SECONDS = 30
SR = 16000
SEED_PCM = 20251114
GEMM_ONLY_MASK = 0x101      # bit 0 = gemm class; bit 8 keeps the value != 1 (1 means 'all classes')
F32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (the kernel used for small launches)
BF16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense BF16 MFMA peak
X6_MFMAS_PER_BLOCK = 6              # gemm_x6.hip: six bf16 MFMAs per fp32 32x32x16 block (exact 3-way operand split; PFHIP_GEMM_X3=0)
X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / X6_MFMAS_PER_BLOCK      # 416.7 TFLOP/s of fp32-equivalent work
X3_MFMAS_PER_BLOCK = 3              # gemm_x3.hip (default): three fp16 MFMAs per block (two fp16 planes per operand, 22-23 bits)
X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / X3_MFMAS_PER_BLOCK      # 833.3 TFLOP/s of fp32-equivalent work (FP16 dense peak = BF16's)


def synth_pcm(index: int, n: int, rng) -> np.ndarray:
    """s16 = round(8000*(0.6 sin(2 pi f_i t) + 0.4 N(0,1))), f_i = 110*2^((i mod 24)/12)."""
    t = np.arange(n, dtype=np.float64) / SR
    f = 110.0 * 2.0 ** ((index % 24) / 12.0)
    x = 8000.0 * (0.6 * np.sin(2 * np.pi * f * t) + 0.4 * rng.standard_normal(n))
    return (np.clip(np.round(x), -32768, 32767) / 32768.0).astype(np.float32)


def shard_utterance_ids(rank: int, world: int, batch: int):
    """Weak scaling: rank r owns the global utterance indices [r*batch, (r+1)*batch) — disjoint shards,
    independent replicas, no data-path collective."""
    return list(range(rank * batch, (rank + 1) * batch))


def max_over_ranks(dt: float, dist, device):
    """The driver's contract: time = MAX over ranks of the barrier-bracketed timed region."""
    import torch
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def pmc_traffic():
    """HBM-side bytes per GEMM launch from the committed PMC passes (profiles/rNN/pmc_hbm_traffic*.json: FETCH_SIZE x2 +
    WRITE_SIZE, collected with rocprofv3 in separate --pmc runs of this same command); PMC counters cannot be read from
    inside the timed run, so this is the last PROFILED value — returned with the file it came from (`traffic_source`) so
    that nobody reads it as measured in this run — or (None, None) when no file is present."""
    for rel in ("r04/pmc_hbm_traffic.json", "r03/pmc_hbm_traffic.json", "r02/pmc_hbm_traffic.json", "r01/pmc_hbm_traffic_e.json", "r01/pmc_hbm_traffic_d.json", "r01/pmc_hbm_traffic_c.json",
                "r01/pmc_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", rel)) as f:
                return json.load(f)["gemm_avg_bytes_per_launch"], "profiles/" + rel + " (rocprofv3 --pmc passes of an earlier run; not measured in this run)"
        except Exception:
            continue
    return None, None


def streaming_block(pkg, model, weight_bytes, rng):
    """BASELINE.json configs[2] (C3) beside the headline: ms per 600-ms chunk of the chunk-streaming path with the same
    Paraformer-large-sized weights, for ONE connection (latency path: pfhip_stream_forward) and for 128 connections advancing
    together (pfhip_stream_forward_batch).  A chunk (or a round of chunks) reads every weight once, so the HBM roofline of a
    chunk is weight_bytes / 8 TB/s; `frac_hbm` = that time / measured time."""
    out = {"chunk_ms_audio": 600, "weight_bytes_per_chunk": weight_bytes, "hbm_peak_GBps": 8000.0}
    for B, rounds, warm in ((1, 60, 5), (128, 14, 2)):
        streams = [pkg.ParaformerOnlineHip(model) for _ in range(B)]
        waves = [synth_pcm(1000 + i, 9600 * (rounds + warm), rng) for i in range(B)]
        tok = 0

        def feed(k):
            if B == 1:
                return [streams[0].Forward(waves[0][k * 9600:(k + 1) * 9600], input_finished=False)]
            return pkg.ParaformerOnlineHip.forward_batch(streams, [w[k * 9600:(k + 1) * 9600] for w in waves], [False] * B)
        for k in range(warm):
            feed(k)
        t0 = time.perf_counter()
        for k in range(warm, warm + rounds):
            tok += sum(len(r) for r in feed(k))
        dt = (time.perf_counter() - t0) / rounds
        for x in streams:
            x.close()
        key = "one_connection" if B == 1 else f"{B}_connections"
        out[key] = {"ms_per_chunk" if B == 1 else "ms_per_round": 1e3 * dt, "x_real_time": B * 0.6 / dt, "tokens": tok,
                    "achieved_GBps": weight_bytes / dt / 1e9, "frac_hbm": weight_bytes / dt / 8e12}
    return out


def c4_block(pkg, weights, utts, seconds, n_fly):
    """BASELINE.json configs[3] (C4): the contextual (hotword) + timestamp Paraformer-large — bias decoder in the last layer, 16
    hotwords compiled on the device, CifPredictorV3 upsampling head with its BLSTM — batch = 32 x 30 s, host buffers in, ids +
    upsampled alphas / peaks out (what TimestampOnnx consumes)."""
    cfg = dict(weights.PARAFORMER_LARGE, contextual=1, timestamp=1)
    man, blob = weights.synth_weights(cfg, seed=1234)
    h = pkg.ParaformerHip().InitAsr((man, blob))
    rng = np.random.default_rng(SEED_PCM + 4)
    hot = [list(rng.integers(2, 8000, int(rng.integers(2, 7)))) for _ in range(16)]           # H = 16, lengths 2-6
    hw = h.CompileHotwordEmbedding(hot)
    h.forward_ids(utts, hw_emb=hw, want_timestamps=True)
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        r = h.forward_ids(utts, hw_emb=hw, want_timestamps=True)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        h.forward_ids(utts, hw_emb=hw)
    dt_no_ts = (time.perf_counter() - t0) / n
    out = {"workload": f"contextual + timestamp Paraformer-large, batch={len(utts)} x {seconds} s, H=16 hotwords (BASELINE.json configs[3])",
           "ms_per_batch": 1e3 * dt, "audio_s_per_s": len(utts) * seconds / dt, "ms_per_batch_without_timestamp_head": 1e3 * dt_no_ts,
           "tokens": int(sum(len(x) for x in r["ids"])), "us_len": int(len(r["us_alphas"][0])),
           "boundary": "host float** buffers in; ids, us_alphas, us_cif_peak out"}
    if n_fly > 1:                   # the same batches from n_fly threads on the one handle (contexts share the weights)
        h.set_inflight(n_fly)
        k = 2

        def calls(_):
            for _ in range(k):
                h.forward_ids(utts, hw_emb=hw, want_timestamps=True)
        calls(0)
        t0 = time.perf_counter()
        th = [threading.Thread(target=calls, args=(i,)) for i in range(n_fly)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dtf = (time.perf_counter() - t0) / (k * n_fly)
        out["ms_per_batch_in_flight"] = 1e3 * dtf
        out["audio_s_per_s_in_flight"] = len(utts) * seconds / dtf
        out["in_flight"] = n_fly
    h.close()
    return out


def c5_block(pkg, weights, model, n_files=8, seconds=600, workers=8):
    """BASELINE.json configs[4] (C5) on ONE GPU: long files (8 x 600 s, seeded 1-s silences every 7-25 s) -> FSMN-VAD over the whole
    file in one pass -> end-point detector (host) -> length-sorted FetchDynamic batches -> the offline model, from 8 concurrent
    decoder workers that share ONE handle (merged by pfhip_set_batching into packed launches over the handle's contexts).
    `model` is the headline Paraformer-large handle (contexts already set)."""
    import importlib
    pipeline = importlib.import_module("asr_2pass_amd.pipeline")
    rng = np.random.default_rng(SEED_PCM + 5)

    def make_file(i):
        parts, t = [], 0.0
        while t < seconds:
            dur = min(float(rng.uniform(7, 25)), seconds - t)
            parts.append(synth_pcm(i, int(dur * SR), rng))
            parts.append(np.zeros(SR, np.float32))          # 1.0-s gaps: longer than the 800-ms end-silence threshold
            t += dur + 1.0
        return np.concatenate(parts)[:seconds * SR]
    files = [make_file(i) for i in range(n_files)]
    vman, vblob = weights.synth_vad_weights()
    vman, vblob = weights.energy_vad_weights(vman, vblob)
    vad = pkg.FsmnVadHip().InitVad((vman, vblob))
    segs_of = [None] * n_files
    ntok = [0] * n_files
    model.set_batching(3000, 96)
    nxt = [0]
    lock = threading.Lock()

    def worker():
        seg = pkg.E2EVadModelHost()
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= n_files:
                break
            ids, frames = pipeline.infer_buffer(files[i], model, vad, seg, batch_size=32)
            segs_of[i] = len(frames)
            ntok[i] = sum(len(x) for x in ids)
        seg.close()
    seg0 = pkg.E2EVadModelHost()
    pipeline.infer_buffer(files[0][:SR * 120], model, vad, seg0, batch_size=32)      # warm-up
    seg0.close()
    before = model.inflight_stats()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker) for _ in range(workers)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    after = model.inflight_stats()
    model.set_batching(0, 32)
    vad.close()
    fw = sum(a["forwards"] - b["forwards"] for a, b in zip(after, before))
    ut = sum(a["utterances"] - b["utterances"] for a, b in zip(after, before))
    return {"workload": f"{n_files} x {seconds}-s files, FSMN-VAD segmented, {workers} concurrent decoder workers on one handle, 1 GPU "
                        f"(BASELINE.json configs[4]; the 8-GPU form shards files over replicas)",
            "xRT": n_files * seconds / dt, "wall_s": dt, "segments": int(sum(segs_of)), "tokens": int(sum(ntok)),
            "packed_forwards": int(fw), "utterances_per_forward": ut / max(1, fw), "in_flight": model.get_inflight()}


def c3_2pass_block(pkg, weights, man, blob):
    """BASELINE.json configs[2] AS STATED (C3): 2-pass mode — streaming encoder chunk_size = [5, 10, 5] + offline rescoring of every
    segment the online VAD closes — through the C++ mirror of the reference's handle API (FunTpassInit / FunTpassOnlineInit /
    FunTpassInferBuffer: onnxruntime/bin/funasr-onnx-2pass-rtf.cpp:45-175 semantics, one thread per connection like the websocket
    handlers), Paraformer-large-sized offline + online models and the FSMN-VAD shaped so that it follows the frame energy.
    One connection streams a 10-minute file in 600-ms messages (latency: p50 / p99 per call); 128 connections stream 60 s each
    (throughput; every device call of a round is merged across connections).  Runs `tpass_bench` as a child process."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_pipeline import shape_vad_weights
    exe = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "tpass_bench")
    d = tempfile.mkdtemp(prefix="c3_2pass_")
    out = {"workload": "2pass: online VAD + streaming Paraformer-large chunk_size=[5,10,5] + offline Paraformer-large rescoring of every "
                       "closed segment, 600-ms messages through the FunTpassInferBuffer mirror (BASELINE.json configs[2])"}
    try:
        rng = np.random.default_rng(SEED_PCM + 9)
        for name, (mm, bb) in (("asr", (man, blob)), ("online", weights.synth_weights(dict(weights.PARAFORMER_LARGE), seed=32))):
            os.mkdir(os.path.join(d, name))
            weights.save(os.path.join(d, name, "model.pfhip"), mm, bb)
        os.mkdir(os.path.join(d, "vad"))
        vman, vblob = shape_vad_weights(*weights.synth_vad_weights())
        weights.save(os.path.join(d, "vad", "model.pfhip"), vman, vblob)

        def stream_file(seconds, path):          # speech-like bursts separated by 1.2-s silences: the VAD closes a segment every few seconds
            parts, total, i = [], 0, 0
            while total < seconds * SR:
                sec = [4.0, 7.5, 2.2, 11.0, 5.3][i % 5]
                parts += [synth_pcm(2000 + i, int(sec * SR), rng), np.zeros(int(1.2 * SR), np.float32)]
                total += len(parts[-2]) + len(parts[-1]); i += 1
            pcm = np.concatenate(parts)[:seconds * SR]
            np.clip(np.round(pcm * 32768.0), -32768, 32767).astype("<i2").tofile(path)
        for key, seconds, conns in (("one_connection", 600, 1), ("128_connections", 60, 128)):
            path = os.path.join(d, f"stream_{seconds}.pcm")
            stream_file(seconds, path)
            r = subprocess.run([exe, os.path.join(d, "asr"), os.path.join(d, "online"), os.path.join(d, "vad"), "-", path, str(conns), "2"],
                               capture_output=True, text=True, timeout=600, env=dict(os.environ, PFHIP_WARMUP_SECONDS="10"))
            if r.returncode != 0:
                out[key] = {"error": (r.stderr or r.stdout)[-500:]}
                continue
            j = json.loads(r.stdout.strip().splitlines()[-1])
            out[key] = {"stream_seconds": seconds, "xRT": j["xrt"], "wall_s": j["wall_s"], "calls": j["calls"], "p50_call_ms": j["p50_call_ms"],
                        "p99_call_ms": j["p99_call_ms"], "worst_call_ms": j["worst_call_ms"], "second_pass_results": j["tpass_results"]}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def cpu_baseline(man, blob, utts, seconds_per_utt):
    """The oracle (CPU restatement, numpy/OpenBLAS fp32) timed in the reference's threading shape:
    W worker threads sharing one model, 1 BLAS thread each, batch 1 per call
    (onnxruntime/src/paraformer.cpp:35,470-473; websocket/run_server_offline.sh:39).  rtf formula of
    onnxruntime/bin/funasr-onnx-offline-rtf.cpp:257-260: max thread compute time / total audio."""
    from threadpoolctl import threadpool_limits
    from oracle import paraformer as oracle_pf
    W = oracle_pf.Weights(man, blob)
    workers = len(utts)
    busy = [0.0] * workers

    def work(i):
        t0 = time.perf_counter()
        oracle_pf.forward_pcm(utts[i], W)
        busy[i] = time.perf_counter() - t0

    with threadpool_limits(limits=1):
        ths = [threading.Thread(target=work, args=(i,)) for i in range(workers)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        wall = time.perf_counter() - t0
    audio = workers * seconds_per_utt
    return {"value": audio / max(busy), "unit": "audio-s/s", "cores": workers, "kind": "port",
            "sample": f"{workers} x {seconds_per_utt}-s utterances, {workers} worker threads x 1 BLAS thread, "
                      f"batch 1 per call, numpy/OpenBLAS fp32 CPU restatement (not onnxruntime); wall {wall:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--seconds", type=int, default=SECONDS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-streaming", action="store_true", help="skip the C3 streaming block (rank 0, N = 1 only)")
    ap.add_argument("--no-c4c5", action="store_true", help="skip the C4 (hotword + timestamp) and C5 (long audio) blocks (rank 0, N = 1 only)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="batches in flight per GPU in the timed region behind `value` (execution contexts of the one handle, one caller "
                         "thread each); 1 = back to back.  The back-to-back figures and the roofline are always measured too")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    pkg = ge.load_package()
    import importlib
    weights = importlib.import_module("asr_2pass_amd.weights")

    cfg = dict(weights.PARAFORMER_LARGE)
    man, blob = weights.synth_weights(cfg, seed=1234)
    model = pkg.ParaformerHip().InitAsr((man, blob), device=local_rank)

    n = args.seconds * SR
    rng = np.random.default_rng(SEED_PCM + rank)
    utts = [synth_pcm(i, n, rng) for i in shard_utterance_ids(rank, world, args.batch)]
    d_pcm = torch.from_numpy(np.concatenate(utts)).cuda(local_rank)
    sample_off = np.arange(args.batch, dtype=np.int64) * n
    n_samples = np.full(args.batch, n, np.int32)
    stream = torch.cuda.current_stream()
    max_tokens = n // 960 + 2

    def step():
        model.enqueue_device(d_pcm.data_ptr(), sample_off, n_samples, stream.cuda_stream)
        return model.fetch(args.batch, max_tokens)

    # Several batches in flight (the reference's serving shape is concurrent decoder threads on ONE shared session:
    # funasr-wss-server.cpp:479-481, paraformer.cpp:35-41,541, run_server_offline.sh:39).  One launch of this workload already fills
    # the chip; what a second and third batch in flight buy is their matrix-core phases over another batch's bandwidth-bound ones
    # (epilogues, FSMN prologue, launch boundaries, the host round trip for the token counts).  They run on execution contexts of
    # the ONE handle (pfhip_set_inflight: one weight set, n workspaces), n_fly caller threads, PCM resident in HBM.
    n_fly = max(1, args.in_flight)
    last_of_thread = [None] * n_fly

    def run_thread(i, k, host):
        for _ in range(k):
            if host:
                last_of_thread[i] = model.forward_ids(utts, max_tokens=max_tokens)
            else:
                last_of_thread[i] = model.forward_resident(d_pcm.data_ptr(), sample_off, n_samples, max_tokens)

    def steps_in_flight(k, host=False):
        share = [k // n_fly + (1 if i < k % n_fly else 0) for i in range(n_fly)]
        th = [threading.Thread(target=run_thread, args=(i, share[i], host)) for i in range(n_fly) if share[i]]
        for t in th:
            t.start()
        for t in th:
            t.join()

    for _ in range(args.warmup):
        res = step()
    # HIP events around the launches of the dominant kernel class only (the fp32 GEMM): bracketing all ~1000
    # launches of a step costs ~6 % of the step, bracketing the 284 GEMMs ~1 %
    model.profile_enable(0 if args.no_profile else GEMM_ONLY_MASK)
    model.profile_read(reset=True)
    # the driver's contract: exactly K steps bracketed by barrier + synchronize on both sides, MAX over ranks below
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()       # the NCCL/RCCL barrier is enqueued on the stream: wait for it too
    dt = time.perf_counter() - t0
    prof = model.profile_read(reset=True)
    model.profile_enable(0)
    # a second, untimed pass with every class bracketed gives the per-class breakdown
    prof_all = None
    if not args.no_profile:
        model.profile_enable(1)
        step()
        prof_all = model.profile_read(reset=True)
        model.profile_enable(0)
    dt = max_over_ranks(dt, dist, torch.device("cuda", local_rank))
    dt_seq = dt
    # the same K steps, n_fly of them in flight on the one handle: this timed region is the one behind `value`
    # (every rank runs the same sequence of collectives: nothing below raises between two barriers)
    try:
        if n_fly > 1:
            model.set_inflight(n_fly)
    except Exception as e:          # contexts that cannot be built (memory on a shared node) must not cost the run its numbers
        print(f"bench.py: 1 batch in flight instead of {n_fly}: {e}", file=sys.stderr)
    n_fly = model.get_inflight()
    steps_in_flight(max(args.warmup, n_fly))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps_in_flight(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt = max_over_ranks(time.perf_counter() - t0, dist, torch.device("cuda", local_rank))
    for got in last_of_thread:        # every context, under concurrency, returns what the back-to-back run returned
        assert got is None or all(list(a) == list(b) for a, b in zip(got["ids"], res["ids"]))
    slots = model.inflight_stats()
    # the same K steps on the reference's own boundary: host float** buffers in (Model::Forward(float** din, ...), H2D inside
    # the timed region where paraformer-torch.cpp:355-358 has it), ids out — reported beside `value`, never as `value`:
    # in flight on the one handle (n_fly caller threads, as the server's decoder threads), then back to back from one thread
    steps_in_flight(min(max(args.warmup, 1), 2) * n_fly, host=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps_in_flight(args.steps, host=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt_host_fly = max_over_ranks(time.perf_counter() - t0, dist, torch.device("cuda", local_rank))
    for got in last_of_thread:
        assert got is None or all(list(a) == list(b) for a, b in zip(got["ids"], res["ids"]))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res_h = model.forward_ids(utts, max_tokens=max_tokens)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt_host = max_over_ranks(time.perf_counter() - t0, dist, torch.device("cuda", local_rank))
    assert all(list(a) == list(b) for a, b in zip(res_h["ids"], res["ids"]))

    x3 = os.environ.get("PFHIP_GEMM_X3", "1") != "0" and os.environ.get("PFHIP_GEMM_X6", "1") != "0"
    audio_per_step = world * args.batch * args.seconds
    value = audio_per_step * args.steps / dt
    tokens = int(sum(len(x) for x in res["ids"]))

    if rank == 0:
        out = {
            "metric": "audio-sec/sec (xRT) Paraformer-large offline, 30s utts",
            "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (operands staged as 2 fp16 planes = 22-23 significant bits, 3 products per block on the FP16 matrix cores, "
                      "fp32 accumulate; error vs fp64 = the fp32 MFMA chain's)" if x3 else
                      "f32 (operands split exactly into 3 bf16 planes, 6 products on the BF16 matrix cores, fp32 accumulate)"),
            "data": "synthetic",
            "value_host_buffers_in_flight": audio_per_step * args.steps / dt_host_fly,
            "ms_per_step_host_buffers_in_flight": 1e3 * dt_host_fly / args.steps,
            "value_host_buffers": audio_per_step * args.steps / dt_host, "ms_per_step_host_buffers": 1e3 * dt_host / args.steps,
            "value_host_buffers_router": audio_per_step * args.steps / dt_host_fly,
            "host_buffers_note": "same steps through pfhip_offline_forward(float** host pcm): the 61 MB H2D copy of the batch is "
                                 "inside the timed region (the reference's Model::Forward boundary); `value` has the PCM resident in HBM. "
                                 "`value_host_buffers_in_flight`: in_flight caller threads on the ONE handle (the unchanged server's decoder "
                                 "threads on one shared model); `value_host_buffers`: one thread, back to back; "
                                 "`value_host_buffers_router`: round-2 key, now the same number as `value_host_buffers_in_flight`",
            "value_one_in_flight": audio_per_step * args.steps / dt_seq, "ms_per_step_one_in_flight": 1e3 * dt_seq / args.steps,
            "in_flight_note": f"`value` / `ms_per_step`: the K steps from {n_fly} caller threads on ONE handle whose {n_fly} execution "
                              "contexts (workspace + stream each) share one weight set (pfhip_set_inflight) — the reference serves with "
                              "concurrent decoder threads on one shared session; `*_one_in_flight`: the same K steps back to back on "
                              "context 0 — the round-over-round comparable figure, the timed region the roofline's HIP events bracket "
                              "(with several batches in flight a launch's event-to-event time is not its own) and the command behind "
                              "profiles/ (`--in-flight 1`)",
            "slots": slots,
            "config": {"workload": f"Paraformer-large offline, batch={args.batch} x {args.seconds} s synthetic 16 kHz "
                                   f"utterances per GPU (BASELINE.json configs[1])",
                       "in_flight": n_fly,
                       "batch_per_gpu": args.batch, "utt_seconds": args.seconds, "lfr_frames_per_utt": int(res["n_frames"][0]),
                       "tokens_per_batch": tokens, "weights": "random-init Paraformer-large (seed 1234), fp32",
                       "parallelism": f"replicas x{world} (no collective)", "rtf": 1.0 / value},
        }
        if not args.no_profile:
            g = prof["gemm"]
            traffic, traffic_source = pmc_traffic()
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
            per_block = X3_MFMAS_PER_BLOCK if x3 else X6_MFMAS_PER_BLOCK
            peak = X3_PEAK_TFLOPS if x3 else X6_PEAK_TFLOPS
            out["roofline"] = {
                # achieved = ALGORITHMIC fp32 flops (2*M*N*K) per second; the dominant kernel executes 3 fp16 MFMAs per block (6 bf16
                # ones with PFHIP_GEMM_X3=0), so its ceiling is the FP16 / BF16 dense peak / 3 (/ 6); executed MFMA rate = 3 x achieved
                "bound": "mfma",
                "kernel": ("gemm_p3_128_kernel (encoder, operands as fp16 plane images) / gemm_f32_f16x3_128_kernel (decoder)"
                           if os.environ.get("PFHIP_PLANES", "1") != "0" else "gemm_f32_f16x3_128_kernel / gemm_f32_f16x3_kernel") if x3 else "gemm_f32_bf16x6_128_kernel / gemm_f32_bf16x6_kernel",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source,
                "note": "flops counted = 2*M*N*K of the GEMMs only (the same launches also carry the LayerNorms folded into them). "
                        "Round 3 halved the MFMAs per block (3 fp16 products instead of 6 bf16 ones: ceiling 417 -> 833 TFLOP/s): the "
                        "class takes 0.7 x the time and `frac` is quoted against the NEW ceiling; against round 2's it would read "
                        f"{ach / X6_PEAK_TFLOPS:.3f}" if x3 else "six bf16 products per block (PFHIP_GEMM_X3=0)",
                "frac_of_round2_ceiling_417": ach / X6_PEAK_TFLOPS,
                "executed_mfma_tflops": ach * per_block, "executed_mfma_peak": BF16_MFMA_PEAK_TFLOPS,
                "fp32_mfma_peak_for_reference": F32_MFMA_PEAK_TFLOPS,
                "avg_launch_ms": g["ms"] / max(1, g["launches"]), "launches_per_step": g["launches"] / args.steps,
                "flops_per_launch": g["flops"] / max(1, g["launches"]),
                "algorithmic_bytes_per_launch": g["bytes"] / max(1, g["launches"]),
                "per_class_ms_untimed_pass": {k: v["ms"] for k, v in prof_all.items()},
                "per_class_tflops_untimed_pass": {k: (v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0)
                                                  for k, v in prof_all.items() if v["flops"] > 0},
            }
        if world == 1 and not args.no_streaming:
            out["streaming"] = streaming_block(pkg, model, int(blob.nbytes), np.random.default_rng(SEED_PCM + 7))
        if world == 1 and not args.no_streaming:
            try:
                out["c3_2pass"] = c3_2pass_block(pkg, weights, man, blob)
            except Exception as e:
                out["c3_2pass"] = {"error": str(e)}
        if world == 1 and not args.no_c4c5:
            try:
                out["c5"] = c5_block(pkg, weights, model)
            except Exception as e:
                out["c5"] = {"error": str(e)}
            try:
                out["c4"] = c4_block(pkg, weights, utts, args.seconds, n_fly)
            except Exception as e:
                out["c4"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            workers = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(man, blob, utts[:workers], args.seconds)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
