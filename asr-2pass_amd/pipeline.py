"""Offline long-audio flow around the hot path — the data flow of `FunOfflineInferBuffer`
(onnxruntime/src/funasrruntime.cpp:208-337) for one decoder worker on one GPU:

    Audio::CutSplit     (audio.cpp:1172-1240)  VAD scores -> segments (ms) -> sample ranges, sorted by length
    Audio::FetchDynamic (audio.cpp:1052-1108)  greedy batches under max_sent = 60 s and max_acc = 300 s of padded audio
    Model::Forward      (funasrruntime.cpp:268) batched acoustic model
    result re-ordering  (funasrruntime.cpp:270-279, index_vector)

MI355X-first differences (results identical, see tests/test_gpu_pipeline.py): the FSMN-VAD network scores the WHOLE
file in one pass instead of 600 one-second Runs per 10 minutes (it is causal, the caches would carry the same state),
only the silence posterior column leaves the GPU, and batches are packed (no padding rows) so `max_acc` bounds memory,
not wasted compute.  Host logic only; all arithmetic is in libpfhip.so.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

SEG_SAMPLE = 16          # samples per millisecond at 16 kHz (audio.cpp: seg_sample = MODEL_SAMPLE_RATE / 1000)


def merge_online_segments(vad_segments: Sequence[Sequence[int]]) -> List[Tuple[int, int]]:
    """audio.cpp:1199-1223 — join the (start,-1) / (-1,end) halves the online detector emits."""
    out, s_i, e_i = [], -1, -1
    for s, e in vad_segments:
        if s != -1:
            s_i = s
        if e != -1:
            e_i = e
        if s_i != -1 and e_i != -1:
            out.append((s_i * SEG_SAMPLE, e_i * SEG_SAMPLE))
            s_i = e_i = -1
    return out


def cut_split(pcm: np.ndarray, vad, segmenter, vad_tail_sil=800, vad_max_len=60000, speech_noise_thres=0.9):
    """Returns (frames, index_vector): frames = [(start_sample, end_sample)] in time order, index_vector = their
    order by increasing length (audio.cpp:1226-1239), i.e. the order they are queued for FetchDynamic."""
    sil = vad.ForwardSil(pcm, is_final=True)
    if sil.size == 0:
        return [], []
    n_used = 400 + 160 * (sil.size - 1)
    segs = segmenter(sil, pcm[:max(n_used, 0)], True, False, vad_tail_sil, vad_max_len, speech_noise_thres)
    frames = [(s * SEG_SAMPLE, min(e * SEG_SAMPLE, len(pcm))) for s, e in segs]
    index_vector = sorted(range(len(frames)), key=lambda i: (frames[i][1] - frames[i][0], i))     # stable like std::sort on ties? see note
    return frames, index_vector


def fetch_dynamic(queue: List[Tuple[int, int]], batch_size: int):
    """Pops one batch off the front of `queue` by the rule of Audio::FetchDynamic (audio.cpp:1056-1084)."""
    max_acc = 300 * 1000 * SEG_SAMPLE
    max_sent = 60 * 1000 * SEG_SAMPLE
    bs_acc, max_len = 0, 0
    max_batch = min(batch_size, len(queue))
    batch = []
    for _ in range(max_batch):
        s, e = queue[0]
        length = e - s
        if length >= max_sent:
            if bs_acc == 0:
                bs_acc += 1
                batch.append(queue.pop(0))
            break
        max_len = max(max_len, length)
        if max_len * (bs_acc + 1) > max_acc:
            break
        bs_acc += 1
        batch.append(queue.pop(0))
    return batch


def infer_buffer(pcm: np.ndarray, asr, vad, segmenter, batch_size=32, vad_tail_sil=800, vad_max_len=60000,
                 speech_noise_thres=0.9):
    """FunOfflineInferBuffer for one file: returns (token ids per segment in TIME order, segments in samples)."""
    frames, index_vector = cut_split(pcm, vad, segmenter, vad_tail_sil, vad_max_len, speech_noise_thres)
    queue = [frames[i] for i in index_vector]
    msgs = []
    while queue:
        batch = fetch_dynamic(queue, batch_size)
        if not batch:
            break
        res = asr.forward_ids([pcm[s:e] for s, e in batch])
        msgs += res["ids"]
    ordered = [None] * len(frames)
    for pos, seg_idx in enumerate(index_vector):          # funasrruntime.cpp:270-279
        ordered[seg_idx] = msgs[pos]
    return ordered, frames
