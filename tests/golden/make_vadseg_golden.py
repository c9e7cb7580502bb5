"""Generates tests/golden/vadseg_*.npz from the REFERENCE's own end-point detector, `funasr::E2EVadModel`
(onnxruntime/src/e2e-vad.h, header-only), compiled in place by oracle/Makefile into oracle/_ref/libe2evad_ref.so
(this container only; the reference cannot travel).  Fixtures are data: per-frame silence posteriors + a waveform recipe
+ a call plan in, the segments every call returned out.

  python tests/golden/make_vadseg_golden.py

A fixture holds
  sil      float32 [T]      silence posterior per 10-ms frame (scores[t][0], e2e-vad.h:601-606)
  amp      float32 [T + 2]  amplitude per 10-ms hop; waveform[i] = amp[i // 160] * pattern[i % 160]  (exact in float32)
  pattern  float32 [160]
  calls    int32 [C, 4]     frame_start, n_frames, is_final, online      (one detector object serves all calls, in order;
                            the waveform slice of a call is samples [160*start, 160*(start+n-1)+400), as the reference's
                            callers hand it over, fsmn-vad.cpp:240-256 / audio.cpp:1183-1196)
  params   float32 [3]      max_end_sil, max_single_segment_time, speech_noise_thres
  segs     int32 [S, 3]     call index, start_ms, end_ms   (-1 = open, online mode)
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load_ref():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libe2evad_ref.so"))
    lib.e2evad_ref_create.restype = ctypes.c_void_p
    lib.e2evad_ref_destroy.argtypes = [ctypes.c_void_p]
    lib.e2evad_ref_feed.restype = ctypes.c_int
    lib.e2evad_ref_feed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    return lib


def waveform(amp, pattern):
    n = 160 * len(amp)
    idx = np.arange(n)
    return (amp[idx // 160] * pattern[idx % 160]).astype(np.float32)


def wave_slice(w, start, n):
    return w[160 * start:160 * (start + n - 1) + 400] if n > 0 else w[:0]


def run_plan(feed, sil, w, calls, params):
    """feed(sil, wave, is_final, online, max_end_sil, max_seg, thres) -> list of [s, e]; returns int32 [S, 3]."""
    out = []
    for ci, (start, n, fin, online) in enumerate(calls):
        for s, e in feed(sil[start:start + n], wave_slice(w, start, n), bool(fin), bool(online), int(params[0]), int(params[1]), float(params[2])):
            out.append((ci, s, e))
    return np.asarray(out, np.int32).reshape(-1, 3)


def scores(rng, runs, noise=0.02, flip=0.03):
    sil = []
    for n, sp in runs:
        base = 0.03 if sp else 0.97
        v = np.clip(base + noise * rng.standard_normal(n), 1e-4, 1 - 1e-4)
        f = rng.random(n) < flip
        v[f] = 1 - v[f]
        sil.append(v)
    return np.concatenate(sil).astype(np.float32)


def chunks(T, sizes, online, final_empty=False):
    calls, pos, k = [], 0, 0
    while pos < T:
        n = min(sizes[k % len(sizes)], T - pos)
        calls.append((pos, n, int(pos + n >= T and not final_empty), int(online)))
        pos += n
        k += 1
    if final_empty:
        calls.append((T, 0, 1, int(online)))
    return calls


def cases():
    rng = np.random.default_rng(20251004)
    pattern = (0.5 * rng.standard_normal(160)).astype(np.float32)
    out = {}

    def add(name, sil, amp, calls, params=(800, 15000, 0.9)):
        out[name] = dict(sil=sil.astype(np.float32), amp=np.asarray(amp, np.float32), pattern=pattern,
                         calls=np.asarray(calls, np.int32).reshape(-1, 4), params=np.asarray(params, np.float32))

    # 1. all silence, zero waveform (decibel = 10 log10(1e-6) = -60 on every frame), offline one-shot
    T = 600
    add("all_silence", scores(rng, [(T, False)]), np.zeros(T + 2), [(0, T, 1, 0)])
    # 2. one burst, offline one-shot (fsmn-vad.cpp:240-256)
    sil = scores(rng, [(150, False), (420, True), (230, False)])
    add("single_burst", sil, 0.1 * np.ones(len(sil) + 2), [(0, len(sil), 1, 0)])
    # 3. a 21-s stretch of speech against max_single_segment_time = 6000 ms (forced splits, e2e-vad.h:672-690)
    sil = scores(rng, [(100, False), (2100, True), (150, False)])
    add("max_segment_split", sil, 0.1 * np.ones(len(sil) + 2), [(0, len(sil), 1, 0)], (800, 6000, 0.9))
    # 4. online, 1-s feeds (audio.cpp:1183-1196), several bursts, open (-1) segment ends across calls
    sil = scores(rng, [(90, False), (400, True), (130, False), (250, True), (60, False), (45, True), (200, False)])
    add("online_1s_feeds", sil, 0.08 * np.ones(len(sil) + 2), chunks(len(sil), [100], True))
    # 5. low energy: stretches of exact digital silence under speech-like scores and under silence scores
    sil = scores(rng, [(80, False), (300, True), (100, False), (200, True), (120, False)])
    amp = 0.1 * np.ones(len(sil) + 2)
    amp[150:260] = 0.0
    amp[400:470] = 0.0
    amp[600:] = 1e-4
    add("low_energy", sil, amp, [(0, len(sil), 1, 0)])
    # 6. re-use after final: two files through ONE object, offline (AllResetDetection, e2e-vad.h:394-423)
    a = scores(rng, [(60, False), (200, True), (140, False)])
    b = scores(rng, [(30, False), (120, True), (90, False), (160, True), (100, False)])
    sil = np.concatenate([a, b])
    add("reuse_offline", sil, 0.1 * np.ones(len(sil) + 2), [(0, len(a), 1, 0), (len(a), len(b), 1, 0)])
    # 7. the same online: file A in 600-ms feeds, final; then file B on the same object
    ca = [(s, n, f, 1) for s, n, f, _ in chunks(len(a), [60], True)]
    cb = [(len(a) + s, n, f, 1) for s, n, f, _ in chunks(len(b), [60], True)]
    add("reuse_online", sil, 0.1 * np.ones(len(sil) + 2), ca + cb)
    # 8. online, irregular feed sizes, short tail silence and a lower threshold
    sil = scores(rng, [(40, False), (180, True), (45, False), (90, True), (35, False), (300, True), (110, False)], noise=0.05, flip=0.06)
    add("online_irregular", sil, 0.1 * np.ones(len(sil) + 2), chunks(len(sil), [30, 170, 5, 61, 100], True), (300, 15000, 0.6))
    # 9. speech runs into the end of the file: the end point is forced by is_final (offline and online)
    sil = scores(rng, [(70, False), (330, True)])
    add("speech_to_eof_offline", sil, 0.1 * np.ones(len(sil) + 2), [(0, len(sil), 1, 0)])
    add("speech_to_eof_online", sil, 0.1 * np.ones(len(sil) + 2), chunks(len(sil), [100], True))
    # 10. online with an EMPTY final call (the 2-pass client's end-of-stream message carries no audio)
    sil = scores(rng, [(50, False), (260, True), (40, False), (150, True)])
    add("online_empty_final", sil, 0.1 * np.ones(len(sil) + 2), chunks(len(sil), [60], True, final_empty=True))
    # 11. offline chunked (is_final only on the last call, online = false): only closed segments come out before the end
    sil = scores(rng, [(100, False), (250, True), (200, False), (300, True), (150, False)])
    add("offline_chunked", sil, 0.1 * np.ones(len(sil) + 2), chunks(len(sil), [100], False))
    return out


def main():
    lib = load_ref()
    for name, c in cases().items():
        h = lib.e2evad_ref_create()

        def feed(sil, wave, fin, online, tail, mx, thr):
            sil = np.ascontiguousarray(sil, np.float32)
            wave = np.ascontiguousarray(wave, np.float32)
            cap = len(sil) + 8
            pairs = np.zeros((cap, 2), np.int32)
            n = lib.e2evad_ref_feed(h, sil.ctypes.data, len(sil), wave.ctypes.data, len(wave), int(fin), int(online), tail, mx, thr, 16000,
                                    pairs.ctypes.data, cap)
            assert n <= cap
            return [list(map(int, p)) for p in pairs[:n]]

        w = waveform(c["amp"], c["pattern"])
        segs = run_plan(feed, c["sil"], w, c["calls"], c["params"])
        lib.e2evad_ref_destroy(h)
        np.savez_compressed(os.path.join(HERE, f"vadseg_{name}.npz"), segs=segs, **c)
        print(f"vadseg_{name}: T={len(c['sil'])} calls={len(c['calls'])} segments={len(segs)} -> {segs[:6].tolist()}")


if __name__ == "__main__":
    sys.exit(main())
