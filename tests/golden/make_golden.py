"""Generates tests/golden/*.npz from the REFERENCE's own kaldi-native-fbank sources, compiled in place
by oracle/Makefile into oracle/_ref/libknf_ref.so (this container only; the reference cannot travel).
Fixtures are data: int16 PCM in, float32 log-mel frames out (options of paraformer.cpp:24-31).

  python tests/golden/make_golden.py
"""
import ctypes
import os
import sys
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libknf_ref.so"))
lib.knf_ref_fbank.restype = ctypes.c_int
lib.knf_ref_fbank.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]


def ref_fbank(w):
    w = np.ascontiguousarray(w, np.float32)
    n = lib.knf_ref_fbank(w.ctypes.data, len(w), 80, None, 0)
    out = np.empty((n, 80), np.float32)
    lib.knf_ref_fbank(w.ctypes.data, len(w), 80, out.ctypes.data, n)
    return out


def main():
    # (1) seeded synthetic PCM, SURVEY §8d recipe, 1.2 s
    rng = np.random.default_rng(20251114)
    n = 19200
    t = np.arange(n) / 16000.0
    pcm = np.clip(np.round(8000 * (0.6 * np.sin(2 * np.pi * 110.0 * t) + 0.4 * rng.standard_normal(n))), -32768, 32767).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "fbank_synth.npz"), pcm=pcm, fbank=ref_fbank(pcm.astype(np.float32) / 32768))
    # (2) real speech: first 1.5 s after 1.0 s of clients/audio/xmov.wav (the reference's own sample)
    wf = wave.open("/root/reference/clients/audio/xmov.wav")
    x = np.frombuffer(wf.readframes(wf.getnframes()), np.int16)
    seg = x[16000:16000 + 24000].copy()
    np.savez_compressed(os.path.join(HERE, "fbank_xmov.npz"), pcm=seg, fbank=ref_fbank(seg.astype(np.float32) / 32768))
    # (3) silence / tiny amplitude edge (log floor FLT_EPSILON, feature-fbank.cc:102-107)
    z = np.zeros(1000, np.int16)
    z[500] = 1
    np.savez_compressed(os.path.join(HERE, "fbank_floor.npz"), pcm=z, fbank=ref_fbank(z.astype(np.float32) / 32768))
    # (4) the reference's other two samples: 1.0-s excerpts (English speech with music bed; spoken digits incl. a loud onset)
    for name, start in (("SteveJobs_10s", 40000), ("number", 30000)):
        wf = wave.open(f"/root/reference/clients/audio/{name}.wav")
        x = np.frombuffer(wf.readframes(wf.getnframes()), np.int16)
        seg = x[start:start + 16000].copy()
        np.savez_compressed(os.path.join(HERE, f"fbank_{name.lower()}.npz"), pcm=seg, fbank=ref_fbank(seg.astype(np.float32) / 32768))
    # (5) full-scale clipping: a +-32767 square wave (largest magnitudes the front end can see)
    sq = np.where((np.arange(4000) // 37) % 2 == 0, 32767, -32768).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "fbank_fullscale.npz"), pcm=sq, fbank=ref_fbank(sq.astype(np.float32) / 32768))
    print("golden fixtures written")


if __name__ == "__main__":
    sys.exit(main())
