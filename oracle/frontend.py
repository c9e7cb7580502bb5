"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU (numpy, fp32 with the FFT in fp64) restatement of the reference's acoustic front end:

  a2  Paraformer::FbankKaldi            onnxruntime/src/paraformer.cpp:309-323
      -> knf::OnlineFbank               third_party/kaldi-native-fbank/.../csrc/online-feature.cc:104-161
      -> ExtractWindow / ProcessWindow  .../csrc/feature-window.cc:121-245
      -> FbankComputer::Compute         .../csrc/feature-fbank.cc:73-118
      -> Rfft (Ooura, in double)        .../csrc/rfft.cc:41-52
      -> ComputePowerSpectrum           .../csrc/feature-functions.cc:28-47
      -> MelBanks ctor / Compute        .../csrc/mel-computations.cc:107-255
  a3  Paraformer::LfrCmvn / LoadCmvn    onnxruntime/src/paraformer.cpp:421-461, 325-360
  a9  sinusoidal position embedding     onnxruntime/src/paraformer-online.cpp:240-268 (and :549-555 scaling)

Parity status: PINNED for fbank — checked against (i) the reference's own knf sources compiled in
place (oracle/_ref/libknf_ref.so, oracle/Makefile), (ii) the knf known-answer test
test-rfft.cc:32-50 and (iii) committed golden vectors under tests/golden/ produced by (i).
LFR/CMVN, PE are pinned by hand-derived known answers (tests/test_oracle_frontend.py).
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math

import numpy as np

F32 = np.float32

# --- defaults: onnxruntime/src/paraformer.h:112-121, com-define.h -----------------------------
SAMPLE_RATE = 16000
FRAME_LEN = 400          # 25 ms  (feature-window.h: WindowSize)
FRAME_SHIFT = 160        # 10 ms  (feature-window.h: WindowShift)
NFFT = 512               # PaddedWindowSize, round_to_power_of_two
N_MELS = 80
LFR_M = 7
LFR_N = 6
PREEMPH = F32(0.97)      # feature-window.h:35
FLT_EPS = np.finfo(np.float32).eps


def num_frames(num_samples: int) -> int:
    """feature-window.cc:84-87 (snip_edges=true)."""
    if num_samples < FRAME_LEN:
        return 0
    return 1 + (num_samples - FRAME_LEN) // FRAME_SHIFT


def hamming_window() -> np.ndarray:
    """feature-window.cc:33-42: computed in double, stored as float."""
    a = 2.0 * math.pi / (FRAME_LEN - 1)
    i = np.arange(FRAME_LEN, dtype=np.float64)
    return (0.54 - 0.46 * np.cos(a * i)).astype(F32)


_LIBM = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_LIBM.logf.restype = ctypes.c_float
_LIBM.logf.argtypes = [ctypes.c_float]


def _logf(x) -> np.float32:
    """glibc logf — the same routine knf's MelScale calls; numpy's SIMD log differs by 1 ulp on
    some inputs, which the triangle-weight subtraction amplifies to ~1e-5 in the log-mel output."""
    return F32(_LIBM.logf(float(F32(x))))


def _mel_scale(freq):
    """mel-computations.h MelScale: 1127*logf(1+f/700), in float."""
    f = np.atleast_1d(np.asarray(freq, dtype=F32))
    out = np.asarray([F32(1127.0) * _logf(F32(1.0) + F32(v) / F32(700.0)) for v in f], dtype=F32)
    return out if np.ndim(freq) else out[0]


def mel_banks(num_bins: int = N_MELS, sample_freq: float = SAMPLE_RATE,
              low_freq: float = 20.0, high_freq: float = 0.0):
    """mel-computations.cc:107-196 (vtln_warp == 1).  Returns (offsets[int32 nb], sizes[int32 nb],
    weights list of float32 arrays)."""
    num_fft_bins = NFFT // 2
    nyquist = F32(0.5) * F32(sample_freq)
    hf = F32(high_freq) if high_freq > 0.0 else F32(nyquist + F32(high_freq))
    lf = F32(low_freq)
    fft_bin_width = F32(sample_freq) / F32(NFFT)
    mel_low = _mel_scale(lf)
    mel_high = _mel_scale(hf)
    mel_delta = F32((mel_high - mel_low) / F32(num_bins + 1))
    mel_fft = _mel_scale(fft_bin_width * np.arange(num_fft_bins, dtype=F32))
    offsets, sizes, weights = [], [], []
    for b in range(num_bins):
        left = F32(mel_low + F32(b) * mel_delta)
        center = F32(mel_low + F32(b + 1) * mel_delta)
        right = F32(mel_low + F32(b + 2) * mel_delta)
        this_bin = np.zeros(num_fft_bins, dtype=F32)
        first, last = -1, -1
        for i in range(num_fft_bins):
            mel = mel_fft[i]
            if mel > left and mel < right:
                if mel <= center:
                    w = F32(F32(mel - left) / F32(center - left))
                else:
                    w = F32(F32(right - mel) / F32(right - center))
                this_bin[i] = w
                if first == -1:
                    first = i
                last = i
        assert first != -1 and last >= first
        offsets.append(first)
        sizes.append(last + 1 - first)
        weights.append(this_bin[first:last + 1].copy())
    return np.asarray(offsets, np.int32), np.asarray(sizes, np.int32), weights


_MEL_CACHE = {}


def _mel(num_bins):
    if num_bins not in _MEL_CACHE:
        _MEL_CACHE[num_bins] = mel_banks(num_bins)
    return _MEL_CACHE[num_bins]


def _seq_sum_f32(x: np.ndarray) -> np.ndarray:
    """Left-to-right float32 accumulation along the last axis (knf sums in a float scalar)."""
    return np.cumsum(x, axis=-1, dtype=F32)[..., -1]


def rfft_packed(frame: np.ndarray) -> np.ndarray:
    """knf::Rfft::Compute (rfft.cc:41-52): double FFT, output packed the Ooura way
    [re0, re(N/2), re1, im1, ...] where the stored imaginary part is the NEGATED numpy one
    (test-rfft.cc:41-49 checks -d[3] == -2.2929 for im = -2.2929j ... i.e. d[3] = +2.2929)."""
    n = frame.shape[-1]
    spec = np.fft.rfft(frame.astype(np.float64), axis=-1)
    out = np.empty(frame.shape, dtype=np.float64)
    out[..., 0] = spec[..., 0].real
    out[..., 1] = spec[..., n // 2].real
    out[..., 2::2] = spec[..., 1:n // 2].real
    out[..., 3::2] = -spec[..., 1:n // 2].imag
    return out.astype(F32)


def fbank(waves: np.ndarray, num_bins: int = N_MELS) -> np.ndarray:
    """waves: float32 in [-1,1) as handed to Model::Forward.  Returns [F, num_bins] float32 log-mel.
    paraformer.cpp:309-323 + the knf chain cited in the module docstring."""
    waves = np.asarray(waves, dtype=F32)
    buf = (waves * F32(32768)).astype(F32)                      # paraformer.cpp:312-314
    nf = num_frames(buf.shape[0])
    if nf == 0:
        return np.zeros((0, num_bins), F32)
    idx = np.arange(nf)[:, None] * FRAME_SHIFT + np.arange(FRAME_LEN)[None, :]
    win = buf[idx]                                              # ExtractWindow, feature-window.cc:147-151
    # RemoveDcOffset feature-window.cc:179-190
    mean = (_seq_sum_f32(win) / F32(FRAME_LEN)).astype(F32)
    win = (win - mean[:, None]).astype(F32)
    # Preemphasize feature-window.cc:200-211 (uses the un-modified d[i-1]: loop runs high->low)
    pre = np.empty_like(win)
    pre[:, 1:] = win[:, 1:] - (PREEMPH * win[:, :-1]).astype(F32)
    pre[:, 0] = win[:, 0] - (PREEMPH * win[:, 0]).astype(F32)
    win = (pre * hamming_window()[None, :]).astype(F32)         # FeatureWindowFunction::Apply :58-64
    padded = np.zeros((nf, NFFT), F32)
    padded[:, :FRAME_LEN] = win
    p = rfft_packed(padded)                                     # feature-fbank.cc:85
    # ComputePowerSpectrum feature-functions.cc:28-47 (float arithmetic)
    half = NFFT // 2
    power = np.empty((nf, half + 1), F32)
    re = p[:, 2::2]
    im = p[:, 3::2]
    power[:, 1:half] = ((re * re).astype(F32) + (im * im).astype(F32)).astype(F32)
    power[:, 0] = (p[:, 0] * p[:, 0]).astype(F32)
    power[:, half] = (p[:, 1] * p[:, 1]).astype(F32)
    # MelBanks::Compute mel-computations.cc:224-247: sequential float accumulation per bin
    offsets, sizes, weights = _mel(num_bins)
    out = np.empty((nf, num_bins), F32)
    for b in range(num_bins):
        e = np.zeros(nf, F32)
        off = int(offsets[b])
        w = weights[b]
        for k in range(int(sizes[b])):
            e = (e + (w[k] * power[:, off + k]).astype(F32)).astype(F32)
        out[:, b] = e
    # feature-fbank.cc:102-107
    out = np.log(np.maximum(out, F32(FLT_EPS))).astype(F32)
    return out


def lfr_cmvn(feats: np.ndarray, means: np.ndarray, istd: np.ndarray,
             lfr_m: int = LFR_M, lfr_n: int = LFR_N) -> np.ndarray:
    """Paraformer::LfrCmvn (paraformer.cpp:421-461).  feats [F, D] -> [ceil(F/n), m*D]."""
    feats = np.asarray(feats, dtype=F32)
    T = feats.shape[0]
    if T == 0:
        return np.zeros((0, lfr_m * feats.shape[1]), F32)
    t_lfr = int(math.ceil(1.0 * T / lfr_n))
    npad = (lfr_m - 1) // 2
    x = np.concatenate([np.repeat(feats[:1], npad, axis=0), feats], axis=0)   # :428-430
    T = T + npad
    rows = []
    for i in range(t_lfr):
        if lfr_m <= T - i * lfr_n:
            rows.append(x[i * lfr_n:i * lfr_n + lfr_m].reshape(-1))
        else:                                                                # :441-452
            num_padding = lfr_m - (T - i * lfr_n)
            part = [x[i * lfr_n:].reshape(-1)]
            part += [x[-1]] * num_padding
            rows.append(np.concatenate(part))
    out = np.stack(rows).astype(F32)
    out = ((out + means[None, :].astype(F32)).astype(F32) * istd[None, :].astype(F32)).astype(F32)  # :455-459
    return out


def parse_cmvn(text: str, scale: float = 1.0):
    """Paraformer::LoadCmvn (paraformer.cpp:325-360): kaldi-nnet am.mvn text -> (means, vars)."""
    means, istd = [], []
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        items = lines[i].split()
        i += 1
        if not items:
            continue
        if items[0] == "<AddShift>" and i < len(lines):
            nxt = lines[i].split()
            i += 1
            if nxt and nxt[0] == "<LearnRateCoef>":
                means = [float(v) for v in nxt[3:len(nxt) - 1]]
        elif items[0] == "<Rescale>" and i < len(lines):
            nxt = lines[i].split()
            i += 1
            if nxt and nxt[0] == "<LearnRateCoef>":
                istd = [float(v) * scale for v in nxt[3:len(nxt) - 1]]
    return np.asarray(means, F32), np.asarray(istd, F32)


PE_SCALE = F32(-0.0330119726594128)     # paraformer-online.cpp:247  (= -ln(1e4)/279 for depth 560)


def pos_emb(timesteps: int, depth: int, start: int = 0) -> np.ndarray:
    """ParaformerOnline::GetPosEmb (paraformer-online.cpp:240-268): row j (0-based absolute index)
    gets sin/cos of (j+1)*exp(i*scale), i < depth/2; float arithmetic.  The offline graph applies the
    same table starting at position 1 (UPSTREAM SinusoidalPositionEncoder)."""
    half = depth // 2
    scale = F32(-math.log(10000.0) / (half - 1)) if depth != 560 else PE_SCALE
    # `float tmptime = exp(i * scale)`: float product, exp evaluated in double, stored to float (:252)
    arg = (np.arange(half, dtype=F32) * scale).astype(F32)
    inv = np.exp(arg.astype(np.float64)).astype(F32)
    pos = np.arange(start + 1, start + timesteps + 1, dtype=F32)
    coe = (inv[None, :] * pos[:, None]).astype(F32)
    return np.concatenate([np.sin(coe), np.cos(coe)], axis=1).astype(F32)


def extract_feats(waves: np.ndarray, means: np.ndarray, istd: np.ndarray) -> np.ndarray:
    """fbank -> LFR -> CMVN, as Paraformer::Forward does before the session Run (paraformer.cpp:475-488)."""
    fb = fbank(waves)
    if fb.shape[0] == 0:
        return np.zeros((0, LFR_M * N_MELS), F32)
    return lfr_cmvn(fb, means, istd)
