"""ORACLE — TEST INFRASTRUCTURE ONLY.  `TimestampOnnx` of the reference (onnxruntime/src/util.cpp:838-963) restated in
Python, float32 arithmetic where the C++ uses float.  Returns [(begin_s, end_s, is_sil)]."""
from __future__ import annotations

import numpy as np

F32 = np.float32


def timestamp_onnx(us_alphas, us_cif_peak, n_chars, begin_time=0.0, total_offset=-1.5):
    if n_chars <= 0:
        return []
    START_END_THRESHOLD, MAX_TOKEN_DURATION = F32(5.0), F32(30.0)
    TIME_RATE = F32(10.0 * 6 / 1000 / 3)
    cif_peak = [F32(v) for v in us_cif_peak]
    num_frames = len(cif_peak)
    fire = [F32(i + total_offset) for i in range(num_frames) if float(cif_peak[i]) > 1.0 - 1e-4]
    if len(fire) != n_chars + 1:
        s = F32(0)
        for a in us_alphas:
            s = F32(s + F32(a))
        scale = F32(s / F32(n_chars + 1))
        if scale == 0:
            return []
        cif_peak = []
        s = F32(0)
        for a in us_alphas:
            a = F32(F32(a) / scale)
            s = F32(s + a)
            cif_peak.append(s)
            if float(s) >= 1.0 - 1e-4:
                s = F32(float(s) - (1.0 - 1e-4))
        idx = len(cif_peak) - 1
        while float(s) >= 1.0 - 1e-4 and idx >= 0:
            if float(cif_peak[idx]) < 1.0 - 1e-4:
                cif_peak[idx] = s
                s = F32(float(s) - (1.0 - 1e-4))
            idx -= 1
        fire = [F32(i + total_offset) for i in range(min(num_frames, len(cif_peak))) if float(cif_peak[i]) > 1.0 - 1e-4]
    if not fire:
        return []
    out = []
    if fire[0] > START_END_THRESHOLD:
        out.append([F32(0.0), F32(fire[0] * TIME_RATE), True])
    for i in range(len(fire) - 1):
        if i == len(fire) - 2 or F32(fire[i + 1] - fire[i]) < MAX_TOKEN_DURATION:
            out.append([F32(fire[i] * TIME_RATE), F32(fire[i + 1] * TIME_RATE), False])
        else:
            split = F32(fire[i] + MAX_TOKEN_DURATION)
            out.append([F32(fire[i] * TIME_RATE), F32(split * TIME_RATE), False])
            out.append([F32(split * TIME_RATE), F32(fire[i + 1] * TIME_RATE), True])
    if not out:
        return []
    if F32(num_frames - fire[-1]) > START_END_THRESHOLD:
        end = F32((float(num_frames) + float(fire[-1])) / 2.0)
        out[-1][1] = F32(end * TIME_RATE)
        out.append([F32(end * TIME_RATE), F32(F32(num_frames) * TIME_RATE), True])
    else:
        out[-1][1] = F32(F32(num_frames) * TIME_RATE)
    if begin_time:
        for t in out:
            t[0] = F32(float(t[0]) + begin_time / 1000.0)
            t[1] = F32(float(t[1]) + begin_time / 1000.0)
    return [(float(a), float(b), bool(c)) for a, b, c in out]


def _is_chinese(ch: str) -> bool:
    b = ch.encode("utf-8")
    if len(b) != 3:
        return False
    return 19968 <= ord(ch) <= 40959 if len(ch) == 1 else False


def post_process(raw_char, stamps):
    """funasr::PostProcess (util.cpp:720-836): hypothesis tokens + (begin, end) stamps -> "text | b0, e0,b1, e1"."""
    merged, words = [], []
    is_pre_english = is_combining = False
    combine, begin = "", -1.0
    n = len(raw_char)
    for i, word in enumerate(raw_char):
        if word in ("<s>", "</s>", "<unk>"):
            continue
        if "@@" in word:
            if i == n - 1 or _is_chinese(raw_char[i + 1]):
                word = word[:-2] + " "
                if is_combining:
                    combine += word
                    is_combining = False
                    word, combine = combine, ""
            else:
                combine += word[:-2]
                if not is_combining:
                    begin = stamps[i][0]
                is_combining = True
                continue
        elif is_combining:
            combine += word
            is_combining = False
            word, combine = combine, ""
        if _is_chinese(word):
            words.append(word)
            merged.append([stamps[i][0], stamps[i][1]])
            is_pre_english = False
        else:
            if is_pre_english:
                words.append(" ")
            words.append(word)
            begin = stamps[i][0] if begin == -1.0 else begin
            merged.append([begin, stamps[i][1]])
            begin = -1.0
            is_pre_english = True
    stamp_str = ""
    for i, (b, e) in enumerate(merged):
        stamp_str += f"{float(np.float32(b)):.6f}, {float(np.float32(e)):.6f}"
        if i != len(merged) - 1:
            stamp_str += ","
    return "".join(words) + " | " + stamp_str
