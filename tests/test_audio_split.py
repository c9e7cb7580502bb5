"""Audio::Split (online, audio.cpp:1257-1424) restated: hand-derived known answers with a scripted VAD."""
import numpy as np

from oracle import audio_split as A


def run(script, n_calls, mode=A.ASR_TWO_PASS, step=9600):
    au = A.TpassAudio()
    calls = [0]

    def vad(w, fin):
        calls[0] += 1
        return script.get(calls[0] - 1, [])

    pcm = np.arange(step * n_calls, dtype=np.float32)
    log = []
    for j in range(n_calls):
        au.LoadPcmwavOnline(pcm[j * step:(j + 1) * step])
        au.Split(vad, 9600, j == n_calls - 1, mode)
        on, off = [], []
        while (f := au.FetchChunck()) is not None:
            on.append((len(f["data"]), f["is_final"], f["global_start"], f["global_end"], float(f["data"][0]) if len(f["data"]) else None))
        while (f := au.FetchTpass()) is not None:
            off.append((len(f["data"]), f["global_start"], f["global_end"], float(f["data"][0])))
        log.append((on, off))
    return log, au


def test_open_segment_streams_chunks_and_closes_with_a_partial_final_chunk():
    # speech starts at 300 ms (reported in call 1) and ends at 2500 ms (reported in call 4); 600-ms calls
    log, au = run({1: [[300, -1]], 4: [[-1, 2500]]}, 6)
    assert log[0] == ([], [])
    assert log[1] == ([(9600, False, 300, 900, 4800.0)], [])           # one full chunk from sample 300*16
    assert log[2] == ([(9600, False, 900, 1500, 14400.0)], [])         # no VAD news: the running segment keeps streaming
    assert log[3] == ([(9600, False, 1500, 2100, 24000.0)], [])
    assert log[4] == ([(6400, True, 2100, 2500, 33600.0)], [(35200, 300, 2500, 4800.0)])    # tail chunk is final; whole segment offline
    assert log[5] == ([], [])
    assert au.speech_start == -1 and au.offset > 0                      # idle: all_samples trimmed to the 2-s cache


def test_whole_segment_inside_one_call_goes_to_both_queues():
    log, _ = run({2: [[1300, 1700]]}, 4)
    assert log[2] == ([(6400, True, 1300, 1700, 20800.0)], [(6400, 1300, 1700, 20800.0)])


def test_modes_select_the_queues():
    log, _ = run({1: [[300, -1]], 3: [[-1, 2000]]}, 5, mode=A.ASR_ONLINE)
    assert all(off == [] for _, off in log) and sum(len(on) for on, _ in log) >= 3
    log, _ = run({1: [[300, -1]], 3: [[-1, 2000]]}, 5, mode=A.ASR_OFFLINE)
    assert all(on == [] for on, _ in log) and [off for _, off in log if off] == [[(27200, 300, 2000, 4800.0)]]


def test_segment_end_before_the_streamed_position_yields_an_empty_final_chunk():
    # the running segment has streamed up to 1500 ms when the VAD reports its end at 1400 ms
    log, _ = run({1: [[300, -1]], 3: [[-1, 1400]]}, 4)
    assert log[3][0] == [(0, True, 1500, 1400, None)]
