"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

The 2-pass audio bookkeeping of the reference, restated on lists (host logic, no arithmetic beyond indices):
  Audio::LoadPcmwavOnline  onnxruntime/src/audio.cpp:821-857   (append to all_samples, one frame_queue entry per call)
  Audio::Split (online)    onnxruntime/src/audio.cpp:1257-1424 (VAD segments -> asr_online_queue chunks of chunk_len and
                                                               asr_offline_queue segments; all_samples trimmed to a 2-s cache)
  Audio::FetchChunck / FetchTpass / ResetIndex   audio.cpp:971-991, audio.h:106-112
Frames are dicts: data (float32 array), is_final, global_start, global_end (ms).
"""
from __future__ import annotations

import numpy as np

SEG_SAMPLE = 16
ASR_OFFLINE, ASR_ONLINE, ASR_TWO_PASS = 0, 1, 2


class TpassAudio:
    def __init__(self, sample_rate=16000):
        self.dest_sample_rate = sample_rate
        self.speech_data = np.zeros(0, np.float32)
        self.frame_queue = []
        self.asr_online_queue = []
        self.asr_offline_queue = []
        self.ResetIndex()

    def ResetIndex(self):
        self.speech_start, self.speech_end, self.speech_offline_start, self.offset = -1, 0, -1, 0
        self.all_samples = np.zeros(0, np.float32)

    def LoadPcmwavOnline(self, pcm_f32):
        self.speech_data = np.asarray(pcm_f32, np.float32)
        self.all_samples = np.concatenate([self.all_samples, self.speech_data]).astype(np.float32)
        self.frame_queue.append(len(self.speech_data))

    def _frame(self, start, n, is_final, gs, ge):
        a = start - self.offset
        return dict(data=self.all_samples[a:a + n].copy(), is_final=is_final, global_start=gs, global_end=ge)

    def Split(self, vad_infer, chunk_len, input_finished, asr_mode):
        sp_len = self.frame_queue.pop(0)
        vad_segments = vad_infer(self.speech_data[:sp_len], input_finished)
        self.speech_end += sp_len // SEG_SAMPLE
        step = chunk_len
        if len(vad_segments) == 0:
            if self.speech_start != -1:
                start, end = self.speech_start * SEG_SAMPLE, self.speech_end * SEG_SAMPLE
                if asr_mode != ASR_OFFLINE and end - start >= step:
                    self.asr_online_queue.append(self._frame(start, step, False, self.speech_start, self.speech_start + step // SEG_SAMPLE))
                    self.speech_start += step // SEG_SAMPLE
        else:
            for seg in vad_segments:
                s_i = seg[0] if seg[0] != -1 else -1
                e_i = seg[1] if seg[1] != -1 else -1
                if s_i != -1 and e_i != -1:                              # [1, 100]
                    start, end = s_i * SEG_SAMPLE, e_i * SEG_SAMPLE
                    if asr_mode != ASR_OFFLINE:
                        self.asr_online_queue.append(self._frame(start, end - start, True, s_i, e_i))
                    if asr_mode != ASR_ONLINE:
                        self.asr_offline_queue.append(self._frame(start, end - start, True, s_i, e_i))
                    self.speech_start = -1
                    self.speech_offline_start = -1
                elif s_i != -1:                                          # [70, -1]
                    self.speech_start = s_i
                    self.speech_offline_start = s_i
                    start, end = self.speech_start * SEG_SAMPLE, self.speech_end * SEG_SAMPLE
                    if asr_mode != ASR_OFFLINE and end - start >= step:
                        self.asr_online_queue.append(self._frame(start, step, False, self.speech_start, self.speech_start + step // SEG_SAMPLE))
                        self.speech_start += step // SEG_SAMPLE
                elif e_i != -1:                                          # [-1, 100]
                    if self.speech_start == -1 or self.speech_offline_start == -1:
                        self.speech_start = 0                            # the reference logs an error and continues (:1354-1357)
                    start = self.speech_start * SEG_SAMPLE
                    offline_start = self.speech_offline_start * SEG_SAMPLE
                    end = e_i * SEG_SAMPLE
                    buff_len = end - start
                    step = chunk_len
                    if asr_mode != ASR_ONLINE:
                        self.asr_offline_queue.append(self._frame(offline_start, end - offline_start, True, self.speech_offline_start, e_i))
                    if asr_mode != ASR_OFFLINE:
                        if buff_len > 0:
                            sample_offset = 0
                            while sample_offset < buff_len:
                                is_final = False
                                if sample_offset + step >= buff_len - 1:
                                    step = buff_len - sample_offset
                                    is_final = True
                                gs = (start + sample_offset) // SEG_SAMPLE
                                self.asr_online_queue.append(self._frame(start + sample_offset, step, is_final, gs, gs + step // SEG_SAMPLE))
                                sample_offset += min(step, buff_len - sample_offset)
                        else:
                            self.asr_online_queue.append(dict(data=np.zeros(0, np.float32), is_final=True,
                                                              global_start=self.speech_start, global_end=e_i))
                    self.speech_start = -1
                    self.speech_offline_start = -1
        # erase all_samples (:1407-1422)
        vector_cache = self.dest_sample_rate * 2
        if self.speech_offline_start == -1:
            if len(self.all_samples) > vector_cache:
                erase = len(self.all_samples) - vector_cache
                self.all_samples = self.all_samples[erase:]
                self.offset += erase
        else:
            offline_start = self.speech_offline_start * SEG_SAMPLE
            if offline_start - self.offset > vector_cache:
                erase = offline_start - self.offset - vector_cache
                self.all_samples = self.all_samples[erase:]
                self.offset += erase

    def FetchChunck(self):
        return self.asr_online_queue.pop(0) if self.asr_online_queue else None

    def FetchTpass(self):
        return self.asr_offline_queue.pop(0) if self.asr_offline_queue else None
