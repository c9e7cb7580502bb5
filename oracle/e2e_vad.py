"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Python restatement of the reference's VAD end-point detector, class for class and branch for branch:
  VADXOptions      onnxruntime/src/e2e-vad.h:46-138   (defaults only)
  WindowDetector   onnxruntime/src/e2e-vad.h:181-266
  E2EVadModel      onnxruntime/src/e2e-vad.h:268-783  (operator() :303-362, GetFrameState :591-640,
                   DetectOneFrame :672-781, PopDataToOutputBuf :470-521 ...)
Pinned by the reference text it transcribes; float/double types follow the C++ expression types.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
NOSTART, INSPEECH, ENDFOUND = 1, 2, 3
SIL, SPEECH, INVALID = 0, 1, -1
S2S, S2SIL, SIL2SIL, SIL2S, CH_INVALID = 0, 1, 2, 3, 5


class WindowDetector:
    def __init__(self, window_size_ms=200, sil_to_speech_time=150, speech_to_sil_time=150, frame_size_ms=10):
        self.win_size_frame = window_size_ms // frame_size_ms
        self.sil_to_speech_frmcnt_thres = sil_to_speech_time // frame_size_ms
        self.speech_to_sil_frmcnt_thres = speech_to_sil_time // frame_size_ms
        self.Reset()

    def Reset(self):
        self.cur_win_pos = 0
        self.win_sum = 0
        self.win_state = [0] * self.win_size_frame
        self.pre_frame_state = SIL

    def DetectOneFrame(self, frame_state):
        if frame_state == SPEECH:
            cur = 1
        elif frame_state == SIL:
            cur = 0
        else:
            return CH_INVALID
        self.win_sum -= self.win_state[self.cur_win_pos]
        self.win_sum += cur
        self.win_state[self.cur_win_pos] = cur
        self.cur_win_pos = (self.cur_win_pos + 1) % self.win_size_frame
        if self.pre_frame_state == SIL and self.win_sum >= self.sil_to_speech_frmcnt_thres:
            self.pre_frame_state = SPEECH
            return SIL2S
        if self.pre_frame_state == SPEECH and self.win_sum <= self.speech_to_sil_frmcnt_thres:
            self.pre_frame_state = SIL
            return S2SIL
        return SIL2SIL if self.pre_frame_state == SIL else S2S


class E2EVadModel:
    def __init__(self):
        # VADXOptions defaults
        self.sample_rate = 16000
        self.detect_mode = 1
        self.max_end_silence_time = 800
        self.max_start_silence_time = 3000
        self.speech_to_sil_time_thres = 150
        self.speech_2_noise_ratio = F32(1.0)
        self.do_extend = 1
        self.lookback_time_start_point = 200
        self.lookahead_time_end_point = 100
        self.max_single_segment_time = 15000
        self.snr_thres = F32(-100.0)
        self.noise_frame_num_used_for_snr = 100
        self.decibel_thres = F32(-100.0)
        self.opt_speech_noise_thres = F32(0.9)
        self.frame_in_ms = 10
        self.frame_length_ms = 25
        self.windows_detector = WindowDetector(200, 150, 150, 10)
        self.data_buf_size = 0
        self.data_buf_all_size = 0
        self.AllResetDetection()

    def AllResetDetection(self):
        self.data_buf_start_frame = 0
        self.frm_cnt = 0
        self.number_end_time_detected = 0
        self.noise_average_decibel = F32(-100.0)
        self.next_seg = True
        self.output_data_buf = []
        self.output_data_buf_offset = 0
        self.max_end_sil_frame_cnt_thresh = self.max_end_silence_time - self.speech_to_sil_time_thres
        self.speech_noise_thres = self.opt_speech_noise_thres
        self.scores = []
        self.idx_pre_chunk = 0
        self.decibel = []
        # data_buf_size / data_buf_all_size: re-declared as locals in the reference (:415-416) -> members unchanged
        self.ResetDetection()

    def ResetDetection(self):
        self.continous_silence_frame_count = 0
        self.latest_confirmed_speech_frame = 0
        self.lastest_confirmed_silence_frame = -1
        self.confirmed_start_frame = -1
        self.confirmed_end_frame = -1
        self.vad_state_machine = NOSTART
        self.windows_detector.Reset()

    def ComputeDecibel(self, waveform):
        fl = int(self.frame_length_ms * self.sample_rate / 1000)
        fs = int(self.frame_in_ms * self.sample_rate / 1000)
        if self.data_buf_all_size == 0:
            self.data_buf_all_size = len(waveform)
            self.data_buf_size = self.data_buf_all_size
        else:
            self.data_buf_all_size += len(waveform)
        w = np.asarray(waveform, F32)
        off = 0
        while off + fl - 1 < len(w):
            sq = (w[off:off + fl] * w[off:off + fl]).astype(F32)
            s = np.cumsum(sq, dtype=F32)[-1]                    # float accumulator, left to right
            self.decibel.append(F32(10 * math.log10(float(s) + 0.000001)))
            off += fs

    def PopDataBufTillFrame(self, frame_idx):
        fs = int(self.frame_in_ms * self.sample_rate / 1000)
        while self.data_buf_start_frame < frame_idx:
            if self.data_buf_size >= fs:
                self.data_buf_start_frame += 1
                self.data_buf_size = self.data_buf_all_size - self.data_buf_start_frame * fs
            else:
                break

    def PopDataToOutputBuf(self, start_frm, frm_cnt, first_is_start, last_is_end, end_point_is_sent_end):
        self.PopDataBufTillFrame(start_frm)
        if len(self.output_data_buf) == 0 or first_is_start:
            self.output_data_buf.append(dict(start_ms=start_frm * self.frame_in_ms, end_ms=start_frm * self.frame_in_ms,
                                             start=False, end=False))
        cur = self.output_data_buf[-1]
        self.data_buf_start_frame += frm_cnt
        cur["end_ms"] = (start_frm + frm_cnt) * self.frame_in_ms
        if first_is_start:
            cur["start"] = True
        if last_is_end:
            cur["end"] = True

    def OnSilenceDetected(self, valid_frame):
        self.lastest_confirmed_silence_frame = valid_frame
        if self.vad_state_machine == NOSTART:
            self.PopDataBufTillFrame(valid_frame)

    def OnVoiceDetected(self, valid_frame):
        self.latest_confirmed_speech_frame = valid_frame
        self.PopDataToOutputBuf(valid_frame, 1, False, False, False)

    def OnVoiceStart(self, start_frame, fake_result=False):
        if self.confirmed_start_frame == -1:
            self.confirmed_start_frame = start_frame
        if not fake_result and self.vad_state_machine == NOSTART:
            self.PopDataToOutputBuf(self.confirmed_start_frame, 1, True, False, False)

    def OnVoiceEnd(self, end_frame, fake_result, is_last_frame):
        for t in range(self.latest_confirmed_speech_frame + 1, end_frame):
            self.OnVoiceDetected(t)
        if self.confirmed_end_frame == -1:
            self.confirmed_end_frame = end_frame
        if not fake_result:
            self.PopDataToOutputBuf(self.confirmed_end_frame, 1, False, True, is_last_frame)
        self.number_end_time_detected += 1

    def MaybeOnVoiceEndIfLastFrame(self, is_final_frame, cur_frm_idx):
        if is_final_frame:
            self.OnVoiceEnd(cur_frm_idx, False, True)
            self.vad_state_machine = ENDFOUND

    def LatencyFrmNumAtStartPoint(self):
        v = self.windows_detector.win_size_frame
        if self.do_extend:
            v += int(self.lookback_time_start_point / self.frame_in_ms)
        return v

    def GetFrameState(self, t):
        cur_decibel = F32(self.decibel[t])
        cur_snr = F32(cur_decibel - self.noise_average_decibel)
        if cur_decibel < self.decibel_thres:
            self.DetectOneFrame(SIL, t, False)
            return SIL
        sum_score = F32(self.scores[t - self.idx_pre_chunk])
        noise_prob = F32(np.log(sum_score) * self.speech_2_noise_ratio)
        sum_score = F32(F32(1.0) - sum_score)
        speech_prob = F32(np.log(sum_score))
        if F32(np.exp(speech_prob)) >= F32(F32(np.exp(noise_prob)) + self.speech_noise_thres):
            if cur_snr >= self.snr_thres and cur_decibel >= self.decibel_thres:
                return SPEECH
            return SIL
        if float(self.noise_average_decibel) < -99.9:
            self.noise_average_decibel = cur_decibel
        else:
            n = self.noise_frame_num_used_for_snr
            self.noise_average_decibel = F32(F32(cur_decibel + F32(self.noise_average_decibel * F32(n - 1))) / F32(n))
        return SIL

    def DetectOneFrame(self, cur_frm_state, cur_frm_idx, is_final_frame):
        state_change = self.windows_detector.DetectOneFrame(cur_frm_state)
        shift = self.frame_in_ms
        m = self
        if state_change == SIL2S:
            m.continous_silence_frame_count = 0
            if m.vad_state_machine == NOSTART:
                start_frame = max(m.data_buf_start_frame, cur_frm_idx - m.LatencyFrmNumAtStartPoint())
                m.OnVoiceStart(start_frame)
                m.vad_state_machine = INSPEECH
                for t in range(start_frame + 1, cur_frm_idx + 1):
                    m.OnVoiceDetected(t)
            elif m.vad_state_machine == INSPEECH:
                for t in range(m.latest_confirmed_speech_frame + 1, cur_frm_idx):
                    m.OnVoiceDetected(t)
                if cur_frm_idx - m.confirmed_start_frame + 1 > m.max_single_segment_time / shift:
                    m.OnVoiceEnd(cur_frm_idx, False, False)
                    m.vad_state_machine = ENDFOUND
                elif not is_final_frame:
                    m.OnVoiceDetected(cur_frm_idx)
                else:
                    m.MaybeOnVoiceEndIfLastFrame(is_final_frame, cur_frm_idx)
        elif state_change in (S2SIL, S2S):
            m.continous_silence_frame_count = 0
            if m.vad_state_machine == INSPEECH:
                if cur_frm_idx - m.confirmed_start_frame + 1 > m.max_single_segment_time / shift:
                    m.OnVoiceEnd(cur_frm_idx, False, False)
                    m.vad_state_machine = ENDFOUND
                elif not is_final_frame:
                    m.OnVoiceDetected(cur_frm_idx)
                else:
                    m.MaybeOnVoiceEndIfLastFrame(is_final_frame, cur_frm_idx)
        elif state_change == SIL2SIL:
            m.continous_silence_frame_count += 1
            if m.vad_state_machine == NOSTART:
                if (m.detect_mode == 0 and m.continous_silence_frame_count * shift > m.max_start_silence_time) or \
                        (is_final_frame and m.number_end_time_detected == 0):
                    for t in range(m.lastest_confirmed_silence_frame + 1, cur_frm_idx):
                        m.OnSilenceDetected(t)
                    m.OnVoiceStart(0, True)
                    m.OnVoiceEnd(0, True, False)
                    m.vad_state_machine = ENDFOUND
                elif cur_frm_idx >= m.LatencyFrmNumAtStartPoint():
                    m.OnSilenceDetected(cur_frm_idx - m.LatencyFrmNumAtStartPoint())
            elif m.vad_state_machine == INSPEECH:
                if m.continous_silence_frame_count * shift >= m.max_end_sil_frame_cnt_thresh:
                    lookback_frame = m.max_end_sil_frame_cnt_thresh // shift if m.max_end_sil_frame_cnt_thresh >= 0 \
                        else -((-m.max_end_sil_frame_cnt_thresh) // shift)
                    if m.do_extend:
                        lookback_frame -= m.lookahead_time_end_point // shift
                        lookback_frame -= 1
                        lookback_frame = max(0, lookback_frame)
                    m.OnVoiceEnd(cur_frm_idx - lookback_frame, False, False)
                    m.vad_state_machine = ENDFOUND
                elif cur_frm_idx - m.confirmed_start_frame + 1 > m.max_single_segment_time / shift:
                    m.OnVoiceEnd(cur_frm_idx, False, False)
                    m.vad_state_machine = ENDFOUND
                elif m.do_extend and not is_final_frame:
                    if m.continous_silence_frame_count <= m.lookahead_time_end_point // shift:
                        m.OnVoiceDetected(cur_frm_idx)
                else:
                    m.MaybeOnVoiceEndIfLastFrame(is_final_frame, cur_frm_idx)
        if m.vad_state_machine == ENDFOUND and m.detect_mode == 1:
            m.ResetDetection()

    def __call__(self, score_sil, waveform, is_final=False, online=False, max_end_sil=800, max_single_segment_time=15000,
                 speech_noise_thres=0.8, sample_rate=16000):
        """score_sil: class-0 score per frame (the only column GetFrameState reads, sil_pdf_ids = {0})."""
        self.max_end_sil_frame_cnt_thresh = max_end_sil - self.speech_to_sil_time_thres
        self.max_single_segment_time = max_single_segment_time
        self.speech_noise_thres = F32(speech_noise_thres)
        self.sample_rate = sample_rate
        self.ComputeDecibel(waveform)
        n = len(score_sil)
        self.frm_cnt += n
        self.scores = list(score_sil)
        if self.vad_state_machine != ENDFOUND:
            for i in range(n - 1, -1, -1):
                t = self.frm_cnt - 1 - i
                fs = self.GetFrameState(t)
                self.DetectOneFrame(fs, t, is_final and i == 0)
            if not is_final:
                self.idx_pre_chunk += n
        out = []
        for i in range(self.output_data_buf_offset, len(self.output_data_buf)):
            p = self.output_data_buf[i]
            if online:
                if not p["start"]:
                    continue
                if not self.next_seg and not p["end"]:
                    continue
                s = p["start_ms"] if self.next_seg else -1
                if p["end"]:
                    e = p["end_ms"]
                    self.next_seg = True
                    self.output_data_buf_offset += 1
                else:
                    e = -1
                    self.next_seg = False
            else:
                if not is_final and (not p["start"] or not p["end"]):
                    continue
                s, e = p["start_ms"], p["end_ms"]
                self.output_data_buf_offset += 1
            out.append([s, e])
        if is_final:
            self.AllResetDetection()
        return out
