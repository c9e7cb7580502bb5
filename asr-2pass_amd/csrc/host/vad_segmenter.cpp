#include "vad_segmenter.h"

#include <algorithm>
#include <cmath>

namespace pfhip_host {

VadSegmenter::VadSegmenter() { ResetAll(); }

void VadSegmenter::WinReset() {
  for (int& v : win_) v = 0;
  win_sum_ = 0; win_pos_ = 0; win_prev_ = Frame::kSil;
}

// WindowDetector::DetectOneFrame (e2e-vad.h:232-261): 200-ms majority window with hysteresis 150/150 ms
VadSegmenter::Change VadSegmenter::WinPush(Frame f) {
  if (f == Frame::kInvalid) return Change::kInvalid;
  const int v = f == Frame::kSpeech ? 1 : 0;
  win_sum_ += v - win_[win_pos_];
  win_[win_pos_] = v;
  win_pos_ = (win_pos_ + 1) % kWin;
  if (win_prev_ == Frame::kSil && win_sum_ >= kSil2Speech) { win_prev_ = Frame::kSpeech; return Change::kSil2Speech; }
  if (win_prev_ == Frame::kSpeech && win_sum_ <= kSpeech2Sil) { win_prev_ = Frame::kSil; return Change::kSpeech2Sil; }
  return win_prev_ == Frame::kSil ? Change::kSil2Sil : Change::kSpeech2Speech;
}

// ResetDetection (e2e-vad.h:421-431)
void VadSegmenter::ResetDetection() {
  sil_run_ = 0; last_speech_frame_ = 0; last_sil_frame_ = -1; start_frame_ = -1; end_frame_ = -1;
  machine_ = Machine::kNoStart;
  WinReset();
}

// AllResetDetection (e2e-vad.h:390-419)
void VadSegmenter::ResetAll() {
  data_start_frame_ = 0; frm_cnt_ = 0; n_end_detected_ = 0; noise_db_ = -100.0f; next_seg_ = true;
  pieces_.clear(); piece_off_ = 0;
  max_end_sil_thresh_ = max_end_silence_time_ - speech_to_sil_time_thres_;
  speech_noise_thres_ = default_speech_noise_thres_;
  sil_.clear(); idx_pre_chunk_ = 0; db_.clear();
  // NB: the reference re-declares data_buf_size / data_buf_all_size as locals in AllResetDetection (:415-416), so the
  // members keep their values across utterances; the sample accounting below is kept the same way on purpose.
  ResetDetection();
}

int VadSegmenter::StartLatencyFrames() const {        // LatencyFrmNumAtStartPoint (:582-588)
  int n = kWin;
  if (do_extend_) n += lookback_time_start_point_ / frame_in_ms_;
  return n;
}

// ComputeDecibel (:433-449): 10*log10(sum x^2 + 1e-6) over 25-ms frames at 10-ms shift
void VadSegmenter::AppendDecibel(const float* w, int n) {
  const int flen = frame_length_ms_ * sample_rate_ / 1000, fshift = frame_in_ms_ * sample_rate_ / 1000;
  if (buf_all_ == 0) { buf_all_ = n; buf_size_ = n; } else { buf_all_ += n; }
  for (int off = 0; off + flen - 1 < n; off += fshift) {
    float s = 0.0f;
    for (int i = 0; i < flen; ++i) s += w[off + i] * w[off + i];
    db_.push_back((float)(10 * std::log10(s + 0.000001)));
  }
}

// GetFrameState (:591-640)
VadSegmenter::Frame VadSegmenter::Classify(int t) {
  const float cur_db = db_[t];
  const float snr = cur_db - noise_db_;
  if (cur_db < decibel_thres_) {
    Step(Frame::kSil, t, false);        // the reference feeds the frame twice in this branch (:596-599)
    return Frame::kSil;
  }
  float sum_score = sil_[t - idx_pre_chunk_];
  const float noise_prob = std::log(sum_score) * speech_2_noise_ratio_;
  sum_score = 1.0f - sum_score;
  const float speech_prob = std::log(sum_score);
  if (std::exp(speech_prob) >= std::exp(noise_prob) + speech_noise_thres_) {
    return (snr >= snr_thres_ && cur_db >= decibel_thres_) ? Frame::kSpeech : Frame::kSil;
  }
  if (noise_db_ < -99.9) noise_db_ = cur_db;
  else noise_db_ = (cur_db + noise_db_ * (noise_frame_num_used_for_snr_ - 1)) / noise_frame_num_used_for_snr_;
  return Frame::kSil;
}

// PopDataBufTillFrame (:459-468)
void VadSegmenter::DropUntil(int frame) {
  const int fs = frame_in_ms_ * sample_rate_ / 1000;
  while (data_start_frame_ < frame) {
    if (buf_size_ >= fs) {
      data_start_frame_ += 1;
      buf_size_ = buf_all_ - (long)data_start_frame_ * fs;
    } else {
      break;      // the reference spins here forever; with consistent inputs the branch is never reached
    }
  }
}

// PopDataToOutputBuf (:470-521), without the sample copies the reference no longer makes either
void VadSegmenter::Emit(int start_frm, int n_frm, bool is_start, bool is_end, bool sent_end) {
  (void)sent_end;
  DropUntil(start_frm);
  if (pieces_.empty() || is_start) {
    Piece p;
    p.start_ms = start_frm * frame_in_ms_;
    p.end_ms = p.start_ms;
    pieces_.push_back(p);
  }
  Piece& cur = pieces_.back();
  data_start_frame_ += n_frm;
  cur.end_ms = (start_frm + n_frm) * frame_in_ms_;
  if (is_start) cur.has_start = true;
  if (is_end) cur.has_end = true;
}

void VadSegmenter::SawSilence(int frame) {            // OnSilenceDetected (:523-530)
  last_sil_frame_ = frame;
  if (machine_ == Machine::kNoStart) DropUntil(frame);
}

void VadSegmenter::SawVoice(int frame) {              // OnVoiceDetected (:532-535)
  last_speech_frame_ = frame;
  Emit(frame, 1, false, false, false);
}

void VadSegmenter::VoiceStart(int frame, bool fake) { // OnVoiceStart (:537-549)
  if (start_frame_ == -1) start_frame_ = frame;
  if (!fake && machine_ == Machine::kNoStart) Emit(start_frame_, 1, true, false, false);
}

void VadSegmenter::VoiceEnd(int frame, bool fake, bool last) {   // OnVoiceEnd (:552-568)
  for (int t = last_speech_frame_ + 1; t < frame; ++t) SawVoice(t);
  if (end_frame_ == -1) end_frame_ = frame;
  if (!fake) Emit(end_frame_, 1, false, true, last);
  ++n_end_detected_;
}

void VadSegmenter::EndIfLast(bool last, int idx) {    // MaybeOnVoiceEndIfLastFrame (:570-575)
  if (last) { VoiceEnd(idx, false, true); machine_ = Machine::kEndFound; }
}

// DetectOneFrame (:672-781)
void VadSegmenter::Step(Frame f, int idx, bool last) {
  const Change ch = WinPush(f);
  const int shift = frame_in_ms_;
  const bool too_long = machine_ == Machine::kInSpeech && idx - start_frame_ + 1 > max_single_segment_time_ / shift;
  auto in_speech_default = [&]() {
    if (too_long) { VoiceEnd(idx, false, false); machine_ = Machine::kEndFound; }
    else if (!last) SawVoice(idx);
    else EndIfLast(last, idx);
  };
  switch (ch) {
    case Change::kSil2Speech:
      sil_run_ = 0;
      if (machine_ == Machine::kNoStart) {
        const int sf = std::max(data_start_frame_, idx - StartLatencyFrames());
        VoiceStart(sf, false);
        machine_ = Machine::kInSpeech;
        for (int t = sf + 1; t <= idx; ++t) SawVoice(t);
      } else if (machine_ == Machine::kInSpeech) {
        for (int t = last_speech_frame_ + 1; t < idx; ++t) SawVoice(t);
        in_speech_default();
      }
      break;
    case Change::kSpeech2Sil:
    case Change::kSpeech2Speech:
      sil_run_ = 0;
      if (machine_ == Machine::kInSpeech) in_speech_default();
      break;
    case Change::kSil2Sil:
      ++sil_run_;
      if (machine_ == Machine::kNoStart) {
        if ((detect_mode_ == 0 && sil_run_ * shift > max_start_silence_time_) || (last && n_end_detected_ == 0)) {
          for (int t = last_sil_frame_ + 1; t < idx; ++t) SawSilence(t);
          VoiceStart(0, true);
          VoiceEnd(0, true, false);
          machine_ = Machine::kEndFound;
        } else if (idx >= StartLatencyFrames()) {
          SawSilence(idx - StartLatencyFrames());
        }
      } else if (machine_ == Machine::kInSpeech) {
        if (sil_run_ * shift >= max_end_sil_thresh_) {
          int back = max_end_sil_thresh_ / shift;
          if (do_extend_) back = std::max(0, back - lookahead_time_end_point_ / shift - 1);
          VoiceEnd(idx - back, false, false);
          machine_ = Machine::kEndFound;
        } else if (too_long) {
          VoiceEnd(idx, false, false);
          machine_ = Machine::kEndFound;
        } else if (do_extend_ && !last) {
          if (sil_run_ <= lookahead_time_end_point_ / shift) SawVoice(idx);
        } else {
          EndIfLast(last, idx);
        }
      }
      break;
    case Change::kInvalid:
      break;
  }
  if (machine_ == Machine::kEndFound && detect_mode_ == 1) ResetDetection();
}

std::vector<VadSegment> VadSegmenter::Feed(const float* sil_prob, int T, const float* waveform, int n_samples,
                                           bool is_final, bool online, int max_end_sil, int max_single_segment_time,
                                           float speech_noise_thres, int sample_rate) {
  max_end_sil_thresh_ = max_end_sil - speech_to_sil_time_thres_;
  max_single_segment_time_ = max_single_segment_time;
  speech_noise_thres_ = speech_noise_thres;
  sample_rate_ = sample_rate;
  AppendDecibel(waveform, n_samples);
  nn_eval_block_size_ = T;                                   // ComputeScores (:451-455)
  frm_cnt_ += T;
  sil_.assign(sil_prob, sil_prob + T);
  // DetectCommonFrames / DetectLastFrames (:642-670)
  if (machine_ != Machine::kEndFound) {
    for (int i = T - 1; i >= 0; --i) {
      const int t = frm_cnt_ - 1 - i;
      const Frame f = Classify(t);
      Step(f, t, is_final && i == 0);
    }
    if (!is_final) idx_pre_chunk_ += T;
  }
  std::vector<VadSegment> out;
  for (size_t i = piece_off_; i < pieces_.size(); ++i) {     // (:320-352)
    const Piece& p = pieces_[i];
    int s, e;
    if (online) {
      if (!p.has_start) continue;
      if (!next_seg_ && !p.has_end) continue;
      s = next_seg_ ? p.start_ms : -1;
      if (p.has_end) { e = p.end_ms; next_seg_ = true; ++piece_off_; }
      else { e = -1; next_seg_ = false; }
    } else {
      if (!is_final && (!p.has_start || !p.has_end)) continue;
      s = p.start_ms; e = p.end_ms; ++piece_off_;
    }
    out.push_back({s, e});
  }
  if (is_final) ResetAll();
  return out;
}

}  // namespace pfhip_host
