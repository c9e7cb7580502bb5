#include "tpass_audio.h"

#include <algorithm>

namespace pfhip_host {

namespace { constexpr int kSegSample = 16; }      // samples per ms at 16 kHz

void TpassAudio::ResetIndex() {
  speech_start_ = -1; speech_end_ = 0; speech_offline_start_ = -1; offset_ = 0;
  all_samples_.clear();
}

bool TpassAudio::LoadPcmwavOnline(const char* buf, int n_buf_len) {
  const int n = n_buf_len / 2;
  speech_data_.resize((size_t)n);
  const uint8_t* b = reinterpret_cast<const uint8_t*>(buf);
  for (int i = 0; i < n; ++i) {
    const int16_t val = (int16_t)((b[2 * i + 1] << 8) | b[2 * i]);
    speech_data_[i] = (float)val / 32768.0f;
  }
  all_samples_.insert(all_samples_.end(), speech_data_.begin(), speech_data_.end());
  frame_queue_.push_back(n);
  return true;
}

TpassFrame TpassAudio::MakeFrame(int start, int n, bool is_final, int gs, int ge) const {
  TpassFrame f;
  f.is_final = is_final; f.global_start = gs; f.global_end = ge;
  const long a = (long)start - offset_;
  if (n > 0 && a >= 0 && a + n <= (long)all_samples_.size()) f.data.assign(all_samples_.begin() + a, all_samples_.begin() + a + n);
  else if (n > 0) f.data.assign((size_t)n, 0.f);      // outside the 2-s cache (the reference would read out of bounds)
  return f;
}

void TpassAudio::Split(const VadInfer& vad, int chunk_len, bool input_finished, AsrType asr_mode) {
  if (frame_queue_.empty()) return;
  const int sp_len = frame_queue_.front();
  frame_queue_.pop_front();
  std::vector<float> pcm_data(speech_data_.begin(), speech_data_.begin() + sp_len);
  const std::vector<std::vector<int>> vad_segments = vad(pcm_data, input_finished);
  speech_end_ += sp_len / kSegSample;
  int step = chunk_len;
  auto push_online_head = [&]() {              // (:1273-1289, :1335-1351) one full chunk from the running segment, if there is one
    const int start = speech_start_ * kSegSample, end = speech_end_ * kSegSample;
    if (asr_mode != kAsrOffline && end - start >= step) {
      asr_online_queue_.push_back(MakeFrame(start, step, false, speech_start_, speech_start_ + step / kSegSample));
      speech_start_ += step / kSegSample;
    }
  };
  if (vad_segments.empty()) {
    if (speech_start_ != -1) push_online_head();
  } else {
    for (const auto& seg : vad_segments) {
      const int s_i = seg[0] != -1 ? seg[0] : -1, e_i = seg[1] != -1 ? seg[1] : -1;
      if (s_i != -1 && e_i != -1) {                                             // [1, 100]
        const int start = s_i * kSegSample, end = e_i * kSegSample;
        if (asr_mode != kAsrOffline) asr_online_queue_.push_back(MakeFrame(start, end - start, true, s_i, e_i));
        if (asr_mode != kAsrOnline) asr_offline_queue_.push_back(MakeFrame(start, end - start, true, s_i, e_i));
        speech_start_ = -1; speech_offline_start_ = -1;
      } else if (s_i != -1) {                                                   // [70, -1]
        speech_start_ = s_i; speech_offline_start_ = s_i;
        push_online_head();
      } else if (e_i != -1) {                                                   // [-1, 100]
        if (speech_start_ == -1 || speech_offline_start_ == -1) speech_start_ = 0;      // logged as an error upstream (:1354-1357)
        const int start = speech_start_ * kSegSample, offline_start = speech_offline_start_ * kSegSample, end = e_i * kSegSample;
        const int buff_len = end - start;
        step = chunk_len;
        if (asr_mode != kAsrOnline)
          asr_offline_queue_.push_back(MakeFrame(offline_start, end - offline_start, true, speech_offline_start_, e_i));
        if (asr_mode != kAsrOffline) {
          if (buff_len > 0) {
            for (int so = 0; so < buff_len; so += std::min(step, buff_len - so)) {
              bool is_final = false;
              if (so + step >= buff_len - 1) { step = buff_len - so; is_final = true; }
              const int gs = (start + so) / kSegSample;
              asr_online_queue_.push_back(MakeFrame(start + so, step, is_final, gs, gs + step / kSegSample));
            }
          } else {
            asr_online_queue_.push_back(MakeFrame(0, 0, true, speech_start_, e_i));
          }
        }
        speech_start_ = -1; speech_offline_start_ = -1;
      }
    }
  }
  // erase all_samples (:1407-1422): keep 2 s behind the newest sample, or behind the running segment's start
  const int vector_cache = dest_sample_rate_ * 2;
  if (speech_offline_start_ == -1) {
    if ((int)all_samples_.size() > vector_cache) {
      const int erase = (int)all_samples_.size() - vector_cache;
      all_samples_.erase(all_samples_.begin(), all_samples_.begin() + erase);
      offset_ += erase;
    }
  } else {
    const int offline_start = speech_offline_start_ * kSegSample;
    if (offline_start - offset_ > vector_cache) {
      const int erase = offline_start - offset_ - vector_cache;
      all_samples_.erase(all_samples_.begin(), all_samples_.begin() + erase);
      offset_ += erase;
    }
  }
}

bool TpassAudio::FetchChunck(TpassFrame& out) {
  if (asr_online_queue_.empty()) return false;
  out = std::move(asr_online_queue_.front());
  asr_online_queue_.pop_front();
  return true;
}

bool TpassAudio::FetchTpass(TpassFrame& out) {
  if (asr_offline_queue_.empty()) return false;
  out = std::move(asr_offline_queue_.front());
  asr_offline_queue_.pop_front();
  return true;
}

}  // namespace pfhip_host
