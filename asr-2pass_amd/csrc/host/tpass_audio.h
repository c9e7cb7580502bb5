// The 2-pass audio bookkeeping of the reference on the host — `funasr::Audio` as `FunTpassInferBuffer` uses it
// (onnxruntime/src/funasrruntime.cpp:491-646):
//   LoadPcmwavOnline  audio.cpp:821-857     s16 LE -> f32 / 32768, appended to all_samples
//   Split             audio.cpp:1257-1424   online VAD segments -> chunks for the streaming model (asr_online_queue) and whole
//                                           segments for the offline model (asr_offline_queue); all_samples keeps a 2-s cache
//   FetchChunck / FetchTpass / ResetIndex   audio.cpp:971-991, audio.h:106-112
// Index arithmetic and queueing only; the VAD is passed in as a callable.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <vector>

namespace pfhip_host {

enum AsrType { kAsrOffline = 0, kAsrOnline = 1, kAsrTwoPass = 2 };     // funasrruntime.h:48-52

struct TpassFrame {
  std::vector<float> data;
  bool is_final = false;
  int global_start = 0, global_end = 0;      // ms on the connection's time axis
};

class TpassAudio {
 public:
  using VadInfer = std::function<std::vector<std::vector<int>>(std::vector<float>& waves, bool input_finished)>;
  explicit TpassAudio(int sample_rate = 16000) : dest_sample_rate_(sample_rate) { ResetIndex(); }
  bool LoadPcmwavOnline(const char* buf, int n_buf_len);
  void Split(const VadInfer& vad, int chunk_len, bool input_finished, AsrType asr_mode);
  bool FetchChunck(TpassFrame& out);
  bool FetchTpass(TpassFrame& out);
  void ResetIndex();
  float GetTimeLen() const { return (float)speech_data_.size() / (float)dest_sample_rate_; }

 private:
  TpassFrame MakeFrame(int start, int n, bool is_final, int gs, int ge) const;
  int dest_sample_rate_;
  std::vector<float> speech_data_, all_samples_;
  std::deque<int> frame_queue_;
  std::deque<TpassFrame> asr_online_queue_, asr_offline_queue_;
  int speech_start_ = -1, speech_end_ = 0, speech_offline_start_ = -1, offset_ = 0;
};

}  // namespace pfhip_host
