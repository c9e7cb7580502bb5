// Host-side end-point detector that turns FSMN-VAD frame scores into speech segments — behaviourally the
// reference's `funasr::E2EVadModel` (onnxruntime/src/e2e-vad.h:268-783) with its `WindowDetector` (:181-266) and
// default `VADXOptions` (:46-138).  It is sequential integer/threshold logic over ~100 frames per second and
// stays on the host like in the reference; the only per-frame quantity it needs from
// the network is the silence posterior (class 0, e2e-vad.h:103,607-609), so the device hands over [T] floats
// instead of the [T, 248] score matrix.
#pragma once
#include <vector>

namespace pfhip_host {

struct VadSegment { int start_ms; int end_ms; };   // -1 = "not yet known" in online mode (e2e-vad.h:322-333)

class VadSegmenter {
 public:
  VadSegmenter();
  // One call = one E2EVadModel::operator() (e2e-vad.h:303-362): sil_prob[T] are the class-0 scores of the new
  // frames, waveform the samples they were computed from (for the per-frame energy, :433-449).
  std::vector<VadSegment> Feed(const float* sil_prob, int T, const float* waveform, int n_samples, bool is_final,
                               bool online, int max_end_sil = 800, int max_single_segment_time = 15000,
                               float speech_noise_thres = 0.8f, int sample_rate = 16000);
  void ResetAll();

 private:
  enum class Machine { kNoStart = 1, kInSpeech = 2, kEndFound = 3 };
  enum class Frame { kInvalid = -1, kSil = 0, kSpeech = 1 };
  enum class Change { kSpeech2Speech, kSpeech2Sil, kSil2Sil, kSil2Speech, kInvalid };
  struct Piece { int start_ms = 0, end_ms = 0; bool has_start = false, has_end = false; };

  // options (VADXOptions defaults, e2e-vad.h:78-107)
  int sample_rate_ = 16000, detect_mode_ = 1, max_end_silence_time_ = 800, max_start_silence_time_ = 3000;
  int speech_to_sil_time_thres_ = 150, do_extend_ = 1, lookback_time_start_point_ = 200, lookahead_time_end_point_ = 100;
  int max_single_segment_time_ = 15000, nn_eval_block_size_ = 8, noise_frame_num_used_for_snr_ = 100;
  float speech_2_noise_ratio_ = 1.0f, snr_thres_ = -100.0f, decibel_thres_ = -100.0f, default_speech_noise_thres_ = 0.9f;
  int frame_in_ms_ = 10, frame_length_ms_ = 25;

  // sliding-window smoother (WindowDetector(200,150,150,10), e2e-vad.h:364)
  static constexpr int kWin = 20, kSil2Speech = 15, kSpeech2Sil = 15;
  int win_[kWin]; int win_sum_ = 0, win_pos_ = 0; Frame win_prev_ = Frame::kSil;
  void WinReset();
  Change WinPush(Frame f);

  // detector state
  int data_start_frame_ = 0, frm_cnt_ = 0, last_speech_frame_ = 0, last_sil_frame_ = -1, sil_run_ = 0;
  Machine machine_ = Machine::kNoStart;
  int start_frame_ = -1, end_frame_ = -1, n_end_detected_ = 0;
  float noise_db_ = -100.0f;
  bool next_seg_ = true;
  std::vector<Piece> pieces_; size_t piece_off_ = 0;
  int max_end_sil_thresh_ = 650; float speech_noise_thres_ = 0.9f;
  std::vector<float> sil_; int idx_pre_chunk_ = 0;
  std::vector<float> db_;
  long buf_size_ = 0, buf_all_ = 0;

  void ResetDetection();
  int StartLatencyFrames() const;
  void AppendDecibel(const float* w, int n);
  Frame Classify(int t);
  void Step(Frame f, int idx, bool last);
  void DropUntil(int frame);
  void Emit(int start_frm, int n_frm, bool is_start, bool is_end, bool sent_end);
  void SawSilence(int frame);
  void SawVoice(int frame);
  void VoiceStart(int frame, bool fake);
  void VoiceEnd(int frame, bool fake, bool last);
  void EndIfLast(bool last, int idx);
};

}  // namespace pfhip_host
