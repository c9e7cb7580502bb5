// Host side of the VAD path in the reference's own language: `FsmnVadHip` / `FsmnVadOnlineHip`, the siblings of
// `funasr::FsmnVad` / `funasr::FsmnVadOnline` behind `class funasr::VadModel` (onnxruntime/include/vad-model.h:10-19).
// Infer = device forward (pfhip_vad_forward_sil / pfhip_vad_stream_infer) + the end-point detector on the host
// (pfhip_vadseg_feed), exactly the split of fsmn-vad.cpp:240-256 and fsmn-vad-online.cpp:135-151.
//
// Built stand-alone against the small interface below; inside the reference tree define PFHIP_WITH_FUNASR to derive from
// the real funasr::VadModel.
#pragma once
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/pfhip.h"

#ifdef PFHIP_WITH_FUNASR
#include "vad-model.h"
namespace funasr {
using VadModelHipBase = VadModel;
}
#else
namespace funasr {
class VadModelHipBase {                       // vad-model.h:10-19, signature for signature
 public:
  virtual ~VadModelHipBase() {}
  virtual void InitVad(const std::string& vad_model, const std::string& vad_cmvn, const std::string& vad_config, int thread_num) = 0;
  virtual std::vector<std::vector<int>> Infer(std::vector<float>& waves, bool input_finished = true) = 0;
  virtual int GetVadSampleRate() = 0;
  virtual void SetConfig(int vad_tail_sil, int vad_max_len) = 0;
};
}  // namespace funasr
#endif

namespace funasr {

class FsmnVadHip : public VadModelHipBase {
 public:
  FsmnVadHip() {}
  ~FsmnVadHip() override;
  // vad_model = <vad-dir>/model.onnx | model_quant.onnx, vad_cmvn = <vad-dir>/am.mvn, vad_config = <vad-dir>/config.yaml as the
  // reference passes them (offline-stream.cpp:12-26) — or a container pair (x.pfhip.bin + its JSON manifest).  model_conf's
  // max_end_silence_time / max_single_segment_time / speech_noise_thres are read as fsmn-vad.cpp:36-38 does (defaults
  // 800 / 60000 / 0.9).  Exits on a load failure like the reference (:30-33, :57-60).
  void InitVad(const std::string& vad_model, const std::string& vad_cmvn, const std::string& vad_config, int thread_num) override;
  // fsmn-vad.cpp:240-256: scores of the whole buffer, a FRESH detector run with is_final = true, online = false.
  // The network caches carry over between calls unless input_finished (Forward, :129-134); Reset() zeroes them.
  std::vector<std::vector<int>> Infer(std::vector<float>& waves, bool input_finished = true) override;
  int GetVadSampleRate() override { return 16000; }
  void SetConfig(int vad_tail_sil, int vad_max_len) override { vad_silence_duration_ = vad_tail_sil; vad_max_len_ = vad_max_len; }
  void Reset();
  void SetDevice(int device) { device_ = device; }
  pfhip_vad* Handle() const { return handle_; }
  int vad_silence_duration_ = 800, vad_max_len_ = 60000;
  float vad_speech_noise_thres_ = 0.9f;

 private:
  pfhip_vad* handle_ = nullptr;
  std::mutex mu_;                        // the offline object keeps per-file caches: one file at a time
  int device_ = 0;
};

class FsmnVadOnlineHip : public VadModelHipBase {
 public:
  explicit FsmnVadOnlineHip(FsmnVadHip* fsmnvad_handle);      // fsmn-vad-online.cpp:206-219: shares the session, copies the config
  ~FsmnVadOnlineHip() override;
  void InitVad(const std::string&, const std::string&, const std::string&, int) override {}     // fsmn-vad-online.h:31
  std::vector<std::vector<int>> Infer(std::vector<float>& waves, bool input_finished = true) override;
  int GetVadSampleRate() override { return 16000; }
  void SetConfig(int vad_tail_sil, int vad_max_len) override { vad_silence_duration_ = vad_tail_sil; vad_max_len_ = vad_max_len; }
  void Reset();                                               // Reset + ResetCache + a fresh detector (:160-163)
  bool ok() const { return stream_ != nullptr && scorer_ != nullptr; }

 private:
  pfhip_vad_stream* stream_ = nullptr;
  pfhip_vadseg* scorer_ = nullptr;
  int vad_silence_duration_, vad_max_len_;
  float vad_speech_noise_thres_;
};

}  // namespace funasr
