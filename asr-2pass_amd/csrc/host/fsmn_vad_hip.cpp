#include "fsmn_vad_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace funasr {
namespace {

bool ReadFile(const std::string& path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

// a top-level-ish numeric value of the manifest ("key": number); absent -> fallback
double ManifestNumber(const std::string& man, const char* key, double fallback) {
  const std::string k = std::string("\"") + key + "\"";
  const size_t at = man.find(k);
  if (at == std::string::npos) return fallback;
  const size_t colon = man.find(':', at + k.size());
  if (colon == std::string::npos) return fallback;
  char* end = nullptr;
  const double v = std::strtod(man.c_str() + colon + 1, &end);
  return end == man.c_str() + colon + 1 ? fallback : v;
}

std::vector<std::vector<int>> Pairs(const std::vector<int32_t>& p, int n) {
  std::vector<std::vector<int>> out;
  for (int i = 0; i < n; ++i) out.push_back({p[2 * i], p[2 * i + 1]});
  return out;
}

}  // namespace

FsmnVadHip::~FsmnVadHip() {
  if (handle_) pfhip_vad_destroy(handle_);
}

void FsmnVadHip::InitVad(const std::string& vad_model, const std::string& vad_cmvn, const std::string& vad_config, int thread_num) {
  (void)thread_num;
  // the reference's own strings (offline-stream.cpp:12-26, tpass-stream.cpp:12-26): <vad-dir>/model.onnx | model_quant.onnx, am.mvn,
  // config.yaml — ReadModel + LoadCmvn + LoadConfigFromYaml (fsmn-vad.cpp:10-50) — or a container pair x.pfhip.bin / .json
  pfhip_container* c = nullptr;
  std::string man;
  if (pfhip_read_model_files("vad", vad_model.c_str(), nullptr, nullptr, vad_cmvn.c_str(), vad_config.c_str(), &c) == PFHIP_OK) {
    size_t bytes = 0;
    const float* blob = pfhip_container_blob(c, &bytes);
    man = pfhip_container_manifest(c);
    if (pfhip_vad_create_from_memory(blob, bytes, man.c_str(), device_, &handle_) != PFHIP_OK) handle_ = nullptr;
    pfhip_container_free(c);
  }
  if (!handle_) {
    std::fprintf(stderr, "Error when load vad hip model: %s\n", pfhip_last_error());       // fsmn-vad.cpp:57-60
    std::exit(-1);
  }
  vad_silence_duration_ = (int)ManifestNumber(man, "max_end_silence_time", 800);             // model_conf (fsmn-vad.cpp:36-38)
  vad_max_len_ = (int)ManifestNumber(man, "max_single_segment_time", 60000);
  vad_speech_noise_thres_ = (float)ManifestNumber(man, "speech_noise_thres", 0.9);
}

void FsmnVadHip::Reset() {
  std::lock_guard<std::mutex> lk(mu_);
  (void)pfhip_vad_reset(handle_);
}

std::vector<std::vector<int>> FsmnVadHip::Infer(std::vector<float>& waves, bool input_finished) {
  const int n = (int)waves.size();
  const int max_frames = n >= 400 ? (n - 400) / 160 + 1 : 0;
  if (max_frames <= 0) return {};                           // no full window: no features (:245-247)
  std::vector<float> sil((size_t)max_frames + 8);
  int T = 0;
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (pfhip_vad_forward_sil(handle_, waves.data(), n, input_finished ? 1 : 0, sil.data(), sil.size(), &T) != PFHIP_OK) {
      std::fprintf(stderr, "FsmnVadHip::Infer: %s\n", pfhip_last_error());
      return {};
    }
  }
  if (T <= 0) return {};
  pfhip_vadseg* seg = nullptr;
  if (pfhip_vadseg_create(&seg) != PFHIP_OK) return {};
  std::vector<int32_t> pairs((size_t)2 * (T / 2 + 8));
  int n_seg = 0;
  const pfhip_status st = pfhip_vadseg_feed(seg, sil.data(), T, waves.data(), std::min(400 + 160 * (T - 1), n), 1, 0, vad_silence_duration_,
                                            vad_max_len_, vad_speech_noise_thres_, 16000, pairs.data(), (int)pairs.size() / 2, &n_seg);
  pfhip_vadseg_destroy(seg);
  if (st != PFHIP_OK) return {};
  return Pairs(pairs, n_seg);
}

FsmnVadOnlineHip::FsmnVadOnlineHip(FsmnVadHip* v)
    : vad_silence_duration_(v->vad_silence_duration_), vad_max_len_(v->vad_max_len_), vad_speech_noise_thres_(v->vad_speech_noise_thres_) {
  if (pfhip_vad_stream_create(v->Handle(), &stream_) != PFHIP_OK || pfhip_vadseg_create(&scorer_) != PFHIP_OK)
    std::fprintf(stderr, "FsmnVadOnlineHip: %s\n", pfhip_last_error());
}

FsmnVadOnlineHip::~FsmnVadOnlineHip() {
  if (stream_) pfhip_vad_stream_destroy(stream_);
  if (scorer_) pfhip_vadseg_destroy(scorer_);
}

void FsmnVadOnlineHip::Reset() {
  if (stream_) (void)pfhip_vad_stream_reset(stream_);
  if (scorer_) (void)pfhip_vadseg_reset(scorer_);
}

std::vector<std::vector<int>> FsmnVadOnlineHip::Infer(std::vector<float>& waves, bool input_finished) {
  if (!ok()) return {};
  std::vector<float> sil(waves.size() / 160 + 16), wv(waves.size() + 4096);
  int nf = 0, nw = 0;
  if (pfhip_vad_stream_infer(stream_, waves.data(), (int)waves.size(), input_finished ? 1 : 0, sil.data(), sil.size(), &nf, wv.data(),
                             wv.size(), &nw) != PFHIP_OK) {
    std::fprintf(stderr, "FsmnVadOnlineHip::Infer: %s\n", pfhip_last_error());
    return {};
  }
  if (nf == 0) return {};                                   // (:140-146)
  std::vector<int32_t> pairs((size_t)2 * (nf + 8));
  int n_seg = 0;
  if (pfhip_vadseg_feed(scorer_, sil.data(), nf, wv.data(), nw, input_finished ? 1 : 0, 1, vad_silence_duration_, vad_max_len_,
                        vad_speech_noise_thres_, 16000, pairs.data(), (int)pairs.size() / 2, &n_seg) != PFHIP_OK)
    return {};
  return Pairs(pairs, n_seg);
}

}  // namespace funasr
