#include "funasrruntime_hip.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <mutex>
#include <numeric>
#include <sstream>

#include "ct_transformer_hip.h"
#include "fsmn_vad_hip.h"
#include "paraformer_hip.h"
#include "tpass_audio.h"

namespace {

struct OfflineStreamHip {
  funasr::ParaformerHip asr;
  std::unique_ptr<funasr::FsmnVadHip> vad;
  std::mutex vad_mu;            // FsmnVad keeps per-file caches: one file at a time, like the reference's per-call Reset
  std::unique_ptr<funasr::PuncModelHipBase> punc;      // OfflineStream::punc_handle (offline-stream.cpp:105-129)
};

struct RecogResult {               // funasr::FUNASR_RECOG_RESULT (com-define.h)
  std::string msg, stamp, tpass_msg;
  float snippet_time = 0.f;
  std::vector<std::vector<int>> seg_ids;
  std::vector<std::pair<int, int>> segs;
  std::vector<int> online_ids;       // ids the streaming chunks of this call emitted (inspection)
};

bool ReadAll(const std::string& path, std::vector<char>& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  f.seekg(0, std::ios::end);
  out.resize((size_t)f.tellg());
  f.seekg(0);
  f.read(out.data(), (std::streamsize)out.size());
  return (bool)f;
}

constexpr int kSegSample = 16;      // samples per ms at 16 kHz (audio.cpp: seg_sample = MODEL_SAMPLE_RATE / 1000)

// Audio::CutSplit (audio.cpp:1172-1240): the FSMN-VAD scores the whole buffer in one pass (it is causal: the reference's
// 1-s slices carry the same cache state), the end-point detector runs on the host, segments come back in ms.
bool CutSplit(OfflineStreamHip* os, std::vector<float>& pcm, int vad_tail_sil, int vad_max_len,
              std::vector<std::pair<int, int>>& frames, std::vector<int>& index_vector) {
  frames.clear();
  index_vector.clear();
  const int n = (int)pcm.size();
  std::vector<std::vector<int>> segs;
  {
    std::lock_guard<std::mutex> lk(os->vad_mu);
    os->vad->Reset();
    os->vad->SetConfig(vad_tail_sil, vad_max_len);
    segs = os->vad->Infer(pcm, true);
  }
  for (const std::vector<int>& sg : segs) frames.emplace_back(sg[0] * kSegSample, std::min(sg[1] * kSegSample, n));
  index_vector.resize(frames.size());
  std::iota(index_vector.begin(), index_vector.end(), 0);
  // audio.cpp:1226-1239: queue by increasing length (ties keep time order)
  std::stable_sort(index_vector.begin(), index_vector.end(), [&](int a, int b) {
    return frames[a].second - frames[a].first < frames[b].second - frames[b].first;
  });
  return true;
}

// Audio::FetchDynamic (audio.cpp:1052-1108): pops one batch off the front of `queue` (positions into `order`)
int FetchDynamic(const std::vector<std::pair<int, int>>& frames, const std::vector<int>& order, size_t& head, int batch_size,
                 std::vector<int>& batch) {
  const long max_acc = 300L * 1000 * kSegSample, max_sent = 60L * 1000 * kSegSample;
  batch.clear();
  long bs_acc = 0, max_len = 0;
  const int max_batch = (int)std::min<size_t>((size_t)batch_size, order.size() - head);
  for (int i = 0; i < max_batch; ++i) {
    const auto& fr = frames[order[head]];
    const long length = fr.second - fr.first;
    if (length >= max_sent) {
      if (bs_acc == 0) { ++bs_acc; batch.push_back(order[head++]); }
      break;
    }
    max_len = std::max(max_len, length);
    if (max_len * (bs_acc + 1) > max_acc) break;
    ++bs_acc;
    batch.push_back(order[head++]);
  }
  return (int)batch.size();
}

}  // namespace

namespace {
bool FileExists(const std::string& path) { return (bool)std::ifstream(path); }
std::string PathAppend(const std::string& dir, const std::string& name) { return dir.empty() || dir.back() == '/' ? dir + name : dir + "/" + name; }
bool IsTrue(std::map<std::string, std::string>& model_path, const char* key) { return model_path.count(key) && model_path[key] == "true"; }
// a directory converted ahead of time holds no model.onnx; the loader finds its container from the ONNX name
bool HasContainer(const std::string& dir, const char* legacy) {
  return FileExists(PathAppend(dir, "model.pfhip.bin")) || FileExists(PathAppend(dir, std::string(legacy) + ".pfhip.bin"));
}
// the VAD block both stream classes open with (offline-stream.cpp:6-29, tpass-stream.cpp:6-29)
std::unique_ptr<funasr::FsmnVadHip> MakeVad(std::map<std::string, std::string>& model_path, int thread_num) {
  if (!model_path.count(VAD_DIR) || model_path[VAD_DIR].empty()) return nullptr;
  const std::string dir = model_path[VAD_DIR];
  const std::string vad_model_path = PathAppend(dir, IsTrue(model_path, VAD_QUANT) ? QUANT_MODEL_NAME : MODEL_NAME);
  const std::string vad_cmvn_path = PathAppend(dir, VAD_CMVN_NAME), vad_config_path = PathAppend(dir, VAD_CONFIG_NAME);
  if (!HasContainer(dir, "vad") && (!FileExists(vad_model_path) || !FileExists(vad_cmvn_path) || !FileExists(vad_config_path))) {
    std::fprintf(stderr, "VAD model file is not exist, skip load vad model.\n");
    return nullptr;
  }
  std::unique_ptr<funasr::FsmnVadHip> vad(new funasr::FsmnVadHip());
  vad->InitVad(vad_model_path, vad_cmvn_path, vad_config_path, thread_num);      // exits on failure
  return vad;
}
}  // namespace

// OfflineStream::OfflineStream (offline-stream.cpp:4-129) on the HIP plug-ins, call for call
FUNASR_HANDLE FunOfflineInit(std::map<std::string, std::string>& model_path, int thread_num, bool use_gpu, int batch_size) {
  auto os = std::make_unique<OfflineStreamHip>();
  os->vad = MakeVad(model_path, thread_num);
  if (model_path.count(MODEL_DIR)) {
    const std::string dir = model_path[MODEL_DIR];
    os->asr.SetBatchSize(batch_size);                                            // :41 (before InitAsr)
    const std::string hw_cpu_model_path = PathAppend(dir, MODEL_EB_NAME), hw_gpu_model_path = PathAppend(dir, TORCH_MODEL_EB_NAME);
    const std::string seg_dict_path = PathAppend(dir, MODEL_SEG_DICT);
    if (FileExists(hw_cpu_model_path)) {                                         // :63-67 if model_eb.onnx exist, hotword enabled
      os->asr.InitHwCompiler(hw_cpu_model_path, thread_num);
      os->asr.InitSegDict(seg_dict_path);
    }
    if (use_gpu && FileExists(hw_gpu_model_path)) {                              // :68-72
      os->asr.InitHwCompiler(hw_gpu_model_path, thread_num);
      os->asr.InitSegDict(seg_dict_path);
    }
    std::string am_model_path = PathAppend(dir, IsTrue(model_path, QUANTIZE) ? QUANT_MODEL_NAME : MODEL_NAME);      // :74-77
    if (use_gpu) am_model_path = PathAppend(dir, TORCH_MODEL_NAME);              // :79-84 (the ONNX file beside it is read)
    std::string token_path = model_path.count(TOKEN_PATH_KEY) ? model_path[TOKEN_PATH_KEY] : PathAppend(dir, TOKEN_PATH);
    if (!FileExists(token_path)) token_path.clear();                             // harness runs on synthetic models: ids as text
    os->asr.InitAsr(am_model_path, PathAppend(dir, AM_CMVN_NAME), PathAppend(dir, AM_CONFIG_NAME), token_path, thread_num);   // :89, exits on failure
  }
  if (model_path.count(PUNC_DIR) && !model_path[PUNC_DIR].empty())               // :105-129, always CTTransformer here
    os->punc.reset(funasr::CreatePuncModelHip(model_path[PUNC_DIR], thread_num, false, IsTrue(model_path, PUNC_QUANT)));
  return os.release();
}

FUNASR_RESULT FunOfflineInferBuffer(FUNASR_HANDLE handle, const char* sz_buf, int n_len, FUNASR_MODE mode, QM_CALLBACK fn_callback,
                                    const std::vector<std::vector<float>>& hw_emb, int sampling_rate, std::string wav_format,
                                    bool itn, int vad_tail_sil, int vad_max_len, FUNASR_DEC_HANDLE dec_handle) {
  (void)mode; (void)itn; (void)dec_handle;
  OfflineStreamHip* os = static_cast<OfflineStreamHip*>(handle);
  if (!os || !sz_buf) return nullptr;
  if (wav_format != "pcm" && wav_format != "PCM") return nullptr;      // the reference decodes other containers with ffmpeg
  if (sampling_rate != os->asr.GetAsrSampleRate()) return nullptr;     // resampling (audio.cpp:230-260) is caller-side here
  // Audio::LoadPcmwav (audio.cpp:787-819): s16 LE -> f32 / 32768
  const int n = n_len / 2;
  std::vector<float> pcm((size_t)n);
  const int16_t* s16 = reinterpret_cast<const int16_t*>(sz_buf);
  for (int i = 0; i < n; ++i) pcm[i] = (float)s16[i] / 32768.f;
  auto res = std::make_unique<RecogResult>();
  res->snippet_time = (float)n / (float)sampling_rate;
  if (n == 0) return res.release();
  std::vector<int> index_vector = {0};
  res->segs.assign(1, {0, n});
  if (os->vad) {
    if (!CutSplit(os, pcm, vad_tail_sil, vad_max_len, res->segs, index_vector)) {
      std::fprintf(stderr, "FunOfflineInferBuffer: %s\n", pfhip_last_error());
      return res.release();
    }
  }
  std::vector<std::string> msgs(index_vector.size());
  std::vector<std::vector<float>> spans(index_vector.size());
  res->seg_ids.assign(index_vector.size(), {});
  size_t head = 0, msg_idx = 0;
  std::vector<int> batch;
  const int batch_size = os->asr.GetBatchSize();
  int step = 0;
  while (FetchDynamic(res->segs, index_vector, head, batch_size, batch) > 0) {
    std::vector<float*> buff(batch.size());
    std::vector<int> len(batch.size());
    for (size_t k = 0; k < batch.size(); ++k) {
      buff[k] = pcm.data() + res->segs[batch[k]].first;
      len[k] = res->segs[batch[k]].second - res->segs[batch[k]].first;
    }
    const std::vector<std::string> msg_batch = os->asr.Forward(buff.data(), len.data(), true, hw_emb, nullptr, (int)batch.size());
    for (size_t k = 0; k < batch.size(); ++k, ++msg_idx) {        // funasrruntime.cpp:270-279
      const int seg = index_vector[msg_idx];
      msgs[seg] = msg_batch[k];
      res->seg_ids[seg] = os->asr.LastTokenIds()[k];
      if (k < os->asr.LastTimestamps().size()) spans[seg] = os->asr.LastTimestamps()[k];
    }
    if (fn_callback) fn_callback(++step, (int)index_vector.size());
  }
  std::string cur_stamp = "[";
  for (size_t idx = 0; idx < msgs.size(); ++idx) {                                     // funasrruntime.cpp:291-312
    if (msgs[idx].empty()) continue;
    const float t0 = (float)res->segs[idx].first / (float)sampling_rate;               // msg_stimes
    const size_t bar = msgs[idx].find(" | ");
    res->msg += msgs[idx].substr(0, bar);
    if (bar != std::string::npos) {                 // "<text> | b0, e0,b1, e1": seconds relative to the segment
      std::vector<float> v;
      std::stringstream ss(msgs[idx].substr(bar + 3));
      std::string item;
      while (std::getline(ss, item, ',')) { try { v.push_back(std::stof(item)); } catch (...) { break; } }
      for (size_t i = 0; i + 1 < v.size(); i += 2)
        cur_stamp += "[" + std::to_string((int)(1000 * (v[i] + t0))) + "," + std::to_string((int)(1000 * (v[i + 1] + t0))) + "],";
    } else {                                        // no vocabulary loaded: ids as text, stamps straight from the adapter
      for (size_t i = 0; i + 2 < spans[idx].size() + 1 && i + 2 <= spans[idx].size(); i += 3) {
        if (spans[idx][i + 2] != 0.f) continue;
        cur_stamp += "[" + std::to_string((int)(1000 * (spans[idx][i] + t0))) + "," + std::to_string((int)(1000 * (spans[idx][i + 1] + t0))) + "],";
      }
    }
  }
  if (cur_stamp != "[") {
    cur_stamp.erase(cur_stamp.size() - 1);
    res->stamp = cur_stamp + "]";
  }
  if (os->punc) res->msg = os->punc->AddPunc(res->msg.c_str(), "zh-cn");               // funasrruntime.cpp:317-320 (re-entrant)
  return res.release();
}

const std::vector<std::vector<float>> CompileHotwordEmbedding(FUNASR_HANDLE handle, std::string& hotwords, ASR_TYPE mode) {
  (void)mode;
  OfflineStreamHip* os = static_cast<OfflineStreamHip*>(handle);
  if (!os) return {};
  return os->asr.CompileHotwordEmbedding(hotwords);
}

const char* FunASRGetResult(FUNASR_RESULT result, int n_index) {
  (void)n_index;
  return result ? static_cast<RecogResult*>(result)->msg.c_str() : nullptr;
}
const char* FunASRGetStamp(FUNASR_RESULT result) { return result ? static_cast<RecogResult*>(result)->stamp.c_str() : nullptr; }
float FunASRGetRetSnippetTime(FUNASR_RESULT result) { return result ? static_cast<RecogResult*>(result)->snippet_time : 0.f; }
void FunASRFreeResult(FUNASR_RESULT result) { delete static_cast<RecogResult*>(result); }
void FunOfflineUninit(FUNASR_HANDLE handle) { delete static_cast<OfflineStreamHip*>(handle); }
const std::vector<std::vector<int>>& FunASRGetSegmentIds(FUNASR_RESULT result) { return static_cast<RecogResult*>(result)->seg_ids; }
const std::vector<std::pair<int, int>>& FunASRGetSegments(FUNASR_RESULT result) { return static_cast<RecogResult*>(result)->segs; }
const std::vector<int>& FunASRGetOnlineIds(FUNASR_RESULT result) { return static_cast<RecogResult*>(result)->online_ids; }
pfhip_model* FunOfflineGetAsrHandle(FUNASR_HANDLE handle) { return handle ? static_cast<OfflineStreamHip*>(handle)->asr.Handle() : nullptr; }


// ---- 2-pass --------------------------------------------------------------------------------------------------------------
namespace {

struct TpassStreamHip {               // funasr::TpassStream (tpass-stream.cpp): the shared models
  funasr::ParaformerHip asr;          // ONE object holds the offline session and the online encoder / decoder (paraformer.cpp:134-154)
  std::unique_ptr<funasr::FsmnVadHip> vad;
  std::unique_ptr<funasr::PuncModelHipBase> punc_online;      // TpassStream::punc_online_handle (tpass-stream.cpp:100-135)
};

struct TpassOnlineStreamHip {         // funasr::TpassOnlineStream (tpass-online-stream.cpp:14-15): per connection
  TpassStreamHip* shared = nullptr;
  std::unique_ptr<funasr::ParaformerOnlineHip> asr_online;      // asr_online_handle (tpass-online-stream.cpp:14-15)
  std::unique_ptr<funasr::FsmnVadOnlineHip> vad_online;
  pfhip_host::TpassAudio audio;
};

}  // namespace

// TpassStream::TpassStream (tpass-stream.cpp:4-135) on the HIP plug-ins, call for call
FUNASR_HANDLE FunTpassInit(std::map<std::string, std::string>& model_path, int thread_num) {
  auto ts = std::make_unique<TpassStreamHip>();
  ts->vad = MakeVad(model_path, thread_num);
  if (!model_path.count(OFFLINE_MODEL_DIR) || !model_path.count(ONLINE_MODEL_DIR)) {                 // :79-82
    std::fprintf(stderr, "Can not find offline-model-dir or online-model-dir\n");
    std::exit(-1);
  }
  const std::string mdir = model_path[MODEL_DIR], odir = model_path[ONLINE_MODEL_DIR];
  const std::string hw_compile_model_path = PathAppend(mdir, MODEL_EB_NAME), seg_dict_path = PathAppend(mdir, MODEL_SEG_DICT);
  if (FileExists(hw_compile_model_path) && FileExists(seg_dict_path)) {                             // :54-60
    ts->asr.InitHwCompiler(hw_compile_model_path, thread_num);
    ts->asr.InitSegDict(seg_dict_path);
  }
  const bool q = IsTrue(model_path, QUANTIZE);                                                       // :62-72
  const std::string am_model_path = PathAppend(model_path[OFFLINE_MODEL_DIR], q ? QUANT_MODEL_NAME : MODEL_NAME);
  const std::string en_model_path = PathAppend(odir, q ? QUANT_ENCODER_NAME : ENCODER_NAME);
  const std::string de_model_path = PathAppend(odir, q ? QUANT_DECODER_NAME : DECODER_NAME);
  auto tokens_or_none = [](const std::string& p) { return FileExists(p) ? p : std::string(); };
  ts->asr.InitAsr(am_model_path, en_model_path, de_model_path, PathAppend(odir, AM_CMVN_NAME), PathAppend(mdir, AM_CONFIG_NAME),
                  tokens_or_none(PathAppend(mdir, TOKEN_PATH)), tokens_or_none(PathAppend(odir, TOKEN_PATH)), thread_num,
                  PathAppend(odir, AM_CONFIG_NAME));                                                 // :76-77, exits on failure
  if (!ts->vad) {                                                                                    // tpass-online-stream.cpp:8-11
    std::fprintf(stderr, "vad_handle is null\n");
    std::exit(-1);
  }
  if (model_path.count(PUNC_DIR) && !model_path[PUNC_DIR].empty())                                   // :100-135, realtime or offline class
    ts->punc_online.reset(funasr::CreatePuncModelHip(model_path[PUNC_DIR], thread_num, true, IsTrue(model_path, PUNC_QUANT)));
  // One handler thread per connection in the server: the VAD's concurrent device calls are merged into batched passes (the
  // online acoustic model's queue is set up by ParaformerHip::InitAsr).  Not keyed on `thread_num`: that is --model-thread-num
  // (onnxruntime intra-op threads, default 1: funasr-wss-server-2pass.cpp:110,569), not the number of handler threads.
  {
    auto knob = [](const char* name, int dflt) { const char* e = std::getenv(name); return e && *e ? std::atoi(e) : dflt; };
    pfhip_set_vad_stream_batching(ts->vad->Handle(), knob("PFHIP_VAD_WAIT_US", 1000), knob("PFHIP_VAD_MAX", 256));
  }
  return ts.release();
}

// TpassOnlineStream::TpassOnlineStream (tpass-online-stream.cpp:4-19)
FUNASR_HANDLE FunTpassOnlineInit(FUNASR_HANDLE tpass_handle, std::vector<int> chunk_size) {
  TpassStreamHip* ts = static_cast<TpassStreamHip*>(tpass_handle);
  if (!ts || chunk_size.size() != 3) return nullptr;
  auto os = std::make_unique<TpassOnlineStreamHip>();
  os->shared = ts;
  os->vad_online.reset(new funasr::FsmnVadOnlineHip(ts->vad.get()));
  os->asr_online.reset(new funasr::ParaformerOnlineHip(&ts->asr, chunk_size));
  if (!os->asr_online->ok() || !os->vad_online->ok()) {
    std::fprintf(stderr, "FunTpassOnlineInit: %s\n", pfhip_last_error());
    return nullptr;
  }
  return os.release();
}

FUNASR_RESULT FunTpassInferBuffer(FUNASR_HANDLE handle, FUNASR_HANDLE online_handle, const char* sz_buf, int n_len,
                                  std::vector<std::vector<std::string>>& punc_cache, bool input_finished, int sampling_rate,
                                  std::string wav_format, ASR_TYPE mode, const std::vector<std::vector<float>>& hw_emb, bool itn,
                                  int vad_tail_sil, int vad_max_len, FUNASR_DEC_HANDLE dec_handle) {
  (void)itn; (void)dec_handle;
  TpassStreamHip* ts = static_cast<TpassStreamHip*>(handle);
  TpassOnlineStreamHip* os = static_cast<TpassOnlineStreamHip*>(online_handle);
  if (!ts || !os || !sz_buf) return nullptr;
  if (wav_format != "pcm" && wav_format != "PCM") return nullptr;                  // funasrruntime.cpp:523-531
  if (sampling_rate != 16000) return nullptr;
  if (ts->punc_online && punc_cache.size() < 2) return nullptr;                    // [0]: online text, [1]: 2nd-pass text
  if (!os->audio.LoadPcmwavOnline(sz_buf, n_len)) return nullptr;
  auto res = std::make_unique<RecogResult>();
  res->snippet_time = os->audio.GetTimeLen();
  // FsmnVadOnline::Infer (fsmn-vad-online.cpp:135-151) as the VAD of Audio::Split
  os->vad_online->SetConfig(vad_tail_sil, vad_max_len);
  auto vad_infer = [&](std::vector<float>& waves, bool fin) { return os->vad_online->Infer(waves, fin); };
  os->audio.Split(vad_infer, os->asr_online->chunk_len, input_finished, (pfhip_host::AsrType)mode);
  pfhip_host::TpassFrame frame;
  while (os->audio.FetchChunck(frame)) {                                            // funasrruntime.cpp:538-566
    const std::string msg = os->asr_online->Forward(frame.data.data(), (int)frame.data.size(), frame.is_final);      // :540
    res->online_ids.insert(res->online_ids.end(), os->asr_online->LastTokenIds().begin(), os->asr_online->LastTokenIds().end());
    if (mode == ASR_ONLINE) {
      os->asr_online->online_res += msg;
      if (frame.is_final) {                                                         // funasrruntime.cpp:543-556
        res->tpass_msg = os->asr_online->online_res;
        if (ts->punc_online) res->tpass_msg = ts->punc_online->AddPunc(os->asr_online->online_res.c_str(), punc_cache[0]);
        os->asr_online->online_res.clear();
      }
      res->msg += msg;
    } else if (mode == ASR_TWO_PASS) {
      res->msg += msg;
    }
  }
  std::string cur_stamp = "[";
  while (os->audio.FetchTpass(frame)) {                                             // funasrruntime.cpp:570-639
    float* buff[1] = {frame.data.data()};
    int len[1] = {(int)frame.data.size()};
    const std::vector<std::string> msgs = ts->asr.Forward(buff, len, true, hw_emb, nullptr, 1);
    std::string msg = msgs.empty() ? "" : msgs[0];
    if (msg.empty()) continue;
    const size_t bar = msg.find(" | ");
    if (bar != std::string::npos) {
      std::vector<float> v;
      std::stringstream ss(msg.substr(bar + 3));
      std::string item;
      while (std::getline(ss, item, ',')) { try { v.push_back(std::stof(item)); } catch (...) { break; } }
      for (size_t i = 0; i + 1 < v.size(); i += 2) {
        const float b = v[i] + (float)frame.global_start / 1000.0f, e = v[i + 1] + (float)frame.global_start / 1000.0f;
        cur_stamp += "[" + std::to_string((int)(1000 * b)) + "," + std::to_string((int)(1000 * e)) + "],";
      }
      msg = msg.substr(0, bar);
    }
    if (cur_stamp != "[") {
      cur_stamp.erase(cur_stamp.size() - 1);
      res->stamp += cur_stamp + "]";
    }
    res->tpass_msg = msg;
    if (ts->punc_online) {                                                          // funasrruntime.cpp:609-614 (re-entrant)
      res->tpass_msg = ts->punc_online->AddPunc(msg.c_str(), punc_cache[1]);
      if (input_finished && ts->punc_online->is_online) res->tpass_msg += "\xE3\x80\x82";      // "。"
    }
  }
  if (input_finished) os->audio.ResetIndex();
  return res.release();
}

const char* FunASRGetTpassResult(FUNASR_RESULT result, int n_index) {
  (void)n_index;
  return result ? static_cast<RecogResult*>(result)->tpass_msg.c_str() : nullptr;
}
void FunTpassOnlineUninit(FUNASR_HANDLE online_handle) { delete static_cast<TpassOnlineStreamHip*>(online_handle); }
void FunTpassUninit(FUNASR_HANDLE handle) { delete static_cast<TpassStreamHip*>(handle); }
