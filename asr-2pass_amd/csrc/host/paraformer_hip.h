// Host side above the C ABI, in the reference's own language (C++): `ParaformerHip`, the sibling of
// `funasr::Paraformer` / `funasr::ParaformerTorch` behind the plug-in seam `class funasr::Model`
// (onnxruntime/include/model.h:12-46) and, like both of them, a `funasr::WfstDecodable`
// (onnxruntime/src/wfst-decodable.h:17-30) so that FunASRWfstDecoderInit's dynamic_cast finds it
// (funasrruntime.cpp:835-850).  Same method names, argument meaning and error behaviour as the reference classes
// (onnxruntime/src/paraformer.cpp:21-53,156-176,463-589; paraformer-torch.cpp:67-90,301-475).
//
// Two builds of the same source:
//   * inside the reference tree: -DPFHIP_WITH_FUNASR derives from the real funasr::Model + funasr::WfstDecodable and uses
//     the real funasr::Vocab / funasr::Decoder (INTEGRATION.md shows the CMake switch and the factory change;
//     tests/test_ref_headers.py compiles this configuration against /root/reference's headers);
//   * stand-alone (the harnesses of this repo): against the small interfaces below, which repeat the virtuals of model.h /
//     decoder.h that this path uses, signature for signature, and the text assembly of host_vocab.h.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../../include/pfhip.h"
#include "hotword_text.h"

#ifdef PFHIP_WITH_FUNASR
#include "model.h"
#include "wfst-decodable.h"
#include "decoder.h"
namespace funasr {
using ParaformerHipBase = Model;
using HipVocab = Vocab;
}
#else
#include "host_vocab.h"
namespace funasr {
// The subset of `class Model` (model.h:12-46) on the offline Paraformer path, signature for signature.
class ParaformerHipBase {
 public:
  virtual ~ParaformerHipBase() {}
  virtual void StartUtterance() = 0;
  virtual void EndUtterance() = 0;
  virtual void Reset() = 0;
  virtual void InitAsr(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config,
                       const std::string& token_file, int thread_num) {}
  virtual void InitAsr(const std::string& en_model, const std::string& de_model, const std::string& am_cmvn, const std::string& am_config,
                       const std::string& token_file, int thread_num) {}
  virtual void InitAsr(const std::string& am_model, const std::string& en_model, const std::string& de_model, const std::string& am_cmvn,
                       const std::string& am_config, const std::string& token_file, const std::string& online_token_file, int thread_num) {}
  virtual void InitAsr(const std::string& am_model, const std::string& en_model, const std::string& de_model, const std::string& am_cmvn,
                       const std::string& am_config, const std::string& token_file, const std::string& online_token_file, int thread_num,
                       const std::string& online_config_file) {}
  virtual std::string Forward(float* din, int len, bool input_finished, const std::vector<std::vector<float>>& hw_emb = {{0.0}},
                              void* wfst_decoder = nullptr) { return ""; }
  virtual void InitLm(const std::string& lm_file, const std::string& lm_config, const std::string& lex_file) {}
  virtual void InitLm(const std::string& lm_file, const std::string& lm_config, const std::string& lex_file,
                      const std::string& lm_units_path) {}
  virtual std::vector<std::string> Forward(float** din, int* len, bool input_finished,
                                           const std::vector<std::vector<float>>& hw_emb = {{0.0}}, void* wfst_decoder = nullptr,
                                           int batch_in = 1) { return std::vector<std::string>(); }
  virtual void InitHwCompiler(const std::string& hw_model, int thread_num) {}
  virtual void InitSegDict(const std::string& seg_dict_model) {}
  virtual std::vector<std::vector<float>> CompileHotwordEmbedding(std::string& hotwords) { return {}; }
  virtual std::string Rescoring() = 0;
  virtual std::string GetLang() { return ""; }
  virtual int GetAsrSampleRate() = 0;
  virtual void SetBatchSize(int batch_size) {}
  virtual int GetBatchSize() { return 0; }
};
// `class Decoder` (onnxruntime/src/decoder.h:10-33): what a FUNASR_DEC_HANDLE points to (funasrruntime.cpp:260,392).
class Decoder {
 public:
  virtual ~Decoder() = default;
  virtual void StartUtterance() {}
  virtual void EndUtterance() {}
  virtual std::string Search(float* in, int len, int64_t token_nums) { return ""; }
  virtual std::string FinalizeDecode(bool is_stamp = false, std::vector<float> us_alphas = {},
                                     std::vector<float> us_cif_peak = {}) { return ""; }
};
using HipVocab = pfhip_host::HostVocab;
}  // namespace funasr
#endif

namespace funasr {

class ParaformerOnlineHip;

class ParaformerHip : public ParaformerHipBase
#ifdef PFHIP_WITH_FUNASR
    , public WfstDecodable
#endif
{
 public:
  ParaformerHip();
  ~ParaformerHip() override;

  // The strings are the reference's own (onnxruntime/include/com-define.h:52-88), read in C++ at server start like
  // Paraformer::InitAsr does (paraformer.cpp:21-53: LoadConfigFromYaml, the session load, Vocab, LoadCmvn):
  //   am_model   <MODEL_DIR>/model.onnx | model_quant.onnx (offline-stream.cpp:74-77) | model.torchscript | model_blade.torchscript
  //              (:79-84, use_gpu: the ONNX file of the same stem beside it is read) | a container x.pfhip.bin
  //   am_cmvn    <dir>/am.mvn        am_config  <dir>/config.yaml (fs, lang + the architecture keys)      token_file  <dir>/tokens.json
  // InitHwCompiler(model_eb.onnx) comes BEFORE InitAsr in both factories (offline-stream.cpp:60-72, tpass-stream.cpp:52-60): the
  // path is kept and the embedder's tensors are loaded with the acoustic model.  A load failure ends the process like the
  // reference's (paraformer.cpp:43-46).  Ends with the warm-up forward of the GPU flavour (paraformer-torch.cpp:59,477-520).
  void InitAsr(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config,
               const std::string& token_file, int thread_num) override;
  // online model alone (model.cpp:45; paraformer.cpp:56-131): encoder + predictor file, decoder file
  void InitAsr(const std::string& en_model, const std::string& de_model, const std::string& am_cmvn, const std::string& am_config,
               const std::string& token_file, int thread_num) override;
  // 2-pass (tpass-stream.cpp:76-77 calls the nine-argument form; paraformer.cpp:134-154): the online model first — its config,
  // its tokens, and am_cmvn, which both models share — then the offline model with am_config and token_file.  The
  // eight-argument virtual of model.h:23-24 (no class of the reference implements it) reads the online config from am_config.
  void InitAsr(const std::string& am_model, const std::string& en_model, const std::string& de_model, const std::string& am_cmvn,
               const std::string& am_config, const std::string& token_file, const std::string& online_token_file, int thread_num) override;
  void InitAsr(const std::string& am_model, const std::string& en_model, const std::string& de_model, const std::string& am_cmvn,
               const std::string& am_config, const std::string& token_file, const std::string& online_token_file, int thread_num,
               const std::string& online_config_file) override;
  // The language model of the WFST path.  Both overloads exist because the two callers differ: offline-stream.cpp:102 passes
  // three arguments (which reaches ParaformerTorch::InitLm, paraformer-torch.cpp:67-90, but NOT Paraformer's four-argument
  // one), tpass-stream.cpp:95-97 passes four.  Inside the reference tree this reads the FST, the LM vocabulary and the phone
  // set exactly like paraformer.cpp:156-176; stand-alone (no openfst) it only records that an LM was configured, which is
  // what routes Forward to the decoder.
  void InitLm(const std::string& lm_file, const std::string& lm_cfg_file, const std::string& lex_file) override;
  void InitLm(const std::string& lm_file, const std::string& lm_cfg_file, const std::string& lex_file,
              const std::string& lm_units_path) override;
  // Returns batch_in strings.  "" for an utterance without one full fbank window (paraformer.cpp:477-480)
  // and for every item when the device call fails (the reference logs and returns "" too, :582-588).
  // Without an LM: GreedySearch on the device + Vocab::Vector2StringV2 (time-stamp models: Vector2String -> TimestampOnnx ->
  // PostProcess), paraformer.cpp:386-408.  With an LM (InitLm succeeded) and a decoder handle: the log-prob rows of every
  // utterance are handed to `((Decoder*)wfst_decoder)->Search(rows, token_num, vocab)` and, when input_finished,
  // `FinalizeDecode(is_stamp, us_alphas, us_peaks)` gives the text — paraformer.cpp:563-579; between the items of a batch the
  // decoder is restarted as paraformer-torch.cpp:464-466 does.
  std::vector<std::string> Forward(float** din, int* len, bool input_finished,
                                   const std::vector<std::vector<float>>& hw_emb = {{0.0}}, void* wfst_decoder = nullptr,
                                   int batch_in = 1) override;
  // offline-stream.cpp:60-72 / paraformer.cpp:243-261: called before InitAsr with <MODEL_DIR>/model_eb.onnx (or
  // model_eb.torchscript); marks the model contextual (use_hotword) and remembers the file, whose tensors (bias_embed,
  // bias_encoder) are read together with the acoustic model's.  InitSegDict reads the "word<TAB>pieces" file Latin hotwords are
  // segmented with.
  void InitHwCompiler(const std::string& hw_model, int thread_num) override { (void)thread_num; hw_model_ = hw_model; use_hotword_ = true; }
  void InitSegDict(const std::string& seg_dict_model) override;
  // Plain model: one zero row of encoder_size, as Paraformer::CompileHotwordEmbedding does when use_hotword is false
  // (paraformer.cpp:594-599).  Contextual model: the reference's string handling (hotword_text.h: space-separated hotwords,
  // all-Chinese ones split into characters, the others through the segmentation dictionary, at most 10 units, out-of-vocabulary
  // hotwords dropped, the [1,0,...] row appended, :600-651), then the device embedder (pfhip_hotword_embed).
  std::vector<std::vector<float>> CompileHotwordEmbedding(std::string& hotwords) override;
  void StartUtterance() override {}
  void EndUtterance() override {}
  void Reset() override {}
  std::string Rescoring() override { return ""; }
  std::string GetLang() override { return language; }
  int GetAsrSampleRate() override;
  void SetBatchSize(int batch_size) override { batch_size_ = batch_size; }
  int GetBatchSize() override { return batch_size_; }

  // Model::Forward(float* din, int len, ...) (model.h:30): one utterance through the batched form
  std::string Forward(float* din, int len, bool input_finished, const std::vector<std::vector<float>>& hw_emb = {{0.0}},
                      void* wfst_decoder = nullptr) override;
#ifdef PFHIP_WITH_FUNASR
  // the SenseVoice overload of model.h:33-34 stays visible (and keeps its empty default)
  using Model::Forward;
  // WfstDecodable (wfst-decodable.h:25-29): what FunASRWfstDecoderInit builds the per-connection decoder from
  std::shared_ptr<fst::Fst<fst::StdArc>> GetLm() const override { return lm_; }
  Vocab* GetVocab() const override { return vocab; }
  PhoneSet* GetPhoneSet() const override { return phone_set_; }
  Vocab* GetLmVocab() const override { return lm_vocab; }
  // the non-const getters of model.h:43-45 answer the same
  Vocab* GetVocab() override { return vocab; }
  PhoneSet* GetPhoneSet() override { return phone_set_; }
  Vocab* GetLmVocab() override { return lm_vocab; }
#endif
  bool HasLm() const { return has_lm_; }

  // token ids of the last Forward, per utterance (what GreedySearch computed, paraformer.cpp:386-395)
  // (of the calling thread: Forward is re-entrant)
  const std::vector<std::vector<int>>& LastTokenIds() const;
  // timestamp models: (begin_s, end_s, is_sil) per span of the last Forward, what TimestampOnnx produced
  // (paraformer.cpp:545-562 + util.cpp:838-963); empty for plain models
  const std::vector<std::vector<float>>& LastTimestamps() const;
  void SetDevice(int device) { device_ = device; }
  // the C-ABI handles underneath — the offline model and, after a 2-pass / online InitAsr, the online model that
  // ParaformerOnlineHip streams are created from (the reference's encoder_session_ / decoder_session_) — and the text mappings
  pfhip_model* Handle() const { return handle_; }
  pfhip_model* OnlineHandle() const { return online_handle_; }
  std::string TokensToString(const std::vector<int>& ids) { return IdsToString(ids); }
  // Paraformer::OnlineGreedySearch's text step: online_vocab->Vector2StringV2(hyps) (paraformer.cpp:362-371)
  std::string OnlineTokensToString(const std::vector<int>& ids);

  std::string language = "zh-cn";          // paraformer.h:97

 private:
  std::string IdsToString(const std::vector<int>& ids);
  void LoadOffline(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config, const std::string& token_file);
  void LoadOnline(const std::string& en_model, const std::string& de_model, const std::string& am_cmvn, const std::string& am_config,
                  const std::string& token_file);
  pfhip_model* handle_ = nullptr;
  pfhip_model* online_handle_ = nullptr;
  HipVocab* online_vocab = nullptr;        // paraformer.cpp:117
  std::string hw_model_;                   // InitHwCompiler's argument
  bool use_hotword_ = false;
  int asr_sample_rate_ = 16000;            // frontend_conf.fs (paraformer.cpp:191)
  int device_ = 0;
  int batch_size_ = 1;
  bool has_lm_ = false;
  HipVocab* vocab = nullptr;               // tokens.json (paraformer.cpp:47-48)
  pfhip_host::SegDictHost* seg_dict_ = nullptr;
};

// `funasr::ParaformerOnline` (onnxruntime/src/paraformer-online.h:11-135): one per connection, built from the shared model
// object like `ParaformerOnline(Model* offline_handle, chunk_size)` (paraformer-online.cpp:12-62; made by
// TpassOnlineStream, tpass-online-stream.cpp:14-15, and CreateModel(void*, chunk_size), model.cpp:53-58).  Its caches (fbank
// splice cache, [5|10|5] window, CIF carry, decoder FSMN caches) live in HBM inside one pfhip_stream.
class ParaformerOnlineHip : public ParaformerHipBase {
 public:
  ParaformerOnlineHip(ParaformerHipBase* offline_handle, std::vector<int> chunk_size, std::string model_type = "Paraformer");
  ~ParaformerOnlineHip() override;
  // ParaformerOnline::Forward (paraformer-online.cpp:525-601): the text of the tokens this call emitted ("" while the chunk is
  // incomplete); a non-empty result of the last chunk ends in a blank (:585-587).  Errors are logged and give "" (:593-597).
  std::string Forward(float* din, int len, bool input_finished, const std::vector<std::vector<float>>& hw_emb = {{0.0}},
                      void* wfst_decoder = nullptr) override;
#ifdef PFHIP_WITH_FUNASR
  using Model::Forward;
#endif
  void StartUtterance() override {}
  void EndUtterance() override {}
  void Reset() override;                                   // Reset + ResetCache (:386-395)
  std::string Rescoring() override { return ""; }          // "Not Imp" in the reference too (:607-611)
  int GetAsrSampleRate() override { return offline_handle_ ? offline_handle_->GetAsrSampleRate() : 16000; }
  bool ok() const { return stream_ != nullptr; }
  const std::vector<int>& LastTokenIds() const { return last_ids_; }
  // 2pass (paraformer-online.h:131-133)
  std::string online_res;
  int chunk_len = 9600;

 private:
  ParaformerHip* offline_handle_ = nullptr;
  pfhip_stream* stream_ = nullptr;
  std::vector<int> last_ids_;
};

}  // namespace funasr
