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

class ParaformerHip : public ParaformerHipBase
#ifdef PFHIP_WITH_FUNASR
    , public WfstDecodable
#endif
{
 public:
  ParaformerHip();
  ~ParaformerHip() override;

  // am_model = weight blob (<dir>/model.pfhip.bin), am_config = its JSON manifest; am_cmvn is folded
  // into the container (cmvn.* tensors); token_file = tokens.json (a JSON array of strings).
  void InitAsr(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config,
               const std::string& token_file, int thread_num) override;
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
  // offline-stream.cpp:65-71: the hotword embedder's weights live in the same container as the acoustic model (bias.* tensors),
  // so InitHwCompiler has nothing to load; InitSegDict reads the "word<TAB>pieces" file Latin hotwords are segmented with.
  void InitHwCompiler(const std::string& hw_model, int thread_num) override { (void)hw_model; (void)thread_num; }
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

#ifdef PFHIP_WITH_FUNASR
  // the overloads of model.h this class does not implement stay visible (and keep their empty defaults)
  using Model::InitAsr;
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
  // the C-ABI handle underneath (streams of the online model are created from it) and the vocabulary mapping
  pfhip_model* Handle() const { return handle_; }
  std::string TokensToString(const std::vector<int>& ids) { return IdsToString(ids); }

  std::string language = "zh-cn";          // paraformer.h:97

 private:
  std::string IdsToString(const std::vector<int>& ids);
  pfhip_model* handle_ = nullptr;
  int device_ = 0;
  int batch_size_ = 1;
  bool has_lm_ = false;
  HipVocab* vocab = nullptr;               // tokens.json (paraformer.cpp:47-48)
  pfhip_host::SegDictHost* seg_dict_ = nullptr;
};

}  // namespace funasr
