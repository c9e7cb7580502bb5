// Host side above the C ABI, in the reference's own language (C++): `ParaformerHip`, the sibling of
// `funasr::Paraformer` / `funasr::ParaformerTorch` behind the plug-in seam `class funasr::Model`
// (onnxruntime/include/model.h:13-46).  Same method names, argument meaning and error behaviour as the
// reference classes (onnxruntime/src/paraformer.cpp:21-53,463-589; paraformer-torch.cpp:301-475).
//
// Built stand-alone here (against the small interface below, which repeats the virtuals of model.h
// that this path uses, because model.h drags in openfst/yaml-cpp/glog headers that are out of scope);
// inside the reference tree define PFHIP_WITH_FUNASR to derive from the real funasr::Model instead
// (INTEGRATION.md shows the CMake switch and the two-line factory change).
#pragma once
#include <string>
#include <vector>

#include "../../../include/pfhip.h"

#ifdef PFHIP_WITH_FUNASR
#include "model.h"
#include "vocab.h"
namespace funasr {
using ParaformerHipBase = Model;
}
#else
namespace funasr {
// The subset of `class Model` (model.h:13-46) on the offline Paraformer path, signature for signature.
class ParaformerHipBase {
 public:
  virtual ~ParaformerHipBase() {}
  virtual void StartUtterance() = 0;
  virtual void EndUtterance() = 0;
  virtual void Reset() = 0;
  virtual void InitAsr(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config,
                       const std::string& token_file, int thread_num) = 0;
  virtual std::vector<std::string> Forward(float** din, int* len, bool input_finished,
                                           const std::vector<std::vector<float>>& hw_emb, void* wfst_decoder,
                                           int batch_in) = 0;
  virtual std::vector<std::vector<float>> CompileHotwordEmbedding(std::string& hotwords) = 0;
  virtual std::string Rescoring() = 0;
  virtual int GetAsrSampleRate() = 0;
  virtual void SetBatchSize(int batch_size) = 0;
  virtual int GetBatchSize() = 0;
};
}  // namespace funasr
#endif

namespace funasr {

class ParaformerHip : public ParaformerHipBase {
 public:
  ParaformerHip();
  ~ParaformerHip() override;

  // am_model = weight blob (<dir>/model.pfhip.bin), am_config = its JSON manifest; am_cmvn is folded
  // into the container (cmvn.* tensors); token_file = tokens.json (a JSON array of strings).
  void InitAsr(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config,
               const std::string& token_file, int thread_num) override;
  // Returns batch_in strings.  "" for an utterance without one full fbank window (paraformer.cpp:477-480)
  // and for every item when the device call fails (the reference logs and returns "" too, :582-588).
  std::vector<std::string> Forward(float** din, int* len, bool input_finished,
                                   const std::vector<std::vector<float>>& hw_emb, void* wfst_decoder,
                                   int batch_in) override;
  // Plain model: one zero row of encoder_size, as Paraformer::CompileHotwordEmbedding does when
  // use_hotword is false (paraformer.cpp:594-599).  Contextual model: whitespace-separated hotwords, each split into
  // vocabulary units (UTF-8 characters looked up in tokens.json; the reference additionally consults seg_dict for
  // Latin words, :601-647 — host text handling, not restated), at most 10 ids each, the [1,0,...] row appended
  // (:648-651), then the device embedder (pfhip_hotword_embed).
  std::vector<std::vector<float>> CompileHotwordEmbedding(std::string& hotwords) override;
  void StartUtterance() override {}
  void EndUtterance() override {}
  void Reset() override {}
  std::string Rescoring() override { return ""; }
  int GetAsrSampleRate() override;
  void SetBatchSize(int batch_size) override { batch_size_ = batch_size; }
  int GetBatchSize() override { return batch_size_; }

  // token ids of the last Forward, per utterance (what GreedySearch computed, paraformer.cpp:386-395)
  // (of the calling thread: Forward is re-entrant)
  const std::vector<std::vector<int>>& LastTokenIds() const;
  // timestamp models: (begin_s, end_s, is_sil) per span of the last Forward, what TimestampOnnx produced
  // (paraformer.cpp:545-562 + util.cpp:838-963); empty for plain models
  const std::vector<std::vector<float>>& LastTimestamps() const;
  void SetDevice(int device) { device_ = device; }
  // the C-ABI handle underneath (streams of the online model are created from it) and the vocabulary mapping
  pfhip_model* Handle() const { return handle_; }
  std::string TokensToString(const std::vector<int>& ids) const { return IdsToString(ids); }

 private:
  std::string IdsToString(const std::vector<int>& ids) const;
  pfhip_model* handle_ = nullptr;
  int device_ = 0;
  int batch_size_ = 1;
  std::vector<std::string> tokens_;
};

}  // namespace funasr
