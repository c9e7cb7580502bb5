// Host side of the punctuation path in the reference's own language: `CTTransformerHip` / `CTTransformerOnlineHip`, the
// siblings of `funasr::CTTransformer` / `funasr::CTTransformerOnline` behind `class funasr::PuncModel`
// (onnxruntime/include/punc-model.h:11-20).  Text in, punctuated text out: CTokenizer::Tokenize and the mini-sentence /
// string assembly of AddPunc (onnxruntime/src/tokenizer.cpp:275-333, ct-transformer.cpp:39-155,
// ct-transformer-online.cpp:40-152) run here on the host; each Infer is one device call (pfhip_punc_infer[_online]).
//
// Built stand-alone against the small interface below (punc-model.h pulls funasrruntime.h and glog); inside the reference tree
// define PFHIP_WITH_FUNASR to derive from the real funasr::PuncModel.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../../include/pfhip.h"

#ifdef PFHIP_WITH_FUNASR
#include "punc-model.h"
namespace funasr {
using PuncModelHipBase = PuncModel;
}
#else
namespace funasr {
class PuncModelHipBase {                      // punc-model.h:11-20, signature for signature
 public:
  virtual ~PuncModelHipBase() {}
  virtual void InitPunc(const std::string& punc_model, const std::string& punc_config, const std::string& token_file,
                        int thread_num) = 0;
  virtual std::string AddPunc(const char* sz_input, std::string language = "zh-cn") { (void)sz_input; (void)language; return ""; }
  virtual std::string AddPunc(const char* sz_input, std::vector<std::string>& arr_cache, std::string language = "zh-cn") {
    (void)sz_input; (void)arr_cache; (void)language; return "";
  }
  bool is_online = false;
};
}  // namespace funasr
#endif

namespace funasr {

// CTokenizer (tokenizer.h / tokenizer.cpp) without the optional cppjieba segmenter (`seg_jieba`, a third-party dependency):
// token list from tokens.json, punctuation list from the manifest's config.punc_list (config.yaml model_conf.punc_list,
// :144-156), default the six entries the constants of com-define.h:128-135 index.
class PuncTokenizerHip {
 public:
  bool Open(const std::string& manifest_json, const std::string& token_file);
  void Tokenize(const char* str_info, std::vector<std::string>& str_out, std::vector<int>& id_out) const;
  const std::string& Id2Punc(int id) const { return id2punc_[(size_t)id]; }
  bool IsPunc(const std::string& s) const { return punc2id_.count(s) != 0; }
  size_t VocabSize() const { return id2token_.size(); }

 private:
  std::vector<std::string> id2token_, id2punc_;
  std::map<std::string, int> token2id_, punc2id_;
};

class CTTransformerHip : public PuncModelHipBase {
 public:
  ~CTTransformerHip() override;
  // punc_model = <punc-dir>/model.onnx | model_quant.onnx, punc_config = <punc-dir>/config.yaml, token_file = <punc-dir>/tokens.json
  // as the reference passes them (offline-stream.cpp:111-127, tpass-stream.cpp:104-134) — or a container pair
  // (x.pfhip.bin + its JSON manifest); a directory that holds only a converted container is found from the ONNX name
  void InitPunc(const std::string& punc_model, const std::string& punc_config, const std::string& token_file,
                int thread_num) override;
  std::string AddPunc(const char* sz_input, std::string language = "zh-cn") override;
  std::string AddPunc(const char* sz_input, std::vector<std::string>& arr_cache, std::string language = "zh-cn") override;
  void SetDevice(int device) { device_ = device; }
  pfhip_punc* Handle() const { return handle_; }

 protected:
  std::vector<int> Infer(const std::vector<int32_t>& ids, int cache_size) const;
  pfhip_punc* handle_ = nullptr;
  PuncTokenizerHip tokenizer_;
  int device_ = 0;
};

class CTTransformerOnlineHip : public CTTransformerHip {
 public:
  CTTransformerOnlineHip() { is_online = true; }                 // ct-transformer-online.cpp:10-13
  std::string AddPunc(const char* sz_input, std::string language = "zh-cn") override { (void)sz_input; (void)language; return ""; }
  std::string AddPunc(const char* sz_input, std::vector<std::string>& arr_cache, std::string language = "zh-cn") override;
};

// tpass-stream.cpp:100-135 / offline-stream.cpp:105-129: <punc_dir>/model.onnx (model_quant.onnx with PUNC_QUANT) + config.yaml +
// tokens.json; the realtime class when the model path contains "realtime" (the reference's test), nullptr when files are missing.
PuncModelHipBase* CreatePuncModelHip(const std::string& punc_dir, int thread_num, bool allow_online, bool quantized = false);

}  // namespace funasr
