// Reader of the reference's OWN model-file contract (onnxruntime/include/com-define.h:52-88): model.onnx / model_quant.onnx
// [+ decoder.onnx, model_eb.onnx] + am.mvn + config.yaml -> the weight container pfhip_create_from_memory takes.  What
// Paraformer::InitAsr / FsmnVad::InitVad / CTTransformer::InitPunc open at server start (onnxruntime/src/paraformer.cpp:21-53,
// 56-154,178-241,325-360; fsmn-vad.cpp:10-70; ct-transformer.cpp:14-37) — there through Ort::Session, YAML::LoadFile and LoadCmvn,
// here through a dependency-free protobuf walk, a YAML subset reader and the same am.mvn row parser.  Host code only (no HIP).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pfhip_files {

struct FormatError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---- ONNX (protobuf wire format) ----------------------------------------------------------------------------------------
struct Initializer {
  std::string name;
  std::vector<int64_t> dims;
  int dtype = 0;                     // TensorProto.DataType
  const uint8_t* raw = nullptr;      // raw_data (points into the file image)
  size_t raw_bytes = 0;
  std::vector<float> f32;            // float_data
  std::vector<double> f64;           // double_data
  std::vector<int64_t> ints;         // int32_data / int64_data / uint64_data
  bool external = false;
  size_t count() const;
};

struct Node {
  std::string op_type, name;
  std::vector<std::string> inputs, outputs;
  std::map<std::string, int64_t> int_attrs;
};

struct OnnxModel {
  std::vector<char> image;           // the file; initializers point into it
  int64_t ir_version = 0;
  std::string producer;
  std::map<std::string, Initializer> initializers;
  std::vector<Node> nodes;
  std::vector<std::string> inputs, outputs, external;
};

void read_onnx(const std::string& path, OnnxModel& m);                      // throws FormatError
void read_onnx_bytes(const void* data, size_t bytes, OnnxModel& m);
std::vector<std::string> check_closed(const OnnxModel& m);                  // node inputs nobody produces ("" = well formed)

// A float32 tensor in torch layout: either a view of the file image (optionally a 2-d transpose of it) or owned storage.
struct Array {
  std::vector<int64_t> dims;         // dims AFTER the transpose
  const uint8_t* view = nullptr;     // little-endian float32, unaligned
  bool transposed = false;           // view holds [dims[1], dims[0]]
  std::shared_ptr<std::vector<float>> own;
  size_t count() const;
  void copy_to(float* dst) const;    // row-major, dims order
};
using State = std::map<std::string, Array>;

// {torch state_dict key: Array}: named initializers pass through, anonymous MatMul / Gemm weights are named after their bias
// sibling or node path and transposed back to [out, in], LSTM W/R/B are split per direction with the gates re-ordered
// (ONNX i,o,f,c -> torch i,f,g,o), onnxruntime's dynamic quantisation is folded back to float32.
void torch_style_state(const OnnxModel& m, State& state);

// ---- am.mvn (paraformer.cpp:325-360) and the YAML subset config.yaml uses -----------------------------------------------------
void parse_am_mvn(const std::string& text, std::vector<float>& shift, std::vector<float>& rescale);

struct YNode {
  enum Kind { NUL, SCALAR, MAP, SEQ } kind = NUL;
  std::string s;
  std::vector<std::pair<std::string, YNode>> map;
  std::vector<YNode> seq;
  const YNode* get(const std::string& key) const;
  double number(const std::string& key, double dflt) const;
  std::string str(const std::string& key, const std::string& dflt) const;
};
YNode parse_yaml(const std::string& text);

// ---- the container ------------------------------------------------------------------------------------------------------------
struct Container {
  std::string manifest;              // JSON: {"config": {...}, "tensors": {name: {"shape": [...], "offset": bytes}}, "total_bytes": n}
  std::vector<float> blob;
  std::vector<std::string> sources;  // files read
  bool from_cache = false;
};

// `model` is whatever string the reference passes as am_model / en_model / vad_model / punc_model: <dir>/model.onnx,
// <dir>/model_quant.onnx, <dir>/model.torchscript (offline-stream.cpp:79-84: the GPU flavour's file; the ONNX file beside it is
// read instead) or a container <dir>/x.pfhip.bin (then `config` is its JSON manifest).  `second` = decoder.onnx of the online
// model, `hotword` = model_eb.onnx (InitHwCompiler's argument), both optional ("").  Throws FormatError / std::runtime_error.
void load_asr(const std::string& model, const std::string& second, const std::string& hotword, const std::string& cmvn,
              const std::string& config, Container& out);
void load_vad(const std::string& model, const std::string& cmvn, const std::string& config, Container& out);
void load_punc(const std::string& model, const std::string& config, Container& out);

}  // namespace pfhip_files
