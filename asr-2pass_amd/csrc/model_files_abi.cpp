// C ABI over model_files.cpp: the file-reading half of Paraformer::InitAsr / FsmnVad::InitVad / CTTransformer::InitPunc
// (onnxruntime/src/paraformer.cpp:21-53,56-154; fsmn-vad.cpp:10-19; ct-transformer.cpp:14-37) with the reference's own arguments.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "internal.h"
#include "model_files.h"

struct pfhip_container {
  pfhip_files::Container c;
};

namespace {
using pfhip_detail::fail;

const char* nz(const char* s) { return s ? s : ""; }

pfhip_status read_files(const std::string& kind, const char* model, const char* second, const char* hotword, const char* cmvn,
                        const char* config, pfhip_files::Container& c) {
  try {
    if (kind == "asr") pfhip_files::load_asr(nz(model), nz(second), nz(hotword), nz(cmvn), nz(config), c);
    else if (kind == "vad") pfhip_files::load_vad(nz(model), nz(cmvn), nz(config), c);
    else if (kind == "punc") pfhip_files::load_punc(nz(model), nz(config), c);
    else return fail(PFHIP_ERR_ARG, "kind must be asr, vad or punc");
  } catch (const pfhip_files::FormatError& e) {
    return fail(PFHIP_ERR_FORMAT, e.what());
  } catch (const std::exception& e) {
    return fail(PFHIP_ERR_FORMAT, e.what());
  }
  return PFHIP_OK;
}
}  // namespace

extern "C" {

pfhip_status pfhip_read_model_files(const char* kind, const char* model, const char* second, const char* hotword, const char* cmvn,
                                    const char* config, pfhip_container** out) {
  if (!kind || !model || !out) return fail(PFHIP_ERR_ARG, "null argument");
  std::unique_ptr<pfhip_container> pc(new pfhip_container);
  const pfhip_status st = read_files(kind, model, second, hotword, cmvn, config, pc->c);
  if (st) return st;
  *out = pc.release();
  return PFHIP_OK;
}

const float* pfhip_container_blob(const pfhip_container* c, size_t* bytes) {
  if (!c) return nullptr;
  if (bytes) *bytes = c->c.blob.size() * sizeof(float);
  return c->c.blob.data();
}
const char* pfhip_container_manifest(const pfhip_container* c) { return c ? c->c.manifest.c_str() : nullptr; }
int pfhip_container_from_cache(const pfhip_container* c) { return c && c->c.from_cache ? 1 : 0; }
void pfhip_container_free(pfhip_container* c) { delete c; }

// {"initializers": n, "initializer_bytes": b, "nodes": n, "open_inputs": n, "float_sum": s}: what the wire-format walk saw in one
// .onnx file (the tests compare it with the Python reader on the genuine files the reference ships)
pfhip_status pfhip_onnx_summary(const char* path, char* out, size_t cap) {
  if (!path || !out || cap == 0) return fail(PFHIP_ERR_ARG, "null argument");
  try {
    pfhip_files::OnnxModel m;
    pfhip_files::read_onnx(path, m);
    size_t bytes = 0;
    double sum = 0;
    pfhip_files::State st;
    pfhip_files::torch_style_state(m, st);
    for (const auto& kv : m.initializers) {
      const pfhip_files::Initializer& t = kv.second;
      const size_t esz = t.dtype == 1 || t.dtype == 6 || t.dtype == 12 ? 4 : t.dtype == 7 || t.dtype == 11 || t.dtype == 13 ? 8 : t.dtype == 10 || t.dtype == 16 || t.dtype == 4 || t.dtype == 5 ? 2 : 1;
      bytes += t.count() * esz;
    }
    for (const auto& kv : st) {
      std::vector<float> v(kv.second.count());
      kv.second.copy_to(v.data());
      for (float x : v) sum += x;
    }
    const std::string js = "{\"initializers\": " + std::to_string(m.initializers.size()) + ", \"initializer_bytes\": " + std::to_string(bytes) +
                           ", \"nodes\": " + std::to_string(m.nodes.size()) + ", \"open_inputs\": " + std::to_string(pfhip_files::check_closed(m).size()) +
                           ", \"state_tensors\": " + std::to_string(st.size()) + ", \"float_sum\": " + std::to_string(sum) + "}";
    if (js.size() + 1 > cap) return fail(PFHIP_ERR_CAPACITY, "summary buffer too small");
    std::memcpy(out, js.c_str(), js.size() + 1);
  } catch (const std::exception& e) {
    return fail(PFHIP_ERR_FORMAT, e.what());
  }
  return PFHIP_OK;
}

pfhip_status pfhip_create_from_files(const char* am_model, const char* second_model, const char* hw_model, const char* am_cmvn,
                                     const char* am_config, int device, pfhip_model** out) {
  if (!am_model || !out) return fail(PFHIP_ERR_ARG, "null argument");
  pfhip_files::Container c;
  const pfhip_status st = read_files("asr", am_model, second_model, hw_model, am_cmvn, am_config, c);
  if (st) return st;
  return pfhip_create_from_memory(c.blob.data(), c.blob.size() * sizeof(float), c.manifest.c_str(), device, out);
}

pfhip_status pfhip_vad_create_from_files(const char* vad_model, const char* vad_cmvn, const char* vad_config, int device, pfhip_vad** out) {
  if (!vad_model || !out) return fail(PFHIP_ERR_ARG, "null argument");
  pfhip_files::Container c;
  const pfhip_status st = read_files("vad", vad_model, nullptr, nullptr, vad_cmvn, vad_config, c);
  if (st) return st;
  return pfhip_vad_create_from_memory(c.blob.data(), c.blob.size() * sizeof(float), c.manifest.c_str(), device, out);
}

pfhip_status pfhip_punc_create_from_files(const char* punc_model, const char* punc_config, int device, pfhip_punc** out) {
  if (!punc_model || !out) return fail(PFHIP_ERR_ARG, "null argument");
  pfhip_files::Container c;
  const pfhip_status st = read_files("punc", punc_model, nullptr, nullptr, nullptr, punc_config, c);
  if (st) return st;
  return pfhip_punc_create_from_memory(c.blob.data(), c.blob.size() * sizeof(float), c.manifest.c_str(), device, out);
}

}  // extern "C"
