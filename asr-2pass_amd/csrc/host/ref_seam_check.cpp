// Compile-only check of the plug-in seam (tests/test_ref_headers.py, -DPFHIP_WITH_FUNASR -fsyntax-only against the
// reference's real headers): the three adapters are concrete classes behind funasr::Model / VadModel / PuncModel, are
// found by the dynamic_cast of FunASRWfstDecoderInit, and the calls the reference makes through the base pointers
// (offline-stream.cpp:89,102; tpass-stream.cpp:95-97; funasrruntime.cpp:260-268,841-850) resolve to the overrides.
#include "ct_transformer_hip.h"
#include "fsmn_vad_hip.h"
#include "paraformer_hip.h"

#include <unistd.h>

#include <map>
#include <memory>
#include <type_traits>

#include "com-define.h"
#include "util.h"

#ifndef PFHIP_WITH_FUNASR
#error "this file only makes sense against the reference headers"
#endif

static_assert(std::is_base_of<funasr::Model, funasr::ParaformerHip>::value, "Model seam");
static_assert(std::is_base_of<funasr::WfstDecodable, funasr::ParaformerHip>::value, "WfstDecodable seam");
static_assert(!std::is_abstract<funasr::ParaformerHip>::value, "ParaformerHip must be instantiable");
static_assert(std::is_base_of<funasr::VadModel, funasr::FsmnVadHip>::value && !std::is_abstract<funasr::FsmnVadHip>::value, "VadModel seam");
static_assert(std::is_base_of<funasr::VadModel, funasr::FsmnVadOnlineHip>::value && !std::is_abstract<funasr::FsmnVadOnlineHip>::value, "VadModel seam (online)");
static_assert(std::is_base_of<funasr::PuncModel, funasr::CTTransformerHip>::value && !std::is_abstract<funasr::CTTransformerHip>::value, "PuncModel seam");
static_assert(std::is_base_of<funasr::PuncModel, funasr::CTTransformerOnlineHip>::value, "PuncModel seam (online)");

// the member-function pointers below only convert if the adapter's signature IS the base's virtual (no silent overload)
using FwdBatch = std::vector<std::string> (funasr::Model::*)(float**, int*, bool, const std::vector<std::vector<float>>&, void*, int);
using InitAsr5 = void (funasr::Model::*)(const std::string&, const std::string&, const std::string&, const std::string&, int);
using InitAsr6 = void (funasr::Model::*)(const std::string&, const std::string&, const std::string&, const std::string&, const std::string&, int);
using InitAsr9 = void (funasr::Model::*)(const std::string&, const std::string&, const std::string&, const std::string&, const std::string&,
                                         const std::string&, const std::string&, int, const std::string&);
using Fwd1 = std::string (funasr::Model::*)(float*, int, bool, const std::vector<std::vector<float>>&, void*);
using InitLm3 = void (funasr::Model::*)(const std::string&, const std::string&, const std::string&);
using InitLm4 = void (funasr::Model::*)(const std::string&, const std::string&, const std::string&, const std::string&);
static FwdBatch kFwd = &funasr::Model::Forward;
static InitAsr5 kInit = &funasr::Model::InitAsr;
static InitAsr6 kInit6 = &funasr::Model::InitAsr;
static InitAsr9 kInit9 = &funasr::Model::InitAsr;
static Fwd1 kFwd1 = &funasr::Model::Forward;
// every InitAsr the factories call is OVERRIDDEN by the adapter, not inherited as model.h's empty default (round 3's gap: the
// nine-argument call of tpass-stream.cpp:76-77 was a silent no-op)
static void (funasr::ParaformerHip::*kOwn5)(const std::string&, const std::string&, const std::string&, const std::string&, int) = &funasr::ParaformerHip::InitAsr;
static void (funasr::ParaformerHip::*kOwn6)(const std::string&, const std::string&, const std::string&, const std::string&, const std::string&, int) = &funasr::ParaformerHip::InitAsr;
static void (funasr::ParaformerHip::*kOwn9)(const std::string&, const std::string&, const std::string&, const std::string&, const std::string&, const std::string&,
                                            const std::string&, int, const std::string&) = &funasr::ParaformerHip::InitAsr;
static_assert(std::is_base_of<funasr::Model, funasr::ParaformerOnlineHip>::value && !std::is_abstract<funasr::ParaformerOnlineHip>::value, "Model seam (online)");
static InitLm3 kLm3 = &funasr::Model::InitLm;
static InitLm4 kLm4 = &funasr::Model::InitLm;

// ---- OfflineStream::OfflineStream's acoustic-model block (offline-stream.cpp:32-103) with ParaformerHip in the factory switch:
// the SAME calls in the SAME order with the SAME strings (file-name macros of com-define.h:52-88, PathAppend of util.h) ----------
std::unique_ptr<funasr::Model> MakeAsrOffline(std::map<std::string, std::string>& model_path, int thread_num, bool use_gpu, int batch_size) {
  using namespace funasr;
  std::unique_ptr<Model> asr_handle;
  std::string am_model_path, am_cmvn_path, am_config_path, token_path, hw_cpu_model_path, hw_gpu_model_path, seg_dict_path;
  asr_handle = std::unique_ptr<Model>(new ParaformerHip());            // <- the one edited line of the factory (:39 / :46-53)
  asr_handle->SetBatchSize(batch_size);                                  // :41
  hw_cpu_model_path = PathAppend(model_path.at(MODEL_DIR), MODEL_EB_NAME);
  hw_gpu_model_path = PathAppend(model_path.at(MODEL_DIR), TORCH_MODEL_EB_NAME);
  seg_dict_path = PathAppend(model_path.at(MODEL_DIR), MODEL_SEG_DICT);
  if (access(hw_cpu_model_path.c_str(), F_OK) == 0) {                    // :63-67: InitHwCompiler BEFORE InitAsr
    asr_handle->InitHwCompiler(hw_cpu_model_path, thread_num);
    asr_handle->InitSegDict(seg_dict_path);
  }
  if (use_gpu && access(hw_gpu_model_path.c_str(), F_OK) == 0) {         // :68-72
    asr_handle->InitHwCompiler(hw_gpu_model_path, thread_num);
    asr_handle->InitSegDict(seg_dict_path);
  }
  am_model_path = PathAppend(model_path.at(MODEL_DIR), MODEL_NAME);      // :74-77
  if (model_path.find(QUANTIZE) != model_path.end() && model_path.at(QUANTIZE) == "true")
    am_model_path = PathAppend(model_path.at(MODEL_DIR), QUANT_MODEL_NAME);
  if (use_gpu) {                                                         // :79-84
    am_model_path = PathAppend(model_path.at(MODEL_DIR), TORCH_MODEL_NAME);
    if (model_path.find(BLADEDISC) != model_path.end() && model_path.at(BLADEDISC) == "true")
      am_model_path = PathAppend(model_path.at(MODEL_DIR), BLADE_MODEL_NAME);
  }
  am_cmvn_path = PathAppend(model_path.at(MODEL_DIR), AM_CMVN_NAME);
  am_config_path = PathAppend(model_path.at(MODEL_DIR), AM_CONFIG_NAME);
  token_path = PathAppend(model_path.at(MODEL_DIR), TOKEN_PATH);
  asr_handle->InitAsr(am_model_path, am_cmvn_path, am_config_path, token_path, thread_num);      // :89
  if (model_path.find(LM_DIR) != model_path.end() && model_path.at(LM_DIR) != "") {             // :92-103
    std::string fst_path = PathAppend(model_path.at(LM_DIR), LM_FST_RES), lm_config_path = PathAppend(model_path.at(LM_DIR), LM_CONFIG_NAME),
                lex_path = PathAppend(model_path.at(LM_DIR), LEX_PATH);
    asr_handle->InitLm(fst_path, lm_config_path, lex_path);
  }
  return asr_handle;
}

// ---- TpassStream::TpassStream's acoustic-model block (tpass-stream.cpp:31-98) -------------------------------------------------
std::unique_ptr<funasr::Model> MakeAsrTpass(std::map<std::string, std::string>& model_path, int thread_num) {
  using namespace funasr;
  std::unique_ptr<Model> asr_handle;
  std::string am_model_path, en_model_path, de_model_path, am_cmvn_path, am_config_path, token_path, online_token_path, online_config_path,
      hw_compile_model_path, seg_dict_path;
  asr_handle = std::unique_ptr<Model>(new ParaformerHip());            // <- the one edited line of the factory (:49)
  hw_compile_model_path = PathAppend(model_path.at(MODEL_DIR), MODEL_EB_NAME);
  seg_dict_path = PathAppend(model_path.at(MODEL_DIR), MODEL_SEG_DICT);
  if ((access(hw_compile_model_path.c_str(), F_OK) == 0) && (access(seg_dict_path.c_str(), F_OK) == 0)) {      // :54-60
    asr_handle->InitHwCompiler(hw_compile_model_path, thread_num);
    asr_handle->InitSegDict(seg_dict_path);
  }
  am_model_path = PathAppend(model_path.at(OFFLINE_MODEL_DIR), MODEL_NAME);                      // :62-72
  en_model_path = PathAppend(model_path.at(ONLINE_MODEL_DIR), ENCODER_NAME);
  de_model_path = PathAppend(model_path.at(ONLINE_MODEL_DIR), DECODER_NAME);
  online_token_path = PathAppend(model_path.at(ONLINE_MODEL_DIR), TOKEN_PATH);
  online_config_path = PathAppend(model_path.at(ONLINE_MODEL_DIR), AM_CONFIG_NAME);
  if (model_path.find(QUANTIZE) != model_path.end() && model_path.at(QUANTIZE) == "true") {
    am_model_path = PathAppend(model_path.at(OFFLINE_MODEL_DIR), QUANT_MODEL_NAME);
    en_model_path = PathAppend(model_path.at(ONLINE_MODEL_DIR), QUANT_ENCODER_NAME);
    de_model_path = PathAppend(model_path.at(ONLINE_MODEL_DIR), QUANT_DECODER_NAME);
  }
  am_cmvn_path = PathAppend(model_path.at(ONLINE_MODEL_DIR), AM_CMVN_NAME);
  am_config_path = PathAppend(model_path.at(MODEL_DIR), AM_CONFIG_NAME);
  token_path = PathAppend(model_path.at(MODEL_DIR), TOKEN_PATH);
  asr_handle->InitAsr(am_model_path, en_model_path, de_model_path, am_cmvn_path, am_config_path, token_path, online_token_path, thread_num,
                      online_config_path);                                                       // :76-77: the NINE-argument overload
  if (model_path.find(LM_DIR) != model_path.end() && model_path.at(LM_DIR) != "") {             // :85-98
    std::string fst_path = PathAppend(model_path.at(LM_DIR), LM_FST_RES), lm_config_path = PathAppend(model_path.at(LM_DIR), LM_CONFIG_NAME),
                lex_path = PathAppend(model_path.at(LM_DIR), LEX_PATH), lm_units_path = PathAppend(model_path.at(LM_DIR), LM_UNITS_PATH);
    if (access(lm_units_path.c_str(), F_OK) != 0) asr_handle->InitLm(fst_path, lm_config_path, lex_path, "");
    else asr_handle->InitLm(fst_path, lm_config_path, lex_path, lm_units_path);
  }
  return asr_handle;
}

// model.cpp:4-58: CreateModel(model_path, thread_num, ASR_ONLINE) calls the six-argument InitAsr; CreateModel(asr_handle,
// chunk_size) builds the per-connection object from the shared one — as TpassOnlineStream does (tpass-online-stream.cpp:14-15)
funasr::Model* MakeOnlineOnly(std::map<std::string, std::string>& model_path, int thread_num) {
  using namespace funasr;
  Model* mm = new ParaformerHip();
  mm->InitAsr(PathAppend(model_path.at(MODEL_DIR), ENCODER_NAME), PathAppend(model_path.at(MODEL_DIR), DECODER_NAME),
              PathAppend(model_path.at(MODEL_DIR), AM_CMVN_NAME), PathAppend(model_path.at(MODEL_DIR), AM_CONFIG_NAME),
              PathAppend(model_path.at(MODEL_DIR), TOKEN_PATH), thread_num);
  return mm;
}
std::unique_ptr<funasr::Model> MakeAsrOnline(funasr::Model* asr_handle, std::vector<int> chunk_size) {
  return std::unique_ptr<funasr::Model>(new funasr::ParaformerOnlineHip(asr_handle, chunk_size));
}
std::string OnlineChunk(funasr::Model* asr_online_handle, float* data, int len, bool is_final) {  // funasrruntime.cpp:540
  return asr_online_handle->Forward(data, len, is_final);
}

// the VAD and punctuation blocks of both factories (offline-stream.cpp:6-29,105-129; tpass-stream.cpp:6-29,100-135)
std::unique_ptr<funasr::VadModel> MakeVadFromDir(std::map<std::string, std::string>& model_path, int thread_num) {
  using namespace funasr;
  std::string vad_model_path = PathAppend(model_path.at(VAD_DIR), MODEL_NAME);
  if (model_path.find(VAD_QUANT) != model_path.end() && model_path.at(VAD_QUANT) == "true")
    vad_model_path = PathAppend(model_path.at(VAD_DIR), QUANT_MODEL_NAME);
  std::unique_ptr<VadModel> vad_handle(new FsmnVadHip());
  vad_handle->InitVad(vad_model_path, PathAppend(model_path.at(VAD_DIR), VAD_CMVN_NAME), PathAppend(model_path.at(VAD_DIR), VAD_CONFIG_NAME), thread_num);
  return vad_handle;
}
std::unique_ptr<funasr::PuncModel> MakePuncFromDir(std::map<std::string, std::string>& model_path, int thread_num) {
  using namespace funasr;
  std::string punc_model_path = PathAppend(model_path.at(PUNC_DIR), MODEL_NAME);
  if (model_path.find(PUNC_QUANT) != model_path.end() && model_path.at(PUNC_QUANT) == "true")
    punc_model_path = PathAppend(model_path.at(PUNC_DIR), QUANT_MODEL_NAME);
  std::unique_ptr<PuncModel> punc_handle;
  if (punc_model_path.find("realtime") != std::string::npos) punc_handle.reset(new CTTransformerOnlineHip());      // tpass-stream.cpp:124-134
  else punc_handle.reset(new CTTransformerHip());
  punc_handle->InitPunc(punc_model_path, PathAppend(model_path.at(PUNC_DIR), PUNC_CONFIG_NAME), PathAppend(model_path.at(PUNC_DIR), TOKEN_PATH), thread_num);
  return punc_handle;
}

funasr::Decoder* MakeDecoder(funasr::Model* asr_handle) {                                       // funasrruntime.cpp:841-850
  auto* paraformer = dynamic_cast<funasr::WfstDecodable*>(asr_handle);
  if (paraformer == nullptr) return nullptr;
  (void)paraformer->GetLm();
  (void)paraformer->GetPhoneSet();
  (void)paraformer->GetLmVocab();
  (void)paraformer->GetVocab();
  return nullptr;
}

std::vector<std::string> Infer(funasr::Model* asr_handle, float** buff, int* len, const std::vector<std::vector<float>>& hw_emb,
                               void* dec_handle, int batch_in) {                                // funasrruntime.cpp:260-268
  funasr::Decoder* wfst_decoder = (funasr::Decoder*)dec_handle;
  if (wfst_decoder) wfst_decoder->StartUtterance();
  std::string hotwords;
  (void)asr_handle->CompileHotwordEmbedding(hotwords);
  (void)(asr_handle->*kFwd)(buff, len, true, hw_emb, dec_handle, batch_in);
  (void)kInit; (void)kInit6; (void)kInit9; (void)kFwd1; (void)kOwn5; (void)kOwn6; (void)kOwn9; (void)kLm3; (void)kLm4;
  return asr_handle->Forward(buff, len, true, hw_emb, dec_handle, batch_in);
}

std::vector<std::vector<int>> Vad(funasr::VadModel* vad, std::vector<float>& waves) {            // audio.cpp:1183-1196
  vad->SetConfig(800, 60000);
  return vad->Infer(waves, true);
}

std::string Punc(funasr::PuncModel* punc, const char* text, std::vector<std::string>& cache) {   // funasrruntime.cpp:609-614
  return punc->is_online ? punc->AddPunc(text, cache) : punc->AddPunc(text);
}

funasr::VadModel* MakeVad() { return new funasr::FsmnVadHip(); }
funasr::VadModel* MakeVadOnline(funasr::FsmnVadHip* h) { return new funasr::FsmnVadOnlineHip(h); }
funasr::PuncModel* MakePunc(bool online) { return online ? static_cast<funasr::PuncModel*>(new funasr::CTTransformerOnlineHip()) : new funasr::CTTransformerHip(); }
