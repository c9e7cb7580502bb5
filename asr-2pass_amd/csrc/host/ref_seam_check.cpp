// Compile-only check of the plug-in seam (tests/test_ref_headers.py, -DPFHIP_WITH_FUNASR -fsyntax-only against the
// reference's real headers): the three adapters are concrete classes behind funasr::Model / VadModel / PuncModel, are
// found by the dynamic_cast of FunASRWfstDecoderInit, and the calls the reference makes through the base pointers
// (offline-stream.cpp:89,102; tpass-stream.cpp:95-97; funasrruntime.cpp:260-268,841-850) resolve to the overrides.
#include "ct_transformer_hip.h"
#include "fsmn_vad_hip.h"
#include "paraformer_hip.h"

#include <memory>
#include <type_traits>

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
using InitLm3 = void (funasr::Model::*)(const std::string&, const std::string&, const std::string&);
using InitLm4 = void (funasr::Model::*)(const std::string&, const std::string&, const std::string&, const std::string&);
static FwdBatch kFwd = &funasr::Model::Forward;
static InitAsr5 kInit = &funasr::Model::InitAsr;
static InitLm3 kLm3 = &funasr::Model::InitLm;
static InitLm4 kLm4 = &funasr::Model::InitLm;

std::unique_ptr<funasr::Model> MakeAsr(const std::string& dir, int threads, int batch) {       // offline-stream.cpp:40-48,89,102
  std::unique_ptr<funasr::Model> asr_handle(new funasr::ParaformerHip());
  asr_handle->SetBatchSize(batch);
  asr_handle->InitAsr(dir + "/model.pfhip.bin", dir + "/am.mvn", dir + "/model.pfhip.json", dir + "/tokens.json", threads);
  asr_handle->InitLm(dir + "/TLG.fst", dir + "/config.yaml", dir + "/lexicon.txt");
  asr_handle->InitLm(dir + "/TLG.fst", dir + "/config.yaml", dir + "/lexicon.txt", "");
  return asr_handle;
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
  (void)kInit; (void)kLm3; (void)kLm4;
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
