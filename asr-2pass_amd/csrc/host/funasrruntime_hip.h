// The offline slice of the reference's handle API (onnxruntime/include/funasrruntime.h:69-115), same names, argument
// meaning and error behaviour, on top of the HIP plug-ins — what the websocket server's decoder threads call
// (websocket/bin/websocket-server.cpp:81-89,173,209-210):
//
//   FunOfflineInit          funasrruntime.cpp:36-40     OfflineStream: VAD + acoustic model (+ batch size)
//   FunOfflineInferBuffer   funasrruntime.cpp:208-337   LoadPcmwav -> CutSplit -> FetchDynamic -> Model::Forward -> re-ordering
//   FunASRGetResult / FunASRGetStamp / FunASRGetRetSnippetTime / FunASRFreeResult   funasrruntime.cpp:540-640
//   CompileHotwordEmbedding funasrruntime.cpp:862-867
//   FunOfflineUninit        funasrruntime.cpp:806-814
//
// Stand-alone mirror: inside the reference tree the real funasrruntime.cpp stays and only the Model classes are swapped
// (INTEGRATION.md).  With PUNC_DIR given the text goes through CTTransformerHip::AddPunc like the reference's punc_handle
// (funasrruntime.cpp:317-320).  What is NOT here: audio container decode (ffmpeg), ITN (WFST text normaliser) and sentence
// stamps — text handling above the hot path.  Result text = the vocabulary strings of the greedy tokens (or space-separated ids without
// a tokens.json); stamps have the reference's "[[b,e],[b,e]...]" millisecond format.
#pragma once
#include <map>
#include <string>
#include <vector>

#define MODEL_DIR "model-dir"          // com-define.h:14-28 keys of the model_path map
#define OFFLINE_MODEL_DIR "model-dir"
#define ONLINE_MODEL_DIR "online-model-dir"
#define VAD_DIR "vad-dir"
#define PUNC_DIR "punc-dir"
#define QUANTIZE "quantize"
#define VAD_QUANT "vad-quant"
#define PUNC_QUANT "punc-quant"
#define TOKEN_PATH_KEY "token-path"    // (not a key of the reference: an optional override of <model-dir>/tokens.json for harness runs)
// com-define.h:52-88 file names inside those directories
#define MODEL_NAME "model.onnx"
#define QUANT_MODEL_NAME "model_quant.onnx"
#define MODEL_EB_NAME "model_eb.onnx"
#define TORCH_MODEL_NAME "model.torchscript"
#define TORCH_MODEL_EB_NAME "model_eb.torchscript"
#define ENCODER_NAME "model.onnx"
#define QUANT_ENCODER_NAME "model_quant.onnx"
#define DECODER_NAME "decoder.onnx"
#define QUANT_DECODER_NAME "decoder_quant.onnx"
#define AM_CMVN_NAME "am.mvn"
#define AM_CONFIG_NAME "config.yaml"
#define VAD_CMVN_NAME "am.mvn"
#define VAD_CONFIG_NAME "config.yaml"
#define MODEL_SEG_DICT "seg_dict"
#define TOKEN_PATH "tokens.json"

typedef void* FUNASR_HANDLE;
typedef void* FUNASR_RESULT;
typedef void* FUNASR_DEC_HANDLE;
typedef enum { RASR_NONE = -1, RASRM_CTC_GREEDY_SEARCH = 0 } FUNASR_MODE;      // funasrruntime.h:30-35 (subset)
typedef enum { ASR_OFFLINE = 0, ASR_ONLINE = 1, ASR_TWO_PASS = 2 } ASR_TYPE;   // funasrruntime.h:48-52
typedef void (*QM_CALLBACK)(int cur_step, int n_total);

// model_path as the reference's servers fill it (funasr-wss-server.cpp:203-320): MODEL_DIR / VAD_DIR / PUNC_DIR are directories in
// the reference's layout (model.onnx | model_quant.onnx with QUANTIZE = "true" | model.torchscript with use_gpu, model_eb.onnx,
// seg_dict, am.mvn, config.yaml, tokens.json); the constructor makes the calls of OfflineStream::OfflineStream
// (offline-stream.cpp:4-129) on the HIP plug-ins.  A directory that holds a converted container (x.pfhip.{bin,json}) instead of
// ONNX files works too.  A model that fails to load ends the process like the reference (paraformer.cpp:43-46).
FUNASR_HANDLE FunOfflineInit(std::map<std::string, std::string>& model_path, int thread_num, bool use_gpu = true,
                             int batch_size = 1);
// sz_buf: n_len BYTES of little-endian int16 PCM (wav_format "pcm"/"PCM"; anything else returns nullptr: the
// reference would hand it to ffmpeg).  Returns nullptr on a bad handle/format, a result with empty msg for empty audio.
FUNASR_RESULT FunOfflineInferBuffer(FUNASR_HANDLE handle, const char* sz_buf, int n_len, FUNASR_MODE mode,
                                    QM_CALLBACK fn_callback, const std::vector<std::vector<float>>& hw_emb,
                                    int sampling_rate = 16000, std::string wav_format = "pcm", bool itn = true,
                                    int vad_tail_sil = 800, int vad_max_len = 60000, FUNASR_DEC_HANDLE dec_handle = nullptr);
const std::vector<std::vector<float>> CompileHotwordEmbedding(FUNASR_HANDLE handle, std::string& hotwords, ASR_TYPE mode = ASR_OFFLINE);
const char* FunASRGetResult(FUNASR_RESULT result, int n_index);
const char* FunASRGetStamp(FUNASR_RESULT result);
float FunASRGetRetSnippetTime(FUNASR_RESULT result);
void FunASRFreeResult(FUNASR_RESULT result);
void FunOfflineUninit(FUNASR_HANDLE handle);

// ---- 2-pass slice (funasrruntime.h:121-132; funasrruntime.cpp:52-63, 491-646, 816-842) ---------------------------------------
//   FunTpassInit         TpassStream: offline model (MODEL_DIR) + online model (ONLINE_MODEL_DIR) + VAD (VAD_DIR), shared
//   FunTpassOnlineInit   TpassOnlineStream: one per connection — ParaformerOnline stream, FsmnVadOnline, Audio
//   FunTpassInferBuffer  LoadPcmwavOnline -> Split (online VAD) -> streaming Forward per chunk -> offline Forward per closed
//                        segment; msg = text of this call's streaming chunks, tpass_msg = 2nd-pass text of a segment that
//                        closed in this call; with PUNC_DIR both go through punc_online_handle->AddPunc with punc_cache[0] /
//                        punc_cache[1] as in the reference (:543-556, :609-614) — the realtime class when the directory
//                        name contains "realtime" (tpass-stream.cpp:124-134)
FUNASR_HANDLE FunTpassInit(std::map<std::string, std::string>& model_path, int thread_num);
FUNASR_HANDLE FunTpassOnlineInit(FUNASR_HANDLE tpass_handle, std::vector<int> chunk_size = {5, 10, 5});
FUNASR_RESULT FunTpassInferBuffer(FUNASR_HANDLE handle, FUNASR_HANDLE online_handle, const char* sz_buf, int n_len,
                                  std::vector<std::vector<std::string>>& punc_cache, bool input_finished = true,
                                  int sampling_rate = 16000, std::string wav_format = "pcm", ASR_TYPE mode = ASR_TWO_PASS,
                                  const std::vector<std::vector<float>>& hw_emb = {{0.0f}}, bool itn = true, int vad_tail_sil = 800,
                                  int vad_max_len = 60000, FUNASR_DEC_HANDLE dec_handle = nullptr);
const char* FunASRGetTpassResult(FUNASR_RESULT result, int n_index);
void FunTpassOnlineUninit(FUNASR_HANDLE online_handle);
void FunTpassUninit(FUNASR_HANDLE handle);

// Inspection for tests: token ids per VAD segment in time order and the segments (samples) of the last result.
const std::vector<std::vector<int>>& FunASRGetSegmentIds(FUNASR_RESULT result);
const std::vector<std::pair<int, int>>& FunASRGetSegments(FUNASR_RESULT result);
// ... the ids the streaming chunks of one FunTpassInferBuffer call emitted
const std::vector<int>& FunASRGetOnlineIds(FUNASR_RESULT result);
// ... and the C-ABI handle of the offline acoustic model behind a FunOfflineInit handle (pfhip_inflight_stats in the harnesses)
struct pfhip_model;
pfhip_model* FunOfflineGetAsrHandle(FUNASR_HANDLE handle);
