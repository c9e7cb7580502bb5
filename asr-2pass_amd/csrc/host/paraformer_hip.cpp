#include "paraformer_hip.h"

#include "postprocess.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <sstream>

namespace funasr {

namespace {
// Forward is re-entrant like the reference's (one handle, many decoder threads): what it leaves behind for inspection is
// per calling thread
thread_local std::vector<std::vector<int>> tl_last_ids;
thread_local std::vector<std::vector<float>> tl_last_spans;
// (Vocab::Vector2StringV2 updates a member, `last_is_complete_english_`, vocab.h:22, and the reference calls it unguarded from
// every decoder thread; the stand-alone HostVocab keeps that flag in an atomic, the in-tree build uses the reference's own
// class as the reference does.  No lock is taken around the text step.)
}  // namespace

ParaformerHip::ParaformerHip() {
#ifdef PFHIP_WITH_FUNASR
  lm_ = nullptr; phone_set_ = nullptr; lm_vocab = nullptr;       // WfstDecodable leaves its raw pointers uninitialised
#endif
}

ParaformerHip::~ParaformerHip() {
  if (handle_) pfhip_destroy(handle_);
  if (online_handle_) pfhip_destroy(online_handle_);
  delete vocab;
  delete online_vocab;
  delete seg_dict_;
#ifdef PFHIP_WITH_FUNASR
  delete lm_vocab;
  delete phone_set_;
#endif
}

namespace {

int Knob(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e && *e ? std::atoi(e) : dflt;
}

// "key": "value" of the manifest's config block ("" when absent)
std::string ManifestString(const char* man, const char* key) {
  const std::string m = man ? man : "", k = std::string("\"") + key + "\": \"";
  const size_t at = m.find(k);
  if (at == std::string::npos) return "";
  const size_t end = m.find('"', at + k.size());
  return end == std::string::npos ? "" : m.substr(at + k.size(), end - at - k.size());
}

// The file-reading half of InitAsr: the reference's own files -> a device model.  Exits on failure like paraformer.cpp:43-46.
pfhip_model* LoadModelFiles(const std::string& model, const std::string& second, const std::string& hotword, const std::string& cmvn,
                            const std::string& config, int device, std::string* lang) {
  pfhip_container* c = nullptr;
  pfhip_model* h = nullptr;
  if (pfhip_read_model_files("asr", model.c_str(), second.empty() ? nullptr : second.c_str(), hotword.empty() ? nullptr : hotword.c_str(),
                             cmvn.c_str(), config.c_str(), &c) != PFHIP_OK) {
    std::fprintf(stderr, "Error when load am hip model: %s\n", pfhip_last_error());
    std::exit(-1);
  }
  size_t bytes = 0;
  const float* blob = pfhip_container_blob(c, &bytes);
  if (lang) *lang = ManifestString(pfhip_container_manifest(c), "lang");
  const pfhip_status st = pfhip_create_from_memory(blob, bytes, pfhip_container_manifest(c), device, &h);
  pfhip_container_free(c);
  if (st != PFHIP_OK) {
    std::fprintf(stderr, "Error when load am hip model: %s\n", pfhip_last_error());
    std::exit(-1);
  }
  return h;
}

HipVocab* LoadVocab(const std::string& token_file) {
  if (token_file.empty()) return nullptr;
#ifdef PFHIP_WITH_FUNASR
  return new Vocab(token_file.c_str());
#else
  HipVocab* v = new HipVocab();
  if (!v->Load(token_file.c_str())) { delete v; v = nullptr; }
  return v;
#endif
}

}  // namespace

void ParaformerHip::LoadOffline(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config,
                                const std::string& token_file) {
  if (handle_) { pfhip_destroy(handle_); handle_ = nullptr; }
  std::string lang;
  handle_ = LoadModelFiles(am_model, "", hw_model_, am_cmvn, am_config, device_, &lang);
  if (!lang.empty()) language = lang;                           // LoadConfigFromYaml (paraformer.cpp:193-196)
  asr_sample_rate_ = pfhip_sample_rate(handle_);                // frontend_conf.fs (:191)
  if (use_hotword_ && !pfhip_is_contextual(handle_))
    std::fprintf(stderr, "ParaformerHip: %s holds no hotword embedder (bias_embed / bias_encoder): hotwords are ignored\n", hw_model_.c_str());
  // What the decoder threads get, whatever --model-thread-num says: their concurrent Forward calls are merged into packed
  // launches (a lone caller never waits: pfhip_set_batching) and dealt to PFHIP_INFLIGHT execution contexts over the one weight set.
  // (Round 3 kept ONE context for models with the timestamp head: their persistent recurrence held a host lock to the end of the
  // caller's stream.  The recurrences of a device now queue on one stream of their own and the lock covers the enqueue only.)
  const int inflight = Knob("PFHIP_INFLIGHT", 3);
  if (inflight > 1 && pfhip_set_inflight(handle_, inflight) != PFHIP_OK)
    std::fprintf(stderr, "ParaformerHip::InitAsr: %s (one forward at a time)\n", pfhip_last_error());
  const int wait_us = Knob("PFHIP_OFFLINE_WAIT_US", 3000);     // 0 = no merging
  if (wait_us > 0) pfhip_set_batching(handle_, wait_us, Knob("PFHIP_OFFLINE_MAX", 96));   // 96 merged utterances: +16 % over 32 on the long-audio flow
  delete vocab;
  vocab = LoadVocab(token_file);                                // paraformer.cpp:47-48
  // ParaformerTorch::WarmUp (paraformer-torch.cpp:59,477-520): a dummy forward inside InitAsr.  Here through every execution
  // context, at the batch size the server was started with (SetBatchSize precedes InitAsr, offline-stream.cpp:41) and
  // PFHIP_WARMUP_SECONDS of audio per utterance (default 30; 0 = no warm-up), so that each context's workspace is sized too.
  const int warm_s = Knob("PFHIP_WARMUP_SECONDS", 30);
  if (warm_s > 0 && pfhip_warm_up(handle_, batch_size_ > 0 ? batch_size_ : 1, warm_s * asr_sample_rate_) != PFHIP_OK)
    std::fprintf(stderr, "ParaformerHip::InitAsr: warm-up failed: %s\n", pfhip_last_error());
}

void ParaformerHip::LoadOnline(const std::string& en_model, const std::string& de_model, const std::string& am_cmvn,
                               const std::string& am_config, const std::string& token_file) {
  if (online_handle_) { pfhip_destroy(online_handle_); online_handle_ = nullptr; }
  // a container pair stands for both files of the online model
  const bool container = en_model.size() > 10 && en_model.compare(en_model.size() - 10, 10, ".pfhip.bin") == 0;
  online_handle_ = LoadModelFiles(en_model, container ? "" : de_model, "", am_cmvn, am_config, device_, nullptr);
  asr_sample_rate_ = pfhip_sample_rate(online_handle_);
  delete online_vocab;
  online_vocab = LoadVocab(token_file);                         // paraformer.cpp:117
  // One handler thread per connection in the server: their concurrent chunk calls are merged into batched passes; a leader
  // stops waiting as soon as every open connection has queued, so a lone connection pays nothing.  0 switches the queue off.
  pfhip_set_stream_batching(online_handle_, Knob("PFHIP_STREAM_WAIT_US", 3000), Knob("PFHIP_STREAM_MAX", 128));
}

// offline (paraformer.cpp:21-53)
void ParaformerHip::InitAsr(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config,
                            const std::string& token_file, int thread_num) {
  // `thread_num` is the server's --model-thread-num — onnxruntime intra-op threads, default 1 (funasr-wss-server.cpp:105-106,
  // 452,511; websocket-server.cpp:433; used at paraformer.cpp:35) — NOT the number of decoder threads that share this handle
  // (--decoder-thread-num, 8, or 16 in run_server_offline.sh:39).  It has no meaning here.
  (void)thread_num;
  LoadOffline(am_model, am_cmvn, am_config, token_file);
}

// online (paraformer.cpp:56-131)
void ParaformerHip::InitAsr(const std::string& en_model, const std::string& de_model, const std::string& am_cmvn,
                            const std::string& am_config, const std::string& token_file, int thread_num) {
  (void)thread_num;
  LoadOnline(en_model, de_model, am_cmvn, am_config, token_file);
}

// 2pass (paraformer.cpp:134-154): online first, then the offline session with the offline model's config and tokens
void ParaformerHip::InitAsr(const std::string& am_model, const std::string& en_model, const std::string& de_model,
                            const std::string& am_cmvn, const std::string& am_config, const std::string& token_file,
                            const std::string& online_token_file, int thread_num, const std::string& online_config_file) {
  (void)thread_num;
  LoadOnline(en_model, de_model, am_cmvn, online_config_file, online_token_file);
  LoadOffline(am_model, am_cmvn, am_config, token_file);
}

void ParaformerHip::InitAsr(const std::string& am_model, const std::string& en_model, const std::string& de_model,
                            const std::string& am_cmvn, const std::string& am_config, const std::string& token_file,
                            const std::string& online_token_file, int thread_num) {
  InitAsr(am_model, en_model, de_model, am_cmvn, am_config, token_file, online_token_file, thread_num, am_config);
}

void ParaformerHip::InitLm(const std::string& lm_file, const std::string& lm_cfg_file, const std::string& lex_file) {
  InitLm(lm_file, lm_cfg_file, lex_file, "");
}

void ParaformerHip::InitLm(const std::string& lm_file, const std::string& lm_cfg_file, const std::string& lex_file,
                           const std::string& lm_units_path) {
#ifdef PFHIP_WITH_FUNASR
  try {                                                          // paraformer.cpp:160-175
    lm_ = std::shared_ptr<fst::Fst<fst::StdArc>>(fst::Fst<fst::StdArc>::Read(lm_file));
    if (lm_) {
      lm_vocab = new Vocab(lm_cfg_file.c_str(), lex_file.c_str());
      if (lm_units_path.size() > 0) phone_set_ = new PhoneSet(lm_units_path.c_str());
    } else {
      std::fprintf(stderr, "Failed to load lm file %s\n", lm_file.c_str());
    }
  } catch (std::exception const& e) {
    std::fprintf(stderr, "Error when load lm file: %s\n", e.what());
    std::exit(0);
  }
  has_lm_ = lm_ != nullptr;
#else
  (void)lm_cfg_file; (void)lex_file; (void)lm_units_path;
  has_lm_ = static_cast<bool>(std::ifstream(lm_file));           // no openfst here: the decoder that is handed in owns the graph
  if (!has_lm_) std::fprintf(stderr, "Failed to load lm file %s\n", lm_file.c_str());
#endif
}

int ParaformerHip::GetAsrSampleRate() { return asr_sample_rate_; }

std::string ParaformerHip::OnlineTokensToString(const std::vector<int>& ids) {
  HipVocab* v = online_vocab ? online_vocab : vocab;
  if (v) return v->Vector2StringV2(ids);                       // OnlineGreedySearch (paraformer.cpp:362-371): no language argument
  std::string s;
  for (size_t i = 0; i < ids.size(); ++i) {
    if (i) s += ' ';
    s += std::to_string(ids[i]);
  }
  return s;
}

std::string ParaformerHip::Forward(float* din, int len, bool input_finished, const std::vector<std::vector<float>>& hw_emb,
                                   void* wfst_decoder) {
  float* buff[1] = {din};
  int lens[1] = {len};
  const std::vector<std::string> r = Forward(buff, lens, input_finished, hw_emb, wfst_decoder, 1);
  return r.empty() ? std::string() : r[0];
}

// Vocab::Vector2StringV2(hyps, language) (paraformer.cpp:396); without a token file (harness runs on synthetic models)
// the ids themselves, space separated
std::string ParaformerHip::IdsToString(const std::vector<int>& ids) {
  if (vocab) return vocab->Vector2StringV2(ids, language);
  std::string s;
  for (size_t i = 0; i < ids.size(); ++i) {
    if (i) s += ' ';
    s += std::to_string(ids[i]);
  }
  return s;
}

std::vector<std::string> ParaformerHip::Forward(float** din, int* len, bool input_finished,
                                                const std::vector<std::vector<float>>& hw_emb, void* wfst_decoder,
                                                int batch_in) {
  std::vector<std::string> results(batch_in > 0 ? batch_in : 0);
  tl_last_ids.assign(results.size(), {});
  tl_last_spans.assign(results.size(), {});
  if (batch_in <= 0 || !handle_) return results;
  // paraformer.cpp:563: the LM decides between GreedySearch and the decoder that came with the call
  Decoder* decoder = has_lm_ ? static_cast<Decoder*>(wfst_decoder) : nullptr;
  int max_len = 0;
  for (int i = 0; i < batch_in; ++i) max_len = len[i] > max_len ? len[i] : max_len;
  const int max_tokens = max_len / 960 + 2;       // at most T+1 CIF fires, T = ceil(frames/6)
  const int V = pfhip_vocab_size(handle_);
  std::vector<int32_t> ids((size_t)batch_in * max_tokens), tn(batch_in), nf(batch_in), usl(batch_in), fr(batch_in);
  std::vector<float> logp;
  pfhip_out out{};
  out.token_ids = ids.data();
  out.token_num = tn.data();
  out.n_fires = nf.data();
  out.n_frames = fr.data();
  out.max_tokens = max_tokens;
  if (decoder) {                                  // the rows Search consumes: [batch][max_tokens][V] log-probabilities
    logp.resize((size_t)batch_in * max_tokens * V);
    out.logp = logp.data();
  }
  const bool with_ts = pfhip_has_timestamp_head(handle_) != 0;      // the reference's outputTensor.size() == 4 (paraformer.cpp:545)
  const int max_us = 3 * max_tokens;
  std::vector<float> usa, usp;
  if (with_ts) {
    usa.resize((size_t)batch_in * max_us);
    usp.resize((size_t)batch_in * max_us);
    out.us_alphas = usa.data(); out.us_peaks = usp.data(); out.us_len = usl.data(); out.max_us = max_us;
  }
  // hw_emb [H][d] as the reference passes it (paraformer.cpp:515-531); plain models ignore it
  std::vector<float> hw;
  int n_hw = 0;
  if (pfhip_is_contextual(handle_)) {
    const size_t d = (size_t)pfhip_d_model(handle_);
    for (const auto& row : hw_emb) {
      if (row.size() != d) continue;
      hw.insert(hw.end(), row.begin(), row.end());
      ++n_hw;
    }
  }
  const pfhip_status st = pfhip_offline_forward(handle_, din, len, batch_in, n_hw ? hw.data() : nullptr, n_hw, &out);
  if (st != PFHIP_OK) {
    std::fprintf(stderr, "ParaformerHip::Forward: %s\n", pfhip_last_error());
    return results;                                // "" per item, as paraformer.cpp:582-588
  }
  for (int i = 0; i < batch_in; ++i) {
    const int n = tn[i] < nf[i] ? tn[i] : nf[i];
    tl_last_ids[i].assign(ids.begin() + (size_t)i * max_tokens, ids.begin() + (size_t)i * max_tokens + n);
    if (decoder && fr[i] > 0) {                   // no feature frame: "" before any decoding (paraformer.cpp:477-480)
      // BeamSearch + FinalizeDecode (paraformer.cpp:410-419, 563-579; per item as paraformer-torch.cpp:431-466): `n` rows of V
      // log-probabilities, of which WfstDecoder::Search feeds the first n - 1 to the lattice decoder (wfst-decoder.cpp:27-58)
      try {
        std::string text = decoder->Search(logp.data() + (size_t)i * max_tokens * V, n, (int64_t)V);
        if (input_finished) {
          if (with_ts)
            text = decoder->FinalizeDecode(true, std::vector<float>(usa.begin() + (size_t)i * max_us, usa.begin() + (size_t)i * max_us + usl[i]),
                                           std::vector<float>(usp.begin() + (size_t)i * max_us, usp.begin() + (size_t)i * max_us + usl[i]));
          else
            text = decoder->FinalizeDecode();
        }
        results[i] = text;
      } catch (std::exception const& e) {
        std::fprintf(stderr, "ParaformerHip::Forward: %s\n", e.what());
      }
      if (batch_in > 1) decoder->StartUtterance();
      continue;
    }
    results[i] = IdsToString(tl_last_ids[i]);
    if (with_ts && n > 0 && usl[i] > 0) {
      // GreedySearch(..., is_stamp = true) -> TimestampOnnx over the characters of the hypothesis (paraformer.cpp:386-395)
      // char_list = all token_num hypotheses (Vocab::Vector2String keeps every id); TimestampOnnx drops a trailing "</s>"
      // (util.cpp:846-849) — id 2 in the FunASR vocabularies when no tokens.json is loaded
      const int last = tl_last_ids[i].back();
      const bool eos = !vocab ? last == 2 : vocab->Id2String(last) == "</s>";
      const int n_chars = n - (eos ? 1 : 0);
      std::vector<float> spans((size_t)3 * (2 * n + 2));
      int n_spans = 0;
      if (n_chars > 0 &&
          pfhip_timestamp_onnx(usa.data() + (size_t)i * max_us, usp.data() + (size_t)i * max_us, usl[i], n_chars, 0.f, -1.5f,
                               spans.data(), (int)spans.size() / 3, &n_spans) == PFHIP_OK)
        tl_last_spans[i].assign(spans.begin(), spans.begin() + (size_t)3 * n_spans);
      if (vocab && n_spans > 0) {
        // time-stamp mode returns "<text> | <stamps>" (paraformer.cpp:398-406: Vector2String -> TimestampOnnx -> PostProcess)
        std::vector<std::string> raw_char;
        vocab->Vector2String(tl_last_ids[i], raw_char);
        std::vector<std::vector<float>> stamps;
        for (int k = 0; k < n_spans; ++k)
          if (spans[3 * k + 2] == 0.f) stamps.push_back({spans[3 * k], spans[3 * k + 1]});       // <sil> spans carry no character (util.cpp:957-961)
        if ((int)stamps.size() >= n_chars) {
          stamps.resize(raw_char.size(), {0.f, 0.f});          // the popped "</s>" is skipped before its stamp is read
          results[i] = pfhip_host::PostProcess(raw_char, stamps);
        }
      }
    }
  }
  return results;
}

void ParaformerHip::InitSegDict(const std::string& seg_dict_model) {
  delete seg_dict_;
  seg_dict_ = new pfhip_host::SegDictHost();
  if (!seg_dict_->Load(seg_dict_model.c_str())) std::fprintf(stderr, "%s open failed !!\n", seg_dict_model.c_str());     // seg_dict.cpp:22-25
}

std::vector<std::vector<float>> ParaformerHip::CompileHotwordEmbedding(std::string& hotwords) {
  const int d = handle_ ? pfhip_d_model(handle_) : 512;
  if (!handle_ || !pfhip_is_contextual(handle_)) return {std::vector<float>(d, 0.f)};      // paraformer.cpp:594-599
  std::vector<int> mat, lens;
  pfhip_host::HotwordIdMatrix(hotwords, seg_dict_, [&](const std::string& unit) { return vocab ? vocab->GetIdByToken(unit) : -1; }, mat,
                              lens);                                                      // :600-651
  const int H = (int)lens.size();
  std::vector<int32_t> mat32(mat.begin(), mat.end()), lens32(lens.begin(), lens.end());
  std::vector<float> emb((size_t)H * d);
  if (pfhip_hotword_embed(handle_, mat32.data(), lens32.data(), H, emb.data()) != PFHIP_OK) {
    std::fprintf(stderr, "ParaformerHip::CompileHotwordEmbedding: %s\n", pfhip_last_error());
    return {std::vector<float>(d, 0.f)};
  }
  std::vector<std::vector<float>> out(H);
  for (int i = 0; i < H; ++i) out[i].assign(emb.begin() + (size_t)i * d, emb.begin() + (size_t)(i + 1) * d);
  return out;
}

}  // namespace funasr

namespace funasr {
const std::vector<std::vector<int>>& ParaformerHip::LastTokenIds() const { return tl_last_ids; }
const std::vector<std::vector<float>>& ParaformerHip::LastTimestamps() const { return tl_last_spans; }
}  // namespace funasr

namespace funasr {

ParaformerOnlineHip::ParaformerOnlineHip(ParaformerHipBase* offline_handle, std::vector<int> chunk_size, std::string model_type) {
  (void)model_type;                                             // MODEL_PARA only (SenseVoice is out of scope)
  offline_handle_ = dynamic_cast<ParaformerHip*>(offline_handle);
  if (!offline_handle_ || !offline_handle_->OnlineHandle() || chunk_size.size() != 3) {
    std::fprintf(stderr, "ParaformerOnlineHip: the shared model holds no online encoder / decoder (InitAsr with en_model, de_model)\n");
    return;
  }
  if (pfhip_stream_create(offline_handle_->OnlineHandle(), chunk_size.data(), &stream_) != PFHIP_OK) {
    std::fprintf(stderr, "ParaformerOnlineHip: %s\n", pfhip_last_error());
    stream_ = nullptr;
  }
  // chunk_len = chunk_size[1] * frame_shift (10 ms) * lfr_n (6) * 16 samples per ms (paraformer-online.cpp:40-43)
  chunk_len = chunk_size[1] * 10 * 6 * (offline_handle_->GetAsrSampleRate() / 1000);
}

ParaformerOnlineHip::~ParaformerOnlineHip() {
  if (stream_) pfhip_stream_destroy(stream_);
}

void ParaformerOnlineHip::Reset() {
  if (stream_) pfhip_stream_reset(stream_);
}

std::string ParaformerOnlineHip::Forward(float* din, int len, bool input_finished, const std::vector<std::vector<float>>& hw_emb,
                                         void* wfst_decoder) {
  (void)hw_emb; (void)wfst_decoder;                              // unused by ParaformerOnline::Forward too
  last_ids_.clear();
  if (!stream_ || len < 0 || (len > 0 && !din)) return "";       // (an empty final frame flushes the look-back cache, :532-540)
  std::vector<int32_t> ids(256);
  int n_ids = 0;
  pfhip_status st = pfhip_stream_forward(stream_, din, len, input_finished ? 1 : 0, ids.data(), (int)ids.size(), &n_ids);
  if (st == PFHIP_ERR_CAPACITY) {                                // *n_tokens = what it needed; the chunk was not consumed
    ids.resize((size_t)n_ids + 16);
    st = pfhip_stream_forward(stream_, din, len, input_finished ? 1 : 0, ids.data(), (int)ids.size(), &n_ids);
  }
  if (st != PFHIP_OK) {
    std::fprintf(stderr, "ParaformerOnlineHip::Forward: %s\n", pfhip_last_error());
    return "";
  }
  last_ids_.assign(ids.begin(), ids.begin() + n_ids);
  std::string result = offline_handle_->OnlineTokensToString(last_ids_);
  if (!result.empty() && pfhip_stream_last_path(stream_) == 2) result.push_back(' ');      // paraformer-online.cpp:585-587
  return result;
}

}  // namespace funasr
