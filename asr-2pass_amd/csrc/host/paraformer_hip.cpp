#include "paraformer_hip.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace funasr {

namespace {
// tokens.json: a flat JSON array of strings (the format of the reference's token file).
std::vector<std::string> LoadTokens(const std::string& path) {
  std::vector<std::string> out;
  std::ifstream f(path);
  if (!f) return out;
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string s = ss.str();
  size_t i = 0;
  while ((i = s.find('"', i)) != std::string::npos) {
    std::string tok;
    for (++i; i < s.size() && s[i] != '"'; ++i) {
      if (s[i] == '\\' && i + 1 < s.size()) ++i;
      tok += s[i];
    }
    ++i;
    out.push_back(tok);
  }
  return out;
}
}  // namespace

ParaformerHip::ParaformerHip() {}

ParaformerHip::~ParaformerHip() {
  if (handle_) pfhip_destroy(handle_);
}

void ParaformerHip::InitAsr(const std::string& am_model, const std::string& am_cmvn, const std::string& am_config,
                            const std::string& token_file, int thread_num) {
  (void)am_cmvn;
  (void)thread_num;
  if (pfhip_create(am_model.c_str(), am_config.c_str(), device_, &handle_) != PFHIP_OK) {
    // the reference exits on a model-load failure (paraformer.cpp:43-46)
    std::fprintf(stderr, "Error when load am hip model: %s\n", pfhip_last_error());
    std::exit(-1);
  }
  if (!token_file.empty()) tokens_ = LoadTokens(token_file);
}

int ParaformerHip::GetAsrSampleRate() { return handle_ ? pfhip_sample_rate(handle_) : 16000; }

std::string ParaformerHip::IdsToString(const std::vector<int>& ids) const {
  std::string s;
  for (size_t i = 0; i < ids.size(); ++i) {
    if (!tokens_.empty() && ids[i] >= 0 && (size_t)ids[i] < tokens_.size()) {
      s += tokens_[ids[i]];
    } else {
      if (i) s += ' ';
      s += std::to_string(ids[i]);
    }
  }
  return s;
}

std::vector<std::string> ParaformerHip::Forward(float** din, int* len, bool input_finished,
                                                const std::vector<std::vector<float>>& hw_emb, void* wfst_decoder,
                                                int batch_in) {
  (void)input_finished;
  (void)hw_emb;
  (void)wfst_decoder;
  std::vector<std::string> results(batch_in > 0 ? batch_in : 0);
  last_ids_.assign(results.size(), {});
  if (batch_in <= 0 || !handle_) return results;
  int max_len = 0;
  for (int i = 0; i < batch_in; ++i) max_len = len[i] > max_len ? len[i] : max_len;
  const int max_tokens = max_len / 960 + 2;       // at most T+1 CIF fires, T = ceil(frames/6)
  std::vector<int32_t> ids((size_t)batch_in * max_tokens), tn(batch_in), nf(batch_in);
  pfhip_out out{};
  out.token_ids = ids.data();
  out.token_num = tn.data();
  out.n_fires = nf.data();
  out.max_tokens = max_tokens;
  const pfhip_status st = pfhip_offline_forward(handle_, din, len, batch_in, nullptr, 0, &out);
  if (st != PFHIP_OK) {
    std::fprintf(stderr, "ParaformerHip::Forward: %s\n", pfhip_last_error());
    return results;                                // "" per item, as paraformer.cpp:582-588
  }
  for (int i = 0; i < batch_in; ++i) {
    const int n = tn[i] < nf[i] ? tn[i] : nf[i];
    last_ids_[i].assign(ids.begin() + (size_t)i * max_tokens, ids.begin() + (size_t)i * max_tokens + n);
    results[i] = IdsToString(last_ids_[i]);
  }
  return results;
}

std::vector<std::vector<float>> ParaformerHip::CompileHotwordEmbedding(std::string& hotwords) {
  (void)hotwords;
  const int d = handle_ ? pfhip_d_model(handle_) : 512;
  return {std::vector<float>(d, 0.f)};
}

}  // namespace funasr
