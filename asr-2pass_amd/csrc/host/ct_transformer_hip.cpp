#include "ct_transformer_hip.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <sstream>

#include "json_strings.h"

namespace funasr {
namespace {

// com-define.h:124-136
constexpr int kTokenLen = 20, kCachePopTriggerLimit = 200;
constexpr int kNotPuncIndex = 1, kCommaIndex = 2, kPeriodIndex = 3, kQuestionIndex = 4, kDunIndex = 5;
const char* const kUnkChar = "<unk>";
const char* const kNotPunc = "_";

bool ReadText(const std::string& path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

bool IsAscii(char c) { return !(c & 0x80); }

// SplitChineseString (tokenizer.cpp:230-246): one UTF-8 sequence per entry, its length = the number of leading one bits
void SplitUtf8(const std::string& s, std::vector<std::string>& out) {
  const int n = (int)s.size();
  for (int i = 0; i < n;) {
    int len = 1;
    for (int j = 0; j < 6 && (s[i] & (0x80 >> j)); ++j) len = j + 1;
    out.push_back(s.substr(i, len));
    i += len;
  }
}

}  // namespace

bool PuncTokenizerHip::Open(const std::string& manifest_json, const std::string& token_file) {
  std::string tok;
  if (!ReadText(token_file, tok) || !pfhip_host::ReadJsonStringArray(tok, 0, id2token_) || id2token_.empty()) return false;
  for (size_t i = 0; i < id2token_.size(); ++i) token2id_[id2token_[i]] = (int)i;
  const size_t at = manifest_json.find("\"punc_list\"");
  if (at == std::string::npos || !pfhip_host::ReadJsonStringArray(manifest_json, at, id2punc_) || id2punc_.size() < 5)
    id2punc_ = {"<unk>", "_", "\xEF\xBC\x8C", "\xE3\x80\x82", "\xEF\xBC\x9F", "\xE3\x80\x81"};   // ， 。 ？ 、
  for (size_t i = 0; i < id2punc_.size(); ++i) punc2id_[id2punc_[i]] = (int)i;
  return true;
}

// Tokenize (tokenizer.cpp:275-333): pieces between blanks; within a piece a run of ASCII bytes is one word, every other
// UTF-8 sequence its own word.  Ids: String2Ids (:188-200) looks the lower-cased word up, <unk> otherwise.
void PuncTokenizerHip::Tokenize(const char* str_info, std::vector<std::string>& str_out, std::vector<int>& id_out) const {
  const std::string all = str_info ? str_info : "";
  std::vector<std::string> pieces;
  if (!all.empty()) {                                       // StrSplit (:255-272)
    size_t from = 0;
    const std::string strs = all + ' ';
    for (size_t pos = strs.find(' '); pos != std::string::npos; pos = strs.find(' ', from)) {
      pieces.push_back(strs.substr(from, pos - from));
      from = pos + 1;
    }
  }
  for (const std::string& item : pieces) {
    std::string eng, chn;
    for (const char ch : item) {
      if (IsAscii(ch)) {
        if (!chn.empty()) { SplitUtf8(chn, str_out); chn.clear(); }
        eng += ch;
      } else {
        if (!eng.empty()) { str_out.push_back(eng); eng.clear(); }
        chn += ch;
      }
    }
    if (!chn.empty()) SplitUtf8(chn, str_out);
    if (!eng.empty()) str_out.push_back(eng);
  }
  id_out.clear();
  const auto unk = token2id_.find(kUnkChar);
  const int unk_id = unk == token2id_.end() ? 0 : unk->second;
  for (std::string item : str_out) {
    std::transform(item.begin(), item.end(), item.begin(), [](unsigned char c) { return c < 0x80 ? (char)std::tolower(c) : (char)c; });
    const auto it = token2id_.find(item);
    id_out.push_back(it == token2id_.end() ? unk_id : it->second);
  }
}

CTTransformerHip::~CTTransformerHip() {
  if (handle_) pfhip_punc_destroy(handle_);
}

void CTTransformerHip::InitPunc(const std::string& punc_model, const std::string& punc_config, const std::string& token_file,
                                int thread_num) {
  (void)thread_num;
  // the reference's own strings (offline-stream.cpp:111-117, tpass-stream.cpp:104-109): <punc-dir>/model.onnx | model_quant.onnx,
  // config.yaml (model_conf.punc_list, tokenizer.cpp:147-158), tokens.json — or a container pair x.pfhip.bin / .json
  pfhip_container* c = nullptr;
  std::string man;
  if (pfhip_read_model_files("punc", punc_model.c_str(), nullptr, nullptr, nullptr, punc_config.c_str(), &c) == PFHIP_OK) {
    size_t bytes = 0;
    const float* blob = pfhip_container_blob(c, &bytes);
    man = pfhip_container_manifest(c);
    if (pfhip_punc_create_from_memory(blob, bytes, man.c_str(), device_, &handle_) != PFHIP_OK) handle_ = nullptr;
    pfhip_container_free(c);
  }
  if (!handle_) {
    // the reference exits on a model-load failure (ct-transformer.cpp:19-26)
    std::fprintf(stderr, "Error when load punc hip model: %s\n", pfhip_last_error());
    std::exit(-1);
  }
  if (!tokenizer_.Open(man, token_file)) {
    std::fprintf(stderr, "Error loading token file, token file error or not exist.\n");
    std::exit(-1);
  }
}

std::vector<int> CTTransformerHip::Infer(const std::vector<int32_t>& ids, int cache_size) const {
  std::vector<int32_t> punc(ids.size());
  if (ids.empty()) return {};
  const pfhip_status st = cache_size < 0 ? pfhip_punc_infer(handle_, ids.data(), (int)ids.size(), punc.data(), nullptr)
                                         : pfhip_punc_infer_online(handle_, ids.data(), (int)ids.size(), cache_size, punc.data(), nullptr);
  if (st != PFHIP_OK) {                                     // the reference logs and returns no punctuation (:198-201)
    std::fprintf(stderr, "punc inference failed: %s\n", pfhip_last_error());
    return std::vector<int>(ids.size(), kNotPuncIndex);
  }
  return std::vector<int>(punc.begin(), punc.end());
}

namespace {
// The not-the-last-mini-sentence branch both AddPunc variants share (ct-transformer.cpp:67-93): cut after the last "。"/"？"
// (searched from the end, positions size-2 .. 1), or — once more than CACHE_POP_TRIGGER_LIMIT words are carried — at the
// last comma, which becomes a period; everything after the cut is carried into the next Infer.
void CutAtSentenceEnd(const PuncTokenizerHip& tk, std::vector<int>& punc, std::vector<std::string>& words, std::vector<int32_t>& ids,
                      std::vector<std::string>& remain_words, std::vector<int32_t>& remain_ids) {
  int sent_end = -1, last_comma = -1;
  for (int k = (int)punc.size() - 2; k > 0; --k) {
    if (tk.Id2Punc(punc[k]) == tk.Id2Punc(kPeriodIndex) || tk.Id2Punc(punc[k]) == tk.Id2Punc(kQuestionIndex)) { sent_end = k; break; }
    if (last_comma < 0 && tk.Id2Punc(punc[k]) == tk.Id2Punc(kCommaIndex)) last_comma = k;
  }
  if (sent_end < 0 && (int)words.size() > kCachePopTriggerLimit && last_comma > 0) {
    sent_end = last_comma;
    punc[sent_end] = kPeriodIndex;
  }
  remain_words.assign(words.begin() + (sent_end + 1), words.end());
  remain_ids.assign(ids.begin() + (sent_end + 1), ids.end());
  words.resize(sent_end + 1);
  punc.resize(sent_end + 1);
}
}  // namespace

std::string CTTransformerHip::AddPunc(const char* sz_input, std::string language) {
  std::vector<std::string> words;
  std::vector<int> ids;
  tokenizer_.Tokenize(sz_input, words, ids);
  const int n = (int)ids.size();
  const int total = (n + kTokenLen - 1) / kTokenLen;
  std::vector<std::string> remain_words, new_string, sentence_out;
  std::vector<int32_t> remain_ids;
  for (int i = 0; i < n; i += kTokenLen) {
    const int take = std::min(kTokenLen, n - i);
    std::vector<int32_t> in_ids(remain_ids);
    in_ids.insert(in_ids.end(), ids.begin() + i, ids.begin() + i + take);
    std::vector<std::string> in_words(remain_words);
    in_words.insert(in_words.end(), words.begin() + i, words.begin() + i + take);
    std::vector<int> punc = Infer(in_ids, -1);
    const int cur = i / kTokenLen;
    if (cur < total - 1) CutAtSentenceEnd(tokenizer_, punc, in_words, in_ids, remain_words, remain_ids);
    for (size_t k = 0; k < in_words.size(); ++k) {
      // (:98-103) a blank between two ASCII words: looks at the previous word as already emitted, never at a mini-sentence start
      if (k > 0 && IsAscii(in_words[k - 1][0]) && IsAscii(in_words[k][0])) in_words[k] = " " + in_words[k];
      new_string.push_back(in_words[k]);
      if (punc[k] != kNotPuncIndex) new_string.push_back(tokenizer_.Id2Punc(punc[k]));
    }
    sentence_out = new_string;
    if (cur == total - 1) {                                 // (:113-128) the text always ends in "。" or "？"
      const std::string& last = new_string.back();
      if (last == tokenizer_.Id2Punc(kCommaIndex) || last == tokenizer_.Id2Punc(kDunIndex)) {
        sentence_out.back() = tokenizer_.Id2Punc(kPeriodIndex);
      } else if (last != tokenizer_.Id2Punc(kPeriodIndex) && last != tokenizer_.Id2Punc(kQuestionIndex)) {
        sentence_out.push_back(tokenizer_.Id2Punc(kPeriodIndex));
      }
    }
  }
  std::string result;
  for (const std::string& s : sentence_out) result += s;
  if (language == "en-bpe") {                               // (:134-149)
    const char* zh[4] = {"\xEF\xBC\x8C", "\xE3\x80\x82", "\xE3\x80\x81", "\xEF\xBC\x9F"};
    const char en[4] = {',', '.', ',', '?'};
    for (int i = 0; i < 4; ++i)
      for (size_t pos = 0; (pos = result.find(zh[i], pos)) != std::string::npos; ++pos) result.replace(pos, 3, 1, en[i]);
  }
  return result;
}

std::string CTTransformerHip::AddPunc(const char* sz_input, std::vector<std::string>& arr_cache, std::string language) {
  (void)arr_cache;
  return AddPunc(sz_input, language);                       // ct-transformer.cpp:157-159
}

std::string CTTransformerOnlineHip::AddPunc(const char* sz_input, std::vector<std::string>& arr_cache, std::string language) {
  (void)language;
  std::string text;
  for (const std::string& s : arr_cache) text += s;
  const char* in = sz_input ? sz_input : "";
  // (:48-50) a blank where cached text ending in an ASCII byte meets input starting with one
  if (!text.empty() && IsAscii(text.back()) && in[0] != '\0' && IsAscii(in[0])) text += " ";
  text += in;
  std::vector<std::string> words;
  std::vector<int> ids;
  tokenizer_.Tokenize(text.c_str(), words, ids);
  const int n = (int)ids.size();
  const int total = (n + kTokenLen - 1) / kTokenLen;
  const size_t n_cache = arr_cache.size();
  std::vector<std::string> remain_words, words_all;
  std::vector<int32_t> remain_ids;
  std::vector<int> punc_all;
  for (int i = 0; i < n; i += kTokenLen) {
    const int take = std::min(kTokenLen, n - i);
    std::vector<int32_t> in_ids(remain_ids);
    in_ids.insert(in_ids.end(), ids.begin() + i, ids.begin() + i + take);
    std::vector<std::string> in_words(remain_words);
    in_words.insert(in_words.end(), words.begin() + i, words.begin() + i + take);
    std::vector<int> punc = Infer(in_ids, (int)n_cache);
    if (i / kTokenLen < total - 1) CutAtSentenceEnd(tokenizer_, punc, in_words, in_ids, remain_words, remain_ids);
    punc_all.insert(punc_all.end(), punc.begin(), punc.end());
    words_all.insert(words_all.end(), in_words.begin(), in_words.end());
  }
  std::vector<std::string> out;
  size_t skip = 0;
  for (size_t i = 0; i < words_all.size(); ++i) {           // (:112-132)
    if (IsAscii(words_all[i][0]) && i + 1 < words_all.size() && IsAscii(words_all[i + 1][0])) words_all[i] += " ";
    if (skip < n_cache) ++skip;
    else out.push_back(words_all[i]);
    if (skip >= n_cache) {                                  // true for the last cached word too: its mark is emitted again
      const std::string& p = tokenizer_.Id2Punc(punc_all[i]);
      if (p != kNotPunc) out.push_back(p);
    }
  }
  int sent_end = -1;
  for (int i = (int)punc_all.size() - 2; i > 0; --i)
    if (punc_all[i] == kPeriodIndex || punc_all[i] == kQuestionIndex) { sent_end = i; break; }
  arr_cache.assign(words_all.begin() + (sent_end + 1), words_all.end());
  if (!out.empty() && tokenizer_.IsPunc(out.back())) out.pop_back();      // (:145-149) a trailing mark is held back
  std::string result;
  for (const std::string& s : out) result += s;
  return result;
}

PuncModelHipBase* CreatePuncModelHip(const std::string& punc_dir, int thread_num, bool allow_online, bool quantized) {
  // tpass-stream.cpp:100-135 / offline-stream.cpp:105-129 with the file names of com-define.h:52-88
  const std::string model = punc_dir + (quantized ? "/model_quant.onnx" : "/model.onnx"), config = punc_dir + "/config.yaml",
                    tok = punc_dir + "/tokens.json";
  const bool container = (bool)std::ifstream(punc_dir + "/punc.pfhip.bin") || (bool)std::ifstream(punc_dir + "/model.pfhip.bin");
  if ((!std::ifstream(model) || !std::ifstream(config)) && !container) {
    std::fprintf(stderr, "PUNC model file is not exist, skip load punc model.\n");
    return nullptr;
  }
  if (!std::ifstream(tok)) {
    std::fprintf(stderr, "PUNC model file is not exist, skip load punc model.\n");
    return nullptr;
  }
  // the reference tests the MODEL PATH for the word (tpass-stream.cpp:124)
  CTTransformerHip* m = allow_online && model.find("realtime") != std::string::npos ? new CTTransformerOnlineHip() : new CTTransformerHip();
  m->InitPunc(model, config, tok, thread_num);
  // One AddPunc per handler thread, each a few Infer calls: merged into packed device passes.  `thread_num` is the server's
  // --model-thread-num (onnxruntime intra-op threads, default 1: funasr-wss-server.cpp:105-106), NOT the number of handler
  // threads, so merging does not depend on it: a caller that finds the model idle runs at once, company queues behind it.
  {
    const char* e = std::getenv("PFHIP_PUNC_WAIT_US");
    const int w = e ? std::atoi(e) : 300;
    if (w > 0) pfhip_set_punc_batching(m->Handle(), w, 128);
  }
  return m;
}

}  // namespace funasr
