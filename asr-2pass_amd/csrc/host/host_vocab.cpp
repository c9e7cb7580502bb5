#include "host_vocab.h"

#include <fstream>
#include <sstream>

#include "json_strings.h"

namespace pfhip_host {

namespace {
const char kPieceMark[] = "@@";
const char kBpeSpace[] = "\xE2\x96\x81";          // U+2581, the sentencepiece word-start mark (vocab.cpp:174)

bool IsSpecial(const std::string& w) { return w == "<s>" || w == "</s>" || w == "<unk>"; }      // vocab.cpp:181

std::string WordFormat(const std::string& w) {                                                   // vocab.cpp:149-162
  if (w == "i") return "I";
  if (w == "i'm") return "I'm";
  if (w == "i've") return "I've";
  if (w == "i'll") return "I'll";
  return w;
}
}  // namespace

bool HostVocab::Load(const char* tokens_json) {
  std::ifstream f(tokens_json);
  if (!f) return false;
  std::stringstream ss;
  ss << f.rdbuf();
  vocab_.clear();
  return ReadJsonStringArray(ss.str(), 0, vocab_);
}

int HostVocab::GetIdByToken(const std::string& token) const {
  for (size_t i = 0; i < vocab_.size(); ++i)
    if (vocab_[i] == token) return (int)i;
  return -1;
}

bool HostVocab::IsChinese(const std::string& ch) {
  if (ch.size() != 3) return false;
  const unsigned char a = (unsigned char)ch[0], b = (unsigned char)ch[1], c = (unsigned char)ch[2];
  if ((a & 0xF0) != 0xE0 || (b & 0xC0) != 0x80 || (c & 0xC0) != 0x80) return false;             // Str2Int -> 0 (vocab.cpp:116-125)
  const unsigned cp = ((a & 0x0Fu) << 12) | ((b & 0x3Fu) << 6) | (c & 0x3Fu);
  return cp >= 0x4E00 && cp <= 0x9FFF;                                                          // 19968..40959 (:142)
}

void HostVocab::Vector2String(const std::vector<int>& in, std::vector<std::string>& preds) const {
  for (int id : in) preds.push_back(Id2String(id));
}

std::string HostVocab::Vector2String(const std::vector<int>& in) const {
  std::string s;
  for (int id : in) s += Id2String(id);
  return s;
}

std::string HostVocab::Vector2StringV2(const std::vector<int>& in, const std::string& language) {
  const size_t n = in.size();
  std::string out;
  // ids outside the table (undefined behaviour in the reference, which indexes the vector unchecked) are skipped like <unk>
  auto token = [&](size_t i) -> std::string { return in[i] < 0 || in[i] >= Size() ? std::string("<unk>") : vocab_[(size_t)in[i]]; };

  if (language == "en-bpe") {                     // sentencepiece pieces: a piece holding U+2581 opens a new word (:183-198, :291-297)
    std::string word;
    auto flush = [&] {
      if (word.empty()) return;
      if (!out.empty()) out += ' ';
      out += WordFormat(word);
    };
    for (size_t i = 0; i < n; ++i) {
      const std::string w = token(i);
      if (IsSpecial(w)) continue;
      if (w.find(kBpeSpace) != std::string::npos) { flush(); word = w.substr(3); }
      else word += w;
    }
    flush();
    return out;
  }

  const bool space_after_chinese = last_is_complete_english_.load(std::memory_order_relaxed);       // what the previous call left behind (:176)
  bool prev_latin = false;
  size_t prev_latin_len = 0;
  // a finished word: Chinese characters are appended bare; a Latin word gets a space in front when it follows a Latin word
  // and either of the two is longer than one letter (spelled-out letters stay glued: "a" "i" -> "ai") (:243-281)
  auto put = [&](const std::string& w) {
    if (IsChinese(w)) { out += w; prev_latin = false; return; }
    if (!prev_latin) { if (space_after_chinese) out += ' '; }
    else if (prev_latin_len > 1 || w.size() > 1) out += ' ';
    out += w;
    prev_latin_len = w.size();
    prev_latin = true;
  };

  std::string glued;
  bool gluing = false, wrote = false, ends_complete_english = false;
  for (size_t i = 0; i < n; ++i) {
    std::string w = token(i);
    if (IsSpecial(w)) continue;
    const bool piece = w.find(kPieceMark) != std::string::npos;
    if (piece) {
      const bool last = i + 1 == n;
      const bool before_chinese = !last && IsChinese(token(i + 1));
      w.erase(w.size() - 2);
      if (!before_chinese && !last) { glued += w; gluing = true; continue; }      // word start / middle (:225-229)
      if (before_chinese) w += ' ';                                              // "lo@@" + Chinese: close the word (:205-214)
    }
    if (gluing) { w = glued + w; glued.clear(); gluing = false; }
    put(w);
    ends_complete_english = i + 1 == n && !IsChinese(w) && !piece;           // (:283-288)
    wrote = true;
  }
  // The reference writes the member after every word (false until the last one); sequentially only the last write is visible.
  // Stored ONCE here, so that a concurrent call of another decoder thread never reads the mid-call "false" of this one.
  if (wrote) last_is_complete_english_.store(ends_complete_english, std::memory_order_relaxed);
  return out;
}

}  // namespace pfhip_host
