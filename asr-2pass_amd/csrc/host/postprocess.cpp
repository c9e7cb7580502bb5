#include "postprocess.h"

namespace pfhip_host {

namespace {
// util.cpp:708-718: a 3-byte UTF-8 sequence in U+4E00..U+9FFF
bool IsChinese(const std::string& ch) {
  if (ch.size() != 3) return false;
  const unsigned char a = (unsigned char)ch[0], b = (unsigned char)ch[1], c = (unsigned char)ch[2];
  if ((a & 0xf0) != 0xe0 || (b & 0xc0) != 0x80 || (c & 0xc0) != 0x80) return false;
  const int u = ((a & 0x0f) << 12) | ((b & 0x3f) << 6) | (c & 0x3f);
  return u >= 19968 && u <= 40959;
}
}  // namespace

std::string PostProcess(const std::vector<std::string>& raw_char, const std::vector<std::vector<float>>& stamps) {
  std::vector<std::vector<float>> merged;
  std::vector<std::string> words;
  bool is_pre_english = false, is_combining = false;
  std::string combine;
  float begin = -1.f;
  const size_t n = raw_char.size();
  for (size_t i = 0; i < n; ++i) {
    std::string word = raw_char[i];
    if (word == "<s>" || word == "</s>" || word == "<unk>") continue;                 // step 1 (:733-735)
    const bool sub_word = word.find("@@") != std::string::npos;                        // step 2 (:737-767)
    if (sub_word) {
      if (i == n - 1 || IsChinese(raw_char[i + 1])) {                                  // "lo@@" followed by a CJK character
        word = word.erase(word.length() - 2) + " ";
        if (is_combining) {
          combine += word;
          is_combining = false;
          word = combine;
          combine.clear();
        }
      } else {
        combine += word.erase(word.length() - 2);
        if (!is_combining) begin = stamps[i][0];
        is_combining = true;
        continue;
      }
    } else if (is_combining) {
      combine += word;
      is_combining = false;
      word = combine;
      combine.clear();
    }
    if (IsChinese(word)) {                                                             // step 3 (:770-815)
      words.push_back(word);
      merged.push_back(stamps[i]);
      is_pre_english = false;
    } else {
      if (is_pre_english) words.push_back(" ");        // both branches on pre_english_len push the same things (:789-811)
      words.push_back(word);
      begin = begin == -1.f ? stamps[i][0] : begin;
      merged.push_back({begin, stamps[i][1]});
      begin = -1.f;
      is_pre_english = true;
    }
  }
  std::string stamp_str;
  for (size_t i = 0; i < merged.size(); ++i) {
    stamp_str += std::to_string(merged[i][0]);
    stamp_str += ", ";
    stamp_str += std::to_string(merged[i][1]);
    if (i != merged.size() - 1) stamp_str += ",";
  }
  std::string text;
  for (const std::string& w : words) text += w;
  return text + " | " + stamp_str;
}

}  // namespace pfhip_host
