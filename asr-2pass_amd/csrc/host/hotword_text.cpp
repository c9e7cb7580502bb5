#include "hotword_text.h"

#include <fstream>

namespace pfhip_host {

namespace {
// UTF-8 -> code points (malformed bytes pass through as themselves: they are never in the CJK range)
std::vector<unsigned> CodePoints(const std::string& s, std::vector<std::string>* pieces = nullptr) {
  std::vector<unsigned> out;
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    size_t n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1;
    if (i + n > s.size()) n = 1;
    unsigned cp = c;
    if (n == 2) cp = ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu);
    else if (n == 3) cp = ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu);
    else if (n == 4) cp = ((c & 0x07u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) | (((unsigned char)s[i + 2] & 0x3Fu) << 6) |
                          ((unsigned char)s[i + 3] & 0x3Fu);
    out.push_back(cp);
    if (pieces) pieces->push_back(s.substr(i, n));
    i += n;
  }
  return out;
}
bool IsCjk(unsigned cp) { return cp >= 0x4E00 && cp <= 0x9FFF; }
}  // namespace

bool SegDictHost::Load(const char* filename) {
  std::ifstream in(filename);
  if (!in) return false;
  std::string line;
  while (std::getline(in, line)) {
    const size_t tab = line.find('\t');
    if (tab == std::string::npos) continue;               // fewer than two tab-separated items (seg_dict.cpp:29)
    const std::string word = line.substr(0, tab);
    std::string segs = line.substr(tab + 1);
    const size_t tab2 = segs.find('\t');
    if (tab2 != std::string::npos) segs.resize(tab2);     // only the second item is read
    // split(segs, ' ') of util.cpp:639-647 (std::getline): empty pieces between two spaces are KEPT (such a word can never
    // be embedded: its empty unit has no id), a trailing space adds nothing
    std::vector<std::string> pieces;
    size_t pos = 0;
    while (pos < segs.size()) {
      const size_t sp = segs.find(' ', pos);
      pieces.push_back(segs.substr(pos, sp == std::string::npos ? std::string::npos : sp - pos));
      if (sp == std::string::npos) break;
      pos = sp + 1;
    }
    dict_[word] = pieces;
  }
  return true;
}

std::vector<std::string> SegDictHost::GetTokensByWord(const std::string& word) const {
  const auto it = dict_.find(word);
  return it == dict_.end() ? std::vector<std::string>() : it->second;
}

bool IsAllChineseCharacter(const std::string& s) {
  if (s.empty()) return false;
  for (unsigned cp : CodePoints(s))
    if (!IsCjk(cp)) return false;
  return true;
}

void KeepChineseCharacterAndSplit(const std::string& s, std::vector<std::string>& out) {
  out.clear();
  std::vector<std::string> pieces;
  const std::vector<unsigned> cps = CodePoints(s, &pieces);
  for (size_t i = 0; i < cps.size(); ++i)
    if (IsCjk(cps[i])) out.push_back(pieces[i]);
}

}  // namespace pfhip_host
