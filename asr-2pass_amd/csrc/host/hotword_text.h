// Hotword strings -> the id matrix the hotword embedder takes: the host half of Paraformer::CompileHotwordEmbedding
// (onnxruntime/src/paraformer.cpp:600-651) with its helpers SegDict (seg_dict.cpp:19-50: "word<TAB>piece piece ..." lines),
// EncodeConverter::IsAllChineseCharactor (encode_converter.cpp:443-458) and KeepChineseCharacterAndSplit (util.cpp:192-209).
//   * the string is split on single spaces; empty pieces are dropped;
//   * a hotword whose code points all lie in U+4E00..U+9FFF becomes one unit per character;
//   * any other hotword is looked up in the segmentation dictionary (its BPE pieces); a word the dictionary does not hold has no
//     units and is skipped, as is every hotword when no dictionary was loaded;
//   * at most 10 units are kept; a unit without an id in the token list drops the whole hotword ("OOV");
//   * the row [1, 0, ..., 0] with length 1 is appended.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace pfhip_host {

class SegDictHost {
 public:
  bool Load(const char* filename);                                   // false if the file cannot be opened
  std::vector<std::string> GetTokensByWord(const std::string& word) const;
  bool empty() const { return dict_.empty(); }

 private:
  std::map<std::string, std::vector<std::string>> dict_;
};

bool IsAllChineseCharacter(const std::string& s);
void KeepChineseCharacterAndSplit(const std::string& s, std::vector<std::string>& out);

// id_of(unit) returns the token id or -1.  matrix: [n_rows][10] row-major, lengths: [n_rows]; always ends with the blank row.
template <typename IdOf>
void HotwordIdMatrix(const std::string& hotwords, const SegDictHost* seg_dict, IdOf id_of, std::vector<int>& matrix,
                     std::vector<int>& lengths) {
  constexpr int kMaxLen = 10;
  matrix.clear();
  lengths.clear();
  size_t pos = 0;
  while (pos <= hotwords.size() && !hotwords.empty()) {
    const size_t sp = hotwords.find(' ', pos);
    const std::string word = hotwords.substr(pos, sp == std::string::npos ? std::string::npos : sp - pos);
    pos = sp == std::string::npos ? hotwords.size() + 1 : sp + 1;
    if (word.empty()) continue;
    std::vector<std::string> units;
    if (IsAllChineseCharacter(word)) KeepChineseCharacterAndSplit(word, units);
    else if (seg_dict) units = seg_dict->GetTokensByWord(word);
    if (units.empty()) continue;
    std::vector<int> row(kMaxLen, 0);
    const int n = (int)units.size() < kMaxLen ? (int)units.size() : kMaxLen;
    bool oov = false;
    for (int i = 0; i < n && !oov; ++i) {
      row[(size_t)i] = id_of(units[(size_t)i]);
      oov = row[(size_t)i] == -1;
    }
    if (oov) continue;
    lengths.push_back(n);
    matrix.insert(matrix.end(), row.begin(), row.end());
  }
  std::vector<int> blank(kMaxLen, 0);
  blank[0] = 1;
  matrix.insert(matrix.end(), blank.begin(), blank.end());
  lengths.push_back(1);
}

}  // namespace pfhip_host
