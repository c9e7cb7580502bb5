// Token ids -> text for the stand-alone build of the adapters: the methods of `funasr::Vocab` that sit between
// GreedySearch / the WFST decoder and the result string (onnxruntime/src/vocab.cpp:47-64 LoadVocabFromJson, :98-114
// Vector2String, :127-147 Id2String / IsChinese, :164-305 Vector2StringV2).  Inside the reference tree
// (-DPFHIP_WITH_FUNASR) the adapters use the real `funasr::Vocab` and this file is not compiled in.
//
// Vector2StringV2 is restated as a small emitter: sub-word pieces ("xx@@") are glued until a piece without the marker
// closes the word; a finished word goes through `put`, which owns the spacing rules between Chinese characters and
// Latin words.  Quirks of the reference are kept on purpose because the output must be byte-identical:
//   * the object remembers whether the previous call ended on a complete Latin word (`last_is_complete_english_`) and,
//     if so, puts a space before EVERY Latin word that follows a Chinese character in the next call — the flag it was
//     copied into is never cleared (vocab.cpp:176, :250-252);
//   * "<s>", "</s>", "<unk>" are skipped without touching that memory (:181-182);
//   * a "xx@@" piece in front of a Chinese character is closed with a trailing space (:205-214);
//   * the last two bytes of a piece that CONTAINS "@@" are dropped wherever the marker sits (:206, :216, :226).
#pragma once
#include <atomic>
#include <string>
#include <vector>

namespace pfhip_host {

class HostVocab {
 public:
  HostVocab() {}
  explicit HostVocab(const char* tokens_json) { Load(tokens_json); }
  bool Load(const char* tokens_json);                  // a flat JSON array of strings
  void Assign(std::vector<std::string> tokens) { vocab_ = std::move(tokens); }
  int Size() const { return (int)vocab_.size(); }
  bool empty() const { return vocab_.empty(); }
  std::string Id2String(int id) const { return id < 0 || id >= Size() ? std::string() : vocab_[(size_t)id]; }
  int GetIdByToken(const std::string& token) const;
  static bool IsChinese(const std::string& ch);        // one 3-byte UTF-8 character in U+4E00..U+9FFF
  void Vector2String(const std::vector<int>& in, std::vector<std::string>& preds) const;
  std::string Vector2String(const std::vector<int>& in) const;
  std::string Vector2StringV2(const std::vector<int>& in, const std::string& language = "");

 private:
  std::vector<std::string> vocab_;
  // carried from call to call as in the reference (vocab.cpp:176); atomic because the decoder threads share the vocabulary
  std::atomic<bool> last_is_complete_english_{false};
};

}  // namespace pfhip_host
