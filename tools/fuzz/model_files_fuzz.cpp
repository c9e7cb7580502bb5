// Robustness harness for the model-file reader (asr-2pass_amd/csrc/model_files.cpp), built by tests/test_model_files.py with
// g++ -fsanitize=address,undefined (CPU only): every iteration damages a copy of a valid model directory's files — a truncation,
// a run of flipped bytes, a spliced length field — and loads it.  Any outcome but "loaded" or "refused with a message" (a crash,
// an out-of-bounds read, a leak of undefined behaviour) fails the run.
//   model_files_fuzz <asr|vad|punc> <dir> <iterations> [seed]
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../asr-2pass_amd/csrc/model_files.h"

namespace {
std::vector<char> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
void spit(const std::string& p, const std::vector<char>& v) {
  std::ofstream f(p, std::ios::binary | std::ios::trunc);
  f.write(v.data(), (std::streamsize)v.size());
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string kind = argv[1], dir = argv[2];
  const int iters = std::atoi(argv[3]);
  unsigned long long s = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 1;
  auto rnd = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)(s >> 33); };
  setenv("PFHIP_MODEL_CACHE", "0", 1);
  const std::vector<std::string> names = kind == "asr" ? std::vector<std::string>{"model.onnx", "am.mvn", "config.yaml"}
                                         : kind == "vad" ? std::vector<std::string>{"model.onnx", "am.mvn", "config.yaml"}
                                                         : std::vector<std::string>{"model.onnx", "config.yaml"};
  std::vector<std::vector<char>> orig;
  for (const auto& n : names) orig.push_back(slurp(dir + "/" + n));
  const std::string work = dir + "/fuzz";
  ::mkdir(work.c_str(), 0755);
  int loaded = 0, refused = 0;
  for (int it = 0; it < iters; ++it) {
    const size_t which = it % 4 == 3 ? 1 + rnd() % (names.size() - 1) : 0;      // mostly the ONNX file
    for (size_t k = 0; k < names.size(); ++k) {
      std::vector<char> v = orig[k];
      if (k == which && !v.empty()) {
        switch (rnd() % 4) {
          case 0: v.resize(rnd() % v.size()); break;                                                   // truncation
          case 1: { const size_t at = rnd() % v.size(), n = 1 + rnd() % 8; for (size_t i = at; i < at + n && i < v.size(); ++i) v[i] = (char)rnd(); break; }
          case 2: { const size_t at = rnd() % v.size(); v[at] = (char)0xFF; if (at + 1 < v.size()) v[at + 1] = (char)0x7F; break; }      // a huge varint
          default: { const size_t at = rnd() % v.size(); v.insert(v.begin() + (long)at, (size_t)(1 + rnd() % 5), (char)rnd()); break; }   // shifted framing
        }
      }
      spit(work + "/" + names[k], v);
    }
    pfhip_files::Container c;
    try {
      if (kind == "asr") pfhip_files::load_asr(work + "/model.onnx", "", "", work + "/am.mvn", work + "/config.yaml", c);
      else if (kind == "vad") pfhip_files::load_vad(work + "/model.onnx", work + "/am.mvn", work + "/config.yaml", c);
      else pfhip_files::load_punc(work + "/model.onnx", work + "/config.yaml", c);
      ++loaded;
    } catch (const std::exception&) {
      ++refused;
    }
  }
  std::printf("{\"iterations\": %d, \"loaded\": %d, \"refused\": %d}\n", iters, loaded, refused);
  return 0;
}
