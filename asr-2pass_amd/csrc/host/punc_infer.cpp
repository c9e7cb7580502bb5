// Harness over the punctuation adapters (ct_transformer_hip.h), the text-in / text-out calls the reference makes as
// CTTransformerInfer (funasrruntime.cpp:170-205):
//   punc_infer <punc_dir> offline [language]      each stdin line -> AddPunc(line, language)
//   punc_infer <punc_dir> online                  each stdin line -> AddPunc(line, arr_cache); a line "<reset>" clears the cache
// Output per line: "out <text>" and, online, "cache <w0>|<w1>|...".
#include <cstdio>
#include <iostream>
#include <memory>
#include <string>

#include "ct_transformer_hip.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s punc_dir offline|online [language]\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1], kind = argv[2], lang = argc > 3 ? argv[3] : "zh-cn";
  std::unique_ptr<funasr::CTTransformerHip> m(kind == "online" ? new funasr::CTTransformerOnlineHip() : new funasr::CTTransformerHip());
  m->InitPunc(dir + "/punc.pfhip.bin", dir + "/punc.pfhip.json", dir + "/tokens.json", 1);
  std::vector<std::string> cache;
  std::string line;
  while (std::getline(std::cin, line)) {
    if (kind == "online") {
      if (line == "<reset>") { cache.clear(); continue; }
      const std::string out = m->AddPunc(line.c_str(), cache, lang);
      std::printf("out %s\ncache ", out.c_str());
      for (size_t i = 0; i < cache.size(); ++i) std::printf("%s%s", i ? "|" : "", cache[i].c_str());
      std::printf("\n");
    } else {
      std::printf("out %s\n", m->AddPunc(line.c_str(), lang).c_str());
    }
  }
  return 0;
}
