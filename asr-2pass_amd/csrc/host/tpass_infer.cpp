// Harness over the 2-pass handle API (funasrruntime_hip.h), one connection fed like the websocket server feeds it
// (websocket/bin/websocket-server-2pass.cpp:135-148: 9600-sample pieces, the last one with input_finished):
//   tpass_infer <offline_model_dir> <online_model_dir> <vad_dir> <pcm_s16_file> [step_samples=9600] [mode=2] [punc_dir]
// One line per call: "call <j> | online <text> | tpass <text> | stamp <stamp>".
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "funasrruntime_hip.h"

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s offline_dir online_dir vad_dir pcm_s16_file [step] [mode]\n", argv[0]);
    return 2;
  }
  std::map<std::string, std::string> paths;
  paths[MODEL_DIR] = argv[1]; paths[ONLINE_MODEL_DIR] = argv[2]; paths[VAD_DIR] = argv[3];
  if (argc > 7) paths[PUNC_DIR] = argv[7];
  const int step = argc > 5 ? std::atoi(argv[5]) : 9600;
  const ASR_TYPE mode = argc > 6 ? (ASR_TYPE)std::atoi(argv[6]) : ASR_TWO_PASS;
  std::ifstream f(argv[4], std::ios::binary);
  std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  FUNASR_HANDLE h = FunTpassInit(paths, 1);
  FUNASR_HANDLE oh = FunTpassOnlineInit(h, {5, 10, 5});
  if (!h || !oh) return 1;
  std::vector<std::vector<std::string>> punc_cache(2);
  const int n_bytes = (int)buf.size(), step_bytes = step * 2;
  int j = 0;
  for (int off = 0; off < n_bytes; off += step_bytes, ++j) {
    const int nb = std::min(step_bytes, n_bytes - off);
    const bool last = off + step_bytes >= n_bytes;
    FUNASR_RESULT r = FunTpassInferBuffer(h, oh, buf.data() + off, nb, punc_cache, last, 16000, "pcm", mode);
    if (!r) { std::fprintf(stderr, "inference failed\n"); return 1; }
    std::printf("call %d | online %s | tpass %s | stamp %s\n", j, FunASRGetResult(r, 0), FunASRGetTpassResult(r, 0), FunASRGetStamp(r));
    FunASRFreeResult(r);
  }
  FunTpassOnlineUninit(oh);
  FunTpassUninit(h);
  return 0;
}
