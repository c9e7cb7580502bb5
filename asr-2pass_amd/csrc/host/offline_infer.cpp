// Harness over the handle-API mirror (funasrruntime_hip.h) with the shape of the reference's offline client
// (onnxruntime/bin/funasr-onnx-offline.cpp: FunOfflineInit -> FunOfflineInferBuffer per file -> FunASRGetResult):
//   offline_infer <model_dir> <vad_dir|-> <pcm_s16_file> [batch=32] [threads=1] [repeat=1]
// Prints one line per VAD segment, time order: "seg <start_sample> <end_sample> : <ids...>", then timing; with threads > 1
// every thread transcribes the same buffer through the shared handle (the server's decoder threads).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>

#include "funasrruntime_hip.h"

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s model_dir vad_dir|- pcm_s16_file [batch] [threads] [repeat] [punc_dir]\n", argv[0]);
    return 2;
  }
  std::map<std::string, std::string> paths;
  paths[MODEL_DIR] = argv[1];
  if (std::string(argv[2]) != "-") paths[VAD_DIR] = argv[2];
  if (argc > 7) paths[PUNC_DIR] = argv[7];
  const int batch = argc > 4 ? std::atoi(argv[4]) : 32, threads = argc > 5 ? std::atoi(argv[5]) : 1, repeat = argc > 6 ? std::atoi(argv[6]) : 1;
  std::ifstream f(argv[3], std::ios::binary);
  std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  FUNASR_HANDLE h = FunOfflineInit(paths, threads, true, batch);
  const std::vector<std::vector<float>> no_hw;
  {
    FUNASR_RESULT r = FunOfflineInferBuffer(h, buf.data(), (int)buf.size(), RASR_NONE, nullptr, no_hw, 16000, "pcm");
    if (!r) { std::fprintf(stderr, "inference failed\n"); return 1; }
    const auto& segs = FunASRGetSegments(r);
    const auto& ids = FunASRGetSegmentIds(r);
    for (size_t i = 0; i < segs.size(); ++i) {
      std::printf("seg %d %d :", segs[i].first, segs[i].second);
      for (int id : ids[i]) std::printf(" %d", id);
      std::printf("\n");
    }
    std::printf("text %s\nstamp %s\n", FunASRGetResult(r, 0), FunASRGetStamp(r));
    const std::string want_text = FunASRGetResult(r, 0), want_stamp = FunASRGetStamp(r);
    std::atomic<int> mismatches{0};
    const float secs = FunASRGetRetSnippetTime(r);
    FunASRFreeResult(r);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
      pool.emplace_back([&] {
        for (int k = 0; k < repeat; ++k) {
          FUNASR_RESULT q = FunOfflineInferBuffer(h, buf.data(), (int)buf.size(), RASR_NONE, nullptr, no_hw, 16000, "pcm");
          if (!q || want_text != FunASRGetResult(q, 0) || want_stamp != FunASRGetStamp(q)) ++mismatches;      // re-entrancy check
          FunASRFreeResult(q);
        }
      });
    for (auto& th : pool) th.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("audio %.1f s x %d files in %.3f s -> %.0f xRT\n", secs, threads * repeat, dt, secs * threads * repeat / dt);
    if (mismatches.load()) { std::fprintf(stderr, "%d concurrent results differ from the single-threaded one\n", mismatches.load()); return 3; }
  }
  FunOfflineUninit(h);
  return 0;
}
