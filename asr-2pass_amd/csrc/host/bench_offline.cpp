// C++ harness with the semantics of the reference's RTF tool
// (onnxruntime/bin/funasr-onnx-offline-rtf.cpp:54-102,257-260): N worker threads share ONE model handle
// and pull utterance indices from an atomic counter, one warm-up inference, only the Forward call is
// timed, total_rtf = max thread compute time / total audio seconds, speedup = 1/rtf.
//   bench_offline <weights.bin> <manifest.json> [threads=1] [utts=8] [seconds=30] [batch=1]
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "paraformer_hip.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s weights.bin manifest.json [threads] [utts] [seconds] [batch]\n", argv[0]);
    return 2;
  }
  const int threads = argc > 3 ? std::atoi(argv[3]) : 1;
  const int utts = argc > 4 ? std::atoi(argv[4]) : 8;
  const int seconds = argc > 5 ? std::atoi(argv[5]) : 30;
  const int batch = argc > 6 ? std::atoi(argv[6]) : 1;
  funasr::ParaformerHip model;
  model.InitAsr(argv[1], "", argv[2], "", 1);
  model.SetBatchSize(batch);
  const int n = seconds * 16000;
  std::vector<std::vector<float>> pcm(utts, std::vector<float>(n));
  unsigned s = 20251114u;
  for (int u = 0; u < utts; ++u)
    for (int i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      pcm[u][i] = 0.15f * std::sin(6.2831853f * 110.f * (1 + u % 24 / 12.f) * i / 16000.f) + ((s >> 9) / 8388608.f - 0.5f) * 0.2f;
    }
  {  // warm-up (funasr-onnx-offline-rtf.cpp:54-61)
    float* d = pcm[0].data();
    int l = n;
    model.Forward(&d, &l, true, {{0.f}}, nullptr, 1);
  }
  std::atomic<int> next{0};
  std::vector<double> busy(threads, 0.0);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      for (;;) {
        const int i0 = next.fetch_add(batch);
        if (i0 >= utts) break;
        const int b = std::min(batch, utts - i0);
        std::vector<float*> d(b);
        std::vector<int> l(b, n);
        for (int k = 0; k < b; ++k) d[k] = pcm[i0 + k].data();
        const auto t0 = std::chrono::steady_clock::now();
        model.Forward(d.data(), l.data(), true, {{0.f}}, nullptr, b);
        busy[t] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
    });
  for (auto& th : pool) th.join();
  double mx = 0;
  for (double b : busy) mx = b > mx ? b : mx;
  const double audio = (double)utts * seconds;
  std::printf("total_time_wav %.1f s, max thread compute %.4f s, total_rtf %.6f, speedup %.1f\n", audio, mx, mx / audio,
              audio / mx);
  return 0;
}
