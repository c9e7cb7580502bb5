// Load harness over the 2-pass handle API: N connections (one thread each, like the websocket server's handlers,
// websocket/bin/websocket-server-2pass.cpp:85-148) share ONE TpassStream and feed the same s16 PCM file in 600-ms pieces as
// fast as the models answer; the concurrent online-VAD, streaming-ASR and 2nd-pass calls are merged into batched device
// passes by the library.  Reports aggregate audio-seconds per second and the distribution of the per-call latency (one call =
// one 600-ms websocket message through FunTpassInferBuffer: online VAD + streaming chunk + any 2nd pass that closed).
//   tpass_bench <offline_dir> <online_dir> <vad_dir> <punc_dir|-> <pcm_s16_file> [connections=64] [mode=2]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <thread>
#include <vector>

#include "funasrruntime_hip.h"

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s offline_dir online_dir vad_dir punc_dir|- pcm_s16_file [connections] [mode]\n", argv[0]);
    return 2;
  }
  std::map<std::string, std::string> paths;
  paths[MODEL_DIR] = argv[1]; paths[ONLINE_MODEL_DIR] = argv[2]; paths[VAD_DIR] = argv[3];
  if (std::string(argv[4]) != "-") paths[PUNC_DIR] = argv[4];
  const int n_conn = argc > 6 ? std::atoi(argv[6]) : 64;
  const ASR_TYPE mode = argc > 7 ? (ASR_TYPE)std::atoi(argv[7]) : ASR_TWO_PASS;
  std::ifstream f(argv[5], std::ios::binary);
  std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (buf.size() < 19200) { std::fprintf(stderr, "pcm file too short\n"); return 2; }
  FUNASR_HANDLE h = FunTpassInit(paths, n_conn);
  if (!h) return 1;
  {  // warm-up: one connection over the first 3 s
    FUNASR_HANDLE oh = FunTpassOnlineInit(h, {5, 10, 5});
    std::vector<std::vector<std::string>> pc(2);
    for (int off = 0; off < 5 * 19200 && off + 19200 <= (int)buf.size(); off += 19200) {
      FUNASR_RESULT r = FunTpassInferBuffer(h, oh, buf.data() + off, 19200, pc, off + 19200 >= 5 * 19200, 16000, "pcm", mode);
      if (r) FunASRFreeResult(r);
    }
    FunTpassOnlineUninit(oh);
  }
  std::mutex mu;
  std::condition_variable cv;
  int ready = 0;
  bool go = false;
  std::atomic<long> online_chars{0}, tpass_chars{0}, tpass_results{0}, calls{0}, failures{0};
  std::vector<double> worst(n_conn, 0.0);
  std::vector<std::vector<float>> lat(n_conn);
  std::vector<std::thread> pool;
  const int n_bytes = (int)buf.size();
  for (int c = 0; c < n_conn; ++c)
    pool.emplace_back([&, c] {
      FUNASR_HANDLE oh = FunTpassOnlineInit(h, {5, 10, 5});
      std::vector<std::vector<std::string>> punc_cache(2);
      {
        std::unique_lock<std::mutex> lk(mu);
        ++ready;
        cv.notify_all();
        cv.wait(lk, [&] { return go; });
      }
      for (int off = 0; off < n_bytes; off += 19200) {
        const int nb = std::min(19200, n_bytes - off);
        const auto t0 = std::chrono::steady_clock::now();
        FUNASR_RESULT r = FunTpassInferBuffer(h, oh, buf.data() + off, nb, punc_cache, off + 19200 >= n_bytes, 16000, "pcm", mode);
        const double call_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        worst[c] = std::max(worst[c], call_s);
        lat[c].push_back((float)(call_s * 1e3));
        ++calls;
        if (!r) { ++failures; continue; }
        online_chars += (long)std::string(FunASRGetResult(r, 0)).size();
        const std::string tp = FunASRGetTpassResult(r, 0);
        if (!tp.empty()) { ++tpass_results; tpass_chars += (long)tp.size(); }
        FunASRFreeResult(r);
      }
      FunTpassOnlineUninit(oh);
    });
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return ready == n_conn; });
    go = true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  cv.notify_all();
  for (auto& th : pool) th.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double audio = (double)n_conn * (double)(n_bytes / 2) / 16000.0;
  double mx = 0;
  for (double w : worst) mx = std::max(mx, w);
  std::vector<float> all;
  for (const auto& v : lat) all.insert(all.end(), v.begin(), v.end());
  std::sort(all.begin(), all.end());
  auto pct = [&](double q) { return all.empty() ? 0.0 : (double)all[std::min(all.size() - 1, (size_t)(q * (double)all.size()))]; };
  std::printf("{\"connections\": %d, \"audio_s\": %.1f, \"wall_s\": %.3f, \"xrt\": %.1f, \"calls\": %ld, \"failures\": %ld, "
              "\"ms_per_round\": %.2f, \"p50_call_ms\": %.2f, \"p99_call_ms\": %.2f, \"worst_call_ms\": %.1f, \"tpass_results\": %ld, "
              "\"online_bytes\": %ld, \"tpass_bytes\": %ld}\n",
              n_conn, audio, dt, audio / dt, calls.load(), failures.load(), dt / ((n_bytes + 19199) / 19200) * 1e3, pct(0.5), pct(0.99), mx * 1e3,
              tpass_results.load(), online_chars.load(), tpass_chars.load());
  FunTpassUninit(h);
  return failures.load() ? 1 : 0;
}
