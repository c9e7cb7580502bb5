// The offline server's decoder-thread shape over the handle-API mirror, with the server's DEFAULT flags:
//   FunOfflineInit(model_path, thread_num = --model-thread-num = 1, ...)     websocket/bin/funasr-wss-server.cpp:105-106,452,511
//   --decoder-thread-num threads that each run do_decoder -> FunOfflineInferBuffer on the one shared handle
//                                                                               funasr-wss-server.cpp:479-481, websocket-server.cpp:60-93,387-403
// Requests (distinct synthetic utterances of different lengths) are first transcribed one after the other from one thread —
// the "separate calls" — then again by the decoder threads pulling from a shared counter, as asio::post onto io_decoder_ does.
// Every concurrent result must equal its separate-call result; the execution-slot statistics of the acoustic model
// (pfhip_inflight_stats) show whether the concurrent calls were merged into packed forwards.  Prints one JSON line.
// "Equal" is counted two ways.  A packed forward of many requests is large enough for the split-precision MFMA GEMMs, a lone
// 5-s request runs the small-grid fp32 kernels: the same sums in another order, 1e-6 apart in the log-probabilities, which can
// turn an argmax at a near-tie.  `mismatches` counts requests that differ by more than ONE substituted token (or in length, or
// in their time stamps with equal tokens) — these fail the run; `near_tie_flips` counts requests with exactly one substituted
// token, reported for the caller to bound.
//   serve_threads <model_dir> <vad_dir|-> [decoder_threads=16] [requests=64] [min_s=3] [max_s=12] [model_thread_num=1]
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/pfhip.h"
#include "funasrruntime_hip.h"

namespace {
struct Totals { long long forwards = 0, calls = 0, utts = 0; int slots = 0, used = 0; };
Totals read_stats(pfhip_model* m) {
  pfhip_slot_stats st[64];
  int n = 0;
  Totals t;
  if (pfhip_inflight_stats(m, st, 64, &n) != PFHIP_OK) return t;
  t.slots = n;
  for (int i = 0; i < n; ++i) { t.forwards += st[i].forwards; t.calls += st[i].calls; t.utts += st[i].utterances; t.used += st[i].forwards > 0; }
  return t;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s model_dir vad_dir|- [decoder_threads] [requests] [min_s] [max_s] [model_thread_num]\n", argv[0]);
    return 2;
  }
  std::map<std::string, std::string> paths;
  paths[MODEL_DIR] = argv[1];
  if (std::string(argv[2]) != "-") paths[VAD_DIR] = argv[2];
  const int threads = argc > 3 ? std::atoi(argv[3]) : 16, requests = argc > 4 ? std::atoi(argv[4]) : 64;
  const double min_s = argc > 5 ? std::atof(argv[5]) : 3.0, max_s = argc > 6 ? std::atof(argv[6]) : 12.0;
  const int model_thread_num = argc > 7 ? std::atoi(argv[7]) : 1;
  FUNASR_HANDLE h = FunOfflineInit(paths, model_thread_num, true, 32);
  pfhip_model* am = FunOfflineGetAsrHandle(h);
  // synthetic s16 requests: a tone + noise, distinct pitch / length / noise per request
  std::vector<std::vector<char>> req(requests);
  unsigned s = 20251114u;
  double audio_s = 0;
  for (int r = 0; r < requests; ++r) {
    const int n = (int)(16000 * (min_s + (max_s - min_s) * ((r * 37) % 101) / 100.0));
    audio_s += n / 16000.0;
    req[r].resize((size_t)n * 2);
    int16_t* p = reinterpret_cast<int16_t*>(req[r].data());
    const float f0 = 110.f * std::pow(2.f, (r % 24) / 12.f);
    for (int i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      const float v = 0.6f * std::sin(6.2831853f * f0 * i / 16000.f) + 0.4f * ((s >> 9) / 4194304.f - 1.f);
      p[i] = (int16_t)std::lrintf(8000.f * v);
    }
  }
  const std::vector<std::vector<float>> no_hw;
  auto infer = [&](int r, std::string& text, std::string& stamp, std::vector<int>& all_ids) {
    FUNASR_RESULT q = FunOfflineInferBuffer(h, req[r].data(), (int)req[r].size(), RASR_NONE, nullptr, no_hw, 16000, "pcm");
    if (!q) return false;
    text = FunASRGetResult(q, 0);
    stamp = FunASRGetStamp(q);
    all_ids.clear();
    for (const auto& ids : FunASRGetSegmentIds(q)) { all_ids.push_back(-1); for (int id : ids) all_ids.push_back(id); }
    FunASRFreeResult(q);
    return true;
  };
  // one call first: the reference's Vocab::Vector2StringV2 carries a flag from call to call (vocab.cpp:176: a Latin word at the
  // start of a result gets a space in front when the PREVIOUS result ended in a complete Latin word), so the very first result of a
  // process differs from the same request served later
  {
    std::string t, st; std::vector<int> v;
    if (!infer(0, t, st, v)) { std::fprintf(stderr, "inference failed\n"); return 1; }
  }
  const Totals z = read_stats(am);
  // separate calls
  std::vector<std::string> want_text(requests), want_stamp(requests);
  std::vector<std::vector<int>> want_ids(requests);
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < requests; ++r)
    if (!infer(r, want_text[r], want_stamp[r], want_ids[r])) { std::fprintf(stderr, "inference failed\n"); return 1; }
  const double dt_seq = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const Totals a = read_stats(am);
  // the decoder threads
  std::atomic<int> next{0}, mismatches{0}, flips{0}, failures{0};
  const auto t1 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&] {
      for (;;) {
        const int r = next.fetch_add(1);
        if (r >= requests) break;
        std::string text, stamp;
        std::vector<int> ids;
        if (!infer(r, text, stamp, ids)) { ++failures; continue; }
        if (ids == want_ids[r]) {
          if (text != want_text[r] || stamp != want_stamp[r]) {
            ++mismatches;
            std::fprintf(stderr, "request %d: equal ids, separate text [%s] stamp [%s], concurrent text [%s] stamp [%s]\n", r,
                         want_text[r].c_str(), want_stamp[r].c_str(), text.c_str(), stamp.c_str());
          }
          continue;
        }
        size_t diff = 0;
        if (ids.size() == want_ids[r].size())
          for (size_t i = 0; i < ids.size(); ++i) diff += ids[i] != want_ids[r][i];
        if (ids.size() == want_ids[r].size() && diff == 1) { ++flips; continue; }
        ++mismatches;
        std::string a, b;
        for (int id : want_ids[r]) a += " " + std::to_string(id);
        for (int id : ids) b += " " + std::to_string(id);
        std::fprintf(stderr, "request %d (%zu samples): separate [%s ] concurrent [%s ]\n", r, req[r].size() / 2, a.c_str(), b.c_str());
      }
    });
  for (auto& th : pool) th.join();
  const double dt_par = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  const Totals b = read_stats(am);
  size_t tokens = 0;
  for (const std::vector<int>& v : want_ids) tokens += v.size();
  std::printf("{\"decoder_threads\": %d, \"model_thread_num\": %d, \"requests\": %d, \"audio_s\": %.1f, \"inflight\": %d, \"slots\": %d, "
              "\"slots_used\": %d, \"separate\": {\"forwards\": %lld, \"calls\": %lld, \"utterances\": %lld, \"wall_s\": %.4f}, "
              "\"concurrent\": {\"forwards\": %lld, \"calls\": %lld, \"utterances\": %lld, \"wall_s\": %.4f}, "
              "\"mismatches\": %d, \"near_tie_flips\": %d, \"failures\": %d, \"id_chars\": %zu}\n",
              threads, model_thread_num, requests, audio_s, pfhip_get_inflight(am), b.slots, b.used, a.forwards - z.forwards, a.calls - z.calls, a.utts - z.utts, dt_seq,
              b.forwards - a.forwards, b.calls - a.calls, b.utts - a.utts, dt_par, mismatches.load(), flips.load(), failures.load(), tokens);
  FunOfflineUninit(h);
  return mismatches.load() || failures.load() ? 3 : 0;
}
