#include "timestamp.h"

#include <numeric>

namespace pfhip_host {

std::vector<TimeSpan> TimestampOnnx(std::vector<float>& us_alphas, const std::vector<float>& us_cif_peak, int n_chars,
                                    float begin_time_ms, float total_offset) {
  std::vector<TimeSpan> spans;
  if (n_chars <= 0) return spans;
  const float kStartEnd = 5.0f, kMaxTok = 30.0f;
  const float kRate = 10.0 * 6 / 1000 / 3;
  const float kPeak = 1.0 - 1e-4;
  std::vector<float> peak = us_cif_peak;
  const int n_frames = (int)peak.size();
  auto collect = [&](std::vector<float>& fires) {
    fires.clear();
    for (int i = 0; i < n_frames && i < (int)peak.size(); ++i)
      if (peak[i] > 1.0 - 1e-4) fires.push_back(i + total_offset);
  };
  std::vector<float> fire;
  collect(fire);
  if ((int)fire.size() != n_chars + 1) {                       // (:872-904) rebuild the peaks from the alphas
    float sum = std::accumulate(us_alphas.begin(), us_alphas.end(), 0.0f);
    const float scale = sum / (n_chars + 1);
    if (scale == 0) return spans;
    peak.clear();
    sum = 0.0f;
    for (float& a : us_alphas) {
      a = a / scale;
      sum += a;
      peak.push_back(sum);
      if (sum >= 1.0 - 1e-4) sum -= kPeak;
    }
    for (int idx = (int)peak.size() - 1; sum >= 1.0 - 1e-4 && idx >= 0; --idx) {
      if (peak[idx] < 1.0 - 1e-4) { peak[idx] = sum; sum -= kPeak; }
    }
    collect(fire);
  }
  const int n_peak = (int)fire.size();
  if (n_peak == 0) return spans;
  if (fire[0] > kStartEnd) spans.push_back({0.0f, fire[0] * kRate, true});
  for (int i = 0; i < n_peak - 1; ++i) {
    if (i == n_peak - 2 || fire[i + 1] - fire[i] < kMaxTok) {
      spans.push_back({fire[i] * kRate, fire[i + 1] * kRate, false});
    } else {
      const float split = fire[i] + kMaxTok;
      spans.push_back({fire[i] * kRate, split * kRate, false});
      spans.push_back({split * kRate, fire[i + 1] * kRate, true});
    }
  }
  if (spans.empty()) return spans;
  if (n_frames - fire.back() > kStartEnd) {
    const float end = (n_frames + fire.back()) / 2.0;
    spans.back().end_s = end * kRate;
    spans.push_back({end * kRate, n_frames * kRate, true});
  } else {
    spans.back().end_s = n_frames * kRate;
  }
  if (begin_time_ms) {
    for (TimeSpan& t : spans) { t.begin_s += begin_time_ms / 1000.0; t.end_s += begin_time_ms / 1000.0; }
  }
  return spans;
}

}  // namespace pfhip_host
