// C ABI of the host-side VAD end-point detector (include/pfhip.h, pfhip_vadseg_*).
#include <new>

#include "../../include/pfhip.h"
#include "host/timestamp.h"
#include "host/postprocess.h"
#include "host/vad_segmenter.h"
#include "internal.h"

struct pfhip_vadseg { pfhip_host::VadSegmenter seg; };

extern "C" {

pfhip_status pfhip_vadseg_create(pfhip_vadseg** out) {
  if (!out) return pfhip_detail::fail(PFHIP_ERR_ARG, "null argument");
  *out = new (std::nothrow) pfhip_vadseg;
  return *out ? PFHIP_OK : pfhip_detail::fail(PFHIP_ERR_ARG, "out of memory");
}
void pfhip_vadseg_destroy(pfhip_vadseg* s) { delete s; }
pfhip_status pfhip_vadseg_reset(pfhip_vadseg* s) {
  if (!s) return pfhip_detail::fail(PFHIP_ERR_ARG, "null handle");
  s->seg.ResetAll();
  return PFHIP_OK;
}
pfhip_status pfhip_vadseg_feed(pfhip_vadseg* s, const float* sil_prob, int n_frames, const float* waveform, int n_samples,
                               int is_final, int online, int max_end_sil, int max_single_segment_time,
                               float speech_noise_thres, int sample_rate, int32_t* segments, int cap_pairs,
                               int* n_segments) {
  if (!s || n_frames < 0 || n_samples < 0 || (n_frames && !sil_prob) || (n_samples && !waveform) || !n_segments)
    return pfhip_detail::fail(PFHIP_ERR_ARG, "bad argument");
  // every scored frame needs its 25-ms energy window (e2e-vad.h:433-449 / :593)
  const int have = n_samples >= 400 ? (n_samples - 400) / 160 + 1 : 0;
  if (have < n_frames) return pfhip_detail::fail(PFHIP_ERR_ARG, "waveform shorter than the scored frames");
  const auto segs = s->seg.Feed(sil_prob, n_frames, waveform, n_samples, is_final != 0, online != 0, max_end_sil,
                                max_single_segment_time, speech_noise_thres, sample_rate);
  *n_segments = (int)segs.size();
  if ((int)segs.size() > cap_pairs) return pfhip_detail::fail(PFHIP_ERR_CAPACITY, "segment buffer too small");
  for (size_t i = 0; i < segs.size(); ++i) { segments[2 * i] = segs[i].start_ms; segments[2 * i + 1] = segs[i].end_ms; }
  return PFHIP_OK;
}

pfhip_status pfhip_timestamp_onnx(float* us_alphas, const float* us_cif_peak, int n_frames3, int n_chars, float begin_time_ms,
                                  float total_offset, float* spans, int cap_spans, int* n_spans) {
  if (!us_alphas || !us_cif_peak || n_frames3 < 0 || !n_spans) return pfhip_detail::fail(PFHIP_ERR_ARG, "bad argument");
  std::vector<float> a(us_alphas, us_alphas + n_frames3), p(us_cif_peak, us_cif_peak + n_frames3);
  const auto r = pfhip_host::TimestampOnnx(a, p, n_chars, begin_time_ms, total_offset);
  for (int i = 0; i < n_frames3; ++i) us_alphas[i] = a[i];
  *n_spans = (int)r.size();
  if ((int)r.size() > cap_spans) return pfhip_detail::fail(PFHIP_ERR_CAPACITY, "span buffer too small");
  for (size_t i = 0; i < r.size(); ++i) { spans[3 * i] = r[i].begin_s; spans[3 * i + 1] = r[i].end_s; spans[3 * i + 2] = r[i].is_sil ? 1.f : 0.f; }
  return PFHIP_OK;
}

pfhip_status pfhip_post_process(const char* const* chars, const float* stamps, int n, char* out, int cap, int* n_out) {
  if ((n > 0 && (!chars || !stamps)) || n < 0 || !out || !n_out) return pfhip_detail::fail(PFHIP_ERR_ARG, "bad argument");
  std::vector<std::string> rc(n);
  std::vector<std::vector<float>> ts(n);
  for (int i = 0; i < n; ++i) { rc[i] = chars[i] ? chars[i] : ""; ts[i] = {stamps[2 * i], stamps[2 * i + 1]}; }
  const std::string r = pfhip_host::PostProcess(rc, ts);
  *n_out = (int)r.size();
  if ((int)r.size() + 1 > cap) return pfhip_detail::fail(PFHIP_ERR_CAPACITY, "output string buffer too small");
  std::memcpy(out, r.c_str(), r.size() + 1);
  return PFHIP_OK;
}

}  // extern "C"
