// TEST INFRASTRUCTURE ONLY: thin extern "C" shim around the reference's end-point detector `funasr::E2EVadModel`
// (onnxruntime/src/e2e-vad.h:269-783, header-only), compiled IN PLACE from /root/reference by oracle/Makefile into
// oracle/_ref/libe2evad_ref.so.  No reference source is copied; the header needs no stand-in (standard library only).
// Used by tests/golden/make_vadseg_golden.py to generate the fixtures that pin oracle/e2e_vad.py and the product's
// csrc/host/vad_segmenter.cpp, and by tests directly where the .so is present.
//
// Scores: the detector reads scores[t][sil_pdf_ids] with sil_pdf_ids = {0} (e2e-vad.h:103,601-606), i.e. only the
// silence posterior of each frame; rows are built with that single column.
#include <vector>

#include "e2e-vad.h"

extern "C" {

void* e2evad_ref_create() { return new funasr::E2EVadModel(); }

void e2evad_ref_destroy(void* h) { delete static_cast<funasr::E2EVadModel*>(h); }

// One call of E2EVadModel::operator() (e2e-vad.h:302-362).  Returns the number of segments; the first `cap` of them are
// written to pairs as (start_ms, end_ms), -1 = open (online mode).
int e2evad_ref_feed(void* h, const float* sil, int n_frames, const float* waveform, int n_samples, int is_final, int online,
                    int max_end_sil, int max_single_segment_time, float speech_noise_thres, int sample_rate, int* pairs, int cap) {
  std::vector<std::vector<float>> score((size_t)n_frames, std::vector<float>(1));
  for (int t = 0; t < n_frames; ++t) score[t][0] = sil[t];
  const std::vector<float> wave(waveform, waveform + n_samples);
  const std::vector<std::vector<int>> segs = (*static_cast<funasr::E2EVadModel*>(h))(
      score, wave, is_final != 0, online != 0, max_end_sil, max_single_segment_time, speech_noise_thres, sample_rate);
  for (size_t i = 0; i < segs.size() && (int)i < cap; ++i) { pairs[2 * i] = segs[i][0]; pairs[2 * i + 1] = segs[i][1]; }
  return (int)segs.size();
}

}  // extern "C"
