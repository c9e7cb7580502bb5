// Token time stamps from the upsampled CIF outputs — behaviourally `funasr::TimestampOnnx`
// (onnxruntime/src/util.cpp:838-963): peaks of us_cif_peak (> 1 - 1e-4) are token boundaries on a 20-ms grid
// (TIME_RATE = 10*6/1000/3 s, :851); when their count is not tokens+1 the peaks are re-derived from the rescaled
// us_alphas (:872-904); long gaps are split into token + <sil> (:918-928); leading / trailing silence (:911-915,
// :936-943).  Host post-processing that consumes two extra model outputs.
#pragma once
#include <vector>

namespace pfhip_host {

struct TimeSpan { float begin_s; float end_s; bool is_sil; };

// n_chars = recognised tokens WITHOUT a trailing "</s>" (the reference pops it, :856-858).  us_alphas is rescaled in
// place like the reference's by-reference parameter.  Returns every span incl. <sil>; the caller keeps the non-sil
// ones as the token stamps (:957-961).
std::vector<TimeSpan> TimestampOnnx(std::vector<float>& us_alphas, const std::vector<float>& us_cif_peak, int n_chars,
                                    float begin_time_ms = 0.0f, float total_offset = -1.5f);

}  // namespace pfhip_host
