// TEST INFRASTRUCTURE ONLY (see oracle/README.md): thin extern "C" shim around the
// reference's vendored kaldi-native-fbank, compiled IN PLACE from /root/reference by
// oracle/Makefile into oracle/_ref/libknf_ref.so.  No reference source is copied.
//
// Options are exactly those Paraformer::InitAsr sets
// (onnxruntime/src/paraformer.cpp:24-31; defaults paraformer.h:112-121) and the x32768
// scaling is Paraformer::FbankKaldi (onnxruntime/src/paraformer.cpp:309-323).
#include <cstdint>
#include <cstring>
#include <vector>

#include "kaldi-native-fbank/csrc/online-feature.h"
#include "kaldi-native-fbank/csrc/rfft.h"

static knf::FbankOptions make_opts(int n_mels, float fs, const char* window, float shift_ms, float len_ms) {
  knf::FbankOptions o;
  o.frame_opts.dither = 0;
  o.mel_opts.num_bins = n_mels;
  o.frame_opts.samp_freq = fs;
  o.frame_opts.window_type = window;
  o.frame_opts.frame_shift_ms = shift_ms;
  o.frame_opts.frame_length_ms = len_ms;
  o.energy_floor = 0;
  o.mel_opts.debug_mel = false;
  return o;
}

extern "C" {

// waves are in [-1,1) floats as Model::Forward receives them; returns frame count,
// writes frames*n_mels floats into out (if out != nullptr and cap is large enough).
int knf_ref_fbank(const float* waves, int len, int n_mels, float* out, int cap_frames) {
  knf::FbankOptions opts = make_opts(n_mels, 16000.f, "hamming", 10.f, 25.f);
  knf::OnlineFbank fbank(opts);
  std::vector<float> buf(len);
  for (int32_t i = 0; i != len; ++i) buf[i] = waves[i] * 32768;
  fbank.AcceptWaveform(16000.f, buf.data(), buf.size());
  int32_t frames = fbank.NumFramesReady();
  if (out) {
    for (int32_t i = 0; i != frames && i < cap_frames; ++i) {
      std::memcpy(out + (size_t)i * n_mels, fbank.GetFrame(i), sizeof(float) * n_mels);
    }
  }
  return frames;
}

// knf::Rfft known-answer entry (third_party/.../csrc/test-rfft.cc:32-50).
void knf_ref_rfft(float* in_out, int n) {
  knf::Rfft fft(n);
  fft.Compute(in_out);
}

}  // extern "C"
