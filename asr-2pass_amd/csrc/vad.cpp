// FSMN-VAD forward on MI355X — the C ABI's pfhip_vad_* family.
// Replaces FsmnVad::FbankKaldi + LfrCmvn + Forward (onnxruntime/src/fsmn-vad.cpp:137-152, 198-238, 72-135): PCM ->
// frame scores [T, 248] with the four [128 x 19] FSMN caches carried in HBM.  The network is causal, so a whole
// file goes through in ONE pass of ~12 launches instead of the reference's 1-s slices (600 Runs per 10 minutes,
// onnxruntime/src/audio.cpp:1183-1196); slice-wise calls give the same scores because the caches carry over.
// Linear layers run on the same fp32 MFMA GEMM kernels as the ASR model; their odd widths (140, 250, 248) are
// zero-padded once at load ([N up to 128][K up to 32]) so that pad outputs are exact zeros.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>

#include "internal.h"
#include "json_min.h"

using namespace pfhip_detail;

struct pfhip_vad {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  int n_mels = 80, lfr_m = 5, lfr_n = 1, input_dim = 400, proj = 128, lorder = 20, layers = 4, n_out = 248, linear = 250;
  FrontendTables ft;
  float *d_mean = nullptr, *d_istd = nullptr;
  Lin in1, in2, out1, out2;
  std::vector<Lin> blk_linear, blk_affine;
  std::vector<float*> fsmn_w;
  Buf pcm, fb, feats, a, b, p, f, probs, sil, meta, cache[2], segs, fbk, ops;
  int cache_cur = 0;
  int* h_pin = nullptr;
  // pinned staging for the online path's PCM (grown on demand); h_wave(n) returns a buffer of at least n floats
  float* h_wave_buf = nullptr; size_t h_wave_cap = 0;
  // merging of concurrent pfhip_vad_stream_infer callers (pfhip_set_vad_stream_batching)
  pfhip_detail::MergeQueue<struct VadReq> mq;
  int q_wait_us = 0, q_max = 1;
  std::atomic<int> live_streams{0};
  float* h_wave(size_t n) {
    if (n > h_wave_cap) {
      if (h_wave_buf) (void)hipHostFree(h_wave_buf);
      h_wave_buf = nullptr; h_wave_cap = 0;
      const size_t cap = std::max<size_t>(n, 70000);
      if (hipHostMalloc((void**)&h_wave_buf, cap * 4, hipHostMallocDefault) == hipSuccess) h_wave_cap = cap;
    }
    return h_wave_buf;
  }
};

namespace {

void lin_gemm(hipStream_t s, const Lin& l, const float* A, int lda, float* C, int ldc, int M, bool relu) {
  pfhip_detail::lin_gemm(s, l, A, lda, C, ldc, nullptr, 0, nullptr, 0, M, relu);
}

}  // namespace

extern "C" {

pfhip_status pfhip_vad_create_from_memory(const void* blob, size_t blob_bytes, const char* manifest_json, int device,
                                          pfhip_vad** out) {
  last_error().clear();
  if (!blob || !manifest_json || !out) return fail(PFHIP_ERR_ARG, "null argument");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(PFHIP_ERR_ARG, "device ordinal out of range");
  HIP_TRY(hipSetDevice(device));
  pfhip::JValue man;
  try { man = pfhip::JParser(manifest_json).parse(); }
  catch (const std::exception& e) { return fail(PFHIP_ERR_FORMAT, e.what()); }
  const pfhip::JValue* jc = man.get("config");
  const pfhip::JValue* jt = man.get("tensors");
  if (!jc || !jt || jt->kind != pfhip::JValue::OBJ) return fail(PFHIP_ERR_FORMAT, "manifest needs config and tensors");
  std::unique_ptr<pfhip_vad> v(new pfhip_vad);
  v->device = device;
  v->n_mels = (int)jc->number("n_mels", 80);
  v->lfr_m = (int)jc->number("lfr_m", 5);
  v->lfr_n = (int)jc->number("lfr_n", 1);
  v->input_dim = (int)jc->number("input_dim", 400);
  v->proj = (int)jc->number("proj", 128);
  v->lorder = (int)jc->number("lorder", 20);
  v->layers = (int)jc->number("layers", 4);
  v->n_out = (int)jc->number("n_out", 248);
  v->linear = (int)jc->number("linear", 250);
  const int affine = (int)jc->number("affine", 140), out_affine = (int)jc->number("out_affine", 140);
  if (v->n_mels != 80 || v->input_dim != v->n_mels * v->lfr_m || v->lorder != 20 || v->proj % 4 || v->layers < 1 ||
      v->layers > 16)
    return fail(PFHIP_ERR_UNSUPPORTED, "FSMN-VAD kernels need 80 mels, left order 20");
  const float* hb = static_cast<const float*>(blob);
  auto get = [&](const std::string& name, std::vector<int> shape, const float** p) -> bool {
    const pfhip::JValue* t = jt->get(name);
    if (!t) { last_error() = "missing tensor " + name; return false; }
    const pfhip::JValue* sh = t->get("shape");
    const pfhip::JValue* of = t->get("offset");
    if (!sh || !of || sh->arr.size() != shape.size()) { last_error() = "tensor " + name + " malformed"; return false; }
    size_t n = 1;
    for (size_t i = 0; i < shape.size(); ++i) {
      if ((int)sh->arr[i].num != shape[i]) { last_error() = "tensor " + name + " has unexpected shape"; return false; }
      n *= (size_t)shape[i];
    }
    const size_t off = (size_t)of->num;
    if (off % 4 || off + n * 4 > blob_bytes) { last_error() = "tensor " + name + " out of blob"; return false; }
    *p = hb + off / 4;
    return true;
  };
  auto lin = [&](const std::string& name, int N, int K, bool bias, Lin* l) -> pfhip_status {
    const float *w = nullptr, *b = nullptr;
    if (!get(name + ".w", {N, K}, &w)) return PFHIP_ERR_FORMAT;
    if (bias && !get(name + ".b", {N}, &b)) return PFHIP_ERR_FORMAT;
    return pack_linear(w, b, N, K, l);
  };
  pfhip_status st;
  const float *mean = nullptr, *istd = nullptr;
  if (!get("cmvn.mean", {v->input_dim}, &mean) || !get("cmvn.istd", {v->input_dim}, &istd)) return PFHIP_ERR_FORMAT;
  HIP_TRY(hipMalloc((void**)&v->d_mean, v->input_dim * 4));
  HIP_TRY(hipMemcpy(v->d_mean, mean, v->input_dim * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&v->d_istd, v->input_dim * 4));
  HIP_TRY(hipMemcpy(v->d_istd, istd, v->input_dim * 4, hipMemcpyHostToDevice));
  if ((st = lin("in1", affine, v->input_dim, true, &v->in1)) || (st = lin("in2", v->linear, affine, true, &v->in2))) return st;
  v->blk_linear.resize(v->layers); v->blk_affine.resize(v->layers); v->fsmn_w.assign(v->layers, nullptr);
  for (int i = 0; i < v->layers; ++i) {
    const std::string p = "blk." + std::to_string(i) + ".";
    if ((st = lin(p + "linear", v->proj, v->linear, false, &v->blk_linear[i])) ||
        (st = lin(p + "affine", v->linear, v->proj, true, &v->blk_affine[i])))
      return st;
    const float* fw = nullptr;
    if (!get(p + "fsmn.w", {v->proj, v->lorder}, &fw)) return PFHIP_ERR_FORMAT;
    HIP_TRY(hipMalloc((void**)&v->fsmn_w[i], (size_t)v->proj * v->lorder * 4));
    HIP_TRY(hipMemcpy(v->fsmn_w[i], fw, (size_t)v->proj * v->lorder * 4, hipMemcpyHostToDevice));
  }
  if ((st = lin("out1", out_affine, v->linear, true, &v->out1)) || (st = lin("out2", v->n_out, out_affine, true, &v->out2))) return st;
  if ((st = build_frontend_tables(v->n_mels, 16000, &v->ft))) return st;
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(v->cache[i].ensure((size_t)v->layers * 19 * v->proj * 4));
    HIP_TRY(hipMemset(v->cache[i].p, 0, (size_t)v->layers * 19 * v->proj * 4));
  }
  HIP_TRY(v->meta.ensure(256));
  HIP_TRY(hipHostMalloc((void**)&v->h_pin, 256, hipHostMallocDefault));
  HIP_TRY(hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking));
  *out = v.release();
  return PFHIP_OK;
}

void pfhip_vad_destroy(pfhip_vad* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  (void)hipDeviceSynchronize();
  for (Buf* b : {&v->pcm, &v->fb, &v->feats, &v->a, &v->b, &v->p, &v->f, &v->probs, &v->meta, &v->cache[0], &v->cache[1], &v->segs, &v->fbk, &v->ops, &v->sil}) b->release();
  auto fl = [](Lin& l) { free_lin(l); };
  fl(v->in1); fl(v->in2); fl(v->out1); fl(v->out2);
  for (auto& l : v->blk_linear) fl(l);
  for (auto& l : v->blk_affine) fl(l);
  for (float* p : v->fsmn_w) if (p) (void)hipFree(p);
  for (void* p : {(void*)v->d_mean, (void*)v->d_istd, (void*)v->ft.d_window, (void*)v->ft.d_tw, (void*)v->ft.d_mel_off,
                  (void*)v->ft.d_mel_size, (void*)v->ft.d_mel_w})
    if (p) (void)hipFree(p);
  if (v->h_pin) (void)hipHostFree(v->h_pin);
  if (v->h_wave_buf) (void)hipHostFree(v->h_wave_buf);
  if (v->stream) (void)hipStreamDestroy(v->stream);
  delete v;
}

pfhip_status pfhip_vad_reset(pfhip_vad* v) {
  last_error().clear();
  if (!v) return fail(PFHIP_ERR_ARG, "null handle");
  std::lock_guard<std::mutex> lk(v->mu);
  HIP_TRY(hipSetDevice(v->device));
  for (int i = 0; i < 2; ++i) HIP_TRY(hipMemsetAsync(v->cache[i].p, 0, (size_t)v->layers * 19 * v->proj * 4, v->stream));
  HIP_TRY(hipStreamSynchronize(v->stream));
  return PFHIP_OK;
}

int pfhip_vad_num_classes(const pfhip_vad* v) { return v ? v->n_out : 0; }

// The FSMN-VAD network on the T packed rows waiting in v->feats (caller holds v->mu and has sized the workspace); scores in
// v->probs.  d_segs: one VadSeg per connection (its rows and caches), max_T = longest segment.
static void vad_network(pfhip_vad* v, hipStream_t s, int T, const pfhip::VadSeg* d_segs, int B, int max_T) {
  lin_gemm(s, v->in1, v->feats.f(), v->in1.Kp, v->a.f(), 256, T, false);
  lin_gemm(s, v->in2, v->a.f(), 256, v->b.f(), 256, T, true);
  for (int i = 0; i < v->layers; ++i) {
    lin_gemm(s, v->blk_linear[i], v->b.f(), 256, v->p.f(), 128, T, false);
    pfhip::launch_fsmn_causal20(v->p.f(), 128, v->fsmn_w[i], d_segs, B, max_T, i, v->f.f(), 128, v->proj, s);
    lin_gemm(s, v->blk_affine[i], v->f.f(), 128, v->b.f(), 256, T, true);
  }
  lin_gemm(s, v->out1, v->b.f(), 256, v->a.f(), 256, T, false);
  lin_gemm(s, v->out2, v->a.f(), 256, v->b.f(), 256, T, false);
  pfhip::launch_softmax_rows(v->b.f(), 256, T, v->n_out, v->probs.f(), v->sil.f(), s);
}

static pfhip_status vad_workspace(pfhip_vad* v, int T) {
  const int Tp = round_up(T, 128);
  HIP_TRY(v->feats.ensure((size_t)Tp * v->in1.Kp * 4));
  HIP_TRY(v->a.ensure((size_t)Tp * 256 * 4));
  HIP_TRY(v->b.ensure((size_t)Tp * 256 * 4));
  HIP_TRY(v->p.ensure((size_t)Tp * 128 * 4));
  HIP_TRY(v->f.ensure((size_t)Tp * 128 * 4));
  HIP_TRY(v->probs.ensure((size_t)T * v->n_out * 4));
  HIP_TRY(v->sil.ensure((size_t)T * 4));
  if (v->in1.Np > 256 || v->in2.Np > 256 || v->out1.Np > 256 || v->out2.Np > 256 || v->proj > 128)
    return fail(PFHIP_ERR_UNSUPPORTED, "FSMN-VAD layer wider than the workspace");
  return PFHIP_OK;
}

static pfhip_status vad_forward_impl(pfhip_vad* v, const float* pcm, int n_samples, int is_final, float* probs,
                                     size_t cap_floats, int* n_frames, bool sil_only);

pfhip_status pfhip_vad_forward(pfhip_vad* v, const float* pcm, int n_samples, int is_final, float* probs,
                               size_t cap_floats, int* n_frames) {
  return vad_forward_impl(v, pcm, n_samples, is_final, probs, cap_floats, n_frames, false);
}

pfhip_status pfhip_vad_forward_sil(pfhip_vad* v, const float* pcm, int n_samples, int is_final, float* sil_prob,
                                   size_t cap_floats, int* n_frames) {
  return vad_forward_impl(v, pcm, n_samples, is_final, sil_prob, cap_floats, n_frames, true);
}

static pfhip_status vad_forward_impl(pfhip_vad* v, const float* pcm, int n_samples, int is_final, float* probs,
                                     size_t cap_floats, int* n_frames, bool sil_only) {
  last_error().clear();
  if (!v || n_samples < 0 || (n_samples > 0 && !pcm) || !n_frames) return fail(PFHIP_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> lk(v->mu);
  HIP_TRY(hipSetDevice(v->device));
  hipStream_t s = v->stream;
  const int F = n_samples < 400 ? 0 : 1 + (n_samples - 400) / 160;
  const int T = (F + v->lfr_n - 1) / v->lfr_n;               // fsmn-vad.cpp:202
  *n_frames = T;
  if (T == 0) return PFHIP_OK;                                // fsmn-vad.cpp:245-247
  if ((size_t)T * (sil_only ? 1 : v->n_out) > cap_floats && probs) return fail(PFHIP_ERR_CAPACITY, "probs buffer too small");
  const int Tp = round_up(T, 128);
  HIP_TRY(v->pcm.ensure((size_t)n_samples * 4));
  HIP_TRY(v->fb.ensure((size_t)F * 80 * 4));
  HIP_TRY(v->feats.ensure((size_t)Tp * v->in1.Kp * 4));
  HIP_TRY(v->a.ensure((size_t)Tp * 256 * 4));
  HIP_TRY(v->b.ensure((size_t)Tp * 256 * 4));
  HIP_TRY(v->p.ensure((size_t)Tp * 128 * 4));
  HIP_TRY(v->f.ensure((size_t)Tp * 128 * 4));
  HIP_TRY(v->probs.ensure((size_t)T * v->n_out * 4));
  HIP_TRY(v->sil.ensure((size_t)T * 4));
  if (v->in1.Np > 256 || v->in2.Np > 256 || v->out1.Np > 256 || v->out2.Np > 256 || v->proj > 128)
    return fail(PFHIP_ERR_UNSUPPORTED, "FSMN-VAD layer wider than the workspace");
  HIP_TRY(hipMemcpyAsync(v->pcm.p, pcm, (size_t)n_samples * 4, hipMemcpyHostToDevice, s));
  {
    int64_t* h64 = reinterpret_cast<int64_t*>(v->h_pin);
    h64[0] = 0;
    int* hm = v->h_pin + 2;
    hm[0] = 0; hm[1] = F; hm[2] = F;
    HIP_TRY(hipMemcpyAsync(v->meta.p, v->h_pin, 32, hipMemcpyHostToDevice, s));
  }
  pfhip::FbankTables tb{v->ft.d_window, v->ft.d_tw, v->ft.d_mel_off, v->ft.d_mel_size, v->ft.d_mel_w, v->d_mean, v->d_istd};
  pfhip::launch_fbank_frames(v->pcm.f(), reinterpret_cast<int64_t*>(v->meta.p), v->meta.i() + 2, v->meta.i() + 4, F, tb,
                             v->fb.f(), s);
  pfhip::launch_lfr_cmvn(v->fb.f(), F, T, v->lfr_m, v->lfr_n, v->n_mels, v->d_mean, v->d_istd, v->feats.f(), v->in1.Kp, s);
  {
    pfhip::VadSeg* hs = reinterpret_cast<pfhip::VadSeg*>(v->h_pin + 16);
    *hs = pfhip::VadSeg{v->cache[v->cache_cur].f(), is_final ? nullptr : v->cache[v->cache_cur ^ 1].f(), 0, T};
    HIP_TRY(v->segs.ensure(sizeof(pfhip::VadSeg)));
    HIP_TRY(hipMemcpyAsync(v->segs.p, hs, sizeof(pfhip::VadSeg), hipMemcpyHostToDevice, s));
    vad_network(v, s, T, static_cast<const pfhip::VadSeg*>(v->segs.p), 1, T);
  }
  if (!is_final) v->cache_cur ^= 1;                             // fsmn-vad.cpp:129-134: caches kept only if not final
  if (probs && !sil_only) HIP_TRY(hipMemcpyAsync(probs, v->probs.p, (size_t)T * v->n_out * 4, hipMemcpyDeviceToHost, s));
  if (probs && sil_only)        // column 0 of the [T, n_out] score matrix
    HIP_TRY(hipMemcpyAsync(probs, v->sil.p, (size_t)T * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return PFHIP_OK;
}


// ---- FsmnVadOnline (onnxruntime/src/fsmn-vad-online.cpp): one object per connection -----------------------------------
// The online feature front end keeps three caches between calls: input_cache_ (samples after the last frame shift,
// :11-38), lfr_splice_cache_ (frames not yet consumed by a full LFR window, :48-70, here in HBM) and reserve_waveforms_
// (the samples that line up with the emitted rows, handed to the scorer for its dB computation, :45-68), plus the four
// network caches.  Host logic is transcribed statement by statement; arithmetic stays on the GPU (fbank, LFR, network).
struct pfhip_vad_stream {
  pfhip_vad* v = nullptr;
  std::vector<float> input_cache, reserve;
  int n_splice = 0;             // frames at the front of fb[fb_cur]
  Buf fb[2], cache[2];
  int fb_cur = 0, cache_cur = 0;
};

namespace {
constexpr int kVadMaxSamples = 64000;     // per call (4 s); the 2-pass server sends 9600, CutSplit one second
constexpr int kVadMaxFrames = 512;

pfhip_status vs_zero_caches(pfhip_vad_stream* vs, hipStream_t s) {
  pfhip_vad* v = vs->v;
  for (int i = 0; i < 2; ++i) HIP_TRY(hipMemsetAsync(vs->cache[i].p, 0, (size_t)v->layers * 19 * v->proj * 4, s));
  return PFHIP_OK;
}
}  // namespace

pfhip_status pfhip_vad_stream_create(pfhip_vad* v, pfhip_vad_stream** out) {
  last_error().clear();
  if (!v || !out) return fail(PFHIP_ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(v->mu);
  HIP_TRY(hipSetDevice(v->device));
  std::unique_ptr<pfhip_vad_stream> vs(new pfhip_vad_stream);
  vs->v = v;
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(vs->fb[i].ensure((size_t)kVadMaxFrames * 80 * 4));
    HIP_TRY(vs->cache[i].ensure((size_t)v->layers * 19 * v->proj * 4));
  }
  pfhip_status st = vs_zero_caches(vs.get(), v->stream);
  if (st) return st;
  HIP_TRY(hipStreamSynchronize(v->stream));
  ++v->live_streams;
  *out = vs.release();
  return PFHIP_OK;
}

void pfhip_vad_stream_destroy(pfhip_vad_stream* vs) {
  if (!vs) return;
  {
    std::lock_guard<std::mutex> lk(vs->v->mu);
    (void)hipSetDevice(vs->v->device);
    (void)hipStreamSynchronize(vs->v->stream);
    for (Buf* b : {&vs->fb[0], &vs->fb[1], &vs->cache[0], &vs->cache[1]}) b->release();
  }
  --vs->v->live_streams;
  delete vs;
}

// Reset() + ResetCache() (fsmn-vad-online.cpp:160-163, fsmn-vad-online.h:59-63)
pfhip_status pfhip_vad_stream_reset(pfhip_vad_stream* vs) {
  last_error().clear();
  if (!vs) return fail(PFHIP_ERR_ARG, "null handle");
  std::lock_guard<std::mutex> lk(vs->v->mu);
  HIP_TRY(hipSetDevice(vs->v->device));
  vs->input_cache.clear(); vs->reserve.clear(); vs->n_splice = 0;
  pfhip_status st = vs_zero_caches(vs, vs->v->stream);
  if (st) return st;
  HIP_TRY(hipStreamSynchronize(vs->v->stream));
  return PFHIP_OK;
}

}  // extern "C"

namespace {

// One connection's share of a (batched) FsmnVadOnline::Infer: the host part of ExtractFeats runs first and leaves a plan,
// the device work of all connections is then issued as a handful of batched launches.
struct VadCall {
  pfhip_vad_stream* vs; const float* pcm; int n_samples; bool fin;
  float* sil_prob; size_t cap_floats; int* n_frames; float* waves_out; size_t waves_cap; int* n_waves;
  // plan.  buf = this call's slice of the pinned sample staging: [reserve_waveforms_ | input_cache_ | new samples]
  float* buf = nullptr;
  size_t fb_off = 0, fb_len = 0;    // samples (relative to buf) whose frames are computed in this call
  size_t out_off = 0, out_len = 0;  // what the scorer gets
  size_t used = 0;
  int frame_number = 0, base = 0, n_rows = 0, Tin = 0, splice = 0;
  bool fresh = false, run_lfr = false;
  int row_off = 0;
};

// ExtractFeats (:40-88) on counters and sample ranges only; the one copy of the samples made here is the one into the
// staging buffer the device reads
pfhip_status vad_plan(VadCall& c) {
  pfhip_vad_stream* vs = c.vs;
  pfhip_vad* v = vs->v;
  const int fl = 400, fs = 160, m = v->lfr_m, n = v->lfr_n;
  const size_t r0 = vs->reserve.size(), ic = vs->input_cache.size();
  if (r0) std::memcpy(c.buf, vs->reserve.data(), r0 * 4);
  if (ic) std::memcpy(c.buf + r0, vs->input_cache.data(), ic * 4);
  if (c.n_samples) std::memcpy(c.buf + r0 + ic, c.pcm, (size_t)c.n_samples * 4);
  c.used = r0 + ic + (size_t)c.n_samples;
  const float* W = c.buf + r0;                              // `waves` = input_cache_ ++ new samples (:43-44)
  const int total = (int)(ic + (size_t)c.n_samples);
  c.frame_number = total >= fl ? (total - fl) / fs + 1 : 0;
  vs->input_cache.assign(W + (size_t)c.frame_number * fs, W + total);
  auto online_lfr = [&](int Tin) {                          // OnlineLfrCmvn (:90-133): counts only
    const int T_lrf = (int)std::ceil((Tin - (m - 1) / 2) / (float)n);
    int splice = T_lrf, n_out = 0;
    for (int i = 0; i < T_lrf; ++i) {
      if (m <= Tin - i * n) ++n_out;
      else if (c.fin) ++n_out;
      else { splice = i; break; }
    }
    c.splice = std::min(Tin - 1, splice * n);
    c.n_rows = n_out; c.Tin = Tin; c.run_lfr = true;
  };
  c.out_off = r0; c.out_len = (size_t)total;
  if (c.frame_number > 0) {
    const size_t wlen = (size_t)(c.frame_number - 1) * fs + fl;
    if (c.frame_number + vs->n_splice + (m - 1) / 2 > kVadMaxFrames) return fail(PFHIP_ERR_ARG, "too many frames in one online VAD call");
    c.fb_off = r0; c.fb_len = wlen;
    c.fresh = vs->n_splice == 0;
    c.base = c.fresh ? (m - 1) / 2 : vs->n_splice;
    const bool had_reserve = r0 > 0;
    c.out_off = 0; c.out_len = r0 + wlen;                   // reserve_waveforms_ goes in front of what the scorer sees
    const int n_splice = c.fresh ? c.base : vs->n_splice;   // (:48-52) fresh: (m-1)/2 copies of the first frame
    if (c.frame_number + n_splice >= m) {
      const int frame_from_waves = ((int)c.out_len - fl) / fs + 1;
      const int minus_frame = had_reserve ? 0 : (m - 1) / 2;
      online_lfr(n_splice + c.frame_number);
      const int reserve_frame_idx = std::abs(c.splice - minus_frame);
      vs->reserve.assign(c.buf + (size_t)reserve_frame_idx * fs, c.buf + (size_t)frame_from_waves * fs);
      c.out_len = (size_t)(frame_from_waves - 1) * fs + fl;
    } else {
      // (:65-69) the splice cache just grows; the reference runs the network on the raw 80-dim frames here (a latent bug
      // reachable only with < 4 frames in a call): no rows in this restatement
      vs->reserve.assign(c.buf + (fl - fs), c.buf + c.out_len);
    }
  } else if (c.fin) {                                       // (:71-83)
    if (r0) { c.out_off = 0; c.out_len = r0; }
    if (vs->n_splice > 0) online_lfr(vs->n_splice);
  }
  return PFHIP_OK;
}

pfhip_status vad_execute(pfhip_vad* v, std::vector<VadCall>& calls) {
  hipStream_t s = v->stream;
  const int B = (int)calls.size(), m = v->lfr_m, n = v->lfr_n;
  // ---- staging: [fbank meta | op lists | segment descriptors | samples of every connection | silence column] ---------------
  size_t cap_samples = 0, cap_rows = 0;
  for (VadCall& c : calls) {
    const size_t have = c.vs->reserve.size() + c.vs->input_cache.size() + (size_t)c.n_samples;
    cap_samples += have;
    cap_rows += have / 160 + (size_t)c.vs->n_splice + 2;
  }
  const size_t meta_bytes = ((size_t)B * 8 + (size_t)(B + 1) * 4 + (size_t)B * 4 + 63) & ~(size_t)63;
  const size_t ops_max = (size_t)(4 * B + 4) * 32;
  const size_t ctl_bytes = meta_bytes + ops_max + (((size_t)B * sizeof(pfhip::VadSeg) + 63) & ~(size_t)63);
  float* hp_f = v->h_wave(ctl_bytes / 4 + cap_samples + cap_rows + 64);
  if (!hp_f) return fail(PFHIP_ERR_HIP, "pinned staging allocation failed");
  char* hp = reinterpret_cast<char*>(hp_f);
  float* h_pcm = reinterpret_cast<float*>(hp + ctl_bytes);
  float* h_sil = h_pcm + cap_samples;
  {
    size_t so = 0;
    for (VadCall& c : calls) {
      c.buf = h_pcm + so;
      so += c.vs->reserve.size() + c.vs->input_cache.size() + (size_t)c.n_samples;
      pfhip_status st = vad_plan(c);
      if (st) return st;
    }
  }
  // ---- fbank of all new audio in one launch ---------------------------------------------------------------------------
  int total_frames = 0, U = 0, T = 0, max_T = 0;
  for (VadCall& c : calls) {
    if (c.frame_number > 0) { total_frames += c.frame_number; ++U; }
    c.row_off = T; T += c.n_rows; max_T = std::max(max_T, c.n_rows);
  }
  if ((size_t)T > cap_rows) return fail(PFHIP_ERR_UNSUPPORTED, "online VAD row bound exceeded");
  std::vector<pfhip::RowsCopyOp> to_fb, fresh, rotate;
  std::vector<pfhip::VadLfrOp> lfr;
  HIP_TRY(v->ops.ensure(ctl_bytes));
  char* dp = static_cast<char*>(v->ops.p);
  if (U > 0) {
    HIP_TRY(v->pcm.ensure((cap_samples + 1024) * 4));
    HIP_TRY(v->fbk.ensure((size_t)total_frames * 80 * 4));
    int64_t* h_soff = reinterpret_cast<int64_t*>(hp);
    int* h_foff = reinterpret_cast<int*>(hp + (size_t)U * 8);
    int* h_nf = h_foff + (U + 1);
    int fo = 0, u = 0;
    for (VadCall& c : calls) {
      if (c.frame_number <= 0) continue;
      h_soff[u] = (int64_t)((c.buf - h_pcm) + c.fb_off); h_foff[u] = fo; h_nf[u] = c.frame_number;
      float* fb = c.vs->fb[c.vs->fb_cur].f();
      to_fb.push_back(pfhip::RowsCopyOp{fb + (size_t)c.base * 80, v->fbk.f() + (size_t)fo * 80, 80, 80, c.frame_number, 80});
      if (c.fresh) fresh.push_back(pfhip::RowsCopyOp{fb, fb + (size_t)c.base * 80, 80, 0, c.base, 80});
      fo += c.frame_number; ++u;
    }
    h_foff[U] = fo;
    HIP_TRY(hipMemcpyAsync(v->pcm.p, h_pcm, cap_samples * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dp, hp, meta_bytes, hipMemcpyHostToDevice, s));
    pfhip::FbankTables tb{v->ft.d_window, v->ft.d_tw, v->ft.d_mel_off, v->ft.d_mel_size, v->ft.d_mel_w, v->d_mean, v->d_istd};
    pfhip::launch_fbank_frames_batch(v->pcm.f(), reinterpret_cast<const int64_t*>(dp), reinterpret_cast<const int*>(dp + (size_t)U * 8),
                                     reinterpret_cast<const int*>(dp + (size_t)U * 8 + (size_t)(U + 1) * 4), U, total_frames, tb,
                                     v->fbk.f(), s);
  }
  pfhip_status ws = vad_workspace(v, std::max(T, 1));
  if (ws) return ws;
  // ---- per-connection frame bookkeeping (splice caches) and the LFR rows, one launch per phase ---------------------------
  std::vector<pfhip::VadSeg> segs;
  for (VadCall& c : calls) {
    pfhip_vad_stream* vs = c.vs;
    if (c.frame_number > 0 && c.fresh) vs->n_splice = c.base;
    if (c.run_lfr) {
      const float* fb = vs->fb[vs->fb_cur].f();
      if (c.n_rows > 0) lfr.push_back(pfhip::VadLfrOp{fb, c.Tin, c.n_rows, c.row_off, 0});
      const int keep = c.Tin - c.splice;                    // lfr_splice_cache_ = frames[splice:]
      rotate.push_back(pfhip::RowsCopyOp{vs->fb[vs->fb_cur ^ 1].f(), fb + (size_t)c.splice * 80, 80, 80, keep, 80});
      vs->fb_cur ^= 1;
      vs->n_splice = keep;
    } else if (c.frame_number > 0) {
      vs->n_splice += c.frame_number;                       // (:65-69) the splice cache just grows
    }
    // (:84-87) a final call: Reset() + ResetCache() run inside ExtractFeats, BEFORE Forward (:143) — scored against
    // zeroed network caches, nothing carried over
    if (c.fin) {
      vs->input_cache.clear(); vs->reserve.clear(); vs->n_splice = 0;
      pfhip_status st = vs_zero_caches(vs, s);
      if (st) return st;
    }
    if (c.n_rows > 0)
      segs.push_back(pfhip::VadSeg{vs->cache[vs->cache_cur].f(), c.fin ? nullptr : vs->cache[vs->cache_cur ^ 1].f(), c.row_off, c.n_rows});
  }
  char* h_ops = hp + meta_bytes;
  char* d_ops = dp + meta_bytes;
  size_t off = 0;
  auto put = [&](const void* src, size_t nb) { const size_t at = off; if (nb) std::memcpy(h_ops + off, src, nb); off += nb; return at; };
  const size_t o_fb = put(to_fb.data(), to_fb.size() * sizeof(pfhip::RowsCopyOp));
  const size_t o_fr = put(fresh.data(), fresh.size() * sizeof(pfhip::RowsCopyOp));
  const size_t o_lfr = put(lfr.data(), lfr.size() * sizeof(pfhip::VadLfrOp));
  const size_t o_rot = put(rotate.data(), rotate.size() * sizeof(pfhip::RowsCopyOp));
  const size_t o_seg = put(segs.data(), segs.size() * sizeof(pfhip::VadSeg));
  if (off) HIP_TRY(hipMemcpyAsync(d_ops, h_ops, off, hipMemcpyHostToDevice, s));
  auto max_rows = [](const std::vector<pfhip::RowsCopyOp>& o) { int mx = 0; for (const auto& x : o) mx = std::max(mx, x.nrows); return mx; };
  pfhip::launch_rows_copy_batch(reinterpret_cast<const pfhip::RowsCopyOp*>(d_ops + o_fb), (int)to_fb.size(), max_rows(to_fb), s);
  pfhip::launch_rows_copy_batch(reinterpret_cast<const pfhip::RowsCopyOp*>(d_ops + o_fr), (int)fresh.size(), max_rows(fresh), s);
  pfhip::launch_lfr_cmvn_online_batch(reinterpret_cast<const pfhip::VadLfrOp*>(d_ops + o_lfr), (int)lfr.size(), max_T, m, n, v->n_mels,
                                      v->d_mean, v->d_istd, v->feats.f(), v->in1.Kp, s);
  pfhip::launch_rows_copy_batch(reinterpret_cast<const pfhip::RowsCopyOp*>(d_ops + o_rot), (int)rotate.size(), max_rows(rotate), s);
  if (T > 0) {
    vad_network(v, s, T, reinterpret_cast<const pfhip::VadSeg*>(d_ops + o_seg), (int)segs.size(), max_T);
    for (VadCall& c : calls)
      if (c.n_rows > 0 && !c.fin) c.vs->cache_cur ^= 1;
  }
  // ---- results: the silence column of all connections in one copy, handed out on the host --------------------------------------
  if (T > 0) HIP_TRY(hipMemcpyAsync(h_sil, v->sil.p, (size_t)T * 4, hipMemcpyDeviceToHost, s));
  for (VadCall& c : calls) {
    if (c.out_len > c.waves_cap && c.waves_out) return fail(PFHIP_ERR_CAPACITY, "waves_out too small");
    if (c.waves_out && c.out_len) std::memcpy(c.waves_out, c.buf + c.out_off, c.out_len * 4);
    *c.n_waves = (int)c.out_len;
    if ((size_t)c.n_rows > c.cap_floats && c.sil_prob) return fail(PFHIP_ERR_CAPACITY, "sil_prob too small");
    *c.n_frames = c.n_rows;
  }
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  for (VadCall& c : calls)
    if (c.n_rows > 0 && c.sil_prob) std::memcpy(c.sil_prob, h_sil + c.row_off, (size_t)c.n_rows * 4);
  return PFHIP_OK;
}

}  // namespace

extern "C" {

pfhip_status pfhip_vad_stream_infer_batch(pfhip_vad_stream* const* streams, int n_streams, const float* const* pcm,
                                          const int* n_samples, const int* input_finished, float* const* sil_prob,
                                          const size_t* cap_floats, int* n_frames, float* const* waves_out,
                                          const size_t* waves_cap, int* n_waves) {
  last_error().clear();
  if (!streams || n_streams <= 0 || !pcm || !n_samples || !input_finished || !sil_prob || !cap_floats || !n_frames || !waves_out ||
      !waves_cap || !n_waves)
    return fail(PFHIP_ERR_ARG, "bad argument");
  pfhip_vad* v = streams[0] ? streams[0]->v : nullptr;
  if (!v) return fail(PFHIP_ERR_ARG, "null stream");
  std::vector<VadCall> calls(n_streams);
  for (int i = 0; i < n_streams; ++i) {
    if (!streams[i] || streams[i]->v != v) return fail(PFHIP_ERR_ARG, "streams of one batch must belong to one VAD model");
    for (int j = 0; j < i; ++j) if (streams[j] == streams[i]) return fail(PFHIP_ERR_ARG, "a stream appears twice in one batch");
    if (n_samples[i] < 0 || (n_samples[i] > 0 && !pcm[i])) return fail(PFHIP_ERR_ARG, "bad pcm buffer");
    if (n_samples[i] > kVadMaxSamples) return fail(PFHIP_ERR_ARG, "more than 64000 samples in one online VAD call");
    calls[i].vs = streams[i]; calls[i].pcm = pcm[i]; calls[i].n_samples = n_samples[i]; calls[i].fin = input_finished[i] != 0;
    calls[i].sil_prob = sil_prob[i]; calls[i].cap_floats = cap_floats[i]; calls[i].n_frames = &n_frames[i];
    calls[i].waves_out = waves_out[i]; calls[i].waves_cap = waves_cap[i]; calls[i].n_waves = &n_waves[i];
    n_frames[i] = 0; n_waves[i] = 0;
  }
  std::lock_guard<std::mutex> lk(v->mu);
  HIP_TRY(hipSetDevice(v->device));
  return vad_execute(v, calls);
}

pfhip_status pfhip_set_vad_stream_batching(pfhip_vad* v, int wait_us, int max_streams) {
  last_error().clear();
  if (!v || wait_us < 0 || max_streams < 1) return fail(PFHIP_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> ql(v->mq.mu);
  v->q_wait_us = wait_us;
  v->q_max = max_streams;
  return PFHIP_OK;
}

}  // extern "C"

// One FsmnVadOnline::Infer per websocket handler thread (funasrruntime.cpp:516-532): with wait_us > 0 the first caller to
// arrive leads, waits up to wait_us for the others and runs ONE pfhip_vad_stream_infer_batch for all of them.
struct VadReq : pfhip_detail::MergeReqBase {
  pfhip_vad_stream* vs; const float* pcm; int n; int fin; float* sil; size_t cap; int* nf; float* wo; size_t wcap; int* nw;
  pfhip_status st = PFHIP_OK; std::string err;
};

namespace {

void vad_run_requests(const std::vector<VadReq*>& reqs) {
  const int n = (int)reqs.size();
  std::vector<pfhip_vad_stream*> ss(n);
  std::vector<const float*> pcm(n);
  std::vector<int> ns(n), fin(n), nf(n), nw(n);
  std::vector<float*> sil(n), wo(n);
  std::vector<size_t> cap(n), wcap(n);
  for (int i = 0; i < n; ++i) {
    const VadReq& r = *reqs[i];
    ss[i] = r.vs; pcm[i] = r.pcm; ns[i] = r.n; fin[i] = r.fin; sil[i] = r.sil; cap[i] = r.cap; wo[i] = r.wo; wcap[i] = r.wcap;
  }
  const pfhip_status st = pfhip_vad_stream_infer_batch(ss.data(), n, pcm.data(), ns.data(), fin.data(), sil.data(), cap.data(),
                                                       nf.data(), wo.data(), wcap.data(), nw.data());
  const std::string err = pfhip_detail::last_error();
  for (int i = 0; i < n; ++i) { *reqs[i]->nf = nf[i]; *reqs[i]->nw = nw[i]; reqs[i]->st = st; reqs[i]->err = err; }
}

pfhip_status vad_infer_queued(pfhip_vad* v, VadReq& me) {
  int wait_us, cap;
  { std::lock_guard<std::mutex> l(v->mq.mu); wait_us = v->q_wait_us; cap = v->q_max; }
  v->mq.submit(
      me, wait_us,
      [&](const std::deque<VadReq*>& q) { return (int)q.size() >= std::min(cap, std::max(1, v->live_streams.load())); },
      [&](std::deque<VadReq*>& q, std::vector<VadReq*>& take) {
        std::deque<VadReq*> later;
        while (!q.empty() && (int)take.size() < cap) {
          VadReq* r = q.front();
          q.pop_front();
          bool dup = false;                       // two queued calls on ONE connection stay in order: the second waits
          for (VadReq* t : take) dup = dup || t->vs == r->vs;
          if (dup) later.push_back(r); else take.push_back(r);
        }
        for (auto it = later.rbegin(); it != later.rend(); ++it) q.push_front(*it);
      },
      [&](std::vector<VadReq*>& take) { vad_run_requests(take); });
  if (me.st != PFHIP_OK) pfhip_detail::last_error() = me.err;
  return me.st;
}

}  // namespace

extern "C" {

pfhip_status pfhip_vad_stream_infer(pfhip_vad_stream* vs, const float* pcm, int n_samples, int input_finished, float* sil_prob,
                                    size_t cap_floats, int* n_frames, float* waves_out, size_t waves_cap, int* n_waves) {
  if (!vs || !n_frames || !n_waves) { last_error().clear(); return fail(PFHIP_ERR_ARG, "bad argument"); }
  pfhip_vad* v = vs->v;
  bool queued;
  { std::lock_guard<std::mutex> ql(v->mq.mu); queued = v->q_wait_us > 0 && v->q_max > 1; }
  if (queued) {
    VadReq me;
    me.vs = vs; me.pcm = pcm; me.n = n_samples; me.fin = input_finished; me.sil = sil_prob; me.cap = cap_floats; me.nf = n_frames;
    me.wo = waves_out; me.wcap = waves_cap; me.nw = n_waves;
    return vad_infer_queued(v, me);
  }
  const float* p[1] = {pcm};
  float* sp[1] = {sil_prob};
  float* wo[1] = {waves_out};
  return pfhip_vad_stream_infer_batch(&vs, 1, p, &n_samples, &input_finished, sp, &cap_floats, n_frames, wo, &waves_cap, n_waves);
}

}  // extern "C"
