// FSMN-VAD forward on MI355X — the C ABI's pfhip_vad_* family.
// Replaces FsmnVad::FbankKaldi + LfrCmvn + Forward (onnxruntime/src/fsmn-vad.cpp:137-152, 198-238, 72-135): PCM ->
// frame scores [T, 248] with the four [128 x 19] FSMN caches carried in HBM.  The network is causal, so a whole
// file goes through in ONE pass of ~12 launches instead of the reference's 1-s slices (600 Runs per 10 minutes,
// onnxruntime/src/audio.cpp:1183-1196); slice-wise calls give the same scores because the caches carry over.
// Linear layers run on the same fp32 MFMA GEMM kernels as the ASR model; their odd widths (140, 250, 248) are
// zero-padded once at load ([N up to 128][K up to 32]) so that pad outputs are exact zeros.
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
  Buf pcm, fb, feats, a, b, p, f, probs, meta, cache[2];
  int cache_cur = 0;
  int* h_pin = nullptr;
  // pinned staging for the online path's PCM (grown on demand); h_wave(n) returns a buffer of at least n floats
  float* h_wave_buf = nullptr; size_t h_wave_cap = 0;
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
  for (Buf* b : {&v->pcm, &v->fb, &v->feats, &v->a, &v->b, &v->p, &v->f, &v->probs, &v->meta, &v->cache[0], &v->cache[1]}) b->release();
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

// The FSMN-VAD network on the T rows waiting in v->feats (caller holds v->mu and has sized the workspace); scores in v->probs.
static void vad_network(pfhip_vad* v, hipStream_t s, int T, const float* cin, float* cout) {
  lin_gemm(s, v->in1, v->feats.f(), v->in1.Kp, v->a.f(), 256, T, false);
  lin_gemm(s, v->in2, v->a.f(), 256, v->b.f(), 256, T, true);
  for (int i = 0; i < v->layers; ++i) {
    lin_gemm(s, v->blk_linear[i], v->b.f(), 256, v->p.f(), 128, T, false);
    pfhip::launch_fsmn_causal20(v->p.f(), 128, v->fsmn_w[i], cin + (size_t)i * 19 * v->proj,
                                cout ? cout + (size_t)i * 19 * v->proj : nullptr, v->f.f(), 128, T, v->proj, s);
    lin_gemm(s, v->blk_affine[i], v->f.f(), 128, v->b.f(), 256, T, true);
  }
  lin_gemm(s, v->out1, v->b.f(), 256, v->a.f(), 256, T, false);
  lin_gemm(s, v->out2, v->a.f(), 256, v->b.f(), 256, T, false);
  pfhip::launch_softmax_rows(v->b.f(), 256, T, v->n_out, v->probs.f(), s);
}

static pfhip_status vad_workspace(pfhip_vad* v, int T) {
  const int Tp = round_up(T, 128);
  HIP_TRY(v->feats.ensure((size_t)Tp * v->in1.Kp * 4));
  HIP_TRY(v->a.ensure((size_t)Tp * 256 * 4));
  HIP_TRY(v->b.ensure((size_t)Tp * 256 * 4));
  HIP_TRY(v->p.ensure((size_t)Tp * 128 * 4));
  HIP_TRY(v->f.ensure((size_t)Tp * 128 * 4));
  HIP_TRY(v->probs.ensure((size_t)T * v->n_out * 4));
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
    const float* cin = v->cache[v->cache_cur].f();
    float* cout = v->cache[v->cache_cur ^ 1].f();
    vad_network(v, s, T, cin, is_final ? nullptr : cout);
  }
  if (!is_final) v->cache_cur ^= 1;                             // fsmn-vad.cpp:129-134: caches kept only if not final
  if (probs && !sil_only) HIP_TRY(hipMemcpyAsync(probs, v->probs.p, (size_t)T * v->n_out * 4, hipMemcpyDeviceToHost, s));
  if (probs && sil_only)        // column 0 of the [T, n_out] score matrix
    HIP_TRY(hipMemcpy2DAsync(probs, 4, v->probs.p, (size_t)v->n_out * 4, 4, T, hipMemcpyDeviceToHost, s));
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

pfhip_status pfhip_vad_stream_infer(pfhip_vad_stream* vs, const float* pcm, int n_samples, int input_finished, float* sil_prob,
                                    size_t cap_floats, int* n_frames, float* waves_out, size_t waves_cap, int* n_waves) {
  last_error().clear();
  if (!vs || n_samples < 0 || (n_samples > 0 && !pcm) || !n_frames || !n_waves) return fail(PFHIP_ERR_ARG, "bad argument");
  if (n_samples > kVadMaxSamples) return fail(PFHIP_ERR_ARG, "more than 64000 samples in one online VAD call");
  pfhip_vad* v = vs->v;
  std::lock_guard<std::mutex> lk(v->mu);
  HIP_TRY(hipSetDevice(v->device));
  hipStream_t s = v->stream;
  const bool fin = input_finished != 0;
  const int fl = 400, fs = 160, m = v->lfr_m, n = v->lfr_n;
  *n_frames = 0; *n_waves = 0;
  // ---- FbankKaldi (:11-38): prepend input_cache_, keep what follows the last frame shift ------------------------------
  std::vector<float> waves(vs->input_cache);
  waves.insert(waves.end(), pcm, pcm + n_samples);
  const int total = (int)waves.size();
  int frame_number = total >= fl ? (total - fl) / fs + 1 : 0;
  vs->input_cache.assign(waves.begin() + (size_t)frame_number * fs, waves.end());
  int n_rows = 0, T = 0;
  float* fb = vs->fb[vs->fb_cur].f();
  auto online_lfr = [&](int Tin) -> pfhip_status {          // OnlineLfrCmvn (:90-133) over the Tin frames at the front of fb
    const int T_lrf = (int)std::ceil((Tin - (m - 1) / 2) / (float)n);
    int splice = T_lrf, n_out = 0;
    for (int i = 0; i < T_lrf; ++i) {
      if (m <= Tin - i * n) ++n_out;
      else if (fin) ++n_out;
      else { splice = i; break; }
    }
    splice = std::min(Tin - 1, splice * n);
    pfhip_status ws = vad_workspace(v, std::max(n_out, 1));
    if (ws) return ws;
    pfhip::launch_lfr_cmvn_online(fb, Tin, n_out, m, n, v->n_mels, v->d_mean, v->d_istd, v->feats.f(), v->in1.Kp, s);
    const int keep = Tin - splice;                          // lfr_splice_cache_ = frames[splice:]
    HIP_TRY(hipMemcpyAsync(vs->fb[vs->fb_cur ^ 1].p, fb + (size_t)splice * 80, (size_t)keep * 80 * 4, hipMemcpyDeviceToDevice, s));
    vs->fb_cur ^= 1;
    vs->n_splice = keep;
    n_rows = n_out;
    T = splice;                                             // lfr_splice_frame_idxs (returned by the reference)
    return PFHIP_OK;
  };
  if (frame_number > 0) {
    waves.resize((size_t)(frame_number - 1) * fs + fl);
    if (frame_number + vs->n_splice + (m - 1) / 2 > kVadMaxFrames) return fail(PFHIP_ERR_ARG, "too many frames in one online VAD call");
    const bool fresh = vs->n_splice == 0;
    const int base = fresh ? (m - 1) / 2 : vs->n_splice;
    HIP_TRY(v->pcm.ensure(waves.size() * 4));
    if (!v->h_wave(waves.size())) return fail(PFHIP_ERR_HIP, "pinned staging allocation failed");
    std::memcpy(v->h_wave(waves.size()), waves.data(), waves.size() * 4);
    HIP_TRY(hipMemcpyAsync(v->pcm.p, v->h_wave(0), waves.size() * 4, hipMemcpyHostToDevice, s));
    {
      int64_t* h64 = reinterpret_cast<int64_t*>(v->h_pin);
      h64[0] = 0;
      int* hm = v->h_pin + 2;
      hm[0] = 0; hm[1] = frame_number; hm[2] = frame_number;
      HIP_TRY(hipMemcpyAsync(v->meta.p, v->h_pin, 32, hipMemcpyHostToDevice, s));
    }
    pfhip::FbankTables tb{v->ft.d_window, v->ft.d_tw, v->ft.d_mel_off, v->ft.d_mel_size, v->ft.d_mel_w, v->d_mean, v->d_istd};
    pfhip::launch_fbank_frames(v->pcm.f(), reinterpret_cast<int64_t*>(v->meta.p), v->meta.i() + 2, v->meta.i() + 4, frame_number,
                               tb, fb + (size_t)base * 80, s);
    // cache deal & online lfr, cmvn (:44-70)
    const bool had_reserve = !vs->reserve.empty();
    if (had_reserve) waves.insert(waves.begin(), vs->reserve.begin(), vs->reserve.end());
    if (fresh) {                                            // lfr_splice_cache_ = (m-1)/2 copies of the first frame (:48-52)
      for (int i = 0; i < base; ++i)
        HIP_TRY(hipMemcpyAsync(fb + (size_t)i * 80, fb + (size_t)base * 80, 80 * 4, hipMemcpyDeviceToDevice, s));
      vs->n_splice = base;
    }
    if (frame_number + vs->n_splice >= m) {
      const int frame_from_waves = ((int)waves.size() - fl) / fs + 1;
      const int minus_frame = had_reserve ? 0 : (m - 1) / 2;
      pfhip_status st = online_lfr(vs->n_splice + frame_number);
      if (st) return st;
      const int reserve_frame_idx = std::abs(T - minus_frame);
      vs->reserve.assign(waves.begin() + (size_t)reserve_frame_idx * fs, waves.begin() + (size_t)frame_from_waves * fs);
      waves.resize((size_t)(frame_from_waves - 1) * fs + fl);
    } else {
      // (:65-69) the splice cache just grows; the reference runs the network on the raw 80-dim frames here (a latent bug
      // reachable only with < 4 frames in a call): no rows in this restatement
      vs->reserve.assign(waves.begin() + (fl - fs), waves.end());
      vs->n_splice += frame_number;
    }
  } else if (fin) {                                         // (:71-83)
    if (!vs->reserve.empty()) waves = vs->reserve;
    if (vs->n_splice > 0) {
      pfhip_status st = online_lfr(vs->n_splice);
      if (st) return st;
    }
  }
  const float* cin = vs->cache[vs->cache_cur].f();
  if (fin) {
    // (:84-87) Reset() + ResetCache() run inside ExtractFeats, BEFORE Forward (:143): the last call of a stream is scored
    // against zeroed network caches, and nothing is carried over
    vs->input_cache.clear(); vs->reserve.clear(); vs->n_splice = 0;
    pfhip_status st = vs_zero_caches(vs, s);
    if (st) return st;
  }
  if ((size_t)waves.size() > waves_cap && waves_out) return fail(PFHIP_ERR_CAPACITY, "waves_out too small");
  if (waves_out) std::memcpy(waves_out, waves.data(), waves.size() * 4);
  *n_waves = (int)waves.size();
  if (n_rows == 0) { HIP_TRY(hipStreamSynchronize(s)); return PFHIP_OK; }
  if ((size_t)n_rows > cap_floats && sil_prob) return fail(PFHIP_ERR_CAPACITY, "sil_prob too small");
  vad_network(v, s, n_rows, cin, fin ? nullptr : vs->cache[vs->cache_cur ^ 1].f());
  if (!fin) vs->cache_cur ^= 1;
  if (sil_prob) HIP_TRY(hipMemcpy2DAsync(sil_prob, 4, v->probs.p, (size_t)v->n_out * 4, 4, n_rows, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  *n_frames = n_rows;
  return PFHIP_OK;
}

}  // extern "C"
