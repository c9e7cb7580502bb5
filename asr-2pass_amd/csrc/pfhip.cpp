// C-ABI implementation (include/pfhip.h): model container + launch sequence of the offline
// Paraformer forward on one MI355X.  Layout in HBM:
//   * all utterances of a batch are PACKED row-major with no padding rows: utterance b owns LFR rows
//     [row_off[b], row_off[b]+T_b) of every [M, *] activation matrix (M = sum T_b); the reference's
//     GPU flavour zero-pads to Tmax instead (onnxruntime/src/paraformer-torch.cpp:342-347);
//   * decoder-side matrices are packed the same way over the CIF-fired tokens (ML = sum fires);
//   * weights stay in the blob's torch [out,in] layout = the K-contiguous B operand of the GEMM.
#include "../../include/pfhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "internal.h"
#include "json_min.h"

namespace pfhip_detail {
std::atomic<uint64_t>& buf_epoch() {
  static std::atomic<uint64_t> e{1};
  return e;
}
std::string& last_error() {
  thread_local std::string e;
  return e;
}
}  // namespace pfhip_detail

namespace {
using namespace pfhip_detail;
#define g_err (pfhip_detail::last_error())

// ---- front-end tables, computed exactly like knf does on the host --------------------------------
void build_mel(int num_bins, float sample_freq, std::vector<int>& off, std::vector<int>& size,
               std::vector<float>& w) {
  // knf mel-computations.cc:107-196 with vtln_warp = 1, low 20 Hz, high = Nyquist (mel-computations.h:33-37)
  auto mel_scale = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };
  const int padded = 512, num_fft_bins = padded / 2;
  const float nyquist = 0.5f * sample_freq;
  const float low_freq = 20.f, high_freq = nyquist + 0.f;
  const float fft_bin_width = sample_freq / padded;
  const float mel_low = mel_scale(low_freq), mel_high = mel_scale(high_freq);
  const float delta = (mel_high - mel_low) / (num_bins + 1);
  off.assign(num_bins, 0); size.assign(num_bins, 0); w.assign((size_t)num_bins * pfhip::kMelW, 0.f);
  for (int bin = 0; bin < num_bins; ++bin) {
    const float left = mel_low + bin * delta, center = mel_low + (bin + 1) * delta,
                right = mel_low + (bin + 2) * delta;
    int first = -1, last = -1;
    std::vector<float> tb(num_fft_bins, 0.f);
    for (int i = 0; i < num_fft_bins; ++i) {
      const float freq = fft_bin_width * i;
      const float mel = mel_scale(freq);
      if (mel > left && mel < right) {
        float weight;
        if (mel <= center) weight = (mel - left) / (center - left);
        else weight = (right - mel) / (right - center);
        tb[i] = weight;
        if (first == -1) first = i;
        last = i;
      }
    }
    off[bin] = first;
    size[bin] = last + 1 - first;
    for (int k = 0; k < size[bin] && k < pfhip::kMelW; ++k) w[(size_t)bin * pfhip::kMelW + k] = tb[first + k];
  }
}

template <typename T>
pfhip_status upload(T** dst, const std::vector<T>& v) {
  HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(v.size(), 1) * sizeof(T)));
  HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return PFHIP_OK;
}

}  // namespace

namespace pfhip_detail {
// knf tables exactly as the reference's host code computes them (feature-window.cc:33-42,
// mel-computations.cc:107-196) + the FFT twiddles of the device kernel.
pfhip_status build_frontend_tables(int n_mels, int sample_rate, FrontendTables* ft) {
  std::vector<float> win(400);
  const double a = 2.0 * M_PI / (400 - 1);
  for (int i = 0; i < 400; ++i) win[i] = (float)(0.54 - 0.46 * cos(a * (double)i));
  std::vector<double> tw(512);
  for (int k = 0; k < 256; ++k) { tw[2 * k] = cos(2.0 * M_PI * k / 512.0); tw[2 * k + 1] = -sin(2.0 * M_PI * k / 512.0); }
  std::vector<int> moff, msz; std::vector<float> mw;
  build_mel(n_mels, (float)sample_rate, moff, msz, mw);
  for (int b = 0; b < n_mels; ++b)
    if (msz[b] > pfhip::kMelW) return fail(PFHIP_ERR_UNSUPPORTED, "mel triangle wider than kMelW");
  pfhip_status st;
  if ((st = upload(&ft->d_window, win)) || (st = upload(&ft->d_tw, tw)) || (st = upload(&ft->d_mel_off, moff)) ||
      (st = upload(&ft->d_mel_size, msz)) || (st = upload(&ft->d_mel_w, mw)))
    return st;
  return PFHIP_OK;
}
}  // namespace pfhip_detail

namespace {
using namespace pfhip_detail;

pfhip_status create_streams(pfhip_model* m);

pfhip_status build_model(const void* blob, size_t blob_bytes, const char* manifest_json, int device,
                         pfhip_model** out) {
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

  std::unique_ptr<pfhip_model> m(new pfhip_model);
  m->device = device;
  Config& c = m->cfg;
  c.d_model = (int)jc->number("d_model", c.d_model);
  c.n_head = (int)jc->number("n_head", c.n_head);
  c.ffn = (int)jc->number("ffn", c.ffn);
  c.enc_layers = (int)jc->number("enc_layers", c.enc_layers);
  c.dec_layers = (int)jc->number("dec_layers", c.dec_layers);
  c.dec_ffn = (int)jc->number("dec_ffn", c.dec_ffn);
  c.kernel = (int)jc->number("kernel", c.kernel);
  c.vocab = (int)jc->number("vocab", c.vocab);
  c.n_mels = (int)jc->number("n_mels", c.n_mels);
  c.lfr_m = (int)jc->number("lfr_m", c.lfr_m);
  c.lfr_n = (int)jc->number("lfr_n", c.lfr_n);
  c.sample_rate = (int)jc->number("fs", c.sample_rate);           // frontend_conf.fs (paraformer.cpp:191)
  if (c.sample_rate != 16000) return fail(PFHIP_ERR_UNSUPPORTED, "the front end is built for 16 kHz models (frontend_conf.fs)");
  c.pred_residual = (int)jc->number("pred_residual", 0);
  c.contextual = (int)jc->number("contextual", 0);
  c.timestamp = (int)jc->number("timestamp", 0);
  c.smooth_factor2 = (float)jc->number("smooth_factor2", c.smooth_factor2);
  c.noise_threshold2 = (float)jc->number("noise_threshold2", c.noise_threshold2);
  if (c.timestamp && (int)jc->number("upsample_times", 3) != 3)
    return fail(PFHIP_ERR_UNSUPPORTED, "timestamp head: upsample_times must be 3");
  c.cif_threshold = (float)jc->number("cif_threshold", c.cif_threshold);
  c.tail_threshold = (float)jc->number("tail_threshold", c.tail_threshold);
  c.smooth_factor = (float)jc->number("smooth_factor", c.smooth_factor);
  c.noise_threshold = (float)jc->number("noise_threshold", c.noise_threshold);

  // what the gfx950 kernels are specialised for
  if (c.d_model % 128 || c.d_model / c.n_head != pfhip::kHeadDim)
    return fail(PFHIP_ERR_UNSUPPORTED, "attention kernel needs d_model/n_head == 128");
  if (c.d_model > 1024) return fail(PFHIP_ERR_UNSUPPORTED, "d_model > 1024");
  if (c.kernel != 11) return fail(PFHIP_ERR_UNSUPPORTED, "FSMN kernel size must be 11");
  if (c.n_mels != 80 || c.lfr_m != 7 || c.lfr_n != 6) return fail(PFHIP_ERR_UNSUPPORTED, "front end needs 80 mels, LFR 7/6");
  if (c.ffn % 128 || c.dec_ffn % 128 || c.ffn > 2048 || c.dec_ffn > 2048)
    return fail(PFHIP_ERR_UNSUPPORTED, "ffn width must be a multiple of 128 and <= 2048");
  if (c.enc_layers < 1 || c.dec_layers < 0) return fail(PFHIP_ERR_FORMAT, "bad layer counts");
  m->feat_dim = c.n_mels * c.lfr_m;
  m->feat_pad = round_up(m->feat_dim, pfhip::kTileK);
  m->vocab_pad = round_up(c.vocab, pfhip::kTileN);

  // ---- weights: upload the blob once; GEMM N-padding reads run into the slack at the end ----------
  const size_t slack = (size_t)pfhip::kTileN * 2048 * sizeof(float);
  HIP_TRY(hipMalloc((void**)&m->d_blob, blob_bytes + slack));
  HIP_TRY(hipMemset(m->d_blob, 0, blob_bytes + slack));
  HIP_TRY(hipMemcpy(m->d_blob, blob, blob_bytes, hipMemcpyHostToDevice));
  const float* hb = static_cast<const float*>(blob);
  for (const auto& kv : jt->obj) {
    const pfhip::JValue* sh = kv.second.get("shape");
    const pfhip::JValue* of = kv.second.get("offset");
    if (!sh || !of || sh->kind != pfhip::JValue::ARR) return fail(PFHIP_ERR_FORMAT, "tensor " + kv.first + ": shape/offset");
    Tensor tt;
    tt.n = 1;
    for (const auto& d : sh->arr) { tt.shape.push_back((int)d.num); tt.n *= (size_t)d.num; }
    const size_t off = (size_t)of->num;
    if (off % 16 || off + tt.n * 4 > blob_bytes) return fail(PFHIP_ERR_FORMAT, "tensor " + kv.first + ": out of blob");
    tt.d = m->d_blob + off / 4;
    tt.h = hb + off / 4;
    m->t.emplace(kv.first, std::move(tt));
  }
  // fp16 two-plane GEMM (gemm_x3.hip): one power-of-two scale per weight matrix, from its largest magnitude
  auto reg_scale = [&](const void* dptr, const float* host, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(host[i]));
    m->wscale[dptr] = pfhip::best_w_scale(mx);
  };
  for (const auto& kv : m->t)
    if (kv.second.shape.size() >= 2) reg_scale(kv.second.d, kv.second.h, kv.second.n);
  // required tensors / shapes
  auto need = [&](const std::string& n, std::vector<int> shape) -> bool {
    auto it = m->t.find(n);
    if (it == m->t.end()) { g_err = "missing tensor " + n; return false; }
    if (it->second.shape != shape) { g_err = "tensor " + n + " has unexpected shape"; return false; }
    return true;
  };
  const int d = c.d_model;
  bool ok = need("cmvn.mean", {m->feat_dim}) && need("cmvn.istd", {m->feat_dim});
  for (int i = 0; ok && i < c.enc_layers; ++i) {
    const std::string p = "enc." + std::to_string(i) + ".";
    const int in = i == 0 ? m->feat_dim : d;
    ok = need(p + "norm1.g", {in}) && need(p + "norm1.b", {in}) && need(p + "qkv.w", {3 * d, in}) &&
         need(p + "qkv.b", {3 * d}) && need(p + "fsmn.w", {d, c.kernel}) && need(p + "out.w", {d, d}) &&
         need(p + "out.b", {d}) && need(p + "norm2.g", {d}) && need(p + "norm2.b", {d}) &&
         need(p + "ffn1.w", {c.ffn, d}) && need(p + "ffn1.b", {c.ffn}) && need(p + "ffn2.w", {d, c.ffn}) &&
         need(p + "ffn2.b", {d});
  }
  ok = ok && need("enc.after_norm.g", {d}) && need("enc.after_norm.b", {d}) && need("pred.conv.w", {d, d, 3}) &&
       need("pred.conv.b", {d}) && need("pred.out.w", {1, d}) && need("pred.out.b", {1});
  auto need_ffn = [&](const std::string& p) {
    return need(p + "norm1.g", {d}) && need(p + "norm1.b", {d}) && need(p + "ffn1.w", {c.dec_ffn, d}) &&
           need(p + "ffn1.b", {c.dec_ffn}) && need(p + "ffn_norm.g", {c.dec_ffn}) &&
           need(p + "ffn_norm.b", {c.dec_ffn}) && need(p + "ffn2.w", {d, c.dec_ffn});
  };
  for (int i = 0; ok && i < c.dec_layers; ++i) {
    const std::string p = "dec." + std::to_string(i) + ".";
    ok = need_ffn(p) && need(p + "norm2.g", {d}) && need(p + "norm2.b", {d}) && need(p + "fsmn.w", {d, c.kernel}) &&
         need(p + "norm3.g", {d}) && need(p + "norm3.b", {d}) && need(p + "q.w", {d, d}) && need(p + "q.b", {d}) &&
         need(p + "kv.w", {2 * d, d}) && need(p + "kv.b", {2 * d}) && need(p + "out.w", {d, d}) && need(p + "out.b", {d});
  }
  ok = ok && need_ffn("dec3.") && need("dec.after_norm.g", {d}) && need("dec.after_norm.b", {d}) &&
       need("dec.out.w", {c.vocab, d}) && need("dec.out.b", {c.vocab});
  if (ok && c.contextual) {      // contextual (hotword) model: bias embedder + bias decoder
    if (c.dec_layers < 1) return fail(PFHIP_ERR_FORMAT, "contextual model needs a decoder");
    ok = need("bias.embed.w", {c.vocab, d}) && need("bias.lstm.w_ih", {4 * d, d}) && need("bias.lstm.w_hh", {4 * d, d}) &&
         need("bias.lstm.b_ih", {4 * d}) && need("bias.lstm.b_hh", {4 * d}) && need("bias.dec.norm3.g", {d}) &&
         need("bias.dec.norm3.b", {d}) && need("bias.dec.q.w", {d, d}) && need("bias.dec.q.b", {d}) &&
         need("bias.dec.kv.w", {2 * d, d}) && need("bias.dec.kv.b", {2 * d}) && need("bias.dec.out.w", {d, d}) &&
         need("bias.dec.out.b", {d}) && need("bias.out.w", {d, 2 * d});
  }
  if (ok && c.timestamp) {      // CifPredictorV3 upsampling head (SURVEY §8a row a6 producer, appendix A)
    ok = need("pred.up.w", {d, d, 3}) && need("pred.up.b", {d}) && need("pred.out2.w", {1, 2 * d}) && need("pred.out2.b", {1});
    for (const char* sfx : {"", "_r"})
      ok = ok && need(std::string("pred.blstm.w_ih") + sfx, {4 * d, d}) && need(std::string("pred.blstm.w_hh") + sfx, {4 * d, d}) &&
           need(std::string("pred.blstm.b_ih") + sfx, {4 * d}) && need(std::string("pred.blstm.b_hh") + sfx, {4 * d});
  }
  if (!ok) return PFHIP_ERR_FORMAT;

  // ---- repacks ---------------------------------------------------------------------------------------
  {
    const Tensor& w0 = m->W("enc.0.qkv.w");
    std::vector<float> p((size_t)3 * d * m->feat_pad, 0.f);
    for (int n = 0; n < 3 * d; ++n)
      std::memcpy(&p[(size_t)n * m->feat_pad], w0.h + (size_t)n * m->feat_dim, sizeof(float) * m->feat_dim);
    pfhip_status st = upload(&m->d_w0qkv, p);
    if (st) return st;
    reg_scale(m->d_w0qkv, p.data(), p.size());
    const Tensor& cw = m->W("pred.conv.w");
    std::vector<float> q((size_t)d * 3 * d);
    for (int n = 0; n < d; ++n)
      for (int ci = 0; ci < d; ++ci)
        for (int j = 0; j < 3; ++j) q[(size_t)n * 3 * d + (size_t)j * d + ci] = cw.h[((size_t)n * d + ci) * 3 + j];
    st = upload(&m->d_predconv, q);
    if (st) return st;
    reg_scale(m->d_predconv, q.data(), q.size());
    {
      // The static half of the fp16-domain guard (kernels.h LaunchCtx).  Everything the two-plane kernels multiply besides the
      // residual stream is the output of a Linear on a LayerNorm-ed row (q, k, v, the FFN hidden layer, the decoder's q / k / v,
      // the logits' input) or an average of such rows (attention context, CIF embeddings), and a LayerNorm-ed row has norm
      // sqrt(K) before gamma:   |LN(x) w_n + b_n| <= sqrt(K) * ||w_n o gamma||_2 + |b_n + w_n . beta|   for ANY input.
      // A model whose bound reaches 32768 (no trained model's does: weights of norm ~1 give a few hundred) runs on the exact
      // kernels altogether; the residual stream itself is checked per forward.
      double worst = 0.0;
      auto ln_linear = [&](const std::string& wn, const std::string& bn, const std::string& ln) {
        if (!m->t.count(wn) || !m->t.count(ln + ".g")) return;
        const Tensor& w = m->W(wn);
        const float* bb = !bn.empty() && m->t.count(bn) ? m->W(bn).h : nullptr;
        const float* g = m->W(ln + ".g").h; const float* be = m->W(ln + ".b").h;
        const int N = w.shape[0], K = w.shape[1];
        for (int n = 0; n < N; ++n) {
          double ss = 0.0, dot = bb ? bb[n] : 0.0;
          for (int k = 0; k < K; ++k) {
            const double wg = (double)w.h[(size_t)n * K + k] * (double)g[k];
            ss += wg * wg;
            dot += (double)w.h[(size_t)n * K + k] * (double)be[k];
          }
          worst = std::max(worst, std::sqrt((double)K) * std::sqrt(ss) + std::fabs(dot));
        }
      };
      for (int i = 0; i < c.enc_layers; ++i) {
        const std::string ep = "enc." + std::to_string(i) + ".";
        ln_linear(ep + "qkv.w", ep + "qkv.b", ep + "norm1");
        ln_linear(ep + "ffn1.w", ep + "ffn1.b", ep + "norm2");
      }
      {
        const float* g = m->W("enc.after_norm.g").h; const float* be = m->W("enc.after_norm.b").h;
        double gm = 0.0, bm = 0.0;
        for (int k = 0; k < d; ++k) { gm = std::max(gm, (double)std::fabs(g[k])); bm = std::max(bm, (double)std::fabs(be[k])); }
        worst = std::max(worst, std::sqrt((double)d) * gm + bm);                 // |enc| (keys / values source, CIF embeddings)
      }
      for (int i = 0; i < c.dec_layers; ++i) {
        const std::string dp = "dec." + std::to_string(i) + ".";
        ln_linear(dp + "ffn1.w", dp + "ffn1.b", dp + "norm1");
        ln_linear(dp + "ffn2.w", "", dp + "ffn_norm");
        ln_linear(dp + "q.w", dp + "q.b", dp + "norm3");
        ln_linear(dp + "kv.w", dp + "kv.b", "enc.after_norm");
      }
      ln_linear("dec3.ffn1.w", "dec3.ffn1.b", "dec3.norm1");
      ln_linear("dec3.ffn2.w", "", "dec3.ffn_norm");
      ln_linear("dec.out.w", "dec.out.b", "dec.after_norm");
      if (c.contextual) ln_linear("bias.dec.q.w", "bias.dec.q.b", "bias.dec.norm3");
      m->static_bound = worst;
      m->always_exact = !(worst < 32768.0);
    }
    if (d == 4 * pfhip::kTileN) {   // LN-on-load needs the residual stream to be exactly four 128-column tiles wide
      // W' = W * gamma[k], b' = b + W beta: LayerNorm's affine part folded into the GEMM that consumes it (offline path,
      // large batches: enqueue_locked).  Products in double, rounded once.
      auto foldk = [&](const std::string& wn, const std::string& bn, const std::string& ln, int N, int K, float* dw, float* db, float* ds) {
        const float* w = m->W(wn).h; const float* bb = bn.empty() ? nullptr : m->W(bn).h;
        const float* g = m->W(ln + ".g").h; const float* be = m->W(ln + ".b").h;
        for (int n = 0; n < N; ++n) {
          double acc = bb ? bb[n] : 0.0, cs = 0.0;
          for (int k = 0; k < K; ++k) {
            const float wf = (float)((double)w[(size_t)n * K + k] * (double)g[k]);
            dw[(size_t)n * K + k] = wf;
            cs += (double)wf;                         // column sum of the weights AS STORED: what the matrix cores will multiply
            acc += (double)w[(size_t)n * K + k] * (double)be[k];
          }
          db[n] = (float)acc;
          ds[n] = (float)cs;
        }
      };
      auto fold = [&](const std::string& wn, const std::string& bn, const std::string& ln, int N, float* dw, float* db, float* ds) {
        foldk(wn, bn, ln, N, d, dw, db, ds);
      };
      const int L = c.enc_layers;
      std::vector<float> wq((size_t)L * 3 * d * d + (size_t)pfhip::kTileN * d, 0.f), bq((size_t)L * 3 * d + pfhip::kTileN, 0.f);
      std::vector<float> wf((size_t)L * c.ffn * d + (size_t)pfhip::kTileN * d, 0.f), bf((size_t)L * c.ffn + pfhip::kTileN, 0.f);
      std::vector<float> sq(bq.size(), 0.f), sf(bf.size(), 0.f);
      for (int i = 0; i < L; ++i) {
        const std::string ep = "enc." + std::to_string(i) + ".";
        if (i > 0) fold(ep + "qkv.w", ep + "qkv.b", ep + "norm1", 3 * d, &wq[(size_t)i * 3 * d * d], &bq[(size_t)i * 3 * d], &sq[(size_t)i * 3 * d]);
        fold(ep + "ffn1.w", ep + "ffn1.b", ep + "norm2", c.ffn, &wf[(size_t)i * c.ffn * d], &bf[(size_t)i * c.ffn], &sf[(size_t)i * c.ffn]);
      }
      st = upload(&m->d_lnw_qkv, wq);
      if (!st) st = upload(&m->d_lnb_qkv, bq);
      if (!st) st = upload(&m->d_lnw_ffn1, wf);
      if (!st) st = upload(&m->d_lnb_ffn1, bf);
      if (!st) st = upload(&m->d_lns_qkv, sq);
      if (!st) st = upload(&m->d_lns_ffn1, sf);
      if (st) return st;
      for (int i = 0; i < L; ++i) {
        reg_scale(m->d_lnw_qkv + (size_t)i * 3 * d * d, &wq[(size_t)i * 3 * d * d], (size_t)3 * d * d);
        reg_scale(m->d_lnw_ffn1 + (size_t)i * c.ffn * d, &wf[(size_t)i * c.ffn * d], (size_t)c.ffn * d);
      }
      // The same four weights of every layer once more as fp16 plane images, pre-multiplied by their scale (gemm_p3.hip): on large
      // batches the encoder's GEMMs stage both operands by LDS-DMA — the activations arrive as plane images from the kernel that
      // produced them (enqueue_locked).  Same bytes as the fp32 copies (0.6 GB for Paraformer-large); PFHIP_PLANES=0 skips them.
      static const bool planes_on = [] { const char* e = getenv("PFHIP_PLANES"); return !(e && e[0] == '0'); }();
      if (planes_on && c.ffn % pfhip::kTileN == 0) {
        const size_t iq = 2 * pfhip::plane_image_bytes(3 * d, d), io = 2 * pfhip::plane_image_bytes(d, d);
        const size_t i1 = 2 * pfhip::plane_image_bytes(c.ffn, d), i2 = 2 * pfhip::plane_image_bytes(d, c.ffn);
        m->wp_off_out = iq; m->wp_off_ffn1 = iq + io; m->wp_off_ffn2 = iq + io + i1;
        const size_t per_layer = iq + io + i1 + i2;
        if (hipMalloc((void**)&m->d_wplanes, per_layer * (size_t)L) != hipSuccess) {
          // no room for the images (+70 % on the weight set): the fp32-operand kernels serve every batch size, nothing is lost
          (void)hipGetLastError();
          m->d_wplanes = nullptr;
        }
        for (int i = 0; i < L && m->d_wplanes; ++i) {
          const std::string ep = "enc." + std::to_string(i) + ".";
          unsigned char* base = m->d_wplanes + per_layer * (size_t)i;
          const float* wqkv = m->d_lnw_qkv + (size_t)i * 3 * d * d;
          const float* wff1 = m->d_lnw_ffn1 + (size_t)i * c.ffn * d;
          const float* wout = m->W(ep + "out.w").d;
          const float* wff2 = m->W(ep + "ffn2.w").d;
          if (i > 0) pfhip::launch_split_planes(wqkv, d, 3 * d, 3 * d, d, m->w_scale_of(wqkv), base, base + iq / 2, nullptr);
          pfhip::launch_split_planes(wout, d, d, d, d, m->w_scale_of(wout), base + m->wp_off_out, base + m->wp_off_out + io / 2, nullptr);
          pfhip::launch_split_planes(wff1, d, c.ffn, c.ffn, d, m->w_scale_of(wff1), base + m->wp_off_ffn1, base + m->wp_off_ffn1 + i1 / 2, nullptr);
          pfhip::launch_split_planes(wff2, c.ffn, d, d, c.ffn, m->w_scale_of(wff2), base + m->wp_off_ffn2, base + m->wp_off_ffn2 + i2 / 2, nullptr);
        }
        if (m->d_wplanes) {
          HIP_TRY(hipGetLastError());
          HIP_TRY(hipDeviceSynchronize());
          m->wp_layer_bytes = per_layer;
        }
      }
      if (!st && c.dec_ffn % pfhip::kTileN == 0) {          // decoder FFNs: layers 0..dec_layers-1 and dec3 (the last entry)
        const int DL = c.dec_layers + 1, f = c.dec_ffn;
        std::vector<float> w1((size_t)DL * f * d + (size_t)pfhip::kTileN * d, 0.f), b1((size_t)DL * f + pfhip::kTileN, 0.f), s1(b1.size(), 0.f);
        std::vector<float> w2((size_t)DL * d * f + (size_t)pfhip::kTileN * f, 0.f), b2((size_t)DL * d + pfhip::kTileN, 0.f), s2(b2.size(), 0.f);
        std::vector<float> w3((size_t)DL * d * d + (size_t)pfhip::kTileN * d, 0.f), b3((size_t)DL * d + pfhip::kTileN, 0.f), s3(b3.size(), 0.f);
        for (int i = 0; i < DL; ++i) {
          const std::string dp = i < c.dec_layers ? "dec." + std::to_string(i) + "." : std::string("dec3.");
          foldk(dp + "ffn1.w", dp + "ffn1.b", dp + "norm1", f, d, &w1[(size_t)i * f * d], &b1[(size_t)i * f], &s1[(size_t)i * f]);
          foldk(dp + "ffn2.w", "", dp + "ffn_norm", d, f, &w2[(size_t)i * d * f], &b2[(size_t)i * d], &s2[(size_t)i * d]);
          if (i < c.dec_layers) foldk(dp + "q.w", dp + "q.b", dp + "norm3", d, d, &w3[(size_t)i * d * d], &b3[(size_t)i * d], &s3[(size_t)i * d]);
        }
        st = upload(&m->d_dlnw3, w3);
        if (!st) st = upload(&m->d_dlnb3, b3);
        if (!st) st = upload(&m->d_dlns3, s3);
        if (st) return st;
        st = upload(&m->d_dlnw1, w1);
        if (!st) st = upload(&m->d_dlnb1, b1);
        if (!st) st = upload(&m->d_dlns1, s1);
        if (!st) st = upload(&m->d_dlnw2, w2);
        if (!st) st = upload(&m->d_dlnb2, b2);
        if (!st) st = upload(&m->d_dlns2, s2);
        if (st) return st;
        for (int i = 0; i < DL; ++i) {
          reg_scale(m->d_dlnw1 + (size_t)i * f * d, &w1[(size_t)i * f * d], (size_t)f * d);
          reg_scale(m->d_dlnw2 + (size_t)i * d * f, &w2[(size_t)i * d * f], (size_t)d * f);
          reg_scale(m->d_dlnw3 + (size_t)i * d * d, &w3[(size_t)i * d * d], (size_t)d * d);
        }
        // Round 4: the decoder's large weights as plane images too (gemm_p3.hip; +0.2 GB for Paraformer-large): FFN1' / FFN2' with
        // their LayerNorms folded in, the K/V projection of the encoder output and the output projection of the cross-attention
        if (m->wp_layer_bytes != 0) {
          const size_t j1 = 2 * pfhip::plane_image_bytes(f, d), j2 = 2 * pfhip::plane_image_bytes(d, f);
          const size_t jk = 2 * pfhip::plane_image_bytes(2 * d, d), jo = 2 * pfhip::plane_image_bytes(d, d);
          m->dwp_off_ffn2 = j1; m->dwp_off_kv = j1 + j2; m->dwp_off_out = j1 + j2 + jk;
          const size_t per = j1 + j2 + jk + jo;
          if (hipMalloc((void**)&m->d_dwplanes, per * (size_t)DL) != hipSuccess) { (void)hipGetLastError(); m->d_dwplanes = nullptr; }
          for (int i = 0; i < DL && m->d_dwplanes; ++i) {
            unsigned char* base = m->d_dwplanes + per * (size_t)i;
            const float* wf1 = m->d_dlnw1 + (size_t)i * f * d;
            const float* wf2 = m->d_dlnw2 + (size_t)i * d * f;
            pfhip::launch_split_planes(wf1, d, f, f, d, m->w_scale_of(wf1), base, base + j1 / 2, nullptr);
            pfhip::launch_split_planes(wf2, f, d, d, f, m->w_scale_of(wf2), base + m->dwp_off_ffn2, base + m->dwp_off_ffn2 + j2 / 2, nullptr);
            if (i < c.dec_layers) {
              const std::string dp = "dec." + std::to_string(i) + ".";
              const float* wkv = m->W(dp + "kv.w").d;
              const float* wo = m->W(dp + "out.w").d;
              pfhip::launch_split_planes(wkv, d, 2 * d, 2 * d, d, m->w_scale_of(wkv), base + m->dwp_off_kv, base + m->dwp_off_kv + jk / 2, nullptr);
              pfhip::launch_split_planes(wo, d, d, d, d, m->w_scale_of(wo), base + m->dwp_off_out, base + m->dwp_off_out + jo / 2, nullptr);
            }
          }
          if (m->d_dwplanes) {
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            m->dwp_layer_bytes = per;
          }
        }
      }
      if (st) return st;
    }
    if (c.dec_layers > 0) {     // streaming latency path: all layers' K/V projections of a window in one launch (stream.cpp)
      std::vector<float> kw((size_t)c.dec_layers * 2 * d * d + (size_t)pfhip::kTileN * d, 0.f), kb((size_t)c.dec_layers * 2 * d + pfhip::kTileN, 0.f);
      for (int i = 0; i < c.dec_layers; ++i) {
        const std::string dp = "dec." + std::to_string(i) + ".";
        std::memcpy(&kw[(size_t)i * 2 * d * d], m->W(dp + "kv.w").h, sizeof(float) * 2 * d * d);
        std::memcpy(&kb[(size_t)i * 2 * d], m->W(dp + "kv.b").h, sizeof(float) * 2 * d);
      }
      st = upload(&m->d_kv_all_w, kw);
      if (!st) st = upload(&m->d_kv_all_b, kb);
      if (st) return st;
      reg_scale(m->d_kv_all_w, kw.data(), kw.size());
    }
    std::vector<float> vb((size_t)m->vocab_pad, 0.f);
    std::memcpy(vb.data(), m->W("dec.out.b").h, sizeof(float) * c.vocab);
    st = upload(&m->d_vocab_bias, vb);
    if (st) return st;
  }
  if (c.timestamp) {
    // ConvTranspose1d(d, d, k = stride = 3): out[3t + j][co] = b[co] + sum_ci x[t][ci] * w[ci][co][j]  ==  one GEMM with
    // the weight laid out [j*d + co][ci]; its [T, 3d] result IS the [3T, d] upsampled sequence.
    const Tensor& uw = m->W("pred.up.w");
    std::vector<float> w2((size_t)3 * d * d), b2((size_t)3 * d);
    for (int ci = 0; ci < d; ++ci)
      for (int co = 0; co < d; ++co)
        for (int j = 0; j < 3; ++j) w2[((size_t)j * d + co) * d + ci] = uw.h[((size_t)ci * d + co) * 3 + j];
    for (int j = 0; j < 3; ++j) std::memcpy(&b2[(size_t)j * d], m->W("pred.up.b").h, sizeof(float) * d);
    pfhip_status st = upload(&m->d_up_w, w2);
    if (!st) st = upload(&m->d_up_b, b2);
    if (st) return st;
    reg_scale(m->d_up_w, w2.data(), w2.size());
    // both directions' input projections in one GEMM (N = 8d), b_ih + b_hh folded; recurrent weights [2][4d][d]
    std::vector<float> wih((size_t)8 * d * d), bih((size_t)8 * d), whh((size_t)8 * d * d);
    int dir = 0;
    for (const char* sfx : {"", "_r"}) {
      std::memcpy(&wih[(size_t)dir * 4 * d * d], m->W(std::string("pred.blstm.w_ih") + sfx).h, sizeof(float) * 4 * d * d);
      std::memcpy(&whh[(size_t)dir * 4 * d * d], m->W(std::string("pred.blstm.w_hh") + sfx).h, sizeof(float) * 4 * d * d);
      const float* bi = m->W(std::string("pred.blstm.b_ih") + sfx).h;
      const float* bh = m->W(std::string("pred.blstm.b_hh") + sfx).h;
      for (int k = 0; k < 4 * d; ++k) bih[(size_t)dir * 4 * d + k] = bi[k] + bh[k];
      ++dir;
    }
    if (!st) st = upload(&m->d_wih, wih);
    if (!st) st = upload(&m->d_bih, bih);
    if (!st) st = upload(&m->d_whh, whh);
    if (st) return st;
    reg_scale(m->d_wih, wih.data(), wih.size());
    m->out2_b = m->W("pred.out2.b").h[0];
  }
  // ---- front-end tables ------------------------------------------------------------------------------
  {
    pfhip_detail::FrontendTables ft;
    pfhip_status st = pfhip_detail::build_frontend_tables(c.n_mels, c.sample_rate, &ft);
    if (st) return st;
    m->d_window = ft.d_window; m->d_tw = ft.d_tw; m->d_mel_off = ft.d_mel_off; m->d_mel_size = ft.d_mel_size;
    m->d_mel_w = ft.d_mel_w;
    const int half = m->feat_dim / 2;
    std::vector<float> inv(half);
    // paraformer-online.cpp:247-252: float scale, exp() in double of a float argument, stored to float
    const float scale = m->feat_dim == 560 ? -0.0330119726594128f : (float)(-std::log(10000.0) / (half - 1));
    for (int i = 0; i < half; ++i) inv[i] = (float)exp((double)(i * scale));
    if ((st = upload(&m->d_inv_ts, inv))) return st;
  }
  {
    pfhip_status st = create_streams(m.get());
    if (st) return st;
  }
  for (auto& kv : m->t) kv.second.h = nullptr;
  *out = m.release();
  return PFHIP_OK;
}

// every device pointer of a model that is read-only after build_model: what an execution context borrows and only the owner frees
#define PFHIP_WEIGHT_PTRS(X)                                                                                                  \
  X(d_blob) X(d_w0qkv) X(d_predconv) X(d_vocab_bias) X(d_kv_all_w) X(d_kv_all_b) X(d_lnw_qkv) X(d_lnb_qkv) X(d_lnw_ffn1)     \
  X(d_lnb_ffn1) X(d_lns_qkv) X(d_lns_ffn1) X(d_dlnw1) X(d_dlnb1) X(d_dlns1) X(d_dlnw2) X(d_dlnb2) X(d_dlns2) X(d_dlnw3)        \
  X(d_dlnb3) X(d_dlns3) X(d_up_w) X(d_up_b) X(d_wih) X(d_bih) X(d_whh) X(d_window) X(d_tw) X(d_mel_off) X(d_mel_size)         \
  X(d_mel_w) X(d_inv_ts) X(d_wplanes) X(d_dwplanes)

pfhip_status create_streams(pfhip_model* m) {
  HIP_TRY(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&m->ev_enc_ready, hipEventDisableTiming));
  m->ev_kv.resize((size_t)m->cfg.dec_layers, nullptr);
  for (auto& e : m->ev_kv) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return PFHIP_OK;
}

// An execution context on `owner`'s device: the reference's decoder threads share ONE session (paraformer.cpp:35-41,541); here
// they share one weight set, and what a forward needs for itself — workspace, streams, events, the state of its last batch —
// is a context.  Costs a few KB until its first forward sizes the workspace.
pfhip_status build_context(pfhip_model* owner, pfhip_model** out) {
  HIP_TRY(hipSetDevice(owner->device));
  std::unique_ptr<pfhip_model> m(new pfhip_model);
  m->device = owner->device;
  m->cfg = owner->cfg;
  m->feat_dim = owner->feat_dim; m->feat_pad = owner->feat_pad; m->vocab_pad = owner->vocab_pad;
  m->t = owner->t;
  m->wscale = owner->wscale;
  m->out2_b = owner->out2_b;
  m->static_bound = owner->static_bound; m->always_exact = owner->always_exact;
  m->wp_layer_bytes = owner->wp_layer_bytes; m->wp_off_out = owner->wp_off_out; m->wp_off_ffn1 = owner->wp_off_ffn1;
  m->wp_off_ffn2 = owner->wp_off_ffn2;
  m->dwp_layer_bytes = owner->dwp_layer_bytes; m->dwp_off_ffn2 = owner->dwp_off_ffn2; m->dwp_off_kv = owner->dwp_off_kv;
  m->dwp_off_out = owner->dwp_off_out;
#define X(f) m->f = owner->f;
  PFHIP_WEIGHT_PTRS(X)
#undef X
  m->weights_of = owner;
  pfhip_status st = create_streams(m.get());
  if (st) { pfhip_destroy(m.release()); return st; }
  *out = m.release();
  return PFHIP_OK;
}

pfhip_status read_file(const char* path, std::vector<char>& out) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) return fail(PFHIP_ERR_ARG, std::string("cannot open ") + path);
  const std::streamsize n = f.tellg();
  f.seekg(0);
  out.resize((size_t)n);
  if (n && !f.read(out.data(), n)) return fail(PFHIP_ERR_ARG, std::string("cannot read ") + path);
  return PFHIP_OK;
}

// ---- the forward ---------------------------------------------------------------------------------------
// Sets the calling thread's launch context (kernels.h) for the kernels of one forward of `m`: its range-flag word and whether this
// is the re-run on the exact kernels.
struct ForwardCtx {
  pfhip::LaunchCtx saved;
  explicit ForwardCtx(pfhip_model* m) : saved(pfhip::launch_ctx()) {
    pfhip::launch_ctx().exact = m->exact_rerun || m->always_exact;
    pfhip::launch_ctx().range_flag = m->d_range_flag;
  }
  ~ForwardCtx() { pfhip::launch_ctx() = saved; }
};

pfhip_status enqueue_body(pfhip_model* m, const float* d_pcm, const int64_t* sample_off, const int* n_samples, int B, hipStream_t s,
                          bool feats_only);
pfhip_status enqueue_locked(pfhip_model* m, const float* d_pcm, const int64_t* sample_off, const int* n_samples,
                            int B, hipStream_t s, bool feats_only) {
  m->d_range_flag = nullptr;                          // until the metadata upload places it
  const pfhip::LaunchCtx saved = pfhip::launch_ctx();
  pfhip::launch_ctx().exact = m->exact_rerun || m->always_exact;
  const pfhip_status st = enqueue_body(m, d_pcm, sample_off, n_samples, B, s, feats_only);
  pfhip::launch_ctx() = saved;
  return st;
}

pfhip_status enqueue_body(pfhip_model* m, const float* d_pcm, const int64_t* sample_off, const int* n_samples,
                          int B, hipStream_t s, bool feats_only) {
  const Config& c = m->cfg;
  const int d = c.d_model, FD = m->feat_dim, FP = m->feat_pad;
  HIP_TRY(hipSetDevice(m->device));
  m->B = B; m->M = 0; m->ML = 0; m->maxT = 0; m->maxL = 0; m->have_logp = false; m->have_ts = false;
  m->T.assign(B, 0); m->row_off.assign(B, 0); m->n_fires.assign(B, 0); m->token_num.assign(B, 0); m->tok_off.assign(B, 0);
  std::vector<int> F(B), frame_off(B + 1, 0);
  for (int b = 0; b < B; ++b) {
    if (n_samples[b] < 0) return fail(PFHIP_ERR_ARG, "negative sample count");
    F[b] = n_samples[b] < 400 ? 0 : 1 + (n_samples[b] - 400) / 160;        // feature-window.cc:84-87
    m->T[b] = (F[b] + c.lfr_n - 1) / c.lfr_n;                               // paraformer.cpp:425
    m->row_off[b] = m->M;
    m->M += m->T[b];
    m->maxT = std::max(m->maxT, m->T[b]);
    frame_off[b + 1] = frame_off[b] + F[b];
  }
  const int M = m->M, total_frames = frame_off[B];
  if (M == 0) return PFHIP_OK;
  const int Mp = round_up(M, pfhip::kTileM);

  // ---- metadata: one pinned staging buffer, one H2D copy ---------------------------------------------
  const size_t n_ints = (size_t)2 * B /*sample_off as int64*/ + (B + 1) + 3 * (size_t)B + 2 * (size_t)M + 8;
  if (n_ints * 4 > m->h_meta_cap) {
    if (m->h_meta) HIP_TRY(hipHostFree(m->h_meta));
    m->h_meta = nullptr; m->h_meta_cap = 0;
    HIP_TRY(hipHostMalloc(&m->h_meta, n_ints * 4 * 2, hipHostMallocDefault));
    m->h_meta_cap = n_ints * 4 * 2;
  }
  HIP_TRY(m->meta.ensure(n_ints * 4));
  {
    int* hm = static_cast<int*>(m->h_meta);
    int* dm = m->meta.i();
    size_t o = 0;
    std::memcpy(hm + o, sample_off, sizeof(int64_t) * B); m->m_sample_off = reinterpret_cast<int64_t*>(dm + o); o += 2 * (size_t)B;
    std::memcpy(hm + o, frame_off.data(), 4 * (B + 1)); m->m_frame_off = dm + o; o += B + 1;
    std::memcpy(hm + o, F.data(), 4 * B); m->m_nframes = dm + o; o += B;
    std::memcpy(hm + o, m->row_off.data(), 4 * B); m->m_row_off = dm + o; o += B;
    std::memcpy(hm + o, m->T.data(), 4 * B); m->m_len = dm + o; o += B;
    if (o & 1) ++o;
    m->m_row_pos = dm + o;
    for (int b = 0; b < B; ++b) for (int t = 0; t < m->T[b]; ++t) hm[o + m->row_off[b] + t] = t;
    o += M;
    m->m_row_len = dm + o;
    for (int b = 0; b < B; ++b) for (int t = 0; t < m->T[b]; ++t) hm[o + m->row_off[b] + t] = m->T[b];
    o += M;
    // the forward's range flag (kernels.h LaunchCtx) is cleared by the same copy; a test may pre-raise it (pfhip_debug_poke)
    if (o & 1) ++o;
    hm[o] = m->debug_range_flag;
    m->debug_range_flag = 0;
    m->d_range_flag = dm + o;
    pfhip::launch_ctx().range_flag = m->d_range_flag;
    ++o;
    HIP_TRY(hipMemcpyAsync(dm, hm, o * 4, hipMemcpyHostToDevice, s));
  }
  // what a re-run on the exact kernels needs (fetch_locked)
  m->last_pcm = d_pcm;
  m->last_off.assign(sample_off, sample_off + B);
  m->last_n.assign(n_samples, n_samples + B);
  m->last_feats_only = feats_only;

  // ---- workspace ---------------------------------------------------------------------------------------
  HIP_TRY(m->feats.ensure((size_t)Mp * FD * 4));
  if (!feats_only) {
    HIP_TRY(m->x0.ensure((size_t)Mp * FP * 4));
    HIP_TRY(m->x.ensure((size_t)Mp * d * 4));
    HIP_TRY(m->y.ensure((size_t)Mp * FP * 4));
    HIP_TRY(m->qkv.ensure((size_t)Mp * 3 * d * 4));
    HIP_TRY(m->mem.ensure((size_t)Mp * d * 4));
    HIP_TRY(m->ctx.ensure((size_t)Mp * d * 4));
    HIP_TRY(m->hbuf.ensure((size_t)(Mp + B + 128) * std::max(c.ffn, d) * 4));
    HIP_TRY(m->enc.ensure((size_t)Mp * d * 4));
    HIP_TRY(m->alphas.ensure((size_t)Mp * 4));
    HIP_TRY(m->counts.ensure((size_t)2 * B * 4));
    if ((size_t)2 * B * 4 > m->h_counts_cap) {
      if (m->h_counts) HIP_TRY(hipHostFree(m->h_counts));
      m->h_counts = nullptr; m->h_counts_cap = 0;
      HIP_TRY(hipHostMalloc((void**)&m->h_counts, (size_t)2 * B * 4 * 2, hipHostMallocDefault));
      m->h_counts_cap = (size_t)2 * B * 4 * 2;
    }
  }

  // ---- a2+a3: fbank -> LFR -> CMVN -----------------------------------------------------------------------
  {
    double in_bytes = 0;
    for (int b = 0; b < B; ++b) in_bytes += 4.0 * n_samples[b];
    Scope sc(m, s, K_FBANK, 0, in_bytes + 4.0 * FD * M);
    pfhip::FbankTables tb{m->d_window, m->d_tw, m->d_mel_off, m->d_mel_size, m->d_mel_w,
                          m->W("cmvn.mean").d, m->W("cmvn.istd").d};
    pfhip::launch_fbank_lfr_cmvn(d_pcm, m->m_sample_off, m->m_frame_off, m->m_nframes, m->m_row_off, B,
                                 total_frames, tb, m->feats.f(), s);
  }
  if (feats_only) { HIP_TRY(hipGetLastError()); return PFHIP_OK; }

  // ---- a4: encoder -------------------------------------------------------------------------------------------
  {
    Scope sc(m, s, K_OTHER, 0, 4.0 * M * (FD + FP));
    pfhip::launch_embed(m->feats.f(), FD, m->x0.f(), FP, m->m_row_pos, M, m->d_inv_ts, sqrtf((float)d), s);
  }
  const float att_scale = 1.0f / sqrtf((float)pfhip::kHeadDim);
  double attn_pairs = 0;
  for (int b = 0; b < B; ++b) attn_pairs += (double)m->T[b] * m->T[b];
  float* x = m->x.f();
  // Large batches: the two LayerNorms of an encoder layer disappear into the GEMMs around them.  The GEMM that writes the
  // residual stream (out-projection, FFN2: N = 512 = four column tiles) leaves per-row statistics of each tile in its
  // epilogue; the GEMM that reads it (FFN1, the next layer's QKV) normalises its operand rows while staging them, with
  // gamma / beta folded into its weights (build_model).  One [M, 512] write + read and one launch per LayerNorm less.
  const bool fuse_ln = m->d_lnw_qkv != nullptr && pfhip::gemm_x6_ln_ok(M);
  const bool mem_in_x = pfhip::attention_fsmn_is_fused(m->maxT);
  if (fuse_ln) HIP_TRY(m->lnstats.ensure((size_t)Mp * 4 * 2 * 4));
  auto gemm_ln = [&](const float* A, const float* Wd, int N, float* Cd, int ldc, const float* bias, const float* R1, const float* R2,
                     bool relu, const float* ln_colsum, bool stats_out, int K) {
    Scope sc(m, s, K_GEMM, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    pfhip::launch_gemm_f32_x6_ln(A, K, Wd, K, Cd, ldc, bias, R1, d, R2, d, M, N, K, relu, ln_colsum ? m->lnstats.f() : nullptr, 4,
                                 ln_colsum, stats_out ? m->lnstats.f() : nullptr, s, m->w_scale_of(Wd));
  };
  // Large batches, second step: the four big GEMMs of a layer take BOTH operands as fp16 plane images staged by LDS-DMA
  // (gemm_p3.hip) — weights split once at load, activations written as planes by the kernel that produces them (the attention's
  // context, the residual stream out of the output projection and FFN2, FFN1's hidden activation).  The fp32 residual stream and
  // the QKV rows stay fp32 (residual adds, the attention's own staging).  Same arithmetic as the in-loop split of gemm_x3.hip.
  struct Img { unsigned char* hi; unsigned char* lo; };
  Img ctxP{nullptr, nullptr}, xP{nullptr, nullptr}, hP{nullptr, nullptr};
  // Below ~3500 rows the fp32-operand kernels stay: measured with the 64-row tile of gemm_p3.hip, 16 x 30 s = 8000 rows 16.95 ms
  // against 17.60, 8 x 30 s = 4000 rows 12.01 against 12.17, 4 x 30 s = 2000 rows 10.08 against 9.94 (PFHIP_PLANES_MIN_ROWS moves the
  // switch, PFHIP_PLANES=0 at load removes the path)
  static const int planes_min_rows = [] { const char* e = getenv("PFHIP_PLANES_MIN_ROWS"); return e && *e ? atoi(e) : 3500; }();
  const bool planes = fuse_ln && mem_in_x && m->wp_layer_bytes != 0 && M >= planes_min_rows && pfhip::gemm_f16_planes_form() &&
                      pfhip::attention_planes_ok(m->maxT);
  if (planes) {
    ++m->plane_forwards;
    const size_t pd = pfhip::plane_image_bytes(Mp, d), pf = pfhip::plane_image_bytes(Mp, c.ffn);
    HIP_TRY(m->ctxP.ensure(2 * pd));
    HIP_TRY(m->xP.ensure(2 * pd));
    HIP_TRY(m->hP.ensure(2 * pf));
    ctxP = {static_cast<unsigned char*>(m->ctxP.p), static_cast<unsigned char*>(m->ctxP.p) + pd};
    xP = {static_cast<unsigned char*>(m->xP.p), static_cast<unsigned char*>(m->xP.p) + pd};
    hP = {static_cast<unsigned char*>(m->hP.p), static_cast<unsigned char*>(m->hP.p) + pf};
  }
  // Third step (round 4, opt-in: PFHIP_KV_PLANES=1): layers 1.. hand K and V to the attention as row-major fp16 planes written by the
  // QKV projection's epilogue (the bytes of the fp32 columns they replace) and staged without the in-loop split (attention_p3.hip); Q
  // stays fp32 in qkv.  Built for VERDICT r3 item 5, bit-identical to the fp32 hand-off, and measured: the attention launch 62-64 us
  // against 66 without the memory block but 74 against 73 with it, the whole step 26.6 ms against 26.3 — not the default.
  const bool kv_planes_on = [] { const char* e = getenv("PFHIP_KV_PLANES"); return e && e[0] == '1'; }();
  const bool kv_planes = planes && kv_planes_on && c.n_head * pfhip::kHeadDim == d;
  Img kvP{nullptr, nullptr};
  if (kv_planes) {
    ++m->kvplane_forwards;
    const size_t pk = (size_t)Mp * 2 * d * 2;
    HIP_TRY(m->kvP.ensure(2 * pk));
    kvP = {static_cast<unsigned char*>(m->kvP.p), static_cast<unsigned char*>(m->kvP.p) + pk};
  }
  struct WImg { const unsigned char* hi; const unsigned char* lo; float scale; };
  auto wimg = [&](int layer, int which) -> WImg {          // 0 qkv', 1 out, 2 ffn1', 3 ffn2
    const unsigned char* base = m->d_wplanes + m->wp_layer_bytes * (size_t)layer;
    const std::string ep = "enc." + std::to_string(layer) + ".";
    switch (which) {
      case 0: return {base, base + pfhip::plane_image_bytes(3 * d, d), m->w_scale_of(m->d_lnw_qkv + (size_t)layer * 3 * d * d)};
      case 1: return {base + m->wp_off_out, base + m->wp_off_out + pfhip::plane_image_bytes(d, d), m->w_scale_of(m->W(ep + "out.w").d)};
      case 2: return {base + m->wp_off_ffn1, base + m->wp_off_ffn1 + pfhip::plane_image_bytes(c.ffn, d),
                      m->w_scale_of(m->d_lnw_ffn1 + (size_t)layer * c.ffn * d)};
      default: return {base + m->wp_off_ffn2, base + m->wp_off_ffn2 + pfhip::plane_image_bytes(d, c.ffn), m->w_scale_of(m->W(ep + "ffn2.w").d)};
    }
  };
  // C (fp32, may be null) and / or plane images of C; LayerNorm folded in when ln_colsum is given (statistics in lnstats)
  auto gemm_pl = [&](const Img& A, const WImg& W, int N, int K, float* Cd, int ldc, const Img* P, const float* bias, const float* R1, bool relu,
                     const float* ln_colsum, bool stats_out) {
    Scope sc(m, s, K_GEMM, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    pfhip::launch_gemm_p3(A.hi, A.lo, Mp, W.hi, W.lo, N, W.scale, Cd, ldc, P ? P->hi : nullptr, P ? P->lo : nullptr, Mp, bias, R1, d, M, N, K, relu,
                          ln_colsum ? m->lnstats.f() : nullptr, 4, ln_colsum, stats_out ? m->lnstats.f() : nullptr, 4, s);
  };
  for (int i = 0; i < c.enc_layers; ++i) {
    const std::string p = "enc." + std::to_string(i) + ".";
    const bool first = i == 0;
    const float* xin = first ? m->x0.f() : x;
    const int ldin = first ? FP : d, Din = first ? FD : d, Kp = first ? FP : d;
    if (kv_planes && !first) {
      // LayerNorm(x) Wqkv'^T on the plane images of x: Q as fp32 rows of qkv, K | V as row-major planes
      Scope sc(m, s, K_GEMM, 2.0 * M * (double)(3 * d) * d, 4.0 * ((double)M * d + 3.0 * d * d + (double)M * 3 * d));
      const WImg W = wimg(i, 0);
      pfhip::launch_gemm_p3(xP.hi, xP.lo, Mp, W.hi, W.lo, 3 * d, W.scale, m->qkv.f(), 3 * d, kvP.hi, kvP.lo, 2 * d,
                            m->d_lnb_qkv + (size_t)i * 3 * d, nullptr, 0, M, 3 * d, d, false, m->lnstats.f(), 4, m->d_lns_qkv + (size_t)i * 3 * d,
                            nullptr, 4, s, 0, d);
    } else if (planes && !first) {
      // LayerNorm(x) Wqkv'^T on the plane images of x that the previous layer's FFN2 left
      gemm_pl(xP, wimg(i, 0), 3 * d, d, m->qkv.f(), 3 * d, nullptr, m->d_lnb_qkv + (size_t)i * 3 * d, nullptr, false,
              m->d_lns_qkv + (size_t)i * 3 * d, false);
    } else if (fuse_ln && !first) {
      gemm_ln(x, m->d_lnw_qkv + (size_t)i * 3 * d * d, 3 * d, m->qkv.f(), 3 * d, m->d_lnb_qkv + (size_t)i * 3 * d, nullptr, nullptr, false,
              m->d_lns_qkv + (size_t)i * 3 * d, false, d);
    } else {
      lnorm(m, s, xin, ldin, m->y.f(), Kp, p + "norm1", M, Din, Kp);
      gemm(m, s, m->y.f(), Kp, first ? m->d_w0qkv : m->W(p + "qkv.w").d, 3 * d, Kp, Din, m->qkv.f(), 3 * d,
           m->W(p + "qkv.b").d, nullptr, 0, nullptr, 0, M, false);
    }
    {
      // FSMN memory of V + self-attention: one launch where the BF16 attention kernel runs (attention_x6.hip), else two
      Scope sc(m, s, K_ATTN, 4.0 * attn_pairs * d + 2.0 * 11 * M * d, 24.0 * M * d);
      // fused launch: the FSMN memory goes straight into the residual stream (x += memory; x = memory in the first layer, which
      // has no residual), so the bandwidth-bound output projection reads ONE residual
      if (kv_planes && !first)
        pfhip::launch_attention_p3(m->qkv.f(), 3 * d, kvP.hi, kvP.lo, 2 * d, d, nullptr, 0, m->m_row_off, m->m_len, m->m_row_off, m->m_len, B,
                                   c.n_head, m->maxT, att_scale, s, m->W(p + "fsmn.w").d, x, d, true, ctxP.hi, ctxP.lo, Mp);
      else
      pfhip::launch_attention_fsmn(m->qkv.f(), 3 * d, m->qkv.f() + d, 3 * d, m->qkv.f() + 2 * d, 3 * d, m->ctx.f(), d, m->m_row_off,
                                   m->m_len, B, c.n_head, m->maxT, att_scale, m->W(p + "fsmn.w").d, mem_in_x ? x : m->mem.f(), d, s,
                                   mem_in_x && !first, planes ? ctxP.hi : nullptr, planes ? ctxP.lo : nullptr, Mp);
    }
    if (planes) {
      // x = ctx Wo^T + b + (x + memory): fp32 for the residual stream, plane images for FFN1, row statistics for its LayerNorm
      gemm_pl(ctxP, wimg(i, 1), d, d, x, d, &xP, m->W(p + "out.b").d, x, false, nullptr, true);
      gemm_pl(xP, wimg(i, 2), c.ffn, d, nullptr, 0, &hP, m->d_lnb_ffn1 + (size_t)i * c.ffn, nullptr, true, m->d_lns_ffn1 + (size_t)i * c.ffn, false);
      const bool more = i + 1 < c.enc_layers;
      gemm_pl(hP, wimg(i, 3), d, c.ffn, x, d, more ? &xP : nullptr, m->W(p + "ffn2.b").d, x, false, nullptr, more);
      continue;
    }
    // x = (first ? 0 : x) + ctx*Wo + b + fsmn_memory
    if (fuse_ln) {
      gemm_ln(m->ctx.f(), m->W(p + "out.w").d, d, x, d, m->W(p + "out.b").d, mem_in_x ? x : m->mem.f(), mem_in_x || first ? nullptr : x, false,
              nullptr, true, d);
      gemm_ln(x, m->d_lnw_ffn1 + (size_t)i * c.ffn * d, c.ffn, m->hbuf.f(), c.ffn, m->d_lnb_ffn1 + (size_t)i * c.ffn, nullptr, nullptr, true,
              m->d_lns_ffn1 + (size_t)i * c.ffn, false, d);
      gemm_ln(m->hbuf.f(), m->W(p + "ffn2.w").d, d, x, d, m->W(p + "ffn2.b").d, x, nullptr, false, nullptr, i + 1 < c.enc_layers, c.ffn);
      continue;
    }
    gemm(m, s, m->ctx.f(), d, m->W(p + "out.w").d, d, d, d, x, d, m->W(p + "out.b").d, mem_in_x ? x : m->mem.f(), d,
         mem_in_x || first ? nullptr : x, d, M, false);
    lnorm(m, s, x, d, m->y.f(), d, p + "norm2", M, d, d);
    gemm(m, s, m->y.f(), d, m->W(p + "ffn1.w").d, c.ffn, d, d, m->hbuf.f(), c.ffn, m->W(p + "ffn1.b").d, nullptr, 0,
         nullptr, 0, M, true);
    gemm(m, s, m->hbuf.f(), c.ffn, m->W(p + "ffn2.w").d, d, c.ffn, c.ffn, x, d, m->W(p + "ffn2.b").d, x, d, nullptr, 0,
         M, false);
  }
  lnorm(m, s, x, d, m->enc.f(), d, "enc.after_norm", M, d, d);

  // ---- predictor + CIF ------------------------------------------------------------------------------------
  float* col = m->qkv.f();          // [Mp, 3d] reused
  float* po = m->ctx.f();           // [Mp, d] reused
  {
    Scope sc(m, s, K_OTHER, 0, 16.0 * M * d);
    pfhip::launch_im2col3(m->enc.f(), d, col, 3 * d, m->m_row_pos, m->m_row_len, M, d, s);
  }
  gemm(m, s, col, 3 * d, m->d_predconv, d, 3 * d, 3 * d, po, d, m->W("pred.conv.b").d,
       c.pred_residual ? m->enc.f() : nullptr, d, nullptr, 0, M, true);
  {
    Scope sc(m, s, K_OTHER, 2.0 * M * d, 4.0 * M * d);
    pfhip::launch_alpha(po, d, m->W("pred.out.w").d, m->W("pred.out.b").d, c.smooth_factor, c.noise_threshold,
                        m->alphas.f(), M, d, s);
  }
  float* stage = m->hbuf.f();       // [(M+B), d] reused
  {
    Scope sc(m, s, K_CIF, 2.0 * M * d, 4.0 * M * d);
    pfhip::launch_cif(m->enc.f(), d, m->alphas.f(), m->m_row_off, m->m_len, B, d, c.cif_threshold, c.tail_threshold,
                      stage, m->counts.i(), m->counts.i() + B, s);
  }
  HIP_TRY(hipMemcpyAsync(m->h_counts, m->counts.p, (size_t)2 * B * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));      // the one host sync: token counts size the decoder launch
  int ML = 0;
  for (int b = 0; b < B; ++b) {
    m->n_fires[b] = m->h_counts[b];
    m->token_num[b] = m->h_counts[B + b];
    m->tok_off[b] = ML;
    ML += m->n_fires[b];
    m->maxL = std::max(m->maxL, m->n_fires[b]);
  }
  m->ML = ML;
  if (ML == 0) { HIP_TRY(hipGetLastError()); return PFHIP_OK; }
  if (c.contextual && m->n_hw <= 0)       // the reference logs "hw_emb is null" and returns empty results (paraformer.cpp:516-520)
    return fail(PFHIP_ERR_ARG, "contextual model needs hotword embeddings (pfhip_set_hotwords / hw_emb)");
  const int MLp = round_up(ML, pfhip::kTileM);

  // ---- decoder-side metadata + workspace -------------------------------------------------------------
  {
    const size_t n = 4 * (size_t)B + ML;
    HIP_TRY(m->dmeta.ensure(n * 4));
    // second half of the pinned staging buffer: never overwritten while an earlier copy may be in flight
    if (m->h_meta_cap / 2 + n * 4 > m->h_meta_cap) return fail(PFHIP_ERR_CAPACITY, "internal: metadata staging too small");
    int* hm = reinterpret_cast<int*>(static_cast<char*>(m->h_meta) + m->h_meta_cap / 2);
    int* dm = m->dmeta.i();
    std::memcpy(hm, m->tok_off.data(), 4 * B); m->m_tok_off = dm;
    std::memcpy(hm + B, m->n_fires.data(), 4 * B); m->m_tok_len = dm + B;
    for (int b = 0; b < B; ++b) { hm[2 * B + b] = 0; hm[3 * B + b] = m->n_hw; }      // every utterance sees the same hotwords
    m->m_hw_off = dm + 2 * B; m->m_hw_len = dm + 3 * B;
    m->m_src_row = dm + 4 * B;
    for (int b = 0; b < B; ++b)
      for (int n2 = 0; n2 < m->n_fires[b]; ++n2) hm[4 * B + m->tok_off[b] + n2] = m->row_off[b] + b + n2;
    HIP_TRY(hipMemcpyAsync(dm, hm, n * 4, hipMemcpyHostToDevice, s));
  }
  HIP_TRY(m->emb.ensure((size_t)MLp * d * 4));
  HIP_TRY(m->xd.ensure((size_t)MLp * d * 4));
  HIP_TRY(m->yd.ensure((size_t)MLp * d * 4));
  HIP_TRY(m->hd.ensure((size_t)MLp * c.dec_ffn * 4));
  HIP_TRY(m->hd2.ensure((size_t)MLp * c.dec_ffn * 4));
  HIP_TRY(m->td.ensure((size_t)MLp * d * 4));
  HIP_TRY(m->t2.ensure((size_t)MLp * d * 4));
  HIP_TRY(m->qd.ensure((size_t)MLp * d * 4));
  HIP_TRY(m->ctxd.ensure((size_t)MLp * d * 4));
  HIP_TRY(m->logits.ensure((size_t)MLp * m->vocab_pad * 4));
  HIP_TRY(m->ids.ensure((size_t)MLp * 4));
  {
    Scope sc(m, s, K_OTHER, 0, 8.0 * ML * d);
    pfhip::launch_compact(stage, m->emb.f(), m->m_src_row, ML, d, s);
    HIP_TRY(hipMemcpyAsync(m->xd.p, m->emb.p, (size_t)ML * d * 4, hipMemcpyDeviceToDevice, s));
  }

  // ---- decoder ----------------------------------------------------------------------------------------------
  float* xd = m->xd.f();
  float* kvbuf = m->qkv.f();        // [Mp, 2d] reused
  // Large batches: the K/V projections of the encoder output (16 x [M, 512] x [512, 1024]: 1000 tiles each, they fill the
  // chip) do not depend on the token side, whose launches (ML ~ 7000 rows: 220 tiles for N = 512) leave most CUs half empty.
  // They can run on a second stream into per-layer buffers; the cross-attention of layer i waits for event i.  No result
  // changes.  Measured: 40.85 -> 40.63 ms per 32 x 30 s step (-0.5 %) — the chip is clock-limited under the GEMMs, so filling the
  // idle CUs buys little — while every overlapped launch measures longer, which muddies the per-kernel roofline accounting.
  // OFF by default (PFHIP_DEC_SIDE=1 turns it on); costs 1 GB of HBM for the per-layer K/V buffers when on.
  static const bool side_on = [] { const char* e = getenv("PFHIP_DEC_SIDE"); return e && e[0] == '1'; }();
  const bool side_kv = side_on && c.dec_layers > 0 && M >= 4096 && m->side_stream;
  if (side_kv) {
    HIP_TRY(m->kvside.ensure((size_t)c.dec_layers * Mp * 2 * d * 4));
    HIP_TRY(hipEventRecord(m->ev_enc_ready, s));
    HIP_TRY(hipStreamWaitEvent(m->side_stream, m->ev_enc_ready, 0));
    for (int i = 0; i < c.dec_layers; ++i) {
      const std::string p = "dec." + std::to_string(i) + ".";
      gemm(m, m->side_stream, m->enc.f(), d, m->W(p + "kv.w").d, 2 * d, d, d, m->kvside.f() + (size_t)i * Mp * 2 * d, 2 * d, m->W(p + "kv.b").d,
           nullptr, 0, nullptr, 0, M, false);
      HIP_TRY(hipEventRecord(m->ev_kv[(size_t)i], m->side_stream));
    }
  }
  double cross_pairs = 0;
  for (int b = 0; b < B; ++b) cross_pairs += (double)m->n_fires[b] * m->T[b];
  // The decoder's FFN LayerNorms fold the same way (rows >= 4096): norm1 into FFN1 — its input is the residual stream the
  // previous layer's output projection wrote (statistics from that epilogue; the first layer's input comes from the CIF, so it
  // keeps its LayerNorm launch) — and the 2048-wide ffn_norm into FFN2 (FFN1's epilogue leaves 16 pairs per row, after its ReLU).
  const bool fuse_dec = m->d_dlnw1 != nullptr && pfhip::gemm_x6_ln_ok(ML);
  const int ftiles = c.dec_ffn / pfhip::kTileN;
  if (fuse_dec) {
    HIP_TRY(m->lnstats.ensure((size_t)std::max(Mp, MLp) * 4 * 2 * 4));
    HIP_TRY(m->lnstats2.ensure((size_t)MLp * ftiles * 2 * 4));
  }
  auto x6ln = [&](const float* A, int K, const float* Wd, int N, float* Cd, const float* bias, const float* R1, bool relu,
                  const float* st_in, int tiles_in, const float* colsum, float* st_out) {
    Scope sc(m, s, K_GEMM, 2.0 * ML * (double)N * K, 4.0 * ((double)ML * K + (double)N * K + (double)ML * N));
    pfhip::launch_gemm_f32_x6_ln(A, K, Wd, K, Cd, N, bias, R1, d, nullptr, 0, ML, N, K, relu, st_in, tiles_in, colsum, st_out, s,
                                 m->w_scale_of(Wd));
  };
  bool xd_has_stats = false;          // lnstats holds the row statistics of the current xd
  // Round 4: from 3500 token rows on (and only where the encoder ran on plane images: same arithmetic form) the decoder's large
  // GEMMs take plane-image operands as well (gemm_p3.hip): the K/V projection reads the images of the encoder output (one split
  // launch per forward), the cross-attention writes its context as images, the output projection leaves the token-side residual
  // stream as fp32 + images + row statistics, FFN1' (norm1 folded) reads those and leaves the hidden activation as fp32 + images +
  // the 16 statistics pairs per row that FFN2' (ffn_norm folded) merges.  The first layer's FFN (its input comes from the CIF),
  // the q projections (their input comes from the FSMN kernel), a contextual model's last layer and the vocabulary projection stay
  // on the in-loop-split kernels.
  static const int dec_planes_min_rows = [] { const char* e = getenv("PFHIP_DEC_PLANES_MIN_ROWS"); return e && *e ? atoi(e) : 3500; }();
  const bool dec_planes = planes && fuse_dec && m->dwp_layer_bytes != 0 && ML >= dec_planes_min_rows && pfhip::attention_planes_ok(m->maxL) &&
                          pfhip::gemm_f16_planes_form();
  Img encP{nullptr, nullptr}, xdP{nullptr, nullptr}, hdP{nullptr, nullptr}, ctxdP{nullptr, nullptr};
  bool xd_has_planes = false;
  if (dec_planes) {
    ++m->dec_plane_forwards;
    const size_t pe = pfhip::plane_image_bytes(Mp, d), pdd = pfhip::plane_image_bytes(MLp, d), pff = pfhip::plane_image_bytes(MLp, c.dec_ffn);
    HIP_TRY(m->encP.ensure(2 * pe));
    HIP_TRY(m->xdP.ensure(2 * pdd));
    HIP_TRY(m->ctxP.ensure(2 * pdd));
    HIP_TRY(m->hP.ensure(2 * pff));
    encP = {static_cast<unsigned char*>(m->encP.p), static_cast<unsigned char*>(m->encP.p) + pe};
    xdP = {static_cast<unsigned char*>(m->xdP.p), static_cast<unsigned char*>(m->xdP.p) + pdd};
    ctxdP = {static_cast<unsigned char*>(m->ctxP.p), static_cast<unsigned char*>(m->ctxP.p) + pdd};
    hdP = {static_cast<unsigned char*>(m->hP.p), static_cast<unsigned char*>(m->hP.p) + pff};
    Scope sc(m, s, K_OTHER, 0, 8.0 * M * d);
    pfhip::launch_split_planes(m->enc.f(), d, M, Mp, d, 1.0f, encP.hi, encP.lo, s);
  }
  auto dwimg = [&](int layer, int which) -> WImg {         // 0 ffn1', 1 ffn2', 2 kv, 3 out
    const unsigned char* base = m->d_dwplanes + m->dwp_layer_bytes * (size_t)layer;
    const std::string dp = "dec." + std::to_string(layer) + ".";
    switch (which) {
      case 0: return {base, base + pfhip::plane_image_bytes(c.dec_ffn, d), m->w_scale_of(m->d_dlnw1 + (size_t)layer * c.dec_ffn * d)};
      case 1: return {base + m->dwp_off_ffn2, base + m->dwp_off_ffn2 + pfhip::plane_image_bytes(d, c.dec_ffn),
                      m->w_scale_of(m->d_dlnw2 + (size_t)layer * d * c.dec_ffn)};
      case 2: return {base + m->dwp_off_kv, base + m->dwp_off_kv + pfhip::plane_image_bytes(2 * d, d), m->w_scale_of(m->W(dp + "kv.w").d)};
      default: return {base + m->dwp_off_out, base + m->dwp_off_out + pfhip::plane_image_bytes(d, d), m->w_scale_of(m->W(dp + "out.w").d)};
    }
  };
  // one gemm_p3 launch over `rows` rows: A image with rows_a rows per K-step, result as fp32 (Cd) and / or images (P, MLp rows)
  auto dgemm_pl = [&](const Img& A, int rows_a, const WImg& W, int rows, int N, int K, float* Cd, int ldc, const Img* P, const float* bias,
                      const float* R1, bool relu, const float* st_in, int tiles_in, const float* colsum, float* st_out) {
    Scope sc(m, s, K_GEMM, 2.0 * rows * (double)N * K, 4.0 * ((double)rows * K + (double)N * K + (double)rows * N));
    pfhip::launch_gemm_p3(A.hi, A.lo, rows_a, W.hi, W.lo, N, W.scale, Cd, ldc, P ? P->hi : nullptr, P ? P->lo : nullptr, MLp, bias, R1, d, rows, N, K,
                          relu, st_in, tiles_in, colsum, st_out, 4, s);
  };
  auto dec_ffn = [&](const std::string& p, int li, const float* xin, float* out) {
    if (dec_planes && xd_has_planes && xd_has_stats) {
      // FFN1' on the images of the residual stream: hidden activation as fp32 (row statistics ride on that pass) + images
      dgemm_pl(xdP, MLp, dwimg(li, 0), ML, c.dec_ffn, d, m->hd.f(), c.dec_ffn, &hdP, m->d_dlnb1 + (size_t)li * c.dec_ffn, nullptr, true,
               m->lnstats.f(), 4, m->d_dlns1 + (size_t)li * c.dec_ffn, m->lnstats2.f());
      dgemm_pl(hdP, MLp, dwimg(li, 1), ML, d, c.dec_ffn, out, d, nullptr, m->d_dlnb2 + (size_t)li * d, nullptr, false, m->lnstats2.f(), ftiles,
               m->d_dlns2 + (size_t)li * d, nullptr);
      return;
    }
    if (fuse_dec) {
      if (xd_has_stats) {
        x6ln(xin, d, m->d_dlnw1 + (size_t)li * c.dec_ffn * d, c.dec_ffn, m->hd.f(), m->d_dlnb1 + (size_t)li * c.dec_ffn, nullptr, true,
             m->lnstats.f(), 4, m->d_dlns1 + (size_t)li * c.dec_ffn, m->lnstats2.f());
      } else {
        lnorm(m, s, xin, d, m->yd.f(), d, p + "norm1", ML, d, d);
        x6ln(m->yd.f(), d, m->W(p + "ffn1.w").d, c.dec_ffn, m->hd.f(), m->W(p + "ffn1.b").d, nullptr, true, nullptr, 0, nullptr,
             m->lnstats2.f());
      }
      x6ln(m->hd.f(), c.dec_ffn, m->d_dlnw2 + (size_t)li * d * c.dec_ffn, d, out, m->d_dlnb2 + (size_t)li * d, nullptr, false, m->lnstats2.f(),
           ftiles, m->d_dlns2 + (size_t)li * d, nullptr);
      return;
    }
    lnorm(m, s, xin, d, m->yd.f(), d, p + "norm1", ML, d, d);
    gemm(m, s, m->yd.f(), d, m->W(p + "ffn1.w").d, c.dec_ffn, d, d, m->hd.f(), c.dec_ffn, m->W(p + "ffn1.b").d, nullptr,
         0, nullptr, 0, ML, true);
    lnorm(m, s, m->hd.f(), c.dec_ffn, m->hd2.f(), c.dec_ffn, p + "ffn_norm", ML, c.dec_ffn, c.dec_ffn);
    gemm(m, s, m->hd2.f(), c.dec_ffn, m->W(p + "ffn2.w").d, d, c.dec_ffn, c.dec_ffn, out, d, nullptr, nullptr, 0, nullptr,
         0, ML, false);
  };
  for (int i = 0; i < c.dec_layers; ++i) {
    const std::string p = "dec." + std::to_string(i) + ".";
    dec_ffn(p, i, xd, m->td.f());
    lnorm(m, s, m->td.f(), d, m->t2.f(), d, p + "norm2", ML, d, d);
    {
      Scope sc(m, s, K_FSMN, 2.0 * 11 * ML * d, 12.0 * ML * d);
      pfhip::launch_fsmn(m->t2.f(), d, m->W(p + "fsmn.w").d, xd, d, xd, d, m->m_tok_off, m->m_tok_len, B, m->maxL, d, s);
    }
    lnorm(m, s, xd, d, m->yd.f(), d, p + "norm3", ML, d, d);
    gemm(m, s, m->yd.f(), d, m->W(p + "q.w").d, d, d, d, m->qd.f(), d, m->W(p + "q.b").d, nullptr, 0, nullptr, 0, ML, false);
    const bool plain_layer = !(c.contextual && i == c.dec_layers - 1);
    if (side_kv) {
      kvbuf = m->kvside.f() + (size_t)i * Mp * 2 * d;
      HIP_TRY(hipStreamWaitEvent(s, m->ev_kv[(size_t)i], 0));
    } else if (dec_planes) {
      Scope sc(m, s, K_GEMM, 2.0 * M * 2.0 * d * d, 4.0 * ((double)M * d + 2.0 * d * d + 2.0 * M * d));
      const WImg W = dwimg(i, 2);
      pfhip::launch_gemm_p3(encP.hi, encP.lo, Mp, W.hi, W.lo, 2 * d, W.scale, kvbuf, 2 * d, nullptr, nullptr, Mp, m->W(p + "kv.b").d, nullptr, 0, M,
                            2 * d, d, false, nullptr, 0, nullptr, nullptr, 4, s);
    } else {
      gemm(m, s, m->enc.f(), d, m->W(p + "kv.w").d, 2 * d, d, d, kvbuf, 2 * d, m->W(p + "kv.b").d, nullptr, 0, nullptr, 0, M,
           false);
    }
    {
      Scope sc(m, s, K_ATTN, 4.0 * cross_pairs * d, 8.0 * ML * d + 8.0 * M * d);
      if (dec_planes && plain_layer)     // the context leaves as plane images for the output projection; no fp32 context is written
        pfhip::launch_attention_x3(m->qd.f(), d, kvbuf, 2 * d, kvbuf + d, 2 * d, m->ctxd.f(), d, m->m_tok_off, m->m_tok_len, m->m_row_off, m->m_len,
                                   B, c.n_head, m->maxL, att_scale, s, nullptr, nullptr, 0, false, ctxdP.hi, ctxdP.lo, MLp);
      else
        pfhip::launch_attention(m->qd.f(), d, kvbuf, 2 * d, kvbuf + d, 2 * d, m->ctxd.f(), d, m->m_tok_off, m->m_tok_len,
                                m->m_row_off, m->m_len, B, c.n_head, m->maxL, att_scale, s);
    }
    xd_has_stats = false;
    xd_has_planes = false;
    if (dec_planes && plain_layer) {
      // xd = ctx Wo^T + b + xd: fp32 for the residual adds, images + row statistics for the next FFN1'
      dgemm_pl(ctxdP, MLp, dwimg(i, 3), ML, d, d, xd, d, &xdP, m->W(p + "out.b").d, xd, false, nullptr, 0, nullptr, m->lnstats.f());
      xd_has_stats = true;
      xd_has_planes = true;
      continue;
    }
    if (!(c.contextual && i == c.dec_layers - 1)) {
      if (fuse_dec) {          // the output projection also leaves the statistics the next norm1 needs
        x6ln(m->ctxd.f(), d, m->W(p + "out.w").d, d, xd, m->W(p + "out.b").d, xd, false, nullptr, 0, nullptr, m->lnstats.f());
        xd_has_stats = true;
      } else {
        gemm(m, s, m->ctxd.f(), d, m->W(p + "out.w").d, d, d, d, xd, d, m->W(p + "out.b").d, xd, d, nullptr, 0, ML, false);
      }
      continue;
    }
    // ---- contextual last layer (UPSTREAM ContextualDecoderLayer + ContextualBiasDecoder + bias_output):
    //      x = x_self_attn + W_b [x_src_attn | cx],  cx = cross-attention of norm3_b(x_self_attn) over the hotword
    //      embeddings; xd holds x_self_attn, cat = [x_src_attn | cx] with row stride 2d.
    HIP_TRY(m->cat.ensure((size_t)MLp * 2 * d * 4));
    float* cat = m->cat.f();
    gemm(m, s, m->ctxd.f(), d, m->W(p + "out.w").d, d, d, d, cat, 2 * d, m->W(p + "out.b").d, nullptr, 0, nullptr, 0, ML, false);
    lnorm(m, s, xd, d, m->yd.f(), d, "bias.dec.norm3", ML, d, d);
    gemm(m, s, m->yd.f(), d, m->W("bias.dec.q.w").d, d, d, d, m->qd.f(), d, m->W("bias.dec.q.b").d, nullptr, 0, nullptr, 0, ML, false);
    // the hotword K/V projection lives in its own buffer, sized by the hotword count and filled once per hotword set
    // (set_hotwords_locked) — the audio-side kv workspace only holds Mp rows
    const float* hwkv = m->hwkv.f();
    {
      Scope sc(m, s, K_ATTN, 4.0 * ML * (double)m->n_hw * d, 8.0 * ML * d);
      pfhip::launch_attention(m->qd.f(), d, hwkv, 2 * d, hwkv + d, 2 * d, m->ctxd.f(), d, m->m_tok_off, m->m_tok_len,
                              m->m_hw_off, m->m_hw_len, B, c.n_head, m->maxL, att_scale, s);
    }
    gemm(m, s, m->ctxd.f(), d, m->W("bias.dec.out.w").d, d, d, d, cat + d, 2 * d, m->W("bias.dec.out.b").d, nullptr, 0, nullptr,
         0, ML, false);
    gemm(m, s, cat, 2 * d, m->W("bias.out.w").d, d, 2 * d, 2 * d, xd, d, nullptr, xd, d, nullptr, 0, ML, false);
  }
  dec_ffn("dec3.", c.dec_layers, xd, m->td.f());
  lnorm(m, s, m->td.f(), d, m->yd.f(), d, "dec.after_norm", ML, d, d);
  gemm(m, s, m->yd.f(), d, m->W("dec.out.w").d, c.vocab, d, d, m->logits.f(), m->vocab_pad, m->d_vocab_bias, nullptr, 0,
       nullptr, 0, ML, false);
  HIP_TRY(hipGetLastError());
  return PFHIP_OK;
}

// ---- a6 producer: CifPredictorV3.get_upsample_timestmap on the encoder output of the batch just run -------------------
pfhip_status ts_head_locked(pfhip_model* m, hipStream_t s, bool stepwise_only = false) {
  ForwardCtx fc(m);
  const Config& c = m->cfg;
  if (!c.timestamp) return fail(PFHIP_ERR_UNSUPPORTED, "model has no timestamp head (us_alphas / us_cif_peak outputs)");
  if (m->have_ts) return PFHIP_OK;
  const int d = c.d_model, B = m->B, M = m->M;
  if (M == 0) { m->have_ts = true; return PFHIP_OK; }
  const int Mp = round_up(M, pfhip::kTileM), R = 3 * M, Rp = 3 * Mp;
  HIP_TRY(m->ts_up.ensure((size_t)Rp * d * 4));
  HIP_TRY(m->ts_gx.ensure((size_t)Rp * 8 * d * 4));
  HIP_TRY(m->ts_y.ensure((size_t)Rp * 2 * d * 4));
  if (m->ts_hx.cap < (size_t)pfhip::kBlstmScratchFloats * 4) {
    HIP_TRY(m->ts_hx.ensure((size_t)pfhip::kBlstmScratchFloats * 4));
    HIP_TRY(hipMemsetAsync(m->ts_hx.p, 0, (size_t)pfhip::kBlstmScratchFloats * 4, s));
  }
  // the kernel's error flag is per call: a barrier time-out of an earlier request must not fail this one
  HIP_TRY(hipMemsetAsync(m->ts_hx.f() + pfhip::kBlstmFlagWord, 0, 4, s));
  if (m->debug_blstm_flag) {          // test hook (pfhip_debug_poke): this request sees the flag raised, as after a barrier time-out
    const unsigned one = 1u;
    HIP_TRY(hipMemcpyAsync(m->ts_hx.f() + pfhip::kBlstmFlagWord, &one, 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    m->debug_blstm_flag = 0;
  }
  HIP_TRY(m->ts_a2.ensure((size_t)R * 4));
  HIP_TRY(m->ts_alphas.ensure((size_t)R * 4));
  HIP_TRY(m->ts_peaks.ensure((size_t)R * 4));
  // upsampled offsets / lengths / token counts per utterance
  HIP_TRY(m->ts_meta.ensure((size_t)3 * B * 4));
  std::vector<int> hm((size_t)3 * B);
  int maxL = 0;
  for (int b = 0; b < B; ++b) {
    hm[b] = 3 * m->row_off[b]; hm[B + b] = 3 * m->T[b]; hm[2 * B + b] = m->token_num[b];
    maxL = std::max(maxL, 3 * m->T[b]);
  }
  HIP_TRY(hipMemcpyAsync(m->ts_meta.p, hm.data(), hm.size() * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));          // hm is a stack buffer
  const int* d_off = m->ts_meta.i();
  const int* d_len = d_off + B;
  const int* d_tok = d_off + 2 * B;
  // ConvTranspose1d: [M, d] x [3d, d]^T -> [M, 3d] == [3M, d]
  gemm(m, s, m->enc.f(), d, m->d_up_w, 3 * d, d, d, m->ts_up.f(), 3 * d, m->d_up_b, nullptr, 0, nullptr, 0, M, false);
  // input projections of both directions: [3M, d] x [8d, d]^T -> [3M, 8d]
  gemm(m, s, m->ts_up.f(), d, m->d_wih, 8 * d, d, d, m->ts_gx.f(), 8 * d, m->d_bih, nullptr, 0, nullptr, 0, R, false);
  {
    Scope sc(m, s, K_OTHER, 2.0 * R * 8 * d * d, 4.0 * R * 10 * d);
    // The recurrence advances up to 32 utterances together.  First the persistent kernel (one launch, 9 us per step); if its
    // step barrier could not be served — the 32 blocks of a direction were not co-resident on one XCD because other work holds
    // CUs there — the request is NOT failed: the same recurrence is redone as one launch per step (blstm.hip), which needs no
    // co-residency and gives the same values bit for bit.  PFHIP_BLSTM_STEPWISE=1 skips the persistent attempt.
    static const bool force_stepwise = [] { const char* e = getenv("PFHIP_BLSTM_STEPWISE"); return e && e[0] == '1'; }();
    auto recurrence = [&](bool stepwise, hipStream_t rs) -> pfhip_status {
      for (int b0 = 0; b0 < B; b0 += 32) {
        const int nb = std::min(32, B - b0);
        int lmax = 0;
        for (int b = b0; b < b0 + nb; ++b) lmax = std::max(lmax, 3 * m->T[b]);
        if (stepwise)
          HIP_TRY(pfhip::launch_blstm_stepwise(m->ts_gx.f(), m->d_whh, m->ts_y.f(), m->ts_hx.f(), m->ts_cst.f(), d_off + b0, d_len + b0, nb,
                                               lmax, rs));
        else
          HIP_TRY(pfhip::launch_blstm(m->ts_gx.f(), m->d_whh, m->ts_y.f(), m->ts_hx.f(), d_off + b0, d_len + b0, nb, lmax, rs));
      }
      return PFHIP_OK;
    };
    HIP_TRY(m->ts_cst.ensure((size_t)2 * 32 * 512 * 4));
    m->ts_persistent = false;
    if (!stepwise_only && !force_stepwise) {
      // The persistent kernel wants the 32 blocks of a direction co-resident on one XCD: two contexts of a device must not run it
      // at the same time (they would time each other out into the per-step form).  Round 3 held a host lock from the launch to the
      // end of the stream — the whole rest of the caller's forward — so the contexts of a timestamp model took turns and three
      // batches in flight were slower than one (bench.c4: 58 ms against 38).  Now every context of a device enqueues its recurrence
      // on ONE stream of that device (events tie it to the caller's stream), so the device orders them and the lock covers the
      // enqueue only; the kernel's error word travels to the host with the results (fetch_once), and a raised one redoes the
      // recurrence per step there.
      pfhip_model* owner = m->weights_of ? m->weights_of : m;
      static std::mutex blstm_mu[64];
      {
        std::lock_guard<std::mutex> bl(blstm_mu[m->device & 63]);
        if (!owner->blstm_stream) HIP_TRY(hipStreamCreateWithFlags(&owner->blstm_stream, hipStreamNonBlocking));
        if (!m->ev_ts_in) {
          HIP_TRY(hipEventCreateWithFlags(&m->ev_ts_in, hipEventDisableTiming));
          HIP_TRY(hipEventCreateWithFlags(&m->ev_ts_out, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(m->ev_ts_in, s));
        HIP_TRY(hipStreamWaitEvent(owner->blstm_stream, m->ev_ts_in, 0));
        pfhip_status st = recurrence(false, owner->blstm_stream);
        if (st) return st;
        HIP_TRY(hipEventRecord(m->ev_ts_out, owner->blstm_stream));
      }
      HIP_TRY(hipStreamWaitEvent(s, m->ev_ts_out, 0));
      m->ts_persistent = true;
    } else {
      pfhip_status st = recurrence(true, s);
      if (st) return st;
    }
    pfhip::launch_alpha2(m->ts_y.f(), 2 * d, m->W("pred.out2.w").d, m->out2_b, c.smooth_factor2, c.noise_threshold2,
                         m->ts_a2.f(), R, 2 * d, s);
    pfhip::launch_us_cif(m->ts_a2.f(), d_off, d_len, d_tok, B, maxL, c.cif_threshold - 1e-4f, m->ts_alphas.f(),
                         m->ts_peaks.f(), s);
  }
  HIP_TRY(hipGetLastError());
  m->have_ts = true;
  return PFHIP_OK;
}

pfhip_status head_locked(pfhip_model* m, hipStream_t s, bool want_logp) {
  if (m->ML == 0) return PFHIP_OK;
  const int V = m->cfg.vocab;
  if (want_logp) HIP_TRY(m->logp.ensure((size_t)m->ML * V * 4));
  {
    Scope sc(m, s, K_HEAD, 0, 4.0 * m->ML * V * (want_logp ? 3 : 2));
    pfhip::launch_logsoftmax_argmax(m->logits.f(), m->vocab_pad, m->ML, V, want_logp ? m->logp.f() : nullptr,
                                    static_cast<int32_t*>(m->ids.p), s, m->d_range_flag);
  }
  m->have_logp = want_logp;
  HIP_TRY(hipGetLastError());
  return PFHIP_OK;
}

// the range flag of the forward crosses to the host with the results (one 4-byte copy in front of the sync that is there anyway)
pfhip_status read_range_flag(pfhip_model* m, hipStream_t s, bool sync) {
  m->range_hit = 0;
  if (!m->d_range_flag) return PFHIP_OK;
  if (!m->h_flag) HIP_TRY(hipHostMalloc((void**)&m->h_flag, 64, hipHostMallocDefault));
  *m->h_flag = 0;
  HIP_TRY(hipMemcpyAsync(m->h_flag, m->d_range_flag, 4, hipMemcpyDeviceToHost, s));
  if (sync) HIP_TRY(hipStreamSynchronize(s));
  return PFHIP_OK;
}

pfhip_status fetch_once(pfhip_model* m, pfhip_out* out, hipStream_t s);
// The guard of the fp16 two-plane domain (kernels.h LaunchCtx): a forward whose range flag came back raised — a LayerNorm-folded
// row outside [2^-8, 2^12] rms, or a log-prob row that is not finite — is redone ONCE, inside the same context, on the exact
// kernels (bf16 three-plane GEMMs and attention, fp32 operands instead of plane images), and counted
// (pfhip_debug_poke "range_fallbacks").  The reference computes in plain fp32 (paraformer.cpp:496-541).
pfhip_status fetch_locked(pfhip_model* m, pfhip_out* out, hipStream_t s) {
  pfhip_status st = fetch_once(m, out, s);
  if (st || !m->range_hit || m->exact_rerun || m->always_exact || !m->last_pcm || m->last_feats_only) return st;
  ++m->range_fallbacks;
  m->exact_rerun = true;
  const std::vector<int64_t> off = m->last_off;
  const std::vector<int> ns = m->last_n;
  st = enqueue_locked(m, m->last_pcm, off.data(), ns.data(), (int)ns.size(), s, false);
  if (!st) st = head_locked(m, s, out->logp != nullptr);
  if (!st) st = fetch_once(m, out, s);
  m->exact_rerun = false;
  return st;
}

pfhip_status fetch_once(pfhip_model* m, pfhip_out* out, hipStream_t s) {
  if (!out) return fail(PFHIP_ERR_ARG, "null out");
  ForwardCtx fc(m);
  const int B = m->B;
  for (int b = 0; b < B; ++b) {
    if (out->token_num) out->token_num[b] = m->token_num[b];
    if (out->n_fires) out->n_fires[b] = m->n_fires[b];
    if (out->n_frames) out->n_frames[b] = m->T[b];
  }
  if (out->us_alphas || out->us_peaks || out->us_len) {
    for (int attempt = 0; attempt < 2; ++attempt) {
      pfhip_status st = ts_head_locked(m, s, attempt == 1);
      if (st) return st;
      for (int b = 0; b < B; ++b) {
        const int L = 3 * m->T[b];
        if (out->us_len) out->us_len[b] = L;
        if ((out->us_alphas || out->us_peaks) && out->max_us < L) return fail(PFHIP_ERR_CAPACITY, "max_us smaller than 3 x frames");
        if (out->us_alphas && L)
          HIP_TRY(hipMemcpyAsync(out->us_alphas + (size_t)b * out->max_us, m->ts_alphas.f() + (size_t)3 * m->row_off[b], (size_t)L * 4,
                                 hipMemcpyDeviceToHost, s));
        if (out->us_peaks && L)
          HIP_TRY(hipMemcpyAsync(out->us_peaks + (size_t)b * out->max_us, m->ts_peaks.f() + (size_t)3 * m->row_off[b], (size_t)L * 4,
                                 hipMemcpyDeviceToHost, s));
      }
      // the persistent recurrence's error word (a step barrier that timed out: its 32 blocks were not co-resident) comes back with
      // the results; raised, the request is NOT failed: the same recurrence is redone as one launch per step, bit for bit the same
      unsigned flag = 0;
      if (m->ts_persistent && m->M > 0) HIP_TRY(hipMemcpyAsync(&flag, m->ts_hx.f() + pfhip::kBlstmFlagWord, 4, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (!flag) break;
      HIP_TRY(hipMemsetAsync(m->ts_hx.f() + pfhip::kBlstmFlagWord, 0, 4, s));
      ++m->blstm_fallbacks;
      m->have_ts = false;
    }
  }
  if (m->ML == 0) {                                  // no token at all: NaN alphas can do that too
    if (m->M > 0) {
      const pfhip_status rs = read_range_flag(m, s, true);
      if (rs) return rs;
      m->range_hit = *m->h_flag;
    }
    return PFHIP_OK;
  }
  if ((out->token_ids || out->logp) && out->max_tokens < m->maxL)
    return fail(PFHIP_ERR_CAPACITY, "max_tokens smaller than the longest token sequence");
  if (out->logp && !m->have_logp) {
    pfhip_status st = head_locked(m, s, true);
    if (st) return st;
  }
  std::vector<int32_t> ids;
  if (out->token_ids) {
    ids.resize(m->ML);
    HIP_TRY(hipMemcpyAsync(ids.data(), m->ids.p, (size_t)m->ML * 4, hipMemcpyDeviceToHost, s));
  }
  const int V = m->cfg.vocab;
  if (out->logp) {
    for (int b = 0; b < B; ++b)
      if (m->n_fires[b])
        HIP_TRY(hipMemcpyAsync(out->logp + (size_t)b * out->max_tokens * V, m->logp.f() + (size_t)m->tok_off[b] * V,
                               (size_t)m->n_fires[b] * V * 4, hipMemcpyDeviceToHost, s));
  }
  {
    const pfhip_status rs = read_range_flag(m, s, false);
    if (rs) return rs;
  }
  HIP_TRY(hipStreamSynchronize(s));
  m->range_hit = m->h_flag ? *m->h_flag : 0;
  if (out->token_ids)
    for (int b = 0; b < B; ++b)
      std::memcpy(out->token_ids + (size_t)b * out->max_tokens, ids.data() + m->tok_off[b], 4 * (size_t)m->n_fires[b]);
  return PFHIP_OK;
}

pfhip_status set_hotwords_locked(pfhip_model* m, const float* hw_emb, int H, hipStream_t s) {
  const int d = m->cfg.d_model;
  if (!m->cfg.contextual) return fail(PFHIP_ERR_UNSUPPORTED, "model has no bias decoder (use_hotword == false)");
  HIP_TRY(m->hw.ensure((size_t)round_up(H, pfhip::kTileM) * d * 4));
  HIP_TRY(hipMemcpyAsync(m->hw.p, hw_emb, (size_t)H * d * 4, hipMemcpyHostToDevice, s));
  // K/V projection of the bias decoder's cross-attention: constant per hotword set, [round_up(H,128), 2d]
  HIP_TRY(m->hwkv.ensure((size_t)round_up(H, pfhip::kTileM) * 2 * d * 4));
  gemm(m, s, m->hw.f(), d, m->W("bias.dec.kv.w").d, 2 * d, d, d, m->hwkv.f(), 2 * d, m->W("bias.dec.kv.b").d, nullptr, 0, nullptr, 0,
       H, false);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  m->n_hw = H;
  return PFHIP_OK;
}

pfhip_status stage_pcm(pfhip_model* m, const float* const* pcm, const int* n_samples, int B, hipStream_t s,
                       std::vector<int64_t>& off) {
  off.assign(B, 0);
  int64_t tot = 0;
  for (int b = 0; b < B; ++b) {
    if (n_samples[b] < 0 || (n_samples[b] > 0 && !pcm[b])) return fail(PFHIP_ERR_ARG, "bad pcm buffer");
    off[b] = tot;
    tot += (n_samples[b] + 3) & ~3;
  }
  HIP_TRY(m->pcm.ensure((size_t)std::max<int64_t>(tot, 4) * 4));
  for (int b = 0; b < B; ++b)
    if (n_samples[b])
      HIP_TRY(hipMemcpyAsync(m->pcm.f() + off[b], pcm[b], (size_t)n_samples[b] * 4, hipMemcpyHostToDevice, s));
  return PFHIP_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* pfhip_last_error(void) { return g_err.c_str(); }

static pfhip_status apply_env_inflight(pfhip_model** out);

pfhip_status pfhip_create_group(const void* blob, size_t blob_bytes, const char* manifest_json, const int* devices, int n_devices,
                                pfhip_model** out) {
  g_err.clear();
  if (!devices || n_devices < 1 || !out) return fail(PFHIP_ERR_ARG, "bad device list");
  pfhip_model* head = nullptr;
  try {
    pfhip_status st = build_model(blob, blob_bytes, manifest_json, devices[0], &head);
    if (st) return st;
    for (int i = 1; i < n_devices; ++i) {
      pfhip_model* r = nullptr;
      st = build_model(blob, blob_bytes, manifest_json, devices[i], &r);
      if (st) { const std::string why = g_err; pfhip_destroy(head); g_err = why; return st; }
      r->group_head = head;
      head->replicas.push_back(r);
    }
  } catch (const std::exception& e) {
    if (head) pfhip_destroy(head);
    return fail(PFHIP_ERR_FORMAT, e.what());
  }
  *out = head;
  return apply_env_inflight(out);
}

// PFHIP_DEVICES="0,1,2,..." turns every model created through pfhip_create / pfhip_create_from_memory into a group with one
// replica per listed device (the `device` argument is then ignored): the unchanged server with its one shared handle and
// `decoder-thread-num` threads (funasr-wss-server.cpp:479-481) uses all of them.
static bool env_devices(std::vector<int>& devs) {
  const char* e = std::getenv("PFHIP_DEVICES");
  if (!e || !*e) return false;
  devs.clear();
  std::stringstream ss(e);
  std::string tok;
  while (std::getline(ss, tok, ',')) {
    if (tok.empty()) continue;
    char* end = nullptr;
    const long v = std::strtol(tok.c_str(), &end, 10);
    if (end == tok.c_str() || *end != '\0' || v < 0) return false;
    devs.push_back((int)v);
  }
  return !devs.empty();
}

// PFHIP_INFLIGHT=n: every handle is created with n execution contexts per device (pfhip_set_inflight)
static pfhip_status apply_env_inflight(pfhip_model** out) {
  const char* e = std::getenv("PFHIP_INFLIGHT");
  if (!e || !*e) return PFHIP_OK;
  const int n = std::atoi(e);
  if (n < 1 || n > 16) return PFHIP_OK;
  const pfhip_status st = pfhip_set_inflight(*out, n);
  if (st) { const std::string why = g_err; pfhip_destroy(*out); *out = nullptr; g_err = why; }
  return st;
}

pfhip_status pfhip_create_from_memory(const void* blob, size_t blob_bytes, const char* manifest_json, int device,
                                      pfhip_model** out) {
  g_err.clear();
  std::vector<int> devs;
  if (env_devices(devs)) return pfhip_create_group(blob, blob_bytes, manifest_json, devs.data(), (int)devs.size(), out);
  try {
    const pfhip_status st = build_model(blob, blob_bytes, manifest_json, device, out);
    return st ? st : apply_env_inflight(out);
  } catch (const std::exception& e) { return fail(PFHIP_ERR_FORMAT, e.what()); }
}

int pfhip_group_size(const pfhip_model* m) { return m ? 1 + (int)m->replicas.size() : 0; }

pfhip_status pfhip_group_stats(pfhip_model* m, int* devices, int64_t* calls, int64_t* utterances, int* open_streams, int cap) {
  g_err.clear();
  if (!m || cap < 1 + (int)m->replicas.size()) return fail(PFHIP_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> l(m->bq.mu);             // pfhip_set_inflight grows `contexts` under this lock
  for (int i = 0; i <= (int)m->replicas.size(); ++i) {
    const pfhip_model* r = i == 0 ? m : m->replicas[(size_t)i - 1];
    if (devices) devices[i] = r->device;
    int64_t nc = r->served_calls.load(), nu = r->served_utts.load();
    for (const pfhip_model* cx : r->contexts) { nc += cx->served_calls.load(); nu += cx->served_utts.load(); }
    if (calls) calls[i] = nc;
    if (utterances) utterances[i] = nu;
    if (open_streams) open_streams[i] = r->live_streams.load();
  }
  return PFHIP_OK;
}

pfhip_status pfhip_create(const char* weight_blob_path, const char* manifest_json_path, int device, pfhip_model** out) {
  g_err.clear();
  if (!weight_blob_path || !manifest_json_path || !out) return fail(PFHIP_ERR_ARG, "null argument");
  std::vector<char> blob, man;
  pfhip_status st = read_file(weight_blob_path, blob);
  if (st) return st;
  st = read_file(manifest_json_path, man);
  if (st) return st;
  man.push_back('\0');
  return pfhip_create_from_memory(blob.data(), blob.size(), man.data(), device, out);
}

void pfhip_destroy(pfhip_model* m) {
  if (!m) return;
  for (pfhip_model* r : m->replicas) pfhip_destroy(r);
  m->replicas.clear();
  for (pfhip_model* cx : m->contexts) pfhip_destroy(cx);
  m->contexts.clear();
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  for (Buf* b : {&m->pcm, &m->meta, &m->feats, &m->x0, &m->x, &m->y, &m->qkv, &m->mem, &m->ctx, &m->hbuf, &m->enc,
                 &m->alphas, &m->counts, &m->emb, &m->xd, &m->yd, &m->hd, &m->hd2, &m->td, &m->t2, &m->qd, &m->ctxd,
                 &m->logits, &m->logp, &m->ids, &m->dmeta, &m->cat, &m->hw, &m->hwkv, &m->ts_up, &m->ts_gx, &m->ts_y, &m->ts_hx, &m->ts_a2,
                 &m->ts_alphas, &m->ts_peaks, &m->ts_meta, &m->sseg, &m->fbk, &m->d_ops, &m->kvall, &m->lnstats, &m->lnstats2, &m->kvside, &m->ts_cst, &m->ctxP, &m->xP, &m->hP, &m->encP, &m->xdP, &m->kvP})
    b->release();
  if (!m->weights_of) {          // a context borrows these
#define X(f) if (m->f) (void)hipFree((void*)m->f);
    PFHIP_WEIGHT_PTRS(X)
#undef X
  }
  if (m->h_meta) (void)hipHostFree(m->h_meta);
  if (m->h_ops) (void)hipHostFree(m->h_ops);
  if (m->h_counts) (void)hipHostFree(m->h_counts);
  if (m->h_flag) (void)hipHostFree(m->h_flag);
  for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
  if (m->ev_enc_ready) (void)hipEventDestroy(m->ev_enc_ready);
  if (m->ev_ts_in) (void)hipEventDestroy(m->ev_ts_in);
  if (m->ev_ts_out) (void)hipEventDestroy(m->ev_ts_out);
  if (m->blstm_stream) (void)hipStreamDestroy(m->blstm_stream);
  for (hipEvent_t e : m->ev_kv) if (e) (void)hipEventDestroy(e);
  delete m;
}

int pfhip_sample_rate(const pfhip_model* m) { return m ? m->cfg.sample_rate : 0; }
int pfhip_vocab_size(const pfhip_model* m) { return m ? m->cfg.vocab : 0; }
int pfhip_feat_dim(const pfhip_model* m) { return m ? m->feat_dim : 0; }
int pfhip_d_model(const pfhip_model* m) { return m ? m->cfg.d_model : 0; }

pfhip_status pfhip_offline_enqueue(pfhip_model* m, const float* d_pcm, const int64_t* sample_off, const int* n_samples,
                                   int batch, void* stream) {
  g_err.clear();
  if (!m || !sample_off || !n_samples || batch <= 0) return fail(PFHIP_ERR_ARG, "bad argument");
  if (!d_pcm) return fail(PFHIP_ERR_ARG, "null device pcm");
  std::lock_guard<std::mutex> lk(m->mu);
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : m->own_stream;
  m->prof_stream = s;
  pfhip_status st = enqueue_locked(m, d_pcm, sample_off, n_samples, batch, s, false);
  if (st) return st;
  return head_locked(m, s, false);
}

pfhip_status pfhip_offline_fetch(pfhip_model* m, pfhip_out* out) {
  g_err.clear();
  if (!m) return fail(PFHIP_ERR_ARG, "null model");
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  return fetch_locked(m, out, m->prof_stream ? m->prof_stream : m->own_stream);
}

static pfhip_status forward_direct(pfhip_model* m, const float* const* pcm, const int* n_samples, int batch,
                                   const float* hw_emb, int n_hotwords, pfhip_out* out) {
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t s = m->own_stream;
  m->prof_stream = s;
  if (m->cfg.contextual) {          // plain models ignore hw_emb (paraformer.cpp:515: use_hotword == false)
    if (!hw_emb || n_hotwords <= 0) return fail(PFHIP_ERR_ARG, "hw_emb is null");          // paraformer.cpp:516-520
    pfhip_status hs = set_hotwords_locked(m, hw_emb, n_hotwords, s);
    if (hs) return hs;
  }
  std::vector<int64_t> off;
  pfhip_status st = stage_pcm(m, pcm, n_samples, batch, s, off);
  if (st) return st;
  st = enqueue_locked(m, m->pcm.f(), off.data(), n_samples, batch, s, false);
  if (st) return st;
  st = head_locked(m, s, out->logp != nullptr);
  if (st) return st;
  return fetch_locked(m, out, s);
}

// ---- execution slots -----------------------------------------------------------------------------------------------
// A handle fronts one or more execution slots: the contexts of its device (pfhip_set_inflight) and of every replica device
// (pfhip_create_group).  `slots` on the head lists them device-major per round (context 0 of every device, then context 1 of
// every device, ...), so that filling them in order spreads calls over the GPUs before it stacks them on one.
static void rebuild_slots_locked(pfhip_model* head) {
  head->slots.clear();
  std::vector<pfhip_model*> devs{head};
  for (pfhip_model* r : head->replicas) devs.push_back(r);
  size_t rounds = 0;
  for (pfhip_model* d : devs) rounds = std::max(rounds, d->contexts.size() + 1);
  for (size_t k = 0; k < rounds; ++k)
    for (pfhip_model* d : devs) {
      if ((int)k >= head->ctx_limit) continue;
      if (k == 0) head->slots.push_back(d);
      else if (k - 1 < d->contexts.size()) head->slots.push_back(d->contexts[k - 1]);
    }
}

// least-loaded slot (ties: round-robin); used by calls that bypass the queue
static pfhip_model* acquire_slot(pfhip_model* head) {
  std::lock_guard<std::mutex> l(head->bq.mu);
  if (head->slots.empty()) rebuild_slots_locked(head);
  const size_t n = head->slots.size();
  const unsigned start = head->rr.fetch_add(1);
  pfhip_model* best = nullptr;
  int best_load = 0;
  for (size_t k = 0; k < n; ++k) {
    pfhip_model* r = head->slots[(start + k) % n];
    const int load = r->inflight.load();
    if (!best || load < best_load) { best = r; best_load = load; }
  }
  ++best->inflight;
  return best;
}
static void release_slot(pfhip_model* head, pfhip_model* slot) {
  --slot->inflight;
  head->bq.slot_freed();
}

// ---- cross-request batching ---------------------------------------------------------------------------------
// The reference server runs `decoder-thread-num` threads that each call Forward on the shared handle with their own
// request (websocket/bin/funasr-wss-server.cpp:479-481); its dynamic batcher only groups segments of ONE request
// (audio.cpp:1056-1084).  On one MI355X a single 60-s segment fills 1/16 of the GEMM grid, so concurrent callers are
// merged here: callers queue at the handle, the one at the front leads — claims an idle execution slot, gathers company
// (PoolQueue: no wait at all when nothing else is in flight), runs ONE packed forward for the utterances it took and hands
// every caller its own slice — while the next caller in line already gathers the next batch for the next idle slot.
// Results are those of separate calls up to the near-tie statement the tests make (packed layout, per-utterance masks:
// tests/test_gpu_forward.py::test_batch_composition_invariance — a merged forward may run another kernel family than a lone one,
// the same sums in another order, 1e-6 apart in the log-probabilities).
struct BatchReq : pfhip_detail::MergeReqBase {
  const float* const* pcm; const int* n; int batch; pfhip_out* out;
  pfhip_status st = PFHIP_OK; std::string err;
};

// one packed forward on `m` for everybody in `take`
static void run_batch_requests(pfhip_model* m, const std::vector<BatchReq*>& take) {
  std::vector<const float*> ptrs; std::vector<int> lens;
  bool want_logp = false, want_us = false; int max_tok = 1, utts = 0;
  for (BatchReq* r : take) {
    for (int i = 0; i < r->batch; ++i) { ptrs.push_back(r->pcm[i]); lens.push_back(r->n[i]); max_tok = std::max(max_tok, r->n[i] / 960 + 2); }
    want_logp = want_logp || r->out->logp != nullptr;
    want_us = want_us || r->out->us_alphas || r->out->us_peaks || r->out->us_len;
    utts += r->batch;
  }
  const int V = m->cfg.vocab, max_us = 3 * max_tok;
  std::vector<int32_t> ids((size_t)utts * max_tok), tn(utts), nf(utts), fr(utts), usl;
  std::vector<float> logp, usa, usp;
  if (want_logp) logp.resize((size_t)utts * max_tok * V);
  pfhip_out all{};
  all.token_ids = ids.data(); all.token_num = tn.data(); all.n_fires = nf.data(); all.n_frames = fr.data();
  all.logp = want_logp ? logp.data() : nullptr; all.max_tokens = max_tok;
  if (want_us) {
    usa.resize((size_t)utts * max_us); usp.resize((size_t)utts * max_us); usl.resize(utts);
    all.us_alphas = usa.data(); all.us_peaks = usp.data(); all.us_len = usl.data(); all.max_us = max_us;
  }
  pfhip_status st = forward_direct(m, ptrs.data(), lens.data(), utts, nullptr, 0, &all);
  const std::string err = g_err;
  int u0 = 0;
  for (BatchReq* r : take) {
    r->st = st; r->err = err;
    for (int i = 0; i < r->batch && st == PFHIP_OK; ++i) {
      const int u = u0 + i;
      pfhip_out* o = r->out;
      if (o->token_num) o->token_num[i] = tn[u];
      if (o->n_fires) o->n_fires[i] = nf[u];
      if (o->n_frames) o->n_frames[i] = fr[u];
      if ((o->token_ids || o->logp) && o->max_tokens < nf[u]) {
        r->st = PFHIP_ERR_CAPACITY; r->err = "max_tokens smaller than the longest token sequence"; break;
      }
      if (o->token_ids) std::memcpy(o->token_ids + (size_t)i * o->max_tokens, ids.data() + (size_t)u * max_tok, 4 * (size_t)nf[u]);
      if (o->logp) std::memcpy(o->logp + (size_t)i * o->max_tokens * V, logp.data() + (size_t)u * max_tok * V, 4 * (size_t)nf[u] * V);
      if (o->us_alphas || o->us_peaks || o->us_len) {
        if (o->us_len) o->us_len[i] = usl[u];
        if ((o->us_alphas || o->us_peaks) && o->max_us < usl[u]) { r->st = PFHIP_ERR_CAPACITY; r->err = "max_us smaller than 3 x frames"; break; }
        if (o->us_alphas) std::memcpy(o->us_alphas + (size_t)i * o->max_us, usa.data() + (size_t)u * max_us, 4 * (size_t)usl[u]);
        if (o->us_peaks) std::memcpy(o->us_peaks + (size_t)i * o->max_us, usp.data() + (size_t)u * max_us, 4 * (size_t)usl[u]);
      }
    }
    u0 += r->batch;
  }
}

namespace { thread_local pfhip_model* tl_last_replica = nullptr; }      // where this thread's last offline forward ran (debug getters)

static pfhip_status forward_batched(pfhip_model* head, const float* const* pcm, const int* n_samples, int batch,
                                    pfhip_out* out) {
  BatchReq me;
  me.pcm = pcm; me.n = n_samples; me.batch = batch; me.out = out;
  int wait_us, max_utts;
  { std::lock_guard<std::mutex> l(head->bq.mu); wait_us = head->batch_wait_us; max_utts = head->batch_max_utts; }
  pfhip_model* ran_on = nullptr;
  head->bq.submit(
      me, wait_us,
      // claim (under the queue lock): an idle slot, in the device-major order of `slots`
      [&]() -> pfhip_model* {
        if (head->slots.empty()) rebuild_slots_locked(head);
        for (pfhip_model* r : head->slots)
          if (r->inflight.load() == 0) { ++r->inflight; return r; }
        return nullptr;
      },
      [&](pfhip_model* r) { --r->inflight; },
      // gather: bounded by time and by utterance count
      [&](const std::deque<BatchReq*>& q) { int u = 0; for (BatchReq* r : q) u += r->batch; return u >= max_utts; },
      [&](std::deque<BatchReq*>& q, std::vector<BatchReq*>& take) {
        int utts = 0;
        while (!q.empty() && (take.empty() || utts + q.front()->batch <= max_utts)) {
          utts += q.front()->batch;
          take.push_back(q.front());
          q.pop_front();
        }
      },
      [&](pfhip_model* r, std::vector<BatchReq*>& take) {
        ran_on = r;
        int utts = 0;
        for (BatchReq* q : take) utts += q->batch;
        run_batch_requests(r, take);
        ++r->served_forwards; r->served_calls += (int64_t)take.size(); r->served_utts += utts;
      });
  if (ran_on) tl_last_replica = ran_on;       // the leader's own slice ran there; followers ask the handle (pfhip_get_tensor: head)
  if (me.st != PFHIP_OK) g_err = me.err;
  return me.st;
}

pfhip_status pfhip_offline_forward(pfhip_model* head, const float* const* pcm, const int* n_samples, int batch,
                                   const float* hw_emb, int n_hotwords, pfhip_out* out) {
  g_err.clear();
  if (!head || !pcm || !n_samples || batch <= 0 || !out) return fail(PFHIP_ERR_ARG, "bad argument");
  for (int i = 0; i < batch; ++i)
    if (n_samples[i] < 0 || (n_samples[i] > 0 && !pcm[i])) return fail(PFHIP_ERR_ARG, "bad pcm buffer");
  // merged with whoever else is calling (plain and timestamp models; hotwords are per connection, so contextual calls are not)
  bool merge;
  { std::lock_guard<std::mutex> l(head->bq.mu); merge = head->batch_wait_us > 0 && batch < head->batch_max_utts; }
  if (merge && !head->cfg.contextual) return forward_batched(head, pcm, n_samples, batch, out);
  pfhip_model* m = acquire_slot(head);                  // the least-loaded execution slot (context / GPU)
  tl_last_replica = m;
  const pfhip_status st = forward_direct(m, pcm, n_samples, batch, hw_emb, n_hotwords, out);
  ++m->served_forwards; ++m->served_calls; m->served_utts += batch;
  release_slot(head, m);
  return st;
}

// pfhip_offline_forward with the PCM already in HBM: same routing over the execution slots, no H2D of the audio
pfhip_status pfhip_offline_forward_resident(pfhip_model* head, const float* d_pcm, const int64_t* sample_off, const int* n_samples,
                                            int batch, pfhip_out* out) {
  g_err.clear();
  if (!head || !d_pcm || !sample_off || !n_samples || batch <= 0 || !out) return fail(PFHIP_ERR_ARG, "bad argument");
  if (head->cfg.contextual && head->n_hw <= 0) return fail(PFHIP_ERR_ARG, "hw_emb is null");
  pfhip_model* m = acquire_slot(head);
  tl_last_replica = m;
  pfhip_status st;
  {
    std::lock_guard<std::mutex> lk(m->mu);
    hipStream_t s = m->own_stream;
    m->prof_stream = s;
    st = enqueue_locked(m, d_pcm, sample_off, n_samples, batch, s, false);
    if (!st) st = head_locked(m, s, out->logp != nullptr);
    if (!st) st = fetch_locked(m, out, s);
  }
  ++m->served_forwards; ++m->served_calls; m->served_utts += batch;
  release_slot(head, m);
  return st;
}

pfhip_status pfhip_set_batching(pfhip_model* m, int wait_us, int max_utterances) {
  g_err.clear();
  if (!m || wait_us < 0 || max_utterances < 1) return fail(PFHIP_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> ql(m->bq.mu);
  m->batch_wait_us = wait_us;
  m->batch_max_utts = max_utterances;
  return PFHIP_OK;
}

// every execution slot of the handle, for calls that configure all of them
static std::vector<pfhip_model*> all_slots(pfhip_model* head) {
  std::lock_guard<std::mutex> l(head->bq.mu);
  if (head->slots.empty()) rebuild_slots_locked(head);
  return head->slots;
}

pfhip_status pfhip_set_inflight(pfhip_model* head, int n) {
  g_err.clear();
  if (!head || n < 1 || n > 16) return fail(PFHIP_ERR_ARG, "in-flight count outside 1..16");
  if (head->group_head || head->weights_of) return fail(PFHIP_ERR_ARG, "not the handle pfhip_create returned");
  std::vector<pfhip_model*> devs{head};
  for (pfhip_model* r : head->replicas) devs.push_back(r);
  std::vector<float> hw;
  { std::lock_guard<std::mutex> l(head->bq.mu); hw = head->hw_host; }
  for (pfhip_model* d : devs) {
    while ((int)d->contexts.size() + 1 < n) {
      pfhip_model* cx = nullptr;
      pfhip_status st = build_context(d, &cx);
      if (st) return st;
      cx->group_head = head;
      cx->ctx_index = (int)d->contexts.size() + 1;
      if (!hw.empty()) {
        std::lock_guard<std::mutex> lk(cx->mu);
        st = set_hotwords_locked(cx, hw.data(), (int)(hw.size() / (size_t)head->cfg.d_model), cx->own_stream);
        if (st) { pfhip_destroy(cx); return st; }
      }
      std::lock_guard<std::mutex> l(head->bq.mu);
      d->contexts.push_back(cx);
    }
  }
  // fewer than before: the extra contexts stay allocated (a call may be running on one) but leave the slot list
  std::lock_guard<std::mutex> l(head->bq.mu);
  head->ctx_limit = n;
  rebuild_slots_locked(head);
  return PFHIP_OK;
}

int pfhip_get_inflight(const pfhip_model* m) {
  if (!m) return 0;
  pfhip_model* head = const_cast<pfhip_model*>(m);
  std::lock_guard<std::mutex> l(head->bq.mu);
  if (head->slots.empty()) rebuild_slots_locked(head);
  int n = 0;
  for (pfhip_model* sl : head->slots) n = std::max(n, sl->ctx_index + 1);
  return n;
}

pfhip_status pfhip_inflight_stats(pfhip_model* head, pfhip_slot_stats* out, int cap, int* n_out) {
  g_err.clear();
  if (!head || !n_out) return fail(PFHIP_ERR_ARG, "bad argument");
  const std::vector<pfhip_model*> sl = all_slots(head);
  *n_out = (int)sl.size();
  if (!out || cap < (int)sl.size()) return fail(PFHIP_ERR_CAPACITY, "slot array too small");
  for (size_t i = 0; i < sl.size(); ++i) {
    out[i].device = sl[i]->device; out[i].context = sl[i]->ctx_index;
    out[i].forwards = sl[i]->served_forwards.load(); out[i].calls = sl[i]->served_calls.load();
    out[i].utterances = sl[i]->served_utts.load();
  }
  return PFHIP_OK;
}

pfhip_status pfhip_set_hotwords(pfhip_model* m, const float* hw_emb, int n_hotwords) {
  g_err.clear();
  if (!m || !hw_emb || n_hotwords <= 0) return fail(PFHIP_ERR_ARG, "bad argument");
  // every context of every device, also those pfhip_set_inflight has taken off the slot list (they come back with a larger n)
  std::vector<pfhip_model*> every;
  {
    std::lock_guard<std::mutex> l(m->bq.mu);
    std::vector<pfhip_model*> devs{m};
    for (pfhip_model* r : m->replicas) devs.push_back(r);
    for (pfhip_model* d : devs) {
      every.push_back(d);
      for (pfhip_model* cx : d->contexts) every.push_back(cx);
    }
  }
  for (pfhip_model* r : every) {
    std::lock_guard<std::mutex> lk(r->mu);
    HIP_TRY(hipSetDevice(r->device));
    pfhip_status st = set_hotwords_locked(r, hw_emb, n_hotwords, r->own_stream);
    if (st) return st;
  }
  std::lock_guard<std::mutex> l(m->bq.mu);
  m->hw_host.assign(hw_emb, hw_emb + (size_t)n_hotwords * m->cfg.d_model);
  return PFHIP_OK;
}

// One synthetic batch through every execution slot: the code objects are loaded, each context's workspace is sized for
// `batch` utterances of `n_samples` samples and its streams have run once, so the first real request pays none of that.
pfhip_status pfhip_warm_up(pfhip_model* m, int batch, int n_samples) {
  g_err.clear();
  if (!m || batch <= 0 || n_samples <= 0 || batch > 1024 || n_samples > 16000 * 120) return fail(PFHIP_ERR_ARG, "bad argument");
  if (m->group_head || m->weights_of) return fail(PFHIP_ERR_ARG, "not the handle pfhip_create returned");
  std::vector<float> pcm((size_t)n_samples);
  uint32_t lcg = 20251114u;
  for (int i = 0; i < n_samples; ++i) {
    lcg = lcg * 1664525u + 1013904223u;
    pcm[i] = 0.15f * sinf(0.0431969f * (float)i) + 0.1f * ((float)(lcg >> 8) / 8388608.f - 1.f);
  }
  std::vector<const float*> ptrs((size_t)batch, pcm.data());
  std::vector<int> lens((size_t)batch, n_samples);
  const int max_tok = n_samples / 960 + 2;
  std::vector<int32_t> ids((size_t)batch * max_tok), tn(batch), nf(batch), fr(batch);
  std::vector<float> hw(m->cfg.contextual ? (size_t)m->cfg.d_model : 0, 0.f);
  for (pfhip_model* r : all_slots(m)) {
    pfhip_out out{};
    out.token_ids = ids.data(); out.token_num = tn.data(); out.n_fires = nf.data(); out.n_frames = fr.data(); out.max_tokens = max_tok;
    const pfhip_status st = forward_direct(r, ptrs.data(), lens.data(), batch, hw.empty() ? nullptr : hw.data(), hw.empty() ? 0 : 1, &out);
    if (st) return st;
  }
  return PFHIP_OK;
}

int pfhip_is_contextual(const pfhip_model* m) { return m ? m->cfg.contextual : 0; }
int pfhip_has_timestamp_head(const pfhip_model* m) { return m ? m->cfg.timestamp : 0; }

// model_eb.onnx Run + row selection (paraformer.cpp:656-685): Embedding -> 1-layer LSTM over the 10 padded positions,
// output of hotword j taken at step lengths[j]-1.
pfhip_status pfhip_hotword_embed(pfhip_model* m, const int32_t* hotword_matrix, const int32_t* lengths, int n_hotwords,
                                 float* out) {
  g_err.clear();
  if (!m || !hotword_matrix || !lengths || n_hotwords <= 0 || !out) return fail(PFHIP_ERR_ARG, "bad argument");
  if (!m->cfg.contextual) return fail(PFHIP_ERR_UNSUPPORTED, "model has no hotword embedder (use_hotword == false)");
  const int H = n_hotwords, L = 10, d = m->cfg.d_model;
  for (int j = 0; j < H; ++j) {
    if (lengths[j] < 1 || lengths[j] > L) return fail(PFHIP_ERR_ARG, "hotword length outside 1..10");
    for (int t = 0; t < L; ++t)
      if (hotword_matrix[j * L + t] < 0 || hotword_matrix[j * L + t] >= m->cfg.vocab) return fail(PFHIP_ERR_ARG, "hotword id outside the vocabulary");
  }
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t s = m->own_stream;
  const int R = L * H, Rp = round_up(R, pfhip::kTileM), Hp = round_up(H, pfhip::kTileM);
  Buf ids, lens, X, GX, G, hc;
  struct Free { Buf* b[6]; ~Free() { for (Buf* x : b) x->release(); } } fr{{&ids, &lens, &X, &GX, &G, &hc}};
  HIP_TRY(ids.ensure((size_t)R * 4)); HIP_TRY(lens.ensure((size_t)H * 4));
  HIP_TRY(X.ensure((size_t)Rp * d * 4)); HIP_TRY(GX.ensure((size_t)(Rp + pfhip::kTileM) * 4 * d * 4)); HIP_TRY(G.ensure((size_t)Hp * 4 * d * 4));
  HIP_TRY(hc.ensure((size_t)3 * Hp * d * 4));
  // time-major ids: row t*H + j = token t of hotword j
  std::vector<int32_t> tm((size_t)R);
  for (int j = 0; j < H; ++j) for (int t = 0; t < L; ++t) tm[(size_t)t * H + j] = hotword_matrix[j * L + t];
  HIP_TRY(hipMemcpyAsync(ids.p, tm.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(lens.p, lengths, (size_t)H * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(hc.p, 0, (size_t)3 * Hp * d * 4, s));
  float* h = hc.f(); float* cst = h + (size_t)Hp * d; float* sel = cst + (size_t)Hp * d;
  pfhip::launch_gather_rows(static_cast<const int32_t*>(ids.p), m->W("bias.embed.w").d, d, X.f(), R, s);
  gemm(m, s, X.f(), d, m->W("bias.lstm.w_ih").d, 4 * d, d, d, GX.f(), 4 * d, m->W("bias.lstm.b_ih").d, nullptr, 0, nullptr, 0, R, false);
  for (int t = 0; t < L; ++t) {
    gemm(m, s, h, d, m->W("bias.lstm.w_hh").d, 4 * d, d, d, G.f(), 4 * d, m->W("bias.lstm.b_hh").d, GX.f() + (size_t)t * H * 4 * d,
         4 * d, nullptr, 0, H, false);
    pfhip::launch_lstm_cell(G.f(), cst, h, static_cast<const int32_t*>(lens.p), t, sel, H, d, s);
  }
  HIP_TRY(hipMemcpyAsync(out, sel, (size_t)H * d * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  HIP_TRY(hipGetLastError());
  return PFHIP_OK;
}

pfhip_status pfhip_extract_feats(pfhip_model* m, const float* const* pcm, const int* n_samples, int batch,
                                 float* feats_out, size_t feats_cap_floats, int32_t* n_frames_out) {
  g_err.clear();
  if (!m || !pcm || !n_samples || batch <= 0) return fail(PFHIP_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t s = m->own_stream;
  m->prof_stream = s;
  std::vector<int64_t> off;
  pfhip_status st = stage_pcm(m, pcm, n_samples, batch, s, off);
  if (st) return st;
  st = enqueue_locked(m, m->pcm.f(), off.data(), n_samples, batch, s, true);
  if (st) return st;
  if (n_frames_out) for (int b = 0; b < batch; ++b) n_frames_out[b] = m->T[b];
  const size_t n = (size_t)m->M * m->feat_dim;
  if (feats_out) {
    if (n > feats_cap_floats) return fail(PFHIP_ERR_CAPACITY, "feats_out too small");
    if (n) HIP_TRY(hipMemcpyAsync(feats_out, m->feats.p, n * 4, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return PFHIP_OK;
}

pfhip_status pfhip_get_tensor(pfhip_model* m, const char* name, float* dst, size_t cap_floats, size_t* n_out) {
  g_err.clear();
  if (!m || !name || !dst) return fail(PFHIP_ERR_ARG, "bad argument");
  // a group: the replica that served this thread's last offline forward holds the state being asked for
  if (!m->replicas.empty() && tl_last_replica && (tl_last_replica == m || tl_last_replica->group_head == m)) m = tl_last_replica;
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t s = m->prof_stream ? m->prof_stream : m->own_stream;
  const std::string nm(name);
  const void* src = nullptr;
  size_t n = 0;
  const int d = m->cfg.d_model;
  if (nm == "feats") { src = m->feats.p; n = (size_t)m->M * m->feat_dim; }
  else if (nm == "enc") { src = m->enc.p; n = (size_t)m->M * d; }
  else if (nm == "alphas") { src = m->alphas.p; n = (size_t)m->M; }
  else if (nm == "emb") { src = m->emb.p; n = (size_t)m->ML * d; }
  else if (nm == "ts_up" && m->have_ts) { src = m->ts_up.p; n = (size_t)3 * m->M * d; }            // timestamp-head stages
  else if (nm == "ts_gx" && m->have_ts) { src = m->ts_gx.p; n = (size_t)3 * m->M * 8 * d; }
  else if (nm == "ts_y" && m->have_ts) { src = m->ts_y.p; n = (size_t)3 * m->M * 2 * d; }
  else if (nm == "logp") {
    if (m->ML && !m->have_logp) { pfhip_status st = head_locked(m, s, true); if (st) return st; }
    src = m->logp.p; n = (size_t)m->ML * m->cfg.vocab;
  } else return fail(PFHIP_ERR_ARG, "unknown tensor name " + nm);
  if (n > cap_floats) return fail(PFHIP_ERR_CAPACITY, "dst too small");
  if (n) HIP_TRY(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n_out) *n_out = n;
  return PFHIP_OK;
}

// Test hook.  "blstm_flag" != 0: the next timestamp request finds the BLSTM error word raised (as after a step-barrier
// time-out) and must fall back to the per-step recurrence; "blstm_fallbacks" returns how often that happened (as the status).
pfhip_status pfhip_debug_poke(pfhip_model* m, const char* what, int value) {
  g_err.clear();
  if (!m || !what) return fail(PFHIP_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> lk(m->mu);
  if (std::string(what) == "blstm_flag") { m->debug_blstm_flag = value; return PFHIP_OK; }
  if (std::string(what) == "blstm_fallbacks") {     // read-out: how often the per-step form ran, over every context of the handle
    long long n = m->blstm_fallbacks;
    for (pfhip_model* cx : m->contexts) n += cx->blstm_fallbacks;
    for (pfhip_model* r : m->replicas) { n += r->blstm_fallbacks; for (pfhip_model* cx : r->contexts) n += cx->blstm_fallbacks; }
    return (pfhip_status)n;
  }
  if (std::string(what) == "plane_forwards") return (pfhip_status)m->plane_forwards;        // read-out: forwards on plane-image operands
  if (std::string(what) == "kvplane_forwards") return (pfhip_status)m->kvplane_forwards;     // ... whose attention took K | V as planes
  if (std::string(what) == "dec_plane_forwards") return (pfhip_status)m->dec_plane_forwards; // ... whose decoder took the plane path too
  if (std::string(what) == "static_bound") return (pfhip_status)std::min(m->static_bound, 2.0e9);      // read-out: the load-time activation bound
  if (std::string(what) == "always_exact") return (pfhip_status)(m->always_exact ? 1 : 0);
  if (std::string(what) == "range_flag") { m->debug_range_flag = value; return PFHIP_OK; }    // the next forward starts with its range flag raised
  if (std::string(what) == "range_fallbacks") {       // read-out: forwards redone on the exact kernels, over every context of the handle
    long long n = m->range_fallbacks;
    for (pfhip_model* cx : m->contexts) n += cx->range_fallbacks;
    for (pfhip_model* r : m->replicas) { n += r->range_fallbacks; for (pfhip_model* cx : r->contexts) n += cx->range_fallbacks; }
    return (pfhip_status)n;
  }
  return fail(PFHIP_ERR_ARG, std::string("unknown debug key ") + what);
}

pfhip_status pfhip_profile_enable(pfhip_model* m, int on) {
  if (!m) return fail(PFHIP_ERR_ARG, "null model");
  std::lock_guard<std::mutex> lk(m->mu);
  m->prof_mask = on < 0 ? 0 : (on == 1 ? 0xff : on);   // 0 off, 1 all classes, else a bit mask of classes << 0
  return PFHIP_OK;
}

pfhip_status pfhip_profile_read(pfhip_model* m, pfhip_profile* out, int reset) {
  g_err.clear();
  if (!m || !out) return fail(PFHIP_ERR_ARG, "bad argument");
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  if (m->prof_stream) HIP_TRY(hipStreamSynchronize(m->prof_stream));
  for (const ProfRec& r : m->prof_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) m->prof.ms[r.cls] += ms;
  }
  m->prof_recs.clear();
  m->ev_used = 0;
  *out = m->prof;
  if (reset) m->prof = pfhip_profile{};
  return PFHIP_OK;
}

}  // extern "C"
