// Chunk-streaming Paraformer on MI355X — the C ABI's pfhip_stream_* family.
//
// Mirrors `funasr::ParaformerOnline` (onnxruntime/src/paraformer-online.cpp) statement by statement for the
// host-side control flow (sample cache, LFR splice cache, [5|10|5] overlap window, first/last-chunk split,
// resets), with every buffer the reference keeps in std::vectors held in HBM per connection instead:
//   fb        fbank frames = lfr_splice_cache_ ++ new frames          (:155-177, 226-228)
//   featc     feats_cache_: the 10 (or 5) scaled+PE'd rows carried into the next window (:397-413)
//   carry_*   CIF hidden_cache_/alphas_cache_                          (:288-293, 329-340)
//   dcache    16 decoder FSMN caches [10][512] (time-major view of the reference's [1,512,10], :374)
// The host keeps only counters (cache lengths, start_idx_cache_, first/last flags) and the < 400 left-over
// PCM samples of input_cache_ (:123-127).  `reserve_waveforms_` is dead state in the reference (it only feeds
// its own index arithmetic, :162-171,180-182) and is not kept.
#include <map>
#include <memory>

#include "internal.h"

using namespace pfhip_detail;

struct pfhip_stream {
  pfhip_model* m = nullptr;
  int chunk_size[3] = {5, 10, 5};
  // host state
  std::vector<float> input_cache;
  int n_splice = 0;            // frames in the splice cache (front of fb[cur])
  int start_idx = 0;           // start_idx_cache_
  bool is_first_chunk = true, is_last_chunk = false;
  int n_featc = 10;            // rows in feats_cache_
  // device state
  Buf pcm, fb[2], rows, featc, chunk, enc, alphas, carry, emb, nfire, dcache, meta, ids, logp;
  int fb_cur = 0;
  int* h_pin = nullptr;        // pinned: [0] n_fire, [1..] ids
  // last chunk (inspection)
  int last_n = 0, last_fires = 0;
  bool last_has_logp = false;
  bool debug = false;
  // hipGraph cache of the two launch sequences of a chunk (encoder+CIF keyed by window rows / last flag, decoder
  // keyed by fired tokens and window rows): a chunk is ~700 launches and launch-bound when issued one by one
  std::map<int, hipGraphExec_t> graphs;
  uint64_t graphs_epoch = 0;
  bool use_graphs = true;
};

namespace {

constexpr int kMaxSamples = 32000;   // per call (2 s); the 2-pass server sends 9600 (websocket-server-2pass.cpp:135-148)
constexpr int kMaxFrames = 256;
constexpr int kMaxRows = 64;         // rows of one encoder window
constexpr int kMaxTok = 72;

pfhip_status stream_alloc(pfhip_stream* s) {
  pfhip_model* m = s->m;
  const int d = m->cfg.d_model, FD = m->feat_dim, FP = m->feat_pad;
  HIP_TRY(s->pcm.ensure((size_t)(kMaxSamples + 1024) * 4));
  for (int i = 0; i < 2; ++i) HIP_TRY(s->fb[i].ensure((size_t)kMaxFrames * 80 * 4));
  HIP_TRY(s->rows.ensure((size_t)kMaxRows * FD * 4));
  HIP_TRY(s->featc.ensure((size_t)16 * FD * 4));
  HIP_TRY(s->chunk.ensure((size_t)128 * FP * 4));
  HIP_TRY(s->enc.ensure((size_t)128 * d * 4));
  HIP_TRY(s->alphas.ensure((size_t)128 * 4));
  HIP_TRY(s->carry.ensure((size_t)(d + 4) * 4));
  HIP_TRY(s->emb.ensure((size_t)128 * d * 4));
  HIP_TRY(s->nfire.ensure(64));
  HIP_TRY(s->dcache.ensure((size_t)std::max(1, m->cfg.dec_layers) * 10 * d * 4));
  HIP_TRY(s->meta.ensure(256));
  HIP_TRY(s->ids.ensure((size_t)128 * 4));
  HIP_TRY(hipHostMalloc((void**)&s->h_pin, 4096, hipHostMallocDefault));
  return PFHIP_OK;
}

// InitCache (paraformer-online.cpp:347-384)
pfhip_status init_cache(pfhip_stream* s, hipStream_t st) {
  pfhip_model* m = s->m;
  s->start_idx = 0;
  s->is_first_chunk = true;
  s->is_last_chunk = false;
  s->n_featc = s->chunk_size[0] + s->chunk_size[2];
  HIP_TRY(hipMemsetAsync(s->carry.p, 0, (size_t)(m->cfg.d_model + 4) * 4, st));
  HIP_TRY(hipMemsetAsync(s->featc.p, 0, (size_t)16 * m->feat_dim * 4, st));
  HIP_TRY(hipMemsetAsync(s->dcache.p, 0, (size_t)std::max(1, m->cfg.dec_layers) * 10 * m->cfg.d_model * 4, st));
  return PFHIP_OK;
}

// ResetCache (:391-395)
void reset_cache(pfhip_stream* s) {
  s->input_cache.clear();
  s->n_splice = 0;
}

void drop_graphs(pfhip_stream* s) {
  for (auto& kv : s->graphs) (void)hipGraphExecDestroy(kv.second);
  s->graphs.clear();
}

// Runs `enqueue` on `st`, through a cached hipGraph when allowed: first use of a key captures the launches,
// later uses replay them (one host call instead of hundreds).
template <typename F>
pfhip_status run_cached(pfhip_stream* s, int key, bool allow, hipStream_t st, F&& enqueue) {
  if (!allow) return enqueue();
  if (s->graphs_epoch != buf_epoch().load()) { drop_graphs(s); s->graphs_epoch = buf_epoch().load(); }
  auto it = s->graphs.find(key);
  if (it == s->graphs.end()) {
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    pfhip_status rc = enqueue();
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(PFHIP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    hipGraphExec_t ge = nullptr;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(PFHIP_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    it = s->graphs.emplace(key, ge).first;
  }
  HIP_TRY(hipGraphLaunch(it->second, st));
  return PFHIP_OK;
}

// ForwardChunk (:415-523) on the window already assembled in s->chunk (n rows).  Appends ids to `out`.
pfhip_status forward_chunk(pfhip_stream* s, int n, hipStream_t st, std::vector<int32_t>& out, bool want_logp) {
  pfhip_model* m = s->m;
  const Config& c = m->cfg;
  const int d = c.d_model, FD = m->feat_dim, FP = m->feat_pad;
  const float att_scale = 1.0f / sqrtf((float)pfhip::kHeadDim);
  s->last_n = n; s->last_fires = 0; s->last_has_logp = false;
  if (n <= 0 || n > 128) return fail(PFHIP_ERR_ARG, "stream window out of range");
  // workspace of the model (serialised by the model lock)
  HIP_TRY(m->y.ensure((size_t)128 * FP * 4));
  HIP_TRY(m->x.ensure((size_t)128 * d * 4));
  HIP_TRY(m->qkv.ensure((size_t)128 * 3 * d * 4));
  HIP_TRY(m->mem.ensure((size_t)128 * d * 4));
  HIP_TRY(m->ctx.ensure((size_t)128 * d * 4));
  HIP_TRY(m->hbuf.ensure((size_t)256 * std::max(c.ffn, d) * 4));
  HIP_TRY(m->yd.ensure((size_t)128 * d * 4));
  HIP_TRY(m->hd.ensure((size_t)128 * c.dec_ffn * 4));
  HIP_TRY(m->hd2.ensure((size_t)128 * c.dec_ffn * 4));
  HIP_TRY(m->td.ensure((size_t)128 * d * 4));
  HIP_TRY(m->t2.ensure((size_t)128 * d * 4));
  HIP_TRY(m->qd.ensure((size_t)128 * d * 4));
  HIP_TRY(m->ctxd.ensure((size_t)128 * d * 4));
  HIP_TRY(m->xd.ensure((size_t)128 * d * 4));
  HIP_TRY(m->logits.ensure((size_t)128 * m->vocab_pad * 4));
  if (want_logp) HIP_TRY(s->logp.ensure((size_t)128 * c.vocab * 4));
  const bool graphs = s->use_graphs && !want_logp && m->prof_mask == 0;
  // device metadata: [0] off=0, [1] len=n (encoder rows), [2] tok_len (set after CIF), [16..] row_pos[n], row_len[n]
  int* dm = s->meta.i();
  int* d_off = dm; int* d_len = dm + 1; int* d_tok = dm + 2;
  {   // pinned staging is filled BEFORE the (possibly replayed) copies read it
    int* hm = s->h_pin + 256;
    hm[0] = 0; hm[1] = n; hm[2] = 0;
    int* hr = s->h_pin + 512;
    for (int t = 0; t < n; ++t) { hr[t] = t; hr[128 + t] = n; }
  }
  const int is_last = s->is_last_chunk ? 1 : 0;
  float* x = m->x.f();

  pfhip_status rc = run_cached(s, (n << 1) | is_last, graphs, st, [&]() -> pfhip_status {
    HIP_TRY(hipMemcpyAsync(dm, s->h_pin + 256, 12, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dm + 16, s->h_pin + 512, 256 * 4, hipMemcpyHostToDevice, st));
    // ---- streaming encoder session (:448): SAN-M stack on the window as given (no scale/PE inside) --------
    for (int i = 0; i < c.enc_layers; ++i) {
      const std::string p = "enc." + std::to_string(i) + ".";
      const bool first = i == 0;
      const float* xin = first ? s->chunk.f() : x;
      const int ldin = first ? FP : d, Din = first ? FD : d, Kp = first ? FP : d;
      lnorm(m, st, xin, ldin, m->y.f(), Kp, p + "norm1", n, Din, Kp);
      gemm(m, st, m->y.f(), Kp, first ? m->d_w0qkv : m->W(p + "qkv.w").d, 3 * d, Kp, Din, m->qkv.f(), 3 * d,
           m->W(p + "qkv.b").d, nullptr, 0, nullptr, 0, n, false);
      pfhip::launch_fsmn(m->qkv.f() + 2 * d, 3 * d, m->W(p + "fsmn.w").d, nullptr, 0, m->mem.f(), d, d_off, d_len, 1, n, d, st);
      pfhip::launch_attention(m->qkv.f(), 3 * d, m->qkv.f() + d, 3 * d, m->qkv.f() + 2 * d, 3 * d, m->ctx.f(), d, d_off,
                              d_len, d_off, d_len, 1, c.n_head, n, att_scale, st);
      gemm(m, st, m->ctx.f(), d, m->W(p + "out.w").d, d, d, d, x, d, m->W(p + "out.b").d, m->mem.f(), d,
           first ? nullptr : x, d, n, false);
      lnorm(m, st, x, d, m->y.f(), d, p + "norm2", n, d, d);
      gemm(m, st, m->y.f(), d, m->W(p + "ffn1.w").d, c.ffn, d, d, m->hbuf.f(), c.ffn, m->W(p + "ffn1.b").d, nullptr, 0,
           nullptr, 0, n, true);
      gemm(m, st, m->hbuf.f(), c.ffn, m->W(p + "ffn2.w").d, d, c.ffn, c.ffn, x, d, m->W(p + "ffn2.b").d, x, d, nullptr, 0,
           n, false);
    }
    lnorm(m, st, x, d, s->enc.f(), d, "enc.after_norm", n, d, d);
    // predictor alphas: conv1d k=3 over the window (zero padded at its ends) -> relu -> linear -> sigmoid
    float* col = m->qkv.f();
    float* po = m->ctx.f();
    pfhip::launch_im2col3(s->enc.f(), d, col, 3 * d, dm + 16, dm + 16 + 128, n, d, st);
    gemm(m, st, col, 3 * d, m->d_predconv, d, 3 * d, 3 * d, po, d, m->W("pred.conv.b").d,
         c.pred_residual ? s->enc.f() : nullptr, d, nullptr, 0, n, true);
    pfhip::launch_alpha(po, d, m->W("pred.out.w").d, m->W("pred.out.b").d, c.smooth_factor, c.noise_threshold,
                        s->alphas.f(), n, d, st);
    // ---- CifSearch (:270-345) ---------------------------------------------------------------------------
    pfhip::launch_cif_stream(s->enc.f(), d, s->alphas.f(), n, s->chunk_size[0], s->chunk_size[0] + s->chunk_size[1], is_last,
                             c.cif_threshold, c.tail_threshold, s->carry.f(), s->carry.f() + d, s->emb.f(), s->nfire.i(), d, st);
    HIP_TRY(hipMemcpyAsync(s->h_pin, s->nfire.p, 4, hipMemcpyDeviceToHost, st));
    return PFHIP_OK;
  });
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  const int N = s->h_pin[0];
  s->last_fires = N;
  if (N <= 0) { HIP_TRY(hipGetLastError()); return PFHIP_OK; }          // :472 decoder only if CIF fired
  if (N > kMaxTok) return fail(PFHIP_ERR_CAPACITY, "more CIF fires in one chunk than the stream workspace holds");
  s->h_pin[256] = N;       // tok_len staging (read by the copy below at execution time)
  // ---- streaming decoder session (:500): FSMN with the 10-frame cache, cross-attention over this window ----
  float* xd = m->xd.f();
  float* kvbuf = m->qkv.f();
  rc = run_cached(s, 0x10000 | (N << 8) | n, graphs, st, [&]() -> pfhip_status {
    HIP_TRY(hipMemcpyAsync(d_tok, s->h_pin + 256, 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(xd, s->emb.p, (size_t)N * d * 4, hipMemcpyDeviceToDevice, st));
    auto dec_ffn = [&](const std::string& p, const float* xin, float* o) {
      lnorm(m, st, xin, d, m->yd.f(), d, p + "norm1", N, d, d);
      gemm(m, st, m->yd.f(), d, m->W(p + "ffn1.w").d, c.dec_ffn, d, d, m->hd.f(), c.dec_ffn, m->W(p + "ffn1.b").d, nullptr,
           0, nullptr, 0, N, true);
      lnorm(m, st, m->hd.f(), c.dec_ffn, m->hd2.f(), c.dec_ffn, p + "ffn_norm", N, c.dec_ffn, c.dec_ffn);
      gemm(m, st, m->hd2.f(), c.dec_ffn, m->W(p + "ffn2.w").d, d, c.dec_ffn, c.dec_ffn, o, d, nullptr, nullptr, 0, nullptr, 0,
           N, false);
    };
    for (int i = 0; i < c.dec_layers; ++i) {
      const std::string p = "dec." + std::to_string(i) + ".";
      dec_ffn(p, xd, m->td.f());
      lnorm(m, st, m->td.f(), d, m->t2.f(), d, p + "norm2", N, d, d);
      pfhip::launch_fsmn_cached(m->t2.f(), m->W(p + "fsmn.w").d, xd, xd, s->dcache.f() + (size_t)i * 10 * d, N, d, st);
      lnorm(m, st, xd, d, m->yd.f(), d, p + "norm3", N, d, d);
      gemm(m, st, m->yd.f(), d, m->W(p + "q.w").d, d, d, d, m->qd.f(), d, m->W(p + "q.b").d, nullptr, 0, nullptr, 0, N, false);
      gemm(m, st, s->enc.f(), d, m->W(p + "kv.w").d, 2 * d, d, d, kvbuf, 2 * d, m->W(p + "kv.b").d, nullptr, 0, nullptr, 0, n,
           false);
      pfhip::launch_attention(m->qd.f(), d, kvbuf, 2 * d, kvbuf + d, 2 * d, m->ctxd.f(), d, d_off, d_tok, d_off, d_len, 1,
                              c.n_head, N, att_scale, st);
      gemm(m, st, m->ctxd.f(), d, m->W(p + "out.w").d, d, d, d, xd, d, m->W(p + "out.b").d, xd, d, nullptr, 0, N, false);
    }
    dec_ffn("dec3.", xd, m->td.f());
    lnorm(m, st, m->td.f(), d, m->yd.f(), d, "dec.after_norm", N, d, d);
    gemm(m, st, m->yd.f(), d, m->W("dec.out.w").d, c.vocab, d, d, m->logits.f(), m->vocab_pad, m->d_vocab_bias, nullptr, 0,
         nullptr, 0, N, false);
    pfhip::launch_logsoftmax_argmax(m->logits.f(), m->vocab_pad, N, c.vocab, want_logp ? s->logp.f() : nullptr,
                                    static_cast<int32_t*>(s->ids.p), st);
    HIP_TRY(hipMemcpyAsync(s->h_pin + 1, s->ids.p, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    return PFHIP_OK;
  });
  if (rc) return rc;
  s->last_has_logp = want_logp;
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  for (int i = 0; i < N; ++i) out.push_back(s->h_pin[1 + i]);        // OnlineGreedySearch paraformer.cpp:362-371
  return PFHIP_OK;
}

// window = feats_cache_ ++ rows[r0, r0+nr) (AddOverlapChunk :397-413); returns the window length in *n_out
pfhip_status add_overlap_chunk(pfhip_stream* s, int r0, int nr, bool input_finished, hipStream_t st, int* n_out) {
  pfhip_model* m = s->m;
  const int FD = m->feat_dim, FP = m->feat_pad;
  const int nc = s->n_featc;
  int n = nc + nr;
  if (n > 128) return fail(PFHIP_ERR_ARG, "stream window too long");
  pfhip::launch_rows_copy(s->chunk.f(), FP, s->featc.f(), FD, nc, FD, st);
  pfhip::launch_rows_copy(s->chunk.f() + (size_t)nc * FP, FP, s->rows.f() + (size_t)r0 * FD, FD, nr, FD, st);
  int keep;
  if (input_finished) {
    keep = s->chunk_size[0];
  } else {
    keep = s->chunk_size[0] + s->chunk_size[2];
  }
  if (keep > n) return fail(PFHIP_ERR_ARG, "stream window shorter than the look-back cache");
  // new feats_cache_ = last `keep` rows of the (unpadded) window; chunk has stride FP, featc FD
  pfhip::launch_rows_copy(s->featc.f(), FD, s->chunk.f() + (size_t)(n - keep) * FP, FP, keep, FD, st);
  s->n_featc = keep;
  if (input_finished && !s->is_last_chunk) {
    const int total = s->chunk_size[0] + s->chunk_size[1] + s->chunk_size[2];
    if (total > n) {                                          // zero rows up to 20 (:402-408)
      pfhip::launch_rows_copy(s->chunk.f() + (size_t)n * FP, FP, nullptr, 0, total - n, 0, st);
      n = total;
    }
  }
  *n_out = n;
  return PFHIP_OK;
}

// OnlineLfrCmvn (:196-238) over the T frames at the front of fb[cur]; emits n rows (CMVN, x sqrt(d), PE) into
// s->rows and rotates the splice cache.
pfhip_status online_lfr_cmvn(pfhip_stream* s, int T, bool input_finished, hipStream_t st, int* n_rows) {
  pfhip_model* m = s->m;
  const int lfr_m = m->cfg.lfr_m, lfr_n = m->cfg.lfr_n;
  const int T_lrf = (int)std::ceil((T - (lfr_m - 1) / 2) / (float)lfr_n);
  int splice = T_lrf, n_out = 0;
  for (int i = 0; i < T_lrf; ++i) {
    if (lfr_m <= T - i * lfr_n) ++n_out;
    else if (input_finished) ++n_out;
    else { splice = i; break; }
  }
  splice = std::min(T - 1, splice * lfr_n);
  if (n_out > kMaxRows) return fail(PFHIP_ERR_ARG, "too many LFR rows in one streaming call");
  const float* fb = s->fb[s->fb_cur].f();
  pfhip::launch_stream_lfr(fb, T, n_out, m->W("cmvn.mean").d, m->W("cmvn.istd").d, sqrtf((float)m->cfg.d_model),
                           m->d_inv_ts, s->start_idx, s->rows.f(), m->feat_dim, st);
  s->start_idx += n_out;                                               // GetPosEmb :242-243
  const int keep = T - splice;                                         // lfr_splice_cache_ = frames[splice:] (:226-228)
  pfhip::launch_rows_copy(s->fb[s->fb_cur ^ 1].f(), 80, fb + (size_t)splice * 80, 80, keep, 80, st);
  s->fb_cur ^= 1;
  s->n_splice = keep;
  *n_rows = n_out;
  return PFHIP_OK;
}

// ExtractFeats (:147-194) + x*sqrt(d) + GetPosEmb (:549-555): leaves `*n_rows` finished LFR rows in s->rows.
pfhip_status extract_feats(pfhip_stream* s, const float* pcm, int len, bool input_finished, hipStream_t st, int* n_rows) {
  pfhip_model* m = s->m;
  *n_rows = 0;
  const int fl = 400, fs = 160, lfr_m = m->cfg.lfr_m;
  // FbankKaldi (:119-145): prepend input_cache_, keep what follows the last frame shift for the next call
  std::vector<float> waves(s->input_cache);
  waves.insert(waves.end(), pcm, pcm + len);
  const int total = (int)waves.size();
  int frame_number = (total - fl) / fs + 1;
  if (!(frame_number >= 1 && total >= fl)) frame_number = 0;            // paraformer-online.h:25-31
  s->input_cache.assign(waves.begin() + (size_t)frame_number * fs, waves.end());
  pfhip_status rc = PFHIP_OK;
  if (frame_number > 0) {
    const int used = (frame_number - 1) * fs + fl;
    const bool fresh = s->n_splice == 0;
    const int base = fresh ? (lfr_m - 1) / 2 : s->n_splice;
    if (used > kMaxSamples + 1024 || base + frame_number > kMaxFrames)
      return fail(PFHIP_ERR_ARG, "too many samples in one streaming call");
    HIP_TRY(hipMemcpyAsync(s->pcm.p, waves.data(), (size_t)used * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));          // `waves` is pageable host memory that dies with this scope
    float* fb = s->fb[s->fb_cur].f();
    {
      int64_t* hm64 = reinterpret_cast<int64_t*>(s->h_pin + 768);
      hm64[0] = 0;
      int* hm = s->h_pin + 772;
      hm[0] = 0; hm[1] = frame_number; hm[2] = frame_number;
      HIP_TRY(hipMemcpyAsync(s->meta.i() + 8, s->h_pin + 768, 32, hipMemcpyHostToDevice, st));
      pfhip::FbankTables tb{m->d_window, m->d_tw, m->d_mel_off, m->d_mel_size, m->d_mel_w, m->W("cmvn.mean").d,
                            m->W("cmvn.istd").d};
      pfhip::launch_fbank_frames(s->pcm.f(), reinterpret_cast<int64_t*>(s->meta.i() + 8), s->meta.i() + 12,
                                 s->meta.i() + 14, frame_number, tb, fb + (size_t)base * 80, st);
    }
    if (fresh) {     // lfr_splice_cache_ = (lfr_m-1)/2 copies of the first frame (:155-158)
      pfhip::launch_rows_copy(fb, 80, fb + (size_t)base * 80, 0, base, 80, st);
      s->n_splice = base;
    }
    if (frame_number + s->n_splice >= lfr_m) {
      rc = online_lfr_cmvn(s, s->n_splice + frame_number, input_finished, st, n_rows);
    } else {
      // (:172-177) the splice cache just grows.  The reference leaves the raw 80-dim frames in wav_feats
      // here and feeds them on — a latent bug only reachable with < 55 ms of audio in a non-final call,
      // which the 2-pass server (9600-sample chunks) never sends; no window is produced.
      s->n_splice += frame_number;
    }
  } else if (input_finished) {
    if (s->n_splice > 0) rc = online_lfr_cmvn(s, s->n_splice, true, st, n_rows);   // (:179-189)
  }
  if (input_finished) reset_cache(s);                                              // (:191-193)
  return rc;
}

pfhip_status copy_out(const void* src, size_t n, float* dst, size_t cap, size_t* n_out, hipStream_t st) {
  if (n > cap) return fail(PFHIP_ERR_CAPACITY, "dst too small");
  if (n) HIP_TRY(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (n_out) *n_out = n;
  return PFHIP_OK;
}

}  // namespace

extern "C" {

pfhip_status pfhip_stream_create(pfhip_model* m, const int* chunk_size, pfhip_stream** out) {
  last_error().clear();
  if (!m || !out) return fail(PFHIP_ERR_ARG, "null argument");
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  std::unique_ptr<pfhip_stream> s(new pfhip_stream);
  s->m = m;
  if (chunk_size) for (int i = 0; i < 3; ++i) s->chunk_size[i] = chunk_size[i];
  if (s->chunk_size[0] < 0 || s->chunk_size[1] <= 0 || s->chunk_size[2] < 0 ||
      s->chunk_size[0] + s->chunk_size[2] > 16 || s->chunk_size[0] + s->chunk_size[1] + s->chunk_size[2] > 64)
    return fail(PFHIP_ERR_ARG, "unsupported chunk_size");
  pfhip_status st = stream_alloc(s.get());
  if (st) return st;
  st = init_cache(s.get(), m->own_stream);
  if (st) return st;
  HIP_TRY(hipStreamSynchronize(m->own_stream));
  *out = s.release();
  return PFHIP_OK;
}

void pfhip_stream_destroy(pfhip_stream* s) {
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(s->m->mu);
    (void)hipSetDevice(s->m->device);
    (void)hipStreamSynchronize(s->m->own_stream);
    for (Buf* b : {&s->pcm, &s->fb[0], &s->fb[1], &s->rows, &s->featc, &s->chunk, &s->enc, &s->alphas, &s->carry, &s->emb,
                   &s->nfire, &s->dcache, &s->meta, &s->ids, &s->logp})
      b->release();
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    drop_graphs(s);
  }
  delete s;
}

pfhip_status pfhip_stream_reset(pfhip_stream* s) {
  last_error().clear();
  if (!s) return fail(PFHIP_ERR_ARG, "null stream");
  std::lock_guard<std::mutex> lk(s->m->mu);
  HIP_TRY(hipSetDevice(s->m->device));
  reset_cache(s);
  pfhip_status st = init_cache(s, s->m->own_stream);
  if (st) return st;
  HIP_TRY(hipStreamSynchronize(s->m->own_stream));
  return PFHIP_OK;
}

pfhip_status pfhip_stream_forward(pfhip_stream* s, const float* pcm, int n_samples, int input_finished,
                                  int32_t* token_ids, int cap, int* n_tokens) {
  last_error().clear();
  if (!s || n_samples < 0 || (n_samples > 0 && !pcm) || !n_tokens) return fail(PFHIP_ERR_ARG, "bad argument");
  if (n_samples > kMaxSamples) return fail(PFHIP_ERR_ARG, "more than 32000 samples in one streaming call");
  pfhip_model* m = s->m;
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = m->own_stream;
  m->prof_stream = st;
  const bool fin = input_finished != 0;
  const bool want_logp = s->debug;
  std::vector<int32_t> out;
  *n_tokens = 0;
  pfhip_status rc = PFHIP_OK;
  auto finish = [&]() -> pfhip_status {
    if ((int)out.size() > cap) return fail(PFHIP_ERR_CAPACITY, "token_ids too small");
    for (size_t i = 0; i < out.size(); ++i) token_ids[i] = out[i];
    *n_tokens = (int)out.size();
    return PFHIP_OK;
  };
  // (:532-540) a short final call after the first chunk: flush the look-back cache as the last chunk
  if (n_samples < 16 * 60 && fin && !s->is_first_chunk) {
    s->is_last_chunk = true;
    const int n = s->n_featc;
    pfhip::launch_rows_copy(s->chunk.f(), m->feat_pad, s->featc.f(), m->feat_dim, n, m->feat_dim, st);
    rc = forward_chunk(s, n, st, out, want_logp);
    reset_cache(s);
    pfhip_status r2 = init_cache(s, st);
    if (rc) return rc;
    if (r2) return r2;
    return finish();
  }
  if (s->is_first_chunk) s->is_first_chunk = false;
  int nr = 0;
  rc = extract_feats(s, pcm, n_samples, fin, st, &nr);
  if (rc) return rc;
  if (nr == 0) return finish();                                        // (:545-547)
  int n = 0;
  if (fin) {
    if (nr + s->chunk_size[2] <= s->chunk_size[1]) {                   // (:557-559)
      s->is_last_chunk = true;
      rc = add_overlap_chunk(s, 0, nr, fin, st, &n);
      if (rc) return rc;
    } else {                                                           // (:560-579) first chunk + last chunk
      rc = add_overlap_chunk(s, 0, nr, fin, st, &n);
      if (rc) return rc;
      rc = forward_chunk(s, n, st, out, want_logp);
      if (rc) return rc;
      s->is_last_chunk = true;
      const int k = nr + s->chunk_size[2] - s->chunk_size[1];
      rc = add_overlap_chunk(s, nr - k, k, fin, st, &n);
      if (rc) return rc;
      rc = forward_chunk(s, n, st, out, want_logp);
      reset_cache(s);
      pfhip_status r2 = init_cache(s, st);
      if (rc) return rc;
      if (r2) return r2;
      return finish();
    }
  } else {
    rc = add_overlap_chunk(s, 0, nr, fin, st, &n);
    if (rc) return rc;
  }
  rc = forward_chunk(s, n, st, out, want_logp);
  if (fin) {                                                           // (:589-593)
    reset_cache(s);
    pfhip_status r2 = init_cache(s, st);
    if (!rc) rc = r2;
  }
  if (rc) return rc;
  return finish();
}

pfhip_status pfhip_stream_set_debug(pfhip_stream* s, int on) {
  if (!s) return fail(PFHIP_ERR_ARG, "null stream");
  s->debug = (on & 1) != 0;          // bit 0: keep log-probs (disables graph replay); bit 1: never use hipGraphs
  s->use_graphs = (on & 2) == 0;
  return PFHIP_OK;
}

pfhip_status pfhip_stream_get_tensor(pfhip_stream* s, const char* name, float* dst, size_t cap_floats, size_t* n_out) {
  last_error().clear();
  if (!s || !name || !dst) return fail(PFHIP_ERR_ARG, "bad argument");
  pfhip_model* m = s->m;
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = m->own_stream;
  const std::string nm(name);
  const int d = m->cfg.d_model;
  if (nm == "chunk") {
    const size_t n = (size_t)s->last_n * m->feat_dim;
    if (n > cap_floats) return fail(PFHIP_ERR_CAPACITY, "dst too small");
    if (n) HIP_TRY(hipMemcpy2DAsync(dst, (size_t)m->feat_dim * 4, s->chunk.p, (size_t)m->feat_pad * 4, (size_t)m->feat_dim * 4,
                                    s->last_n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_out) *n_out = n;
    return PFHIP_OK;
  }
  if (nm == "enc") return copy_out(s->enc.p, (size_t)s->last_n * d, dst, cap_floats, n_out, st);
  if (nm == "alphas") return copy_out(s->alphas.p, (size_t)s->last_n, dst, cap_floats, n_out, st);
  if (nm == "emb") return copy_out(s->emb.p, (size_t)s->last_fires * d, dst, cap_floats, n_out, st);
  if (nm == "logp") {
    if (!s->last_has_logp && s->last_fires > 0) return fail(PFHIP_ERR_ARG, "enable pfhip_stream_set_debug before the call");
    return copy_out(s->logp.p, (size_t)s->last_fires * m->cfg.vocab, dst, cap_floats, n_out, st);
  }
  return fail(PFHIP_ERR_ARG, "unknown tensor name " + nm);
}

}  // extern "C"
