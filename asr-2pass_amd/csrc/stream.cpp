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
// PCM samples of input_cache_ (:123-127).
//
// MI355X-first: a 20-row window is weight-streaming- and launch-latency-bound (0.88 GB of weights and ~700 launches
// per chunk whatever the row count), so the windows of ALL connections that have a chunk ready are packed into one
// forward (pfhip_stream_forward_batch): encoder / predictor / decoder run once over sum(rows) rows with per-connection
// segments (attention, FSMN), the CIF scan and the cached decoder FSMN take a per-connection descriptor array
// (StreamSeg) that points at each connection's own carry and caches.  pfhip_stream_forward is the batch of one.  `reserve_waveforms_` is dead state in the reference (it only feeds
// its own index arithmetic, :162-171,180-182) and is not kept.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <thread>

#include "internal.h"

using namespace pfhip_detail;

struct pfhip_stream {
  pfhip_model* m = nullptr;
  int chunk_size[3] = {5, 10, 5};
  // host state
  std::vector<float> input_cache;
  std::vector<float> waves;    // input_cache_ ++ this call's samples, alive until the batched upload has copied it
  int n_splice = 0;            // frames in the splice cache (front of fb[cur])
  int start_idx = 0;           // start_idx_cache_
  bool is_first_chunk = true, is_last_chunk = false;
  int last_path = 0;           // which branch of ParaformerOnline::Forward the last call took (pfhip_stream_last_path)
  int n_featc = 10;            // rows in feats_cache_
  // device state
  Buf fb[2], rows, featc, chunk, carry, dcache;
  int fb_cur = 0;
  // the window waiting in `chunk` for the next batched forward
  int win_n = 0;
  // last chunk (inspection): where its rows / tokens sit in the model's packed workspace
  int last_n = 0, last_fires = 0, last_row_off = 0, last_tok_off = 0, last_slot = 0;
  bool last_has_logp = false;
  bool debug = false;
};

namespace {

constexpr int kMaxSamples = 32000;   // per call (2 s); the 2-pass server sends 9600 (websocket-server-2pass.cpp:135-148)
constexpr int kMaxFrames = 256;
constexpr int kMaxRows = 64;         // rows of one encoder window
constexpr int kMaxTok = 72;

pfhip_status stream_alloc(pfhip_stream* s) {
  pfhip_model* m = s->m;
  const int d = m->cfg.d_model, FD = m->feat_dim, FP = m->feat_pad;
  for (int i = 0; i < 2; ++i) HIP_TRY(s->fb[i].ensure((size_t)kMaxFrames * 80 * 4));
  HIP_TRY(s->rows.ensure((size_t)kMaxRows * FD * 4));
  HIP_TRY(s->featc.ensure((size_t)16 * FD * 4));
  HIP_TRY(s->chunk.ensure((size_t)128 * FP * 4));
  HIP_TRY(s->carry.ensure((size_t)(d + 4) * 4));
  HIP_TRY(s->dcache.ensure((size_t)std::max(1, m->cfg.dec_layers) * 10 * d * 4));
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

// The front-end work of one batched call, recorded per connection while its host control flow runs and flushed as a
// handful of batched launches (one per phase) instead of ~8 tiny launches per connection.  Operations of one connection
// keep their order because the phases are flushed in program order; operations of different connections are independent.
struct Recorder {
  enum { kFramesToFb = 0, kFreshReplicate, kSpliceRotate, kWindow, kFeatc, kZeroPad, kPack, kCopyPhases };
  std::vector<pfhip::RowsCopyOp> copies[kCopyPhases];
  std::vector<pfhip::StreamLfrOp> lfr;
  // fbank batch: PCM of every connection that has new frames, back to back
  std::vector<const float*> pcm_src;
  std::vector<int> pcm_len, nframes;
  struct FrameDst { float* dst; int n; };
  std::vector<FrameDst> frame_dst;
  bool empty() const {
    if (!lfr.empty() || !pcm_src.empty()) return false;
    for (const auto& c : copies) if (!c.empty()) return false;
    return true;
  }
  void copy(int phase, float* dst, int ldd, const float* src, int lds_, int nrows, int ncols) {
    if (nrows > 0) copies[phase].push_back(pfhip::RowsCopyOp{dst, src, ldd, lds_, nrows, ncols});
  }
};

pfhip_status flush(pfhip_model* m, Recorder& r, hipStream_t st) {
  if (r.empty()) return PFHIP_OK;
  const Config& c = m->cfg;
  // ---- fbank of all new audio in one launch: PCM staged back to back, frames into a batch buffer -----------------
  const int U = (int)r.pcm_src.size();
  size_t total_samples = 0;
  int total_frames = 0;
  for (int u = 0; u < U; ++u) { total_samples += (size_t)r.pcm_len[u]; total_frames += r.nframes[u]; }
  size_t n_ops = r.lfr.size();
  for (const auto& cp : r.copies) n_ops += cp.size();
  n_ops += (size_t)U;                                      // frames -> fb copies are appended below
  const size_t meta_bytes = ((size_t)U * 8 + (size_t)(U + 1) * 4 + (size_t)U * 4 + 63) & ~(size_t)63;
  const size_t ops_bytes = n_ops * sizeof(pfhip::RowsCopyOp) + 64;
  const size_t pin_bytes = meta_bytes + ops_bytes + total_samples * 4 + 256;
  if (pin_bytes > m->h_ops_cap) {
    if (m->h_ops) HIP_TRY(hipHostFree(m->h_ops));
    m->h_ops = nullptr; m->h_ops_cap = 0;
    HIP_TRY(hipHostMalloc(&m->h_ops, pin_bytes * 2, hipHostMallocDefault));
    m->h_ops_cap = pin_bytes * 2;
  }
  HIP_TRY(m->d_ops.ensure(meta_bytes + ops_bytes));
  char* hp = static_cast<char*>(m->h_ops);
  char* dp = static_cast<char*>(m->d_ops.p);
  if (U > 0) {
    HIP_TRY(m->pcm.ensure((total_samples + 1024) * 4));
    HIP_TRY(m->fbk.ensure((size_t)total_frames * 80 * 4));
    int64_t* h_soff = reinterpret_cast<int64_t*>(hp);
    int* h_foff = reinterpret_cast<int*>(hp + (size_t)U * 8);
    int* h_nf = h_foff + (U + 1);
    float* h_pcm = reinterpret_cast<float*>(hp + meta_bytes + ops_bytes);
    size_t so = 0; int fo = 0;
    for (int u = 0; u < U; ++u) {
      h_soff[u] = (int64_t)so; h_foff[u] = fo; h_nf[u] = r.nframes[u];
      std::memcpy(h_pcm + so, r.pcm_src[u], (size_t)r.pcm_len[u] * 4);
      r.copy(Recorder::kFramesToFb, r.frame_dst[u].dst, 80, m->fbk.f() + (size_t)fo * 80, 80, r.frame_dst[u].n, 80);
      so += (size_t)r.pcm_len[u]; fo += r.nframes[u];
    }
    h_foff[U] = fo;
    HIP_TRY(hipMemcpyAsync(m->pcm.p, h_pcm, total_samples * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dp, hp, meta_bytes, hipMemcpyHostToDevice, st));
    pfhip::FbankTables tb{m->d_window, m->d_tw, m->d_mel_off, m->d_mel_size, m->d_mel_w, m->W("cmvn.mean").d, m->W("cmvn.istd").d};
    pfhip::launch_fbank_frames_batch(m->pcm.f(), reinterpret_cast<const int64_t*>(dp), reinterpret_cast<const int*>(dp + (size_t)U * 8),
                                     reinterpret_cast<const int*>(dp + (size_t)U * 8 + (size_t)(U + 1) * 4), U, total_frames, tb,
                                     m->fbk.f(), st);
  }
  // ---- descriptor arrays: one upload, then the phases in program order ----------------------------------------------
  char* h_ops = hp + meta_bytes;
  char* d_ops = dp + meta_bytes;
  size_t off = 0;
  size_t phase_off[Recorder::kCopyPhases];
  for (int ph = 0; ph < Recorder::kCopyPhases; ++ph) {
    phase_off[ph] = off;
    const size_t nb = r.copies[ph].size() * sizeof(pfhip::RowsCopyOp);
    if (nb) std::memcpy(h_ops + off, r.copies[ph].data(), nb);
    off += nb;
  }
  const size_t lfr_off = off;
  if (!r.lfr.empty()) std::memcpy(h_ops + off, r.lfr.data(), r.lfr.size() * sizeof(pfhip::StreamLfrOp));
  off += r.lfr.size() * sizeof(pfhip::StreamLfrOp);
  if (off) HIP_TRY(hipMemcpyAsync(d_ops, h_ops, off, hipMemcpyHostToDevice, st));
  auto run_copies = [&](int ph) {
    int mx = 0;
    for (const auto& o : r.copies[ph]) mx = std::max(mx, o.nrows);
    pfhip::launch_rows_copy_batch(reinterpret_cast<const pfhip::RowsCopyOp*>(d_ops + phase_off[ph]), (int)r.copies[ph].size(), mx, st);
  };
  run_copies(Recorder::kFramesToFb);
  run_copies(Recorder::kFreshReplicate);
  {
    int mx = 0;
    for (const auto& o : r.lfr) mx = std::max(mx, o.n_rows);
    pfhip::launch_stream_lfr_batch(reinterpret_cast<const pfhip::StreamLfrOp*>(d_ops + lfr_off), (int)r.lfr.size(), mx,
                                   m->W("cmvn.mean").d, m->W("cmvn.istd").d, sqrtf((float)c.d_model), m->d_inv_ts, m->feat_dim, st);
  }
  run_copies(Recorder::kSpliceRotate);
  run_copies(Recorder::kWindow);
  run_copies(Recorder::kFeatc);
  run_copies(Recorder::kZeroPad);
  run_copies(Recorder::kPack);
  HIP_TRY(hipGetLastError());
  r = Recorder();
  return PFHIP_OK;
}

// ForwardChunk (:415-523) for every stream in `ss` at once: stream b's window (win_n rows) is waiting in its `chunk`
// buffer.  Appends each stream's ids to outs[b].
pfhip_status forward_windows(pfhip_model* m, const std::vector<pfhip_stream*>& ss, hipStream_t st,
                             const std::vector<std::vector<int32_t>*>& outs, bool want_logp, Recorder& rec) {
  const Config& c = m->cfg;
  const int d = c.d_model, FD = m->feat_dim, FP = m->feat_pad, B = (int)ss.size();
  const float att_scale = 1.0f / sqrtf((float)pfhip::kHeadDim);
  if (B == 0) return flush(m, rec, st);
  std::vector<pfhip::StreamSeg> segs(B);
  int M = 0;
  for (int b = 0; b < B; ++b) {
    pfhip_stream* s = ss[b];
    const int n = s->win_n;
    s->last_n = n; s->last_fires = 0; s->last_has_logp = false; s->last_row_off = M; s->last_tok_off = 0; s->last_slot = b;
    if (n <= 0 || n > 128) return fail(PFHIP_ERR_ARG, "stream window out of range");
    segs[b] = pfhip::StreamSeg{s->carry.f(), s->dcache.f(), M, n, s->is_last_chunk ? 1 : 0, s->chunk_size[0],
                               s->chunk_size[0] + s->chunk_size[1], 0, 0, 0};
    M += n;
  }
  m->B = 0; m->M = 0; m->ML = 0; m->have_ts = false;      // the offline results in this workspace are gone
  const int Mp = round_up(M, pfhip::kTileM);
  HIP_TRY(m->x0.ensure((size_t)Mp * FP * 4));
  HIP_TRY(m->y.ensure((size_t)Mp * FP * 4));
  HIP_TRY(m->x.ensure((size_t)Mp * d * 4));
  HIP_TRY(m->qkv.ensure((size_t)Mp * 3 * d * 4));
  HIP_TRY(m->mem.ensure((size_t)Mp * d * 4));
  HIP_TRY(m->ctx.ensure((size_t)Mp * d * 4));
  HIP_TRY(m->hbuf.ensure((size_t)Mp * std::max(c.ffn, d) * 4));
  HIP_TRY(m->enc.ensure((size_t)Mp * d * 4));
  HIP_TRY(m->alphas.ensure((size_t)Mp * 4));
  HIP_TRY(m->emb.ensure((size_t)B * kMaxTok * d * 4));
  HIP_TRY(m->counts.ensure((size_t)2 * B * 4));
  // device metadata: off[B] len[B] tok_off[B] tok_len[B] row_pos[M] row_len[M] src_row[B*kMaxTok]; segs in their own buffer
  const size_t n_meta = 4 * (size_t)B + 2 * (size_t)M + (size_t)B * kMaxTok;
  HIP_TRY(m->dmeta.ensure(n_meta * 4));
  HIP_TRY(m->sseg.ensure((size_t)B * sizeof(pfhip::StreamSeg)));
  {   // pinned staging: first half = this upload, second half = the post-CIF upload
    const size_t half = ((n_meta * 4 + 15) & ~(size_t)15) + (size_t)B * sizeof(pfhip::StreamSeg) + 64;
    pfhip_status ps = ensure_h_meta(m, 2 * half);
    if (ps) return ps;
    ps = ensure_h_counts(m, (size_t)2 * B * 4);
    if (ps) return ps;
  }
  int* hm = static_cast<int*>(m->h_meta);
  int* dm = m->dmeta.i();
  int* d_off = dm; int* d_len = dm + B; int* d_tok_off = dm + 2 * B; int* d_tok_len = dm + 3 * B;
  int* d_row_pos = dm + 4 * B; int* d_row_len = d_row_pos + M; int* d_src_row = d_row_len + M;
  for (int b = 0; b < B; ++b) {
    hm[b] = segs[b].row_off; hm[B + b] = segs[b].n; hm[2 * B + b] = 0; hm[3 * B + b] = 0;
    for (int t = 0; t < segs[b].n; ++t) { hm[4 * B + segs[b].row_off + t] = t; hm[4 * B + M + segs[b].row_off + t] = segs[b].n; }
  }
  pfhip::StreamSeg* h_segs = reinterpret_cast<pfhip::StreamSeg*>(static_cast<char*>(m->h_meta) + ((n_meta * 4 + 15) & ~(size_t)15));
  if (((n_meta * 4 + 15) & ~(size_t)15) + (size_t)B * sizeof(pfhip::StreamSeg) > m->h_meta_cap)
    return fail(PFHIP_ERR_CAPACITY, "too many stream rows in one batch");
  std::memcpy(h_segs, segs.data(), (size_t)B * sizeof(pfhip::StreamSeg));
  HIP_TRY(hipMemcpyAsync(dm, hm, (4 * (size_t)B + 2 * (size_t)M) * 4, hipMemcpyHostToDevice, st));
  pfhip::StreamSeg* d_segs = static_cast<pfhip::StreamSeg*>(m->sseg.p);
  HIP_TRY(hipMemcpyAsync(d_segs, h_segs, (size_t)B * sizeof(pfhip::StreamSeg), hipMemcpyHostToDevice, st));
  // pack the windows (one batched copy; also flushes whatever front-end work the callers recorded)
  for (int b = 0; b < B; ++b)
    rec.copy(Recorder::kPack, m->x0.f() + (size_t)segs[b].row_off * FP, FP, ss[b]->chunk.f(), FP, segs[b].n, FP);
  {
    pfhip_status fs = flush(m, rec, st);
    if (fs) return fs;
  }
  int maxn = 0;
  for (int b = 0; b < B; ++b) maxn = std::max(maxn, segs[b].n);
  float* x = m->x.f();
  // One connection, one window: the latency path.  Launch count, not bytes or flops, sets its time (stream_fused.hip), so
  // row-wise operators are folded into the GEMMs that consume them.  PFHIP_STREAM_FUSED=0 keeps the general path.
  static const bool fused_on = [] { const char* e = getenv("PFHIP_STREAM_FUSED"); return !(e && e[0] == '0'); }();
  const bool lean = fused_on && B == 1 && M <= 32;
  static const bool att_out_on = [] { const char* e = getenv("PFHIP_STREAM_ATT_OUT"); return e && e[0] == '1'; }();
  auto ln_gemm = [&](const float* X, int ldx, int D, const std::string& norm, const float* Wd, int ldw, float* Cd, int ldc,
                     const float* bias, const float* R1, int ldr1, const float* R2, int ldr2, const float* fv, int ldv,
                     const float* fw, int rows, int N, int K, bool relu) {
    pfhip::launch_fused_ln_gemm(X, ldx, D, norm.empty() ? nullptr : m->W(norm + ".g").d, norm.empty() ? nullptr : m->W(norm + ".b").d,
                                1e-12f, Wd, ldw, Cd, ldc, bias, R1, ldr1, R2, ldr2, fv, ldv, fw, rows, N, K, relu, st);
  };
  // one-trip form of the same launches where the shape allows (stream_fused.hip: every operand requested at once, LayerNorm
  // applied algebraically on the gamma/beta-folded weights built at load); false -> the caller uses ln_gemm
  auto gemv1 = [&](const float* X, int ldx, const float* Wd, int ldw, float* Cd, int ldc, const float* bias, const float* colsum,
                   const float* R1, int ldr1, const float* fv, int ldv, const float* fw, int rows, int N, int K, bool relu) {
    return Wd && pfhip::launch_fused_gemv_1trip(X, ldx, Wd, ldw, Cd, ldc, bias, colsum, 1e-12f, R1, ldr1, fv, ldv, fw, rows, N, K, relu, st);
  };
  const bool fuse_ln_s = !lean && m->d_lnw_qkv != nullptr && pfhip::gemm_x6_ln_ok(M);
  if (fuse_ln_s) HIP_TRY(m->lnstats.ensure((size_t)(M + 256) * 4 * 2 * 4));
  // ---- streaming encoder session (:448): SAN-M stack on the windows as given (no scale/PE inside) --------
  for (int i = 0; i < c.enc_layers; ++i) {
    const std::string p = "enc." + std::to_string(i) + ".";
    const bool first = i == 0;
    const float* xin = first ? m->x0.f() : x;
    const int ldin = first ? FP : d, Din = first ? FD : d, Kp = first ? FP : d;
    if (lean) {
      // 5 launches: LN1+QKV | attention | out-projection + FSMN memory + residual | LN2+FFN1 | FFN2 + residual
      if (first || !m->d_lnw_qkv ||
          !gemv1(x, d, m->d_lnw_qkv + (size_t)i * 3 * d * d, d, m->qkv.f(), 3 * d, m->d_lnb_qkv + (size_t)i * 3 * d,
                 m->d_lns_qkv + (size_t)i * 3 * d, nullptr, 0, nullptr, 0, nullptr, M, 3 * d, d, false))
        ln_gemm(xin, ldin, Din, p + "norm1", first ? m->d_w0qkv : m->W(p + "qkv.w").d, Kp, m->qkv.f(), 3 * d, m->W(p + "qkv.b").d,
                nullptr, 0, nullptr, 0, nullptr, 0, nullptr, M, 3 * d, Kp, false);
      // attention + output projection + FSMN memory + residual as one launch (every workgroup redoes the attention) is opt-in:
      // measured slower than the two launches (stream_fused.hip, launch_fused_att_out)
      if (!att_out_on || !pfhip::launch_fused_att_out(m->qkv.f(), 3 * d, m->qkv.f() + d, 3 * d, m->qkv.f() + 2 * d, 3 * d, M, M, c.n_head, att_scale,
                                       m->W(p + "out.w").d, d, x, d, m->W(p + "out.b").d, first ? nullptr : x, d, m->qkv.f() + 2 * d,
                                       3 * d, m->W(p + "fsmn.w").d, d, st)) {
        if (!pfhip::launch_window_attention(m->qkv.f(), 3 * d, m->qkv.f() + d, 3 * d, m->qkv.f() + 2 * d, 3 * d, m->ctx.f(), d, M, M,
                                            c.n_head, att_scale, st))
          pfhip::launch_attention(m->qkv.f(), 3 * d, m->qkv.f() + d, 3 * d, m->qkv.f() + 2 * d, 3 * d, m->ctx.f(), d, d_off,
                                  d_len, d_off, d_len, B, c.n_head, maxn, att_scale, st);
        if (!gemv1(m->ctx.f(), d, m->W(p + "out.w").d, d, x, d, m->W(p + "out.b").d, nullptr, first ? nullptr : x, d, m->qkv.f() + 2 * d,
                   3 * d, m->W(p + "fsmn.w").d, M, d, d, false))
          ln_gemm(m->ctx.f(), d, 0, "", m->W(p + "out.w").d, d, x, d, m->W(p + "out.b").d, first ? nullptr : x, d, nullptr, 0,
                  m->qkv.f() + 2 * d, 3 * d, m->W(p + "fsmn.w").d, M, d, d, false);
      }
      if (!m->d_lnw_ffn1 ||
          !gemv1(x, d, m->d_lnw_ffn1 + (size_t)i * c.ffn * d, d, m->hbuf.f(), c.ffn, m->d_lnb_ffn1 + (size_t)i * c.ffn,
                 m->d_lns_ffn1 + (size_t)i * c.ffn, nullptr, 0, nullptr, 0, nullptr, M, c.ffn, d, true))
        ln_gemm(x, d, d, p + "norm2", m->W(p + "ffn1.w").d, d, m->hbuf.f(), c.ffn, m->W(p + "ffn1.b").d, nullptr, 0, nullptr, 0,
                nullptr, 0, nullptr, M, c.ffn, d, true);
      if (!gemv1(m->hbuf.f(), c.ffn, m->W(p + "ffn2.w").d, c.ffn, x, d, m->W(p + "ffn2.b").d, nullptr, x, d, nullptr, 0, nullptr, M, d,
                 c.ffn, false))
        ln_gemm(m->hbuf.f(), c.ffn, 0, "", m->W(p + "ffn2.w").d, c.ffn, x, d, m->W(p + "ffn2.b").d, x, d, nullptr, 0, nullptr, 0,
                nullptr, M, d, c.ffn, false);
      continue;
    }
    // rounds of many connections (>= 1536 rows): the two LayerNorms of a layer folded into the GEMMs around them, as offline
    // (pfhip.cpp enqueue_locked: row statistics from the producing epilogue, algebraic normalisation in the consumer's)
    if (fuse_ln_s && !first)
      pfhip::launch_gemm_f32_x6_ln(x, d, m->d_lnw_qkv + (size_t)i * 3 * d * d, d, m->qkv.f(), 3 * d, m->d_lnb_qkv + (size_t)i * 3 * d, nullptr, 0,
                                   nullptr, 0, M, 3 * d, d, false, m->lnstats.f(), 4, m->d_lns_qkv + (size_t)i * 3 * d, nullptr, st,
                                   m->w_scale_of(m->d_lnw_qkv + (size_t)i * 3 * d * d));
    else {
      lnorm(m, st, xin, ldin, m->y.f(), Kp, p + "norm1", M, Din, Kp);
      gemm(m, st, m->y.f(), Kp, first ? m->d_w0qkv : m->W(p + "qkv.w").d, 3 * d, Kp, Din, m->qkv.f(), 3 * d,
           m->W(p + "qkv.b").d, nullptr, 0, nullptr, 0, M, false);
    }
    // windows of <= 32 rows: one small workgroup per (head, connection) instead of the long-sequence kernel, the FSMN memory of V
    // written by the same launch
    if (!pfhip::launch_window_attention_segments(m->qkv.f(), 3 * d, m->qkv.f() + d, 3 * d, m->qkv.f() + 2 * d, 3 * d, m->ctx.f(), d, d_off,
                                                 d_len, d_off, d_len, B, c.n_head, maxn, maxn, att_scale, st, m->W(p + "fsmn.w").d,
                                                 m->mem.f(), d)) {
      pfhip::launch_fsmn(m->qkv.f() + 2 * d, 3 * d, m->W(p + "fsmn.w").d, nullptr, 0, m->mem.f(), d, d_off, d_len, B, maxn, d, st);
      pfhip::launch_attention(m->qkv.f(), 3 * d, m->qkv.f() + d, 3 * d, m->qkv.f() + 2 * d, 3 * d, m->ctx.f(), d, d_off,
                              d_len, d_off, d_len, B, c.n_head, maxn, att_scale, st);
    }
    if (fuse_ln_s) {
      pfhip::launch_gemm_f32_x6_ln(m->ctx.f(), d, m->W(p + "out.w").d, d, x, d, m->W(p + "out.b").d, m->mem.f(), d, first ? nullptr : x, d, M,
                                   d, d, false, nullptr, 4, nullptr, m->lnstats.f(), st, m->w_scale_of(m->W(p + "out.w").d));
      pfhip::launch_gemm_f32_x6_ln(x, d, m->d_lnw_ffn1 + (size_t)i * c.ffn * d, d, m->hbuf.f(), c.ffn, m->d_lnb_ffn1 + (size_t)i * c.ffn,
                                   nullptr, 0, nullptr, 0, M, c.ffn, d, true, m->lnstats.f(), 4, m->d_lns_ffn1 + (size_t)i * c.ffn, nullptr, st,
                                   m->w_scale_of(m->d_lnw_ffn1 + (size_t)i * c.ffn * d));
      pfhip::launch_gemm_f32_x6_ln(m->hbuf.f(), c.ffn, m->W(p + "ffn2.w").d, c.ffn, x, d, m->W(p + "ffn2.b").d, x, d, nullptr, 0, M, d, c.ffn,
                                   false, nullptr, 4, nullptr, i + 1 < c.enc_layers ? m->lnstats.f() : nullptr, st,
                                   m->w_scale_of(m->W(p + "ffn2.w").d));
      continue;
    }
    gemm(m, st, m->ctx.f(), d, m->W(p + "out.w").d, d, d, d, x, d, m->W(p + "out.b").d, m->mem.f(), d,
         first ? nullptr : x, d, M, false);
    lnorm(m, st, x, d, m->y.f(), d, p + "norm2", M, d, d);
    gemm(m, st, m->y.f(), d, m->W(p + "ffn1.w").d, c.ffn, d, d, m->hbuf.f(), c.ffn, m->W(p + "ffn1.b").d, nullptr, 0,
         nullptr, 0, M, true);
    gemm(m, st, m->hbuf.f(), c.ffn, m->W(p + "ffn2.w").d, d, c.ffn, c.ffn, x, d, m->W(p + "ffn2.b").d, x, d, nullptr, 0,
         M, false);
  }
  lnorm(m, st, x, d, m->enc.f(), d, "enc.after_norm", M, d, d);
  // predictor alphas: conv1d k=3 over each window (zero padded at its ends) -> relu -> linear -> sigmoid
  float* col = m->qkv.f();
  float* po = m->ctx.f();
  pfhip::launch_im2col3(m->enc.f(), d, col, 3 * d, d_row_pos, d_row_len, M, d, st);
  if (!lean || !gemv1(col, 3 * d, m->d_predconv, 3 * d, po, d, m->W("pred.conv.b").d, nullptr, c.pred_residual ? m->enc.f() : nullptr, d,
                      nullptr, 0, nullptr, M, d, 3 * d, true))
    gemm(m, st, col, 3 * d, m->d_predconv, d, 3 * d, 3 * d, po, d, m->W("pred.conv.b").d,
         c.pred_residual ? m->enc.f() : nullptr, d, nullptr, 0, M, true);
  pfhip::launch_alpha(po, d, m->W("pred.out.w").d, m->W("pred.out.b").d, c.smooth_factor, c.noise_threshold,
                      m->alphas.f(), M, d, st);
  // ---- CifSearch (:270-345), one block per connection ---------------------------------------------------
  pfhip::launch_cif_stream(m->enc.f(), d, m->alphas.f(), d_segs, B, c.cif_threshold, c.tail_threshold, m->emb.f(), kMaxTok,
                           m->counts.i(), d, st);
  HIP_TRY(hipMemcpyAsync(m->h_counts, m->counts.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  static const bool timing = [] { const char* e = getenv("PFHIP_STREAM_TIMING"); return e && e[0] == '1'; }();
  const auto t_sync0 = std::chrono::steady_clock::now();
  HIP_TRY(hipStreamSynchronize(st));
  if (timing) {
    static double wait_us = 0, enq_us = 0; static int n = 0;
    static auto t_last = t_sync0;
    const auto t1 = std::chrono::steady_clock::now();
    wait_us += std::chrono::duration<double, std::micro>(t1 - t_sync0).count();
    (void)enq_us; (void)t_last;
    if (++n % 50 == 0) { std::fprintf(stderr, "[stream timing] encoder: avg wait in sync %.1f us over %d chunks\n", wait_us / n, n); }
  }
  const int* fires = m->h_counts;
  int ML = 0, maxN = 0;
  for (int b = 0; b < B; ++b) {
    const int N = fires[b];
    if (N > kMaxTok) return fail(PFHIP_ERR_CAPACITY, "more CIF fires in one chunk than the stream workspace holds");
    ss[b]->last_fires = N;
    ss[b]->last_tok_off = ML;
    segs[b].tok_off = ML; segs[b].n_tok = N;
    ML += N;
    maxN = std::max(maxN, N);
  }
  if (ML <= 0) { HIP_TRY(hipGetLastError()); return PFHIP_OK; }          // :472 decoder only if CIF fired
  const int MLp = round_up(ML, pfhip::kTileM);
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
  if (want_logp) HIP_TRY(m->logp.ensure((size_t)MLp * c.vocab * 4));
  // second half of the pinned staging buffer (the first half may still be read by the copies above)
  {
    int* hm2 = reinterpret_cast<int*>(static_cast<char*>(m->h_meta) + m->h_meta_cap / 2);
    const size_t need = (2 * (size_t)B + ML) * 4 + 16 + (size_t)B * sizeof(pfhip::StreamSeg);
    if (m->h_meta_cap / 2 + need > m->h_meta_cap) return fail(PFHIP_ERR_CAPACITY, "too many stream tokens in one batch");
    for (int b = 0; b < B; ++b) { hm2[b] = segs[b].tok_off; hm2[B + b] = segs[b].n_tok; }
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < segs[b].n_tok; ++k) hm2[2 * B + segs[b].tok_off + k] = b * kMaxTok + k;
    pfhip::StreamSeg* h2 = reinterpret_cast<pfhip::StreamSeg*>(reinterpret_cast<char*>(hm2) + (((2 * (size_t)B + ML) * 4 + 15) & ~(size_t)15));
    std::memcpy(h2, segs.data(), (size_t)B * sizeof(pfhip::StreamSeg));
    HIP_TRY(hipMemcpyAsync(d_tok_off, hm2, 2 * (size_t)B * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_src_row, hm2 + 2 * B, (size_t)ML * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_segs, h2, (size_t)B * sizeof(pfhip::StreamSeg), hipMemcpyHostToDevice, st));
  }
  // ---- streaming decoder session (:500): FSMN with the 10-frame caches, cross-attention over each stream's window ----
  float* xd = m->xd.f();
  float* kvbuf = m->qkv.f();
  pfhip::launch_compact(m->emb.f(), xd, d_src_row, ML, d, st);
  const bool lean_dec = lean && ML <= 32;
  auto dec_ffn = [&](const std::string& p, int li, const float* xin, float* o) {
    if (lean_dec) {          // LN1+FFN1 | ffn_norm+FFN2
      const int f = c.dec_ffn;
      if (!m->d_dlnw1 || !gemv1(xin, d, m->d_dlnw1 + (size_t)li * f * d, d, m->hd.f(), f, m->d_dlnb1 + (size_t)li * f,
                                m->d_dlns1 + (size_t)li * f, nullptr, 0, nullptr, 0, nullptr, ML, f, d, true))
        ln_gemm(xin, d, d, p + "norm1", m->W(p + "ffn1.w").d, d, m->hd.f(), c.dec_ffn, m->W(p + "ffn1.b").d, nullptr, 0, nullptr, 0,
                nullptr, 0, nullptr, ML, c.dec_ffn, d, true);
      if (!m->d_dlnw2 || !gemv1(m->hd.f(), f, m->d_dlnw2 + (size_t)li * d * f, f, o, d, m->d_dlnb2 + (size_t)li * d,
                                m->d_dlns2 + (size_t)li * d, nullptr, 0, nullptr, 0, nullptr, ML, d, f, false))
        ln_gemm(m->hd.f(), c.dec_ffn, c.dec_ffn, p + "ffn_norm", m->W(p + "ffn2.w").d, c.dec_ffn, o, d, nullptr, nullptr, 0, nullptr, 0,
                nullptr, 0, nullptr, ML, d, c.dec_ffn, false);
      return;
    }
    lnorm(m, st, xin, d, m->yd.f(), d, p + "norm1", ML, d, d);
    gemm(m, st, m->yd.f(), d, m->W(p + "ffn1.w").d, c.dec_ffn, d, d, m->hd.f(), c.dec_ffn, m->W(p + "ffn1.b").d, nullptr,
         0, nullptr, 0, ML, true);
    lnorm(m, st, m->hd.f(), c.dec_ffn, m->hd2.f(), c.dec_ffn, p + "ffn_norm", ML, c.dec_ffn, c.dec_ffn);
    gemm(m, st, m->hd2.f(), c.dec_ffn, m->W(p + "ffn2.w").d, d, c.dec_ffn, c.dec_ffn, o, d, nullptr, nullptr, 0, nullptr, 0,
         ML, false);
  };
  const int kv_ld = c.dec_layers * 2 * d;
  if (lean_dec) {            // the window is the same for every layer: all K/V projections in one launch
    HIP_TRY(m->kvall.ensure((size_t)32 * (kv_ld + pfhip::kTileN) * 4));
    if (!gemv1(m->enc.f(), d, m->d_kv_all_w, d, m->kvall.f(), kv_ld, m->d_kv_all_b, nullptr, nullptr, 0, nullptr, 0, nullptr, M, kv_ld, d,
               false))
      ln_gemm(m->enc.f(), d, 0, "", m->d_kv_all_w, d, m->kvall.f(), kv_ld, m->d_kv_all_b, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, M,
              kv_ld, d, false);
  }
  for (int i = 0; i < c.dec_layers; ++i) {
    const std::string p = "dec." + std::to_string(i) + ".";
    dec_ffn(p, i, xd, m->td.f());
    lnorm(m, st, m->td.f(), d, m->t2.f(), d, p + "norm2", ML, d, d);
    pfhip::launch_fsmn_cached(m->t2.f(), m->W(p + "fsmn.w").d, xd, xd, d_segs, B, i, d, st);
    if (lean_dec) {
      if (!m->d_dlnw3 || !gemv1(xd, d, m->d_dlnw3 + (size_t)i * d * d, d, m->qd.f(), d, m->d_dlnb3 + (size_t)i * d,
                                m->d_dlns3 + (size_t)i * d, nullptr, 0, nullptr, 0, nullptr, ML, d, d, false))
        ln_gemm(xd, d, d, p + "norm3", m->W(p + "q.w").d, d, m->qd.f(), d, m->W(p + "q.b").d, nullptr, 0, nullptr, 0, nullptr, 0,
                nullptr, ML, d, d, false);
      const float* kvl = m->kvall.f() + (size_t)i * 2 * d;
      if (att_out_on && pfhip::launch_fused_att_out(m->qd.f(), d, kvl, kv_ld, kvl + d, kv_ld, ML, M, c.n_head, att_scale, m->W(p + "out.w").d, d, xd, d,
                                      m->W(p + "out.b").d, xd, d, nullptr, 0, nullptr, d, st))
        continue;
      if (!pfhip::launch_window_attention(m->qd.f(), d, kvl, kv_ld, kvl + d, kv_ld, m->ctxd.f(), d, ML, M, c.n_head, att_scale, st))
        pfhip::launch_attention(m->qd.f(), d, kvl, kv_ld, kvl + d, kv_ld, m->ctxd.f(), d, d_tok_off, d_tok_len, d_off, d_len, B,
                                c.n_head, maxN, att_scale, st);
      if (!gemv1(m->ctxd.f(), d, m->W(p + "out.w").d, d, xd, d, m->W(p + "out.b").d, nullptr, xd, d, nullptr, 0, nullptr, ML, d, d, false))
        ln_gemm(m->ctxd.f(), d, 0, "", m->W(p + "out.w").d, d, xd, d, m->W(p + "out.b").d, xd, d, nullptr, 0, nullptr, 0, nullptr,
                ML, d, d, false);
      continue;
    }
    lnorm(m, st, xd, d, m->yd.f(), d, p + "norm3", ML, d, d);
    gemm(m, st, m->yd.f(), d, m->W(p + "q.w").d, d, d, d, m->qd.f(), d, m->W(p + "q.b").d, nullptr, 0, nullptr, 0, ML, false);
    gemm(m, st, m->enc.f(), d, m->W(p + "kv.w").d, 2 * d, d, d, kvbuf, 2 * d, m->W(p + "kv.b").d, nullptr, 0, nullptr, 0, M,
         false);
    if (!pfhip::launch_window_attention_segments(m->qd.f(), d, kvbuf, 2 * d, kvbuf + d, 2 * d, m->ctxd.f(), d, d_tok_off, d_tok_len, d_off,
                                                 d_len, B, c.n_head, maxN, maxn, att_scale, st))
      pfhip::launch_attention(m->qd.f(), d, kvbuf, 2 * d, kvbuf + d, 2 * d, m->ctxd.f(), d, d_tok_off, d_tok_len, d_off, d_len, B,
                              c.n_head, maxN, att_scale, st);
    gemm(m, st, m->ctxd.f(), d, m->W(p + "out.w").d, d, d, d, xd, d, m->W(p + "out.b").d, xd, d, nullptr, 0, ML, false);
  }
  dec_ffn("dec3.", c.dec_layers, xd, m->td.f());
  if (lean_dec) {
    ln_gemm(m->td.f(), d, d, "dec.after_norm", m->W("dec.out.w").d, d, m->logits.f(), m->vocab_pad, m->d_vocab_bias, nullptr, 0,
            nullptr, 0, nullptr, 0, nullptr, ML, c.vocab, d, false);
  } else {
    lnorm(m, st, m->td.f(), d, m->yd.f(), d, "dec.after_norm", ML, d, d);
    gemm(m, st, m->yd.f(), d, m->W("dec.out.w").d, c.vocab, d, d, m->logits.f(), m->vocab_pad, m->d_vocab_bias, nullptr, 0,
         nullptr, 0, ML, false);
  }
  pfhip::launch_logsoftmax_argmax(m->logits.f(), m->vocab_pad, ML, c.vocab, want_logp ? m->logp.f() : nullptr,
                                  static_cast<int32_t*>(m->ids.p), st);
  std::vector<int32_t> ids(ML);
  HIP_TRY(hipMemcpyAsync(ids.data(), m->ids.p, (size_t)ML * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  for (int b = 0; b < B; ++b) {
    ss[b]->last_has_logp = want_logp;
    for (int k = 0; k < segs[b].n_tok; ++k) outs[b]->push_back(ids[segs[b].tok_off + k]);        // OnlineGreedySearch paraformer.cpp:362-371
  }
  return PFHIP_OK;
}

// window = feats_cache_ ++ rows[r0, r0+nr) (AddOverlapChunk :397-413); returns the window length in *n_out
pfhip_status add_overlap_chunk(pfhip_stream* s, int r0, int nr, bool input_finished, Recorder& rec, int* n_out) {
  pfhip_model* m = s->m;
  const int FD = m->feat_dim, FP = m->feat_pad;
  const int nc = s->n_featc;
  int n = nc + nr;
  if (n > 128) return fail(PFHIP_ERR_ARG, "stream window too long");
  rec.copy(Recorder::kWindow, s->chunk.f(), FP, s->featc.f(), FD, nc, FD);
  rec.copy(Recorder::kWindow, s->chunk.f() + (size_t)nc * FP, FP, s->rows.f() + (size_t)r0 * FD, FD, nr, FD);
  int keep;
  if (input_finished) {
    keep = s->chunk_size[0];
  } else {
    keep = s->chunk_size[0] + s->chunk_size[2];
  }
  if (keep > n) return fail(PFHIP_ERR_ARG, "stream window shorter than the look-back cache");
  // new feats_cache_ = last `keep` rows of the (unpadded) window; chunk has stride FP, featc FD
  rec.copy(Recorder::kFeatc, s->featc.f(), FD, s->chunk.f() + (size_t)(n - keep) * FP, FP, keep, FD);
  s->n_featc = keep;
  if (input_finished && !s->is_last_chunk) {
    const int total = s->chunk_size[0] + s->chunk_size[1] + s->chunk_size[2];
    if (total > n) {                                          // zero rows up to 20 (:402-408)
      rec.copy(Recorder::kZeroPad, s->chunk.f() + (size_t)n * FP, FP, nullptr, 0, total - n, 0);
      n = total;
    }
  }
  *n_out = n;
  return PFHIP_OK;
}

// OnlineLfrCmvn (:196-238) over the T frames at the front of fb[cur]; emits n rows (CMVN, x sqrt(d), PE) into
// s->rows and rotates the splice cache.
pfhip_status online_lfr_cmvn(pfhip_stream* s, int T, bool input_finished, Recorder& rec, int* n_rows) {
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
  if (n_out > 0) rec.lfr.push_back(pfhip::StreamLfrOp{fb, s->rows.f(), T, n_out, s->start_idx, 0});
  s->start_idx += n_out;                                               // GetPosEmb :242-243
  const int keep = T - splice;                                         // lfr_splice_cache_ = frames[splice:] (:226-228)
  rec.copy(Recorder::kSpliceRotate, s->fb[s->fb_cur ^ 1].f(), 80, fb + (size_t)splice * 80, 80, keep, 80);
  s->fb_cur ^= 1;
  s->n_splice = keep;
  *n_rows = n_out;
  return PFHIP_OK;
}

// ExtractFeats (:147-194) + x*sqrt(d) + GetPosEmb (:549-555): leaves `*n_rows` finished LFR rows in s->rows.
pfhip_status extract_feats(pfhip_stream* s, const float* pcm, int len, bool input_finished, Recorder& rec, int* n_rows) {
  pfhip_model* m = s->m;
  *n_rows = 0;
  const int fl = 400, fs = 160, lfr_m = m->cfg.lfr_m;
  // FbankKaldi (:119-145): prepend input_cache_, keep what follows the last frame shift for the next call
  std::vector<float>& waves = s->waves;            // lives until the flush of this call has copied it
  waves.assign(s->input_cache.begin(), s->input_cache.end());
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
    float* fb = s->fb[s->fb_cur].f();
    rec.pcm_src.push_back(waves.data());
    rec.pcm_len.push_back(used);
    rec.nframes.push_back(frame_number);
    rec.frame_dst.push_back(Recorder::FrameDst{fb + (size_t)base * 80, frame_number});
    if (fresh) {     // lfr_splice_cache_ = (lfr_m-1)/2 copies of the first frame (:155-158)
      rec.copy(Recorder::kFreshReplicate, fb, 80, fb + (size_t)base * 80, 0, base, 80);
      s->n_splice = base;
    }
    if (frame_number + s->n_splice >= lfr_m) {
      rc = online_lfr_cmvn(s, s->n_splice + frame_number, input_finished, rec, n_rows);
    } else {
      // (:172-177) the splice cache just grows.  The reference leaves the raw 80-dim frames in wav_feats
      // here and feeds them on — a latent bug only reachable with < 55 ms of audio in a non-final call,
      // which the 2-pass server (9600-sample chunks) never sends; no window is produced.
      s->n_splice += frame_number;
    }
  } else if (input_finished) {
    if (s->n_splice > 0) rc = online_lfr_cmvn(s, s->n_splice, true, rec, n_rows);   // (:179-189)
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
  m = pfhip_detail::route_stream(m);           // a group: the connection lives on the replica (GPU) with the fewest open streams
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
  ++m->live_streams;
  *out = s.release();
  return PFHIP_OK;
}

void pfhip_stream_destroy(pfhip_stream* s) {
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(s->m->mu);
    (void)hipSetDevice(s->m->device);
    (void)hipStreamSynchronize(s->m->own_stream);
    for (Buf* b : {&s->fb[0], &s->fb[1], &s->rows, &s->featc, &s->chunk, &s->carry, &s->dcache})
      b->release();
  }
  --s->m->live_streams;
  delete s;
}

int pfhip_stream_last_path(const pfhip_stream* s) { return s ? s->last_path : 0; }

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

}  // extern "C"

namespace {

// One connection's share of a (batched) ParaformerOnline::Forward call (:525-601): the host control flow is cut where
// the reference calls ForwardChunk, so that the chunks of many connections can run as one packed forward.
struct Call {
  pfhip_stream* s;
  const float* pcm;
  int n_samples;
  bool fin;
  std::vector<int32_t> out;
  bool has_window = false;     // a window is waiting in s->chunk
  bool second = false;         // after this window: the (:560-579) last-chunk window
  bool reinit = false;         // after the last window: Reset + InitCache (:589-593 / :532-540)
  int nr = 0;
};

// up to the first ForwardChunk
pfhip_status prepare_first(Call& c, Recorder& rec) {
  pfhip_stream* s = c.s;
  pfhip_model* m = s->m;
  // (:532-540) a short final call after the first chunk: flush the look-back cache as the last chunk
  s->last_path = 0;
  if (c.n_samples < 16 * 60 && c.fin && !s->is_first_chunk) {
    s->is_last_chunk = true;
    s->last_path = 1;
    s->win_n = s->n_featc;
    rec.copy(Recorder::kWindow, s->chunk.f(), m->feat_pad, s->featc.f(), m->feat_dim, s->win_n, m->feat_dim);
    c.has_window = true;
    c.reinit = true;
    return PFHIP_OK;
  }
  if (s->is_first_chunk) s->is_first_chunk = false;
  pfhip_status rc = extract_feats(s, c.pcm, c.n_samples, c.fin, rec, &c.nr);
  if (rc) return rc;
  if (c.nr == 0) return PFHIP_OK;                                      // (:545-547)
  if (c.fin) {
    if (c.nr + s->chunk_size[2] <= s->chunk_size[1]) { s->is_last_chunk = true; s->last_path = 2; }      // (:557-559)
    else { c.second = true; s->last_path = 3; }                                     // (:560-579) first chunk + last chunk
    c.reinit = true;
  }
  rc = add_overlap_chunk(s, 0, c.nr, c.fin, rec, &s->win_n);
  if (rc) return rc;
  c.has_window = true;
  return PFHIP_OK;
}

// (:566-579) the last-chunk window of a final call that did not fit one chunk
pfhip_status prepare_second(Call& c, Recorder& rec) {
  pfhip_stream* s = c.s;
  s->is_last_chunk = true;
  const int k = c.nr + s->chunk_size[2] - s->chunk_size[1];
  return add_overlap_chunk(s, c.nr - k, k, c.fin, rec, &s->win_n);
}

pfhip_status forward_calls(pfhip_model* m, std::vector<Call>& calls, hipStream_t st) {
  bool want_logp = false;
  Recorder rec;
  for (Call& c : calls) {
    pfhip_status rc = prepare_first(c, rec);
    if (rc) return rc;
    want_logp = want_logp || c.s->debug;
  }
  for (int round = 0; round < 2; ++round) {
    std::vector<pfhip_stream*> ss;
    std::vector<std::vector<int32_t>*> outs;
    for (Call& c : calls) {
      if (round == 0 ? !c.has_window : !c.second) continue;
      if (round == 1) { pfhip_status rc = prepare_second(c, rec); if (rc) return rc; }
      ss.push_back(c.s);
      outs.push_back(&c.out);
    }
    pfhip_status rc = forward_windows(m, ss, st, outs, want_logp, rec);
    if (rc) return rc;
  }
  pfhip_status rc = PFHIP_OK;
  for (Call& c : calls)
    if (c.reinit) {
      reset_cache(c.s);
      pfhip_status r2 = init_cache(c.s, st);
      if (!rc) rc = r2;
    }
  return rc;
}

// The batched call with one status PER connection: a connection whose token buffer is too small gets PFHIP_ERR_CAPACITY
// (n_tokens = the count it needed) and the others still receive their ids; only a failure of the shared forward itself
// fails everybody, and then every stream of the batch is re-initialised (Reset + InitCache) so that its caches do not
// keep a half-advanced chunk.  Returns the first non-OK status.
pfhip_status forward_batch_each(pfhip_stream* const* streams, int n_streams, const float* const* pcm, const int* n_samples,
                                const int* input_finished, int32_t* const* token_ids, const int* cap, int* n_tokens,
                                pfhip_status* each, std::string* each_err) {
  auto all = [&](pfhip_status st) {
    if (each) for (int i = 0; i < n_streams; ++i) { each[i] = st; if (each_err) each_err[i] = last_error(); }
    return st;
  };
  if (!streams || n_streams <= 0 || !pcm || !n_samples || !input_finished || !token_ids || !cap || !n_tokens)
    return fail(PFHIP_ERR_ARG, "bad argument");
  pfhip_model* m = streams[0] ? streams[0]->m : nullptr;
  if (!m) return all(fail(PFHIP_ERR_ARG, "null stream"));
  std::vector<Call> calls(n_streams);
  for (int i = 0; i < n_streams; ++i) {
    pfhip_stream* s = streams[i];
    if (!s || s->m != m) return all(fail(PFHIP_ERR_ARG, "streams of one batch must belong to one model"));
    for (int j = 0; j < i; ++j) if (streams[j] == s) return all(fail(PFHIP_ERR_ARG, "a stream appears twice in one batch"));
    if (n_samples[i] < 0 || (n_samples[i] > 0 && !pcm[i])) return all(fail(PFHIP_ERR_ARG, "bad pcm buffer"));
    if (n_samples[i] > kMaxSamples) return all(fail(PFHIP_ERR_ARG, "more than 32000 samples in one streaming call"));
    calls[i].s = s; calls[i].pcm = pcm[i]; calls[i].n_samples = n_samples[i]; calls[i].fin = input_finished[i] != 0;
    n_tokens[i] = 0;
  }
  std::lock_guard<std::mutex> lk(m->mu);
  HIP_TRY(hipSetDevice(m->device));
  hipStream_t st = m->own_stream;
  m->prof_stream = st;
  pfhip_status rc = forward_calls(m, calls, st);
  if (rc) {
    const std::string why = last_error();
    for (Call& c : calls) { reset_cache(c.s); (void)init_cache(c.s, st); }
    (void)hipStreamSynchronize(st);
    last_error() = why + " (every stream of the batch was reset)";
    return all(rc);
  }
  pfhip_status first = PFHIP_OK;
  std::string first_err;
  for (int i = 0; i < n_streams; ++i) {
    n_tokens[i] = (int)calls[i].out.size();
    pfhip_status si = PFHIP_OK;
    if ((int)calls[i].out.size() > cap[i]) {
      si = fail(PFHIP_ERR_CAPACITY, "token_ids too small (n_tokens holds the count needed; the chunk's ids are lost)");
      if (!first) { first = si; first_err = last_error(); }
    } else {
      for (size_t k = 0; k < calls[i].out.size(); ++k) token_ids[i][k] = calls[i].out[k];
    }
    if (each) { each[i] = si; if (each_err) each_err[i] = si ? last_error() : std::string(); }
  }
  if (first) last_error() = first_err;
  return first;
}
}  // namespace

extern "C" {

pfhip_status pfhip_stream_forward_batch(pfhip_stream* const* streams, int n_streams, const float* const* pcm, const int* n_samples,
                                        const int* input_finished, int32_t* const* token_ids, const int* cap, int* n_tokens) {
  last_error().clear();
  if (!streams || n_streams <= 0 || !pcm || !n_samples || !input_finished || !token_ids || !cap || !n_tokens)
    return fail(PFHIP_ERR_ARG, "bad argument");
  // streams of a replica group (pfhip_create_group) may sit on different devices: one sub-batch per replica, run concurrently
  std::vector<pfhip_model*> models;
  for (int i = 0; i < n_streams; ++i) {
    if (!streams[i]) return fail(PFHIP_ERR_ARG, "null stream");
    if (std::find(models.begin(), models.end(), streams[i]->m) == models.end()) models.push_back(streams[i]->m);
  }
  if (models.size() == 1)
    return forward_batch_each(streams, n_streams, pcm, n_samples, input_finished, token_ids, cap, n_tokens, nullptr, nullptr);
  auto head_of = [](pfhip_model* m) { return m->group_head ? m->group_head : m; };
  for (pfhip_model* m : models)
    if (head_of(m) != head_of(models[0])) return fail(PFHIP_ERR_ARG, "streams of one batch must belong to one model");
  struct Part { std::vector<int> idx; pfhip_status st = PFHIP_OK; std::string err; };
  std::vector<Part> parts(models.size());
  for (int i = 0; i < n_streams; ++i)
    parts[(size_t)(std::find(models.begin(), models.end(), streams[i]->m) - models.begin())].idx.push_back(i);
  std::vector<std::thread> pool;
  for (Part& p : parts)
    pool.emplace_back([&, pp = &p] {
      const size_t n = pp->idx.size();
      std::vector<pfhip_stream*> ss(n); std::vector<const float*> pc(n); std::vector<int> ns(n), fin(n), cp(n), nt(n); std::vector<int32_t*> ids(n);
      for (size_t k = 0; k < n; ++k) {
        const int i = pp->idx[k];
        ss[k] = streams[i]; pc[k] = pcm[i]; ns[k] = n_samples[i]; fin[k] = input_finished[i]; cp[k] = cap[i]; ids[k] = token_ids[i];
      }
      last_error().clear();
      pp->st = forward_batch_each(ss.data(), (int)n, pc.data(), ns.data(), fin.data(), ids.data(), cp.data(), nt.data(), nullptr, nullptr);
      pp->err = last_error();
      for (size_t k = 0; k < n; ++k) n_tokens[pp->idx[k]] = nt[k];
    });
  for (std::thread& t : pool) t.join();
  for (const Part& p : parts)
    if (p.st) { last_error() = p.err; return p.st; }
  return PFHIP_OK;
}

}  // extern "C"

// ---- cross-connection batching behind the per-connection call --------------------------------------------------------
// The 2-pass server runs one strand per connection, each calling ParaformerOnline::Forward on its own stream object
// (websocket-server-2pass.cpp:266-297).  With pfhip_set_stream_batching(wait_us > 0) concurrent pfhip_stream_forward callers on
// streams of one model are merged like the offline callers (pfhip_set_batching): the first to arrive leads, waits up to
// wait_us for others, runs ONE batched forward (forward_calls) and hands every caller its ids.
struct StreamReq : pfhip_detail::MergeReqBase {
  pfhip_stream* s; const float* pcm; int n; int fin; int32_t* ids; int cap; int* n_out;
  pfhip_status st = PFHIP_OK; std::string err;
};

namespace {

void run_requests(const std::vector<StreamReq*>& reqs) {
  const int n = (int)reqs.size();
  std::vector<pfhip_stream*> ss(n);
  std::vector<const float*> pcm(n);
  std::vector<int> ns(n), fin(n), cap(n), nt(n);
  std::vector<int32_t*> ids(n);
  for (int i = 0; i < n; ++i) { ss[i] = reqs[i]->s; pcm[i] = reqs[i]->pcm; ns[i] = reqs[i]->n; fin[i] = reqs[i]->fin; ids[i] = reqs[i]->ids; cap[i] = reqs[i]->cap; }
  std::vector<pfhip_status> each(n, PFHIP_OK);
  std::vector<std::string> err(n);
  pfhip_detail::last_error().clear();
  const pfhip_status st = forward_batch_each(ss.data(), n, pcm.data(), ns.data(), fin.data(), ids.data(), cap.data(), nt.data(),
                                             each.data(), err.data());
  if (st && !each.empty() && each[0] == PFHIP_OK && err[0].empty()) {      // argument-level failure before `each` was filled
    bool any = false;
    for (int i = 0; i < n; ++i) any = any || each[i] != PFHIP_OK;
    if (!any) for (int i = 0; i < n; ++i) { each[i] = st; err[i] = pfhip_detail::last_error(); }
  }
  for (int i = 0; i < n; ++i) { *reqs[i]->n_out = nt[i]; reqs[i]->st = each[i]; reqs[i]->err = err[i]; }
}

pfhip_status stream_forward_queued(pfhip_model* m, StreamReq& me) {
  int wait_us, cap;
  { std::lock_guard<std::mutex> l(m->sq.mu); wait_us = m->stream_wait_us; cap = m->stream_max; }
  m->sq.submit(
      me, wait_us,
      // wait for company: until every open stream has a call queued (connections advancing together), the cap, or the deadline
      [&](const std::deque<StreamReq*>& q) { return (int)q.size() >= std::min(cap, std::max(1, m->live_streams.load())); },
      [&](std::deque<StreamReq*>& q, std::vector<StreamReq*>& take) {
        std::deque<StreamReq*> later;
        while (!q.empty() && (int)take.size() < cap) {
          StreamReq* r = q.front();
          q.pop_front();
          bool dup = false;                       // two queued calls on ONE stream must stay in order: the second waits
          for (StreamReq* t : take) dup = dup || t->s == r->s;
          if (dup) later.push_back(r); else take.push_back(r);
        }
        for (auto it = later.rbegin(); it != later.rend(); ++it) q.push_front(*it);
      },
      [&](std::vector<StreamReq*>& take) { run_requests(take); });
  if (me.st != PFHIP_OK) pfhip_detail::last_error() = me.err;
  return me.st;
}

}  // namespace

extern "C" {

pfhip_status pfhip_set_stream_batching(pfhip_model* m, int wait_us, int max_streams) {
  last_error().clear();
  if (!m || wait_us < 0 || max_streams < 1) return fail(PFHIP_ERR_ARG, "bad argument");
  for (size_t i = 0; i <= m->replicas.size(); ++i) {
    pfhip_model* r = i == 0 ? m : m->replicas[i - 1];
    std::lock_guard<std::mutex> ql(r->sq.mu);
    r->stream_wait_us = wait_us;
    r->stream_max = max_streams;
  }
  return PFHIP_OK;
}

pfhip_status pfhip_stream_forward(pfhip_stream* s, const float* pcm, int n_samples, int input_finished,
                                  int32_t* token_ids, int cap, int* n_tokens) {
  if (!s || !n_tokens) { last_error().clear(); return fail(PFHIP_ERR_ARG, "bad argument"); }
  if (s->m->stream_wait_us > 0 && !s->debug) {
    last_error().clear();
    if (n_samples < 0 || (n_samples > 0 && !pcm) || n_samples > kMaxSamples) return fail(PFHIP_ERR_ARG, "bad pcm buffer");
    StreamReq me;
    me.s = s; me.pcm = pcm; me.n = n_samples; me.fin = input_finished; me.ids = token_ids; me.cap = cap; me.n_out = n_tokens;
    *n_tokens = 0;
    return stream_forward_queued(s->m, me);
  }
  int32_t* ids[1] = {token_ids};
  const float* p[1] = {pcm};
  return pfhip_stream_forward_batch(&s, 1, p, &n_samples, &input_finished, ids, &cap, n_tokens);
}

pfhip_status pfhip_stream_set_debug(pfhip_stream* s, int on) {
  if (!s) return fail(PFHIP_ERR_ARG, "null stream");
  s->debug = (on & 1) != 0;          // bit 0: keep log-probs of the chunks this stream takes part in
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
  // the last window of this stream, where the last batched forward left it in the model's packed workspace (valid
  // until the next forward on this model)
  if (nm == "chunk") {
    const size_t n = (size_t)s->last_n * m->feat_dim;
    if (n > cap_floats) return fail(PFHIP_ERR_CAPACITY, "dst too small");
    if (n) HIP_TRY(hipMemcpy2DAsync(dst, (size_t)m->feat_dim * 4, s->chunk.p, (size_t)m->feat_pad * 4, (size_t)m->feat_dim * 4,
                                    s->last_n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (n_out) *n_out = n;
    return PFHIP_OK;
  }
  if (nm == "enc") return copy_out(m->enc.f() + (size_t)s->last_row_off * d, (size_t)s->last_n * d, dst, cap_floats, n_out, st);
  if (nm == "alphas") return copy_out(m->alphas.f() + s->last_row_off, (size_t)s->last_n, dst, cap_floats, n_out, st);
  if (nm == "emb") return copy_out(m->emb.f() + (size_t)s->last_slot * kMaxTok * d, (size_t)s->last_fires * d, dst, cap_floats, n_out, st);
  if (nm == "logp") {
    if (!s->last_has_logp && s->last_fires > 0) return fail(PFHIP_ERR_ARG, "enable pfhip_stream_set_debug before the call");
    return copy_out(m->logp.f() + (size_t)s->last_tok_off * m->cfg.vocab, (size_t)s->last_fires * m->cfg.vocab, dst, cap_floats, n_out, st);
  }
  return fail(PFHIP_ERR_ARG, "unknown tensor name " + nm);
}

}  // extern "C"
