// Internal definitions shared by pfhip.cpp (offline forward) and stream.cpp (chunk-streaming forward).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pfhip.h"
#include "kernels.h"
#include "merge_queue.h"

namespace pfhip_detail {

std::string& last_error();          // thread-local, defined in pfhip.cpp

inline pfhip_status fail(pfhip_status st, const std::string& msg) {
  last_error() = msg;
  return st;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess)                                                                   \
      return pfhip_detail::fail(PFHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

// bumped whenever a workspace buffer is (re)allocated or freed: cached hipGraphs hold raw pointers
std::atomic<uint64_t>& buf_epoch();          // pfhip.cpp

struct Buf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    buf_epoch().fetch_add(1);
    if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
    bytes = (bytes + (1u << 20) - 1) & ~((size_t)(1u << 20) - 1);
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() { if (p) { (void)hipFree(p); buf_epoch().fetch_add(1); } p = nullptr; cap = 0; }
  float* f() const { return static_cast<float*>(p); }
  int* i() const { return static_cast<int*>(p); }
};

struct Tensor {
  const float* d = nullptr;   // device
  const float* h = nullptr;   // host (only valid during create)
  std::vector<int> shape;
  size_t n = 0;
};

struct Config {
  int d_model = 512, n_head = 4, ffn = 2048, enc_layers = 50, dec_layers = 16, dec_ffn = 2048;
  int kernel = 11, vocab = 8404, n_mels = 80, lfr_m = 7, lfr_n = 6, pred_residual = 0, contextual = 0, timestamp = 0;
  float smooth_factor2 = 0.25f, noise_threshold2 = 0.01f;      // CifPredictorV3 timestamp head
  float cif_threshold = 1.0f, tail_threshold = 0.45f, smooth_factor = 1.0f, noise_threshold = 0.0f;
  int sample_rate = 16000;
};

struct FrontendTables {
  float* d_window = nullptr; double* d_tw = nullptr; int* d_mel_off = nullptr; int* d_mel_size = nullptr;
  float* d_mel_w = nullptr;
};
pfhip_status build_frontend_tables(int n_mels, int sample_rate, FrontendTables* ft);   // pfhip.cpp

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// A linear layer repacked for the GEMM kernels: W zero-padded to [ceil(N/128)*128][ceil(K/32)*32], bias to the
// padded N, so pad outputs are exact zeros (odd widths: FSMN-VAD 140/250/248, punctuation head 6).
struct Lin { float* w = nullptr; float* b = nullptr; int N = 0, K = 0, Np = 0, Kp = 0; float ws = 1.0f; };
inline pfhip_status pack_linear(const float* w, const float* bias, int N, int K, Lin* out) {
  out->N = N; out->K = K; out->Np = round_up(N, 128); out->Kp = round_up(K, 32);
  std::vector<float> pw((size_t)out->Np * out->Kp, 0.f), pb((size_t)out->Np, 0.f);
  for (int n = 0; n < N; ++n) std::memcpy(&pw[(size_t)n * out->Kp], w + (size_t)n * K, sizeof(float) * K);
  if (bias) std::memcpy(pb.data(), bias, sizeof(float) * N);
  float mx = 0.f;
  for (float v : pw) mx = std::max(mx, std::fabs(v));
  out->ws = pfhip::best_w_scale(mx);
  HIP_TRY(hipMalloc((void**)&out->w, pw.size() * 4));
  HIP_TRY(hipMemcpy(out->w, pw.data(), pw.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&out->b, pb.size() * 4));
  HIP_TRY(hipMemcpy(out->b, pb.data(), pb.size() * 4, hipMemcpyHostToDevice));
  return PFHIP_OK;
}
inline void lin_gemm(hipStream_t s, const Lin& l, const float* A, int lda, float* C, int ldc, const float* R1, int ldr1,
                     const float* R2, int ldr2, int M, bool relu) {
  pfhip::launch_gemm_f32(A, lda, l.w, l.Kp, C, ldc, l.b, R1, ldr1, R2, ldr2, M, l.Np, l.Kp, relu, false, s, l.ws);
}
inline void free_lin(Lin& l) { if (l.w) (void)hipFree(l.w); if (l.b) (void)hipFree(l.b); l.w = l.b = nullptr; }

struct ProfRec { int cls; hipEvent_t e0, e1; };

}  // namespace pfhip_detail

using pfhip_detail::Buf;
using pfhip_detail::Config;
using pfhip_detail::ProfRec;
using pfhip_detail::Tensor;

struct BatchReq;
struct StreamReq;

struct pfhip_model {
  int device = 0;
  hipStream_t own_stream = nullptr;
  // decoder K/V projections of a large batch run beside the (under-filled) token-side launches: pfhip.cpp enqueue_locked
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_enc_ready = nullptr;
  std::vector<hipEvent_t> ev_kv;
  std::mutex mu;
  Config cfg;
  int feat_dim = 560, feat_pad = 576, vocab_pad = 8448;

  float* d_blob = nullptr;
  std::map<std::string, Tensor> t;
  float* d_w0qkv = nullptr;     // enc.0.qkv.w K-padded to feat_pad
  float* d_predconv = nullptr;  // [d][3*d] im2col order
  float* d_vocab_bias = nullptr;  // dec.out.b padded to vocab_pad
  float* d_kv_all_w = nullptr;    // the decoder layers' kv.w stacked [layers * 2d][d] (+ bias): one launch projects a streaming window for all layers
  float* d_kv_all_b = nullptr;
  // LayerNorm folded into its consumer GEMM (gemm_x6.hip LN-on-load): per encoder layer W * gamma (per input column) and
  // bias + W beta, for qkv (layers >= 1) and ffn1; [layers][N][d] / [layers][N]
  float* d_lnw_qkv = nullptr; float* d_lnb_qkv = nullptr; float* d_lnw_ffn1 = nullptr; float* d_lnb_ffn1 = nullptr;
  float* d_lns_qkv = nullptr; float* d_lns_ffn1 = nullptr;      // column sums of the folded weights [layers][N]
  // fp16 plane images (gemm_p3.hip) of the encoder's four large weights per layer, scale baked in: [layer] { qkv' | out | ffn1' |
  // ffn2 }, each image hi plane then lo plane.  wp_layer_bytes = 0: not built (the fp32 path serves every batch size).
  unsigned char* d_wplanes = nullptr;
  size_t wp_layer_bytes = 0, wp_off_out = 0, wp_off_ffn1 = 0, wp_off_ffn2 = 0;
  // guard of the fp16 two-plane domain (pfhip.cpp fetch_locked): the forward's flag word (inside `meta`, cleared by the metadata
  // upload), its pinned host mirror, what a re-run on the exact kernels needs, and the count of such re-runs
  double static_bound = 0.0;          // load-time bound on |Linear(LayerNorm(x))| over the model's layers
  bool always_exact = false;          // that bound reaches fp16's range: every forward runs the exact kernels
  int* d_range_flag = nullptr;
  int* h_flag = nullptr;
  int range_hit = 0, debug_range_flag = 0;
  bool exact_rerun = false, last_feats_only = false;
  long long range_fallbacks = 0;
  const float* last_pcm = nullptr;
  std::vector<int64_t> last_off;
  std::vector<int> last_n;
  int plane_forwards = 0;                                       // forwards of this context that took the plane path (debug read-out)
  Buf ctxP, xP, hP;                                             // activation plane images of a large batch: context, residual stream, FFN hidden
  Buf kvP;                                                      // K | V of an encoder layer as row-major fp16 planes (attention_p3.hip)
  int kvplane_forwards = 0;                                     // forwards whose encoder attention took K | V as planes (debug read-out)
  Buf encP, xdP;                                                // decoder on plane operands: images of the encoder output and of the token-side residual stream
  // fp16 plane images of the decoder's large weights per layer (dec3 = the last entry: FFN only): [layer] { ffn1' | ffn2' | kv | out }
  unsigned char* d_dwplanes = nullptr;
  size_t dwp_layer_bytes = 0, dwp_off_ffn2 = 0, dwp_off_kv = 0, dwp_off_out = 0;
  int dec_plane_forwards = 0;                                   // forwards whose decoder took the plane path (debug read-out)
  // the same for the decoder's FFN: ffn1 with norm1, ffn2 with ffn_norm; [dec_layers + 1] entries (the last one is dec3)
  float* d_dlnw1 = nullptr; float* d_dlnb1 = nullptr; float* d_dlns1 = nullptr;
  float* d_dlnw2 = nullptr; float* d_dlnb2 = nullptr; float* d_dlns2 = nullptr;
  float* d_dlnw3 = nullptr; float* d_dlnb3 = nullptr; float* d_dlns3 = nullptr;     // norm3 -> q projection (streaming latency path)
  // timestamp head repacks: ConvTranspose1d as [3d][d] + tiled bias, both LSTM directions' input weights [8d][d] + summed
  // biases, recurrent weights [2][4d][d]
  float* d_up_w = nullptr; float* d_up_b = nullptr; float* d_wih = nullptr; float* d_bih = nullptr; float* d_whh = nullptr;
  // front-end tables
  float* d_window = nullptr; double* d_tw = nullptr; int* d_mel_off = nullptr; int* d_mel_size = nullptr;
  float* d_mel_w = nullptr; float* d_inv_ts = nullptr;

  // workspace
  Buf pcm, meta, feats, x0, x, y, qkv, mem, ctx, hbuf, enc, alphas, counts;
  Buf emb, xd, yd, hd, hd2, td, t2, qd, ctxd, logits, logp, ids, dmeta, cat, hw, hwkv;
  Buf sseg;                     // StreamSeg descriptors of a streaming batch
  Buf kvside;                   // [dec_layers][Mp][2d]: every decoder layer's K/V projection of the encoder output (side stream)
  Buf lnstats2;                 // the decoder's second hand-off (FFN1 -> ffn_norm -> FFN2): [ML][dec_ffn / 128][2]
  Buf lnstats;                  // per-row LayerNorm statistics handed from a producing GEMM's epilogue to the consumer [M][4][2]
  Buf kvall;                    // one window's K/V projections of every decoder layer [32][layers * 2d]
  Buf fbk, d_ops;               // streaming batch: fbank frames of all connections, operation descriptors
  void* h_ops = nullptr; size_t h_ops_cap = 0;       // pinned staging of the same (+ the batch's PCM)
  Buf ts_up, ts_gx, ts_y, ts_hx, ts_a2, ts_alphas, ts_peaks, ts_meta, ts_cst;
  bool have_ts = false;
  int debug_blstm_flag = 0;        // pfhip_debug_poke
  hipStream_t blstm_stream = nullptr;      // per DEVICE (on the weight owner): every context's persistent recurrence runs here, in turn
  hipEvent_t ev_ts_in = nullptr, ev_ts_out = nullptr;
  bool ts_persistent = false;              // the last timestamp head ran the persistent kernel: its error word is read with the results
  int blstm_fallbacks = 0;         // timestamp requests served by the per-step recurrence after a barrier time-out
  float out2_b = 0.f;
  int n_hw = 0;                  // hotword embeddings resident in `hw` ([n_hw, d])
  void* h_meta = nullptr; size_t h_meta_cap = 0;     // pinned
  int* h_counts = nullptr;                            // pinned [2*B]
  size_t h_counts_cap = 0;

  // state of the last forward
  int B = 0, M = 0, ML = 0, maxT = 0, maxL = 0;
  std::vector<int> T, row_off, n_fires, token_num, tok_off;
  bool have_logp = false;
  // device views into meta / dmeta
  int *m_frame_off = nullptr, *m_nframes = nullptr, *m_row_off = nullptr, *m_len = nullptr,
      *m_row_pos = nullptr, *m_row_len = nullptr;
  int64_t* m_sample_off = nullptr;
  int *m_tok_off = nullptr, *m_tok_len = nullptr, *m_src_row = nullptr, *m_hw_off = nullptr, *m_hw_len = nullptr;

  // cross-request batching (pfhip_set_batching): callers queue at the handle they hold (ONE queue for every execution slot of
  // the handle: contexts on this device, replicas on other devices); the caller at the front leads a merged forward on an idle
  // slot while the next one already gathers the next batch (merge_queue.h PoolQueue)
  pfhip_detail::PoolQueue<BatchReq, pfhip_model> bq;
  int batch_wait_us = 0, batch_max_utts = 32;
  // the same for streaming calls: concurrent pfhip_stream_forward callers (one thread per connection) are merged
  pfhip_detail::MergeQueue<StreamReq> sq;
  int stream_wait_us = 0, stream_max = 128;
  std::atomic<int> live_streams{0};     // open pfhip_streams: a leader stops waiting once all of them have queued

  // profiling
  int prof_mask = 0;             // bit c set -> launches of kernel class c are bracketed by events
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  pfhip_profile prof{};
  hipStream_t prof_stream = nullptr;

  // in-process multi-GPU router (replicas only): the handle the caller holds is replica 0; `replicas` are full
  // models on the other devices of PFHIP_DEVICES / pfhip_create_group.  Offline calls go to the replica with the fewest calls
  // in flight, a new stream to the one with the fewest open streams (a connection stays on its device for life).
  std::vector<pfhip_model*> replicas;
  pfhip_model* group_head = nullptr;        // replica / context -> the handle the caller holds (nullptr on the head itself)
  std::atomic<int> inflight{0};
  std::atomic<int64_t> served_calls{0}, served_utts{0}, served_forwards{0};
  std::atomic<unsigned> rr{0};
  // Execution contexts (pfhip_set_inflight / PFHIP_INFLIGHT): further pfhip_model objects on THIS device that borrow every
  // weight pointer of `weights_of` (the blob, the repacks, the LayerNorm-folded copies, the front-end tables) and own only a
  // workspace, streams and events — the reference's one shared Ort::Session under many decoder threads (paraformer.cpp:35-41,541).
  std::vector<pfhip_model*> contexts;       // on a device replica: its extra contexts (the replica itself is context 0)
  pfhip_model* weights_of = nullptr;        // on a context: whose weights it borrows
  int ctx_index = 0;
  int ctx_limit = 1;                        // on the head: contexts per device that take calls (pfhip_set_inflight)
  std::vector<pfhip_model*> slots;          // on the head: every execution slot of the handle, device-major round order; guarded by bq.mu
  std::vector<float> hw_host;               // on the head: the resident hotword set (pfhip_set_hotwords), for contexts created later

  // per weight matrix (device pointer of its first element): the power-of-two scale the fp16 two-plane GEMM stages it with
  // (kernels.h best_w_scale), fixed at load from its largest magnitude; contexts copy the table
  std::unordered_map<const void*, float> wscale;
  float w_scale_of(const void* w) const { auto it = wscale.find(w); return it == wscale.end() ? 1.0f : it->second; }

  const Tensor& W(const std::string& n) const { return t.at(n); }
};

namespace pfhip_detail {
inline pfhip_model* route_stream(pfhip_model* m) {
  pfhip_model* best = m;
  for (pfhip_model* r : m->replicas)
    if (r->live_streams.load() < best->live_streams.load()) best = r;
  return best;
}
}  // namespace pfhip_detail

namespace pfhip_detail {

using pfhip::launch_gemm_f32;

struct Scope {
  pfhip_model* m; hipStream_t s; int cls; hipEvent_t e1 = nullptr;
  Scope(pfhip_model* m_, hipStream_t s_, int cls_, double flops, double bytes) : m(m_), s(s_), cls(cls_) {
    if (!((m->prof_mask >> cls) & 1)) return;
    if (m->ev_used + 2 > m->ev_pool.size()) {
      for (int i = 0; i < 256; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; m->ev_pool.push_back(e); }
    }
    hipEvent_t e0 = m->ev_pool[m->ev_used++];
    e1 = m->ev_pool[m->ev_used++];
    (void)hipEventRecord(e0, s);
    m->prof_recs.push_back({cls, e0, e1});
    m->prof.flops[cls] += flops;
    m->prof.bytes[cls] += bytes;
    m->prof.launches[cls] += 1;
  }
  ~Scope() { if (e1) (void)hipEventRecord(e1, s); }
};

enum { K_GEMM = 0, K_ATTN = 1, K_LN = 2, K_FSMN = 3, K_FBANK = 4, K_CIF = 5, K_HEAD = 6, K_OTHER = 7 };

inline void gemm(pfhip_model* m, hipStream_t s, const float* A, int lda, const float* Wd, int N, int K, int Ktrue,
          float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* R2, int ldr2,
          int M, bool relu) {
  Scope sc(m, s, K_GEMM, 2.0 * M * (double)N * Ktrue, 4.0 * ((double)M * Ktrue + (double)N * Ktrue + (double)M * N));
  launch_gemm_f32(A, lda, Wd, K, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, /*guard=*/false, s, m->w_scale_of(Wd));
}
// pinned staging, grown on demand (callers have no copy in flight from the old block: every forward ends with a sync)
inline pfhip_status ensure_h_meta(pfhip_model* m, size_t bytes) {
  if (bytes <= m->h_meta_cap) return PFHIP_OK;
  if (m->h_meta) HIP_TRY(hipHostFree(m->h_meta));
  m->h_meta = nullptr; m->h_meta_cap = 0;
  HIP_TRY(hipHostMalloc(&m->h_meta, bytes * 2, hipHostMallocDefault));
  m->h_meta_cap = bytes * 2;
  return PFHIP_OK;
}
inline pfhip_status ensure_h_counts(pfhip_model* m, size_t bytes) {
  if (bytes <= m->h_counts_cap) return PFHIP_OK;
  if (m->h_counts) HIP_TRY(hipHostFree(m->h_counts));
  m->h_counts = nullptr; m->h_counts_cap = 0;
  HIP_TRY(hipHostMalloc((void**)&m->h_counts, bytes * 2, hipHostMallocDefault));
  m->h_counts_cap = bytes * 2;
  return PFHIP_OK;
}
inline void lnorm(pfhip_model* m, hipStream_t s, const float* x, int ldx, float* y, int ldy, const std::string& name,
           int M, int D, int Dout) {
  Scope sc(m, s, K_LN, 8.0 * M * D, 8.0 * M * D);
  pfhip::launch_layernorm(x, ldx, y, ldy, m->W(name + ".g").d, m->W(name + ".b").d, M, D, Dout, 1e-12f, s);
}


}  // namespace pfhip_detail
