// Internal launcher declarations for the gfx950 kernels behind include/pfhip.h.
// All pointers are device pointers unless a name says host.  All launches are asynchronous on `s`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pfhip {

constexpr int kTileM = 128;      // GEMM block tile rows   (activation buffers are allocated in multiples)
constexpr int kTileN = 128;      // GEMM block tile cols   (weights are repacked/padded to multiples)
constexpr int kTileK = 32;       // GEMM k-step            (K is padded to multiples)
constexpr int kHeadDim = 128;    // attention kernel is specialised for d_k = 128
constexpr int kMelW = 32;        // max taps per mel triangle (80 bins @ 512-pt FFT need <= 19)

// ---- front end (SURVEY §8a rows a2,a3) -------------------------------------------------------
struct FbankTables {
  const float* window;     // [400] hamming, feature-window.cc:33-42
  const double* tw512;     // [256][2] cos/-sin of 2*pi*k/512
  const int* mel_off;      // [n_mels]
  const int* mel_size;     // [n_mels]
  const float* mel_w;      // [n_mels][kMelW]
  const float* cmvn_mean;  // [lfr_m*n_mels]
  const float* cmvn_istd;  // [lfr_m*n_mels]
};
// pcm: utterances back to back; sample_off[b] (int64), frame_off[b] prefix of fbank frame counts
// (B+1 entries), nframes[b], row_off[b] (first LFR row of utterance b in feats).
void launch_fbank_lfr_cmvn(const float* pcm, const int64_t* sample_off, const int* frame_off,
                           const int* nframes, const int* row_off, int B, int total_frames,
                           FbankTables tb, float* feats, hipStream_t s);

// Streaming form (ParaformerOnline::FbankKaldi, paraformer-online.cpp:119-145): one utterance, raw
// log-mel frames [total_frames, 80] out, no LFR/CMVN.  sample_off/frame_off/nframes: 1/2/1 entries.
void launch_fbank_frames(const float* pcm, const int64_t* sample_off, const int* frame_off, const int* nframes,
                         int total_frames, FbankTables tb, float* fb_out, hipStream_t s);
// the same for B utterances back to back in `pcm` (frame_off has B + 1 entries); frames land back to back in fb_out
void launch_fbank_frames_batch(const float* pcm, const int64_t* sample_off, const int* frame_off, const int* nframes, int B,
                               int total_frames, FbankTables tb, float* fb_out, hipStream_t s);

// x0[row][0..D) = feats*scale + PE(row_pos[row]+1); columns D..ldx are zeroed.
void launch_embed(const float* feats, int D, float* x0, int ldx, const int* row_pos, int M,
                  const float* inv_timescale, float scale, hipStream_t s);

// ---- the fp16 two-plane domain guard ---------------------------------------------------------------------------------------
// The default large-GEMM / attention forms stage their operands as two fp16 planes (gemm_x3.hip header): |a| >= 65504 overflows,
// rows whose magnitude is far below 1 lose relative precision (absolute error 2^-25 per element).  A forward runs with a
// per-thread launch context: `range_flag` (device word) is raised by the LayerNorm-folding kernels when a row's rms leaves
// [2^-8, 2^11] (bit 1: no element of such a row of <= 512 values can reach 65504), by the LayerNorm kernel and by the head when
// a row is not finite (bit 0: the check sits on the residual stream because ReLU swallows NaN); `exact` routes every split-operand launch of
// the calling thread to the bf16 three-plane kernels (gemm_x6.hip / attention_x6.hip: fp32's exponent range, exact split) and
// keeps the encoder off the plane images.  The reference computes in plain fp32 (onnxruntime/src/paraformer.cpp:496-541).
struct LaunchCtx {
  bool exact = false;
  int* range_flag = nullptr;
};
LaunchCtx& launch_ctx();                      // thread-local (gemm.hip)
constexpr float kLnRstdMin = 1.0f / 2048.0f, kLnRstdMax = 256.0f;

// ---- dense ops --------------------------------------------------------------------------------
// C[M,N] = A[M,K] * W[N,K]^T (+bias[N]) (+R1[M,N]) (+R2[M,N]) (ReLU)   — fp32 MFMA 32x32x2.
// A rows must be allocated up to a multiple of 128, K % 32 == 0, W readable for ceil(N/128)*128 rows.
// guard == false (the product path): C/R1/R2 are allocated for ceil(M/128)*128 rows and
// ceil(N/128)*128 columns and bias is readable to the padded N, so the epilogue has no bounds
// branches (pad outputs are junk nobody reads).  guard == true bounds-checks every element.
// C may alias R1 or R2 (in-place residual update).
void launch_gemm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc,
                     const float* bias, const float* R1, int ldr1, const float* R2, int ldr2,
                     int M, int N, int K, bool relu, bool guard, hipStream_t s, float w_scale = 1.0f);
// w_scale: power of two with max|W| * w_scale < 65504, used by the fp16 two-plane kernels (gemm_x3.hip) to stage W * w_scale
// (best_w_scale below; 1 is always valid for |W| < 65504 and costs precision only for weights of very small magnitude)
float best_w_scale(float max_abs);
// kind: 0 = pick by M, 1 = the 128 x 128 tiled kernel, 2 = the weight-streaming kernel (dev / tests)
// fp32-grade GEMM on the BF16 matrix cores (three-way bf16 split of both operands, six MFMAs per block): gemm_x6.hip
// stats_out (N == tiles_n * 128 exactly): the epilogue also leaves per-row LayerNorm statistics of its 128-column tile at
// stats_out[row][tile column][2] = (mean of the tile's columns, sum of squared deviations from it).
// ln_stats (with ln_tiles, ln_colsum): LayerNorm folded in — A is the raw residual stream x, W must be W * gamma, bias must be
// bias + W beta and ln_colsum[n] = sum_k W[n][k] gamma[k]; the epilogue forms rstd_i * (x W'^T - mean_i * colsum) + bias with the
// statistics merged from the ln_tiles pairs per row the producer left (eps 1e-12).
void launch_gemm_f32_bf16x6(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                            int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, int gw, hipStream_t s, bool small_tile = false,
                            const float* ln_stats = nullptr, int ln_tiles = 0, float* stats_out = nullptr, bool half_tile = false,
                            const float* ln_colsum = nullptr);
// the same tilings with TWO fp16 planes and THREE products per block (gemm_x3.hip); sw: the power-of-two weight scale
void launch_gemm_f32_f16x3(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                           int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, int gw, hipStream_t s, bool small_tile,
                           const float* ln_stats, int ln_tiles, float* stats_out, bool half_tile, const float* ln_colsum, float sw);
// ---- pre-split operands (gemm_p3.hip): the three-product fp16 scheme with the split taken out of the K-loop -------------------
// Plane image of X[rows, K]: two arrays (hi, lo) of plane_image_bytes(rows, K) bytes each, laid out [K / 16][rows padded to 128][16]
// fp16 with the 16-byte halves of a row swapped where row bit 3 is set.  launch_split_planes writes the images of an fp32 matrix
// (times `scale`); launch_gemm_p3 multiplies A images by W images (N % 128 == 0, K % 16 == 0; W pre-multiplied by w_scale) and
// writes fp32 C (bias, LayerNorm-fold finish, residual R1, ReLU, row statistics) and / or the plane images of C (rows_p rows).
size_t plane_image_bytes(int rows, int K);
void launch_split_planes(const float* X, int ld, int rows_valid, int rows, int K, float scale, void* hi, void* lo, hipStream_t s);
void launch_gemm_p3(const void* Ah, const void* Al, int rows_a, const void* Wh, const void* Wl, int rows_w, float w_scale, float* C, int ldc,
                    void* Ph, void* Pl, int rows_p, const float* bias, const float* R1, int ldr1, int M, int N, int K, bool relu,
                    const float* ln_stats, int ln_tiles, const float* ln_colsum, float* stats_out, int gw, hipStream_t s,
                    int tile_rows = 0,       // 0: 64-row tiles when 128-row tiles would fill less than a round; 64 / 128 force one
                    int row_planes_from = 0);
// row_planes_from = c > 0 (the encoder's QKV projection, C given): columns < c leave as fp32 rows of C, columns >= c as ROW-MAJOR fp16
// planes Ph / Pl [M][rows_p] (rows_p = elements per plane row; column n at element n - c) — the K | V operand of attention_p3.hip.
// c % 128 == 0; 128- and 64-row tiles only.
// Row-major planes of an fp32 matrix (tests, tools): hi / lo [rows][ldp] fp16, cols % 8 == 0.
void launch_split_rows(const float* X, int ld, int rows, int cols, void* hi, void* lo, int ldp, hipStream_t s);
// attention_x3.hip's fused attention on K / V given as row-major planes (row stride ldkv elements, K at column 0, V at column v_col of
// each plane; head h in columns 128 h ..): tiles staged by LDS-DMA, V read through gfx950's transposing LDS read.  d_k = 128.
void launch_attention_p3(const float* Q, int ldq, const void* kv_hi, const void* kv_lo, int ldkv, int v_col, float* O, int ldo, const int* q_off,
                         const int* q_len, const int* kv_off, const int* kv_len, int B, int H, int max_q_len, float scale, hipStream_t s,
                         const float* fsmn_w = nullptr, float* mem = nullptr, int ldmem = 0, bool mem_accumulate = false,
                         void* planes_hi = nullptr, void* planes_lo = nullptr, int plane_rows = 0);
// The product-path form of the two options above: the BF16-split kernels with the tile / column-group choice of launch_gemm_f32.
// gemm_x6_ln_ok(M): whether launch_gemm_f32 would put the N = 512 launches of M rows on these kernels (both sides of a
// statistics hand-off must).
bool gemm_x6_ln_ok(int M);
bool gemm_f16_planes_form();      // the large GEMMs are on the fp16 two-plane form (not PFHIP_GEMM_X3=0 / PFHIP_GEMM_X6=0)
void launch_gemm_f32_x6_ln(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                           int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, const float* ln_stats, int ln_tiles,
                           const float* ln_colsum, float* stats_out, hipStream_t s, float w_scale = 1.0f);
void launch_gemm_f32_kind(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                          int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, bool guard, int kind,
                          hipStream_t s, float w_scale = 1.0f);

// y[row][0..D) = LN(x[row][0..D)) * g + b; columns D..Dout zeroed.  D % 4 == 0, Dout <= 2048.
void launch_layernorm(const float* x, int ldx, float* y, int ldy, const float* g, const float* b,
                      int M, int D, int Dout, float eps, hipStream_t s);

// out[t][c] = (res? res[t][c] : 0) + v[t][c] + sum_j w[c][j] * v[t + j - (k-1)/2][c], per utterance
// segment [off[b], off[b]+len[b]) with zero padding outside the segment.  C % 4 == 0, k == 11.
void launch_fsmn(const float* v, int ldv, const float* w, const float* res, int ldres, float* out,
                 int ldo, const int* off, const int* len, int B, int max_len, int C, hipStream_t s);
// Same with the window shifted into the past by `shift` frames (UPSTREAM sanm_shfit): taps cover
// [t - 5 - shift, t + 5 - shift]; shift is 0 or 5 (5 = fully causal, the vad-realtime punctuation model).
void launch_fsmn_shift(const float* v, int ldv, const float* w, const float* res, int ldres, float* out,
                       int ldo, const int* off, const int* len, int B, int max_len, int C, int shift, hipStream_t s);

// Multi-head attention, d_k = 128: O[q, h*128:(h+1)*128] = softmax(scale * Q_h K_h^T) V_h over the
// utterance's own keys.  q segments (q_off,q_len), kv segments (kv_off,kv_len), all device arrays.
void launch_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                      float* O, int ldo, const int* q_off, const int* q_len, const int* kv_off,
                      const int* kv_len, int B, int H, int max_q_len, float scale, hipStream_t s);

// Same kernel with the head dimension chosen at run time: 128 or 32 (CT-Transformer: 256 / 8 heads).
void launch_attention_hd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                         const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                         int max_q_len, float scale, int head_dim, hipStream_t s);

// With a per-query key limit: query row i (packed index) only sees keys [0, min(kv_len, q_kv_limit[i])) — the
// prefix mask CTTransformerOnline::VadMask builds (ct-transformer-online.cpp:225-240).
// both attention products on the BF16 matrix cores (exact three-way split, attention_x6.hip); d_k = 128, no per-query limits
// fsmn_w != nullptr (self-attention only: q segments == kv segments): the kernel also writes the encoder layer's FSMN memory
// mem = V + depthwise conv k = 11 over time (what launch_fsmn computes, bit for bit) for its rows and its head's channels
void launch_attention_x6(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                         const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H, int max_q_len,
                         float scale, hipStream_t s, const float* fsmn_w = nullptr, float* mem = nullptr, int ldmem = 0,
                         bool mem_accumulate = false);
// the same with two fp16 planes and three products per block (attention_x3.hip)
void launch_attention_x3(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                         const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H, int max_q_len,
                         float scale, hipStream_t s, const float* fsmn_w = nullptr, float* mem = nullptr, int ldmem = 0,
                         bool mem_accumulate = false, void* planes_hi = nullptr, void* planes_lo = nullptr, int plane_rows = 0);
// Encoder-layer pair: FSMN memory of V (into mem) + self-attention (into O).  One launch where the BF16 attention kernel runs
// (d_k = 128, more than 64 queries per utterance), otherwise launch_fsmn + launch_attention.  C = V's channel count (H * 128).
void launch_attention_fsmn(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                           const int* off, const int* len, int B, int H, int max_len, float scale, const float* fsmn_w, float* mem,
                           int ldmem, hipStream_t s, bool mem_accumulate = false, void* planes_hi = nullptr, void* planes_lo = nullptr,
                           int plane_rows = 0);
// whether launch_attention_fsmn can write the context as fp16 plane images (gemm_p3.hip's A operand) instead of fp32 rows: the
// fused launch on attention_x3.hip only.  With planes_hi set, O is not written.
bool attention_planes_ok(int max_len);
// whether launch_attention_fsmn will be the single fused launch (then, and only then, mem_accumulate is honoured: the caller may
// pass the residual stream as `mem` and drop the memory term from the output projection)
bool attention_fsmn_is_fused(int max_len);
void launch_attention_masked(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                             const int* q_off, const int* q_len, const int* kv_off, const int* kv_len,
                             const int* q_kv_limit, int B, int H, int max_q_len, float scale, int head_dim, hipStream_t s);

// ---- timestamp head (SURVEY §8a row a6 producer; blstm.hip) -------------------------------------------------------
// Bidirectional LSTM, hidden 512, over packed sequences (off/len in frames, B <= 32): gx [rows, 4096] = input projections
// + biases (forward i,f,g,o | backward i,f,g,o), whh [2][2048][512], y [rows, 1024]; hx = kBlstmScratchFloats floats of
// scratch, ZEROED ONCE by the caller; its word kBlstmFlagWord is an error flag the kernel sets (1 = step barrier timed
// out, 2 = the blocks of a direction were not all on one XCD) — check it after the stream has drained.
constexpr int kBlstmScratchFloats = 2 * 4 * 32 * 512 + 8;      // exchange ring [2 dir][4][32][512], then 8 state words
constexpr int kBlstmFlagWord = 2 * 4 * 32 * 512 + 2;
hipError_t launch_blstm(const float* gx, const float* whh, float* y, float* hx, const int* off, const int* len, int B, int Lmax,
                        hipStream_t s);
// The same recurrence as one launch per time step (no inter-block exchange inside a launch: works whatever else runs on the
// device); cst = 2 * 32 * 512 floats of cell state.  Same results bit for bit.
hipError_t launch_blstm_stepwise(const float* gx, const float* whh, float* y, float* hx, float* cst, const int* off, const int* len, int B,
                                 int Lmax, hipStream_t s);
void launch_alpha2(const float* y, int ldy, const float* w, float b, float smooth, float noise, float* a2, int rows, int D,
                   hipStream_t s);
void launch_us_cif(const float* a2, const int* off, const int* len, const int* token_num, int B, int max_len, float threshold,
                   float* us_alphas, float* us_peaks, hipStream_t s);

// ---- predictor / CIF (SURVEY §8a rows a4,a12) -------------------------------------------------
// col[row] = [h[t-1] | h[t] | h[t+1]] with zeros outside the utterance.  row_pos/row_len give the
// local index and utterance length of every packed row.
void launch_im2col3(const float* h, int ldh, float* col, int ldc, const int* row_pos,
                    const int* row_len, int M, int D, hipStream_t s);
// alphas[row] = relu(sigmoid(dot(o[row], w) + b) * smooth - noise)
void launch_alpha(const float* o, int ldo, const float* w, const float* b, float smooth,
                  float noise, float* alphas, int M, int D, hipStream_t s);
// CIF integrate-and-fire per utterance (paraformer-online.cpp:301-327) with the tail slot
// (alpha = tail, hidden = 0) appended.  Writes fired frames to stage[(row_off[b]+b+n)][0..D),
// n_fires[b], token_num[b] = floor(sequential fp32 sum of alphas incl. tail).
void launch_cif(const float* hidden, int ldh, const float* alphas, const int* row_off,
                const int* len, int B, int D, float threshold, float tail, float* stage,
                int* n_fires, int* token_num, hipStream_t s);
// emb[tok_off[b]+n] = stage[row_off[b]+b+n]
void launch_compact(const float* stage, float* emb, const int* tok_row_src, int ML, int D,
                    hipStream_t s);

// ---- hotword embedder ------------------------------------------------------------
// out[r] = table[ids[r]]  (Embedding lookup, D % 4 == 0)
void launch_gather_rows(const int32_t* ids, const float* table, int D, float* out, int R, hipStream_t s);
// One LSTM step on pre-activations G [H, 4D] (torch gate order i,f,g,o): c = sig(f)*c + sig(i)*tanh(g);
// h = sig(o)*tanh(c); rows with lens[j]-1 == t also copy h into sel[j].
void launch_lstm_cell(const float* G, float* c, float* h, const int32_t* lens, int t, float* sel, int H, int D,
                      hipStream_t s);

// ---- head -----------------------------------------------------------------
// per row: log-softmax over V logits, argmax (first max wins, util.cpp:63-74).  logp may be null.
// range_flag (may be null): bit 0 is raised when a row's log-sum-exp is not finite (NaN / Inf reached the logits).
void launch_logsoftmax_argmax(const float* logits, int ldl, int ML, int V, float* logp, int32_t* ids,
                              hipStream_t s, int* range_flag = nullptr);

// ---- chunk-streaming pieces ----------------------------------------------
// OnlineLfrCmvn + x*sqrt(d) + GetPosEmb (paraformer-online.cpp:196-238, 549-555, 240-268) for `n_rows`
// LFR rows: row i = frames [6i, 6i+7) of fb (T frames incl. the splice cache), the tail replicated with
// the last frame; out[i] = ((x + mean) * istd) * scale + PE(pos0 + i + 1).  out has row stride ldo.
void launch_stream_lfr(const float* fb, int T, int n_rows, const float* mean, const float* istd, float scale,
                       const float* inv_ts, int pos0, float* out, int ldo, hipStream_t s);
// dst[r][0..ncols) = src[r][0..ncols) (or 0 when src == nullptr), columns ncols..ldd zeroed.
void launch_rows_copy(float* dst, int ldd, const float* src, int lds_, int nrows, int ncols, hipStream_t s);
// The same two for many connections in one launch: one descriptor per operation (arrays in HBM).
struct RowsCopyOp { float* dst; const float* src; int ldd, lds, nrows, ncols; };      // src == nullptr: zeros; lds == 0: row 0 repeated
struct StreamLfrOp { const float* fb; float* out; int T, n_rows, pos0, pad_; };
void launch_rows_copy_batch(const RowsCopyOp* ops, int n_ops, int max_rows, hipStream_t s);
void launch_stream_lfr_batch(const StreamLfrOp* ops, int n_ops, int max_rows, const float* mean, const float* istd, float scale,
                             const float* inv_ts, int ldo, hipStream_t s);
// One connection of a streaming batch (device-side descriptor): its window in the packed encoder matrices, its tokens in
// the packed decoder matrices (filled in after the CIF counts are known), and its persistent state.
struct StreamSeg {
  float* carry;        // CIF hidden_cache_ [D] followed by alphas_cache_ [1]
  float* dcache;       // decoder FSMN caches [layers][10][D]
  int row_off, n;      // window rows
  int is_last, pre, suf;       // CifSearch: last chunk flag, alphas outside [pre, suf) are zeroed (paraformer-online.cpp:279-286)
  int tok_off, n_tok;  // fired tokens
  int pad_;
};
// CifSearch (paraformer-online.cpp:270-345) for B connections at once; stream b's fires land in emb_all[b * emb_rows ..],
// their count in n_fire[b] (counts above emb_rows are reported but not stored: the caller checks).
void launch_cif_stream(const float* enc, int lde, const float* alphas, const StreamSeg* segs, int B, float threshold, float tail,
                       float* emb_all, int emb_rows, int* n_fire, int D, hipStream_t s);
// Decoder FSMN with the 10-frame cache (paraformer-online.cpp:374, 500) for the packed tokens of B connections.
void launch_fsmn_cached(const float* t2, const float* w, const float* res, float* out, const StreamSeg* segs, int B, int layer,
                        int C, hipStream_t s);

// ---- fused kernels for ONE streaming window (stream_fused.hip): M <= 32 rows ---------------------------------------------
// C[M,N] = act(LN?(X) W^T + bias) (+R1) (+R2) (+ FSMN memory of fsmn_v: v + depthwise conv k = 11 over the M rows).
// g == nullptr: no LayerNorm; D = LN width (<= K; columns D..K of the normalised operand are 0).  One workgroup per 32 columns.
void launch_fused_ln_gemm(const float* X, int ldx, int D, const float* g, const float* b, float eps, const float* W, int ldw,
                          float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* R2, int ldr2,
                          const float* fsmn_v, int ldv, const float* fsmn_w, int M, int N, int K, bool relu, hipStream_t s);
// Attention of ONE window (Lq, Lk <= 32 rows starting at the given pointers; d_k = 128; H heads): one small workgroup per head,
// operands requested at kernel start (stream_fused.hip).  Returns false for shapes it does not take.
bool launch_window_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int Lq, int Lk,
                             int H, float scale, hipStream_t s);
bool launch_window_attention_segments(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                                      const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                                      int max_q_len, int max_kv_len, float scale, hipStream_t s, const float* fsmn_w = nullptr,
                                      float* mem = nullptr, int ldmem = 0);
// The window's attention AND the projection of its context by W [N, 512] (+bias, +R1, + the FSMN memory of fsmn_v) in one launch:
// every workgroup redoes the attention and keeps the context in LDS (stream_fused.hip).  H = 4 heads of 128, Lq <= 20, Lk <= 32.
bool launch_fused_att_out(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int Lq, int Lk, int H, float scale,
                          const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* fsmn_v,
                          int ldfv, const float* fsmn_w, int N, hipStream_t s);
// the same operator with every operand requested in one trip and LayerNorm applied algebraically (stream_fused.hip): W / bias are
// the gamma/beta-folded ones and ln_colsum their column sums when the LayerNorm is wanted.  Returns false for shapes it does not take.
bool launch_fused_gemv_1trip(const float* X, int ldx, const float* W, int ldw, float* C, int ldc, const float* bias, const float* ln_colsum,
                             float eps, const float* R1, int ldr1, const float* fsmn_v, int ldv, const float* fsmn_w, int M, int N, int K,
                             bool relu, hipStream_t s);

// ---- FSMN-VAD pieces --------------------------------------------------------------
// Generic LfrCmvn over raw fbank frames fb [F, n_mels] -> out [T = ceil(F/n), ldo] (columns >= m*n_mels zeroed).
void launch_lfr_cmvn(const float* fb, int F, int T, int m, int n, int n_mels, const float* mean, const float* istd,
                     float* out, int ldo, hipStream_t s);
// OnlineLfrCmvn form (fsmn-vad-online.cpp:90-133): row i = frames [i*n, i*n + m) of fb (which starts with the splice cache),
// the tail replicated with the last frame; no left padding.
void launch_lfr_cmvn_online(const float* fb, int F, int T, int m, int n, int n_mels, const float* mean, const float* istd,
                            float* out, int ldo, hipStream_t s);
// Memory block with left order 20: out = p + causal depthwise conv over [cache(19 rows); p]; cache_out (may be
// null) receives the last 19 rows of [cache_in; p] and must not alias cache_in.
// One connection's share of a packed FSMN-VAD forward: rows [row_off, row_off + T), network caches [layers][19][C]
// (cache_out == nullptr: do not advance, the final call of fsmn-vad.cpp:129-134).
struct VadSeg { const float* cache_in; float* cache_out; int row_off, T; };
struct VadLfrOp { const float* fb; int Tin, n_out, row_off, pad_; };
void launch_fsmn_causal20(const float* p, int ldp, const float* w, const VadSeg* segs, int B, int max_T, int layer, float* out,
                          int ldo, int C, hipStream_t s);
void launch_lfr_cmvn_online_batch(const VadLfrOp* ops, int n_ops, int max_rows, int m, int n, int n_mels, const float* mean,
                                  const float* istd, float* out, int ldo, hipStream_t s);
void launch_softmax_rows(const float* x, int ldx, int M, int N, float* y, float* col0, hipStream_t s);

}  // namespace pfhip
