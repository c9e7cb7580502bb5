/*
 * pfhip_ops.h — operator-level C ABI over the same gfx950 kernels that pfhip_offline_forward launches.
 * Device pointers in, device pointers out, asynchronous on `stream` (hipStream_t, NULL = default).
 * Exists so that each kernel can be parity-tested and timed in isolation against the oracle; every
 * entry names the node of the reference graph it stands for (the graph itself is the opaque
 * `m_session_->Run`, onnxruntime/src/paraformer.cpp:541; architecture per SURVEY.md appendix A).
 * All return 0 on success, a hipError_t value otherwise.
 */
#ifndef PFHIP_OPS_H_
#define PFHIP_OPS_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* MatMul/Gemm (+Add bias, +residual Adds, Relu): C[M,N] = A[M,K] W[N,K]^T ...; guard=1 bounds-checks. */
int pfhip_op_gemm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                      const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int relu,
                      int guard, void* stream);
/* Same, with the kernel forced: kind 0 = by size (as above), 1 = 128x128 tiled, 2 = weight-streaming (M-small) kernel,
 * 3 = 64x128 tiled; 4 / 5 / 7 = bf16 three-plane kernels (256x128, 128x128, 64x128 tile); 8 / 9 / 10 = fp16 two-plane kernels. */
int pfhip_op_gemm_f32_kind(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                           const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int relu,
                           int guard, int kind, void* stream);
/* Same; w_scale = the power-of-two scale the fp16 two-plane kernels (kinds 8 / 9 / 10, and kind 0 by default) stage W with:
 * pfhip_op_best_w_scale(max |W|) keeps max |W| * w_scale <= 32768 (the model does this per weight matrix at load). */
int pfhip_op_gemm_f32_scaled(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                             const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int relu,
                             int guard, int kind, float w_scale, void* stream);
float pfhip_op_best_w_scale(float max_abs);
/* Pre-split operands (csrc/gemm_p3.hip): plane images — two fp16 planes of an fp32 matrix, [K/16][rows][16] with the 16-byte halves
 * of a row swapped where row bit 3 is set; rows a multiple of 128 — and the GEMM that consumes and produces them:
 * C = A W^T (x 1 / w_scale, LayerNorm-fold finish, +bias, +R1, ReLU) as fp32 (C != NULL) and / or as plane images (Ph / Pl != NULL).
 * tile_rows: 0 = by grid size, 64 / 128 = that tile height (the results are bit-identical). */
size_t pfhip_op_plane_image_bytes(int rows, int K);
int pfhip_op_split_planes(const float* X, int ld, int rows_valid, int rows, int K, float scale, void* hi, void* lo, void* stream);
int pfhip_op_gemm_p3(const void* Ah, const void* Al, int rows_a, const void* Wh, const void* Wl, int rows_w, float w_scale, float* C, int ldc,
                     void* Ph, void* Pl, int rows_p, const float* bias, const float* R1, int ldr1, int M, int N, int K, int relu,
                     const float* ln_stats, int ln_tiles, const float* ln_colsum, float* stats_out, int tile_rows, void* stream);
/* LayerNormalization over the last axis. */
int pfhip_op_layernorm(const float* x, int ldx, float* y, int ldy, const float* g, const float* b, int M, int D,
                       int Dout, float eps, void* stream);
/* FSMN memory block: depthwise Conv1d k=11 over time + identity (+residual), per utterance segment. */
int pfhip_op_fsmn(const float* v, int ldv, const float* w, const float* res, int ldres, float* out, int ldo,
                  const int* off, const int* len, int B, int max_len, int C, void* stream);
/* MatMul-Softmax-MatMul of one attention block, d_k = 128. */
int pfhip_op_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                       const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                       int max_q_len, float scale, void* stream);
/* Same, head dimension 32 or 128 chosen at run time (CT-Transformer: 256 / 8 heads). */
int pfhip_op_attention_hd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                          const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                          int max_q_len, float scale, int head_dim, void* stream);
/* The same block (d_k = 128, attention_x3.hip) with the context written as the two fp16 plane images that pfhip_op_gemm_p3 takes as
 * its A operand ([K / 16][plane_rows][16] per plane, K = H * 128; row = q_off[b] + t) instead of fp32 rows: the encoder's
 * attention -> output-projection hand-off on large batches.  total_q_rows = the number of rows the offsets span (<= plane_rows,
 * plane_rows a multiple of 128). */
int pfhip_op_attention_planes(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* planes_hi, void* planes_lo,
                              int plane_rows, const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                              int max_q_len, int total_q_rows, float scale, void* stream);
/* Row-major fp16 planes of an fp32 matrix: hi = fp16_rtz(x), lo = fp16_rn(x - hi), [rows][ldp] each (cols % 8 == 0). */
int pfhip_op_split_rows(const float* X, int ld, int rows, int cols, void* hi, void* lo, int ldp, void* stream);
/* The encoder's QKV projection on plane-image operands (pfhip_op_gemm_p3) whose result leaves in two forms: columns < q_cols as fp32
 * rows of C, columns >= q_cols (K | V) as row-major planes kv_hi / kv_lo [M][ldkv] (column n at element n - q_cols) — what
 * pfhip_op_attention_kvplanes stages by LDS-DMA.  q_cols % 128 == 0; tile_rows 0 / 64 / 128. */
int pfhip_op_gemm_p3_qkv(const void* Ah, const void* Al, int rows_a, const void* Wh, const void* Wl, int rows_w, float w_scale, float* C, int ldc,
                         void* kv_hi, void* kv_lo, int ldkv, int q_cols, const float* bias, int M, int N, int K, const float* ln_stats,
                         int ln_tiles, const float* ln_colsum, int tile_rows, void* stream);
/* MatMul-Softmax-MatMul (d_k = 128) with K and V given as row-major fp16 planes (attention_p3.hip: row stride ldkv elements, K at
 * column 0 and V at column v_col of each plane, head h in columns 128 h ..; total_kv_rows rows).  Context as fp32 rows (O) or as the
 * plane images of pfhip_op_attention_planes (planes_hi / planes_lo / plane_rows).  fsmn_w != NULL (self-attention): also the SAN-M
 * memory block of V into mem (+= when mem_accumulate).  Same arithmetic as pfhip_op_attention on the values hi + lo. */
int pfhip_op_attention_kvplanes(const float* Q, int ldq, const void* kv_hi, const void* kv_lo, int ldkv, int v_col, int total_kv_rows, float* O,
                                int ldo, void* planes_hi, void* planes_lo, int plane_rows, const int* q_off, const int* q_len,
                                const int* kv_off, const int* kv_len, int B, int H, int max_q_len, int total_q_rows, float scale,
                                const float* fsmn_w, float* mem, int ldmem, int mem_accumulate, void* stream);
/* CIF integrate-and-fire (onnxruntime/src/paraformer-online.cpp:301-327) + tail slot. */
int pfhip_op_cif(const float* hidden, int ldh, const float* alphas, const int* row_off, const int* len, int B, int D,
                 float threshold, float tail, float* stage, int* n_fires, int* token_num, void* stream);
/* LogSoftmax + ArgMax (GreedySearch/FindMax, onnxruntime/src/paraformer.cpp:386-395, util.cpp:63-74). */
int pfhip_op_logsoftmax_argmax(const float* logits, int ldl, int ML, int V, float* logp, int32_t* ids, void* stream);

/* One streaming window (M <= 32 rows): LayerNormalization (g != NULL; width D <= K) -> MatMul/Gemm (+bias, +residual Adds, Relu)
 * (+ the SAN-M FSMN memory of fsmn_v over the M rows, k = 11) in ONE launch — stream_fused.hip. */
int pfhip_op_fused_ln_gemm(const float* X, int ldx, int D, const float* g, const float* b, float eps, const float* W, int ldw,
                           float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* R2, int ldr2,
                           const float* fsmn_v, int ldv, const float* fsmn_w, int M, int N, int K, int relu, void* stream);

/* The same node group with every operand requested in ONE trip to memory and the LayerNormalization applied algebraically
 * (M <= 20 rows, K = 64..2048 in the window shapes): when ln_colsum != NULL, W / bias are the gamma / beta-folded weights
 * (W' = W gamma, b' = b + W beta) and ln_colsum[n] = sum_k W'[n][k]; the LayerNorm is over exactly the K operand columns.
 * Returns hipErrorInvalidValue for shapes the kernel does not take (callers use pfhip_op_fused_ln_gemm there). */
int pfhip_op_fused_gemv_1trip(const float* X, int ldx, const float* W, int ldw, float* C, int ldc, const float* bias,
                              const float* ln_colsum, float eps, const float* R1, int ldr1, const float* fsmn_v, int ldv,
                              const float* fsmn_w, int M, int N, int K, int relu, void* stream);

/* MatMul-Softmax-MatMul of ONE streaming window: Lq <= 32 queries against Lk <= 32 keys (rows 0.. of the given pointers), H heads of
 * d_k = 128 (the streaming encoder's self-attention over its 20-row window, the decoder's tokens against it:
 * onnxruntime/src/paraformer-online.cpp:426-515).  hipErrorInvalidValue for other shapes (callers use pfhip_op_attention). */
int pfhip_op_window_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int Lq, int Lk,
                              int H, float scale, void* stream);

/* The window's MatMul-Softmax-MatMul AND the MatMul/Gemm that projects its context (W [N, 512]; +bias, +residual Add, + the SAN-M
 * FSMN memory of fsmn_v over the Lq rows) in ONE launch: H = 4 heads of 128, Lq <= 20, Lk <= 32.  hipErrorInvalidValue otherwise. */
int pfhip_op_fused_att_out(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int Lq, int Lk, int H, float scale,
                           const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* fsmn_v,
                           int ldfv, const float* fsmn_w, int N, void* stream);

#ifdef __cplusplus
}
#endif
#endif
