// Operator-level C ABI (include/pfhip_ops.h): thin wrappers over the kernel launchers.
#include "../../include/pfhip_ops.h"

#include "kernels.h"

namespace {
inline hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }
inline int done() { return (int)hipGetLastError(); }
}  // namespace

extern "C" {

int pfhip_op_gemm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                      const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int relu,
                      int guard, void* stream) {
  if (K % pfhip::kTileK) return (int)hipErrorInvalidValue;
  pfhip::launch_gemm_f32(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu != 0, guard != 0, S(stream));
  return done();
}
int pfhip_op_gemm_f32_kind(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                           const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int relu,
                           int guard, int kind, void* stream) {
  if (K % pfhip::kTileK || kind < 0 || kind > 10 || kind == 6) return (int)hipErrorInvalidValue;
  pfhip::launch_gemm_f32_kind(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu != 0, guard != 0, kind, S(stream));
  return done();
}
int pfhip_op_fused_ln_gemm(const float* X, int ldx, int D, const float* g, const float* b, float eps, const float* W, int ldw,
                           float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* R2, int ldr2,
                           const float* fsmn_v, int ldv, const float* fsmn_w, int M, int N, int K, int relu, void* stream) {
  if (M < 1 || M > 32 || K % 8 || (g && (D % 4 || D > K || D > 2048))) return (int)hipErrorInvalidValue;
  pfhip::launch_fused_ln_gemm(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu != 0,
                              S(stream));
  return done();
}
// dev hook (tools/fused_gemv_bench.py): n launches back to back over a ring of weight copies, so the host loop is not Python
int pfhip_dev_fused_ln_gemm_bench(const float* X, int ldx, int D, const float* g, const float* b, const float* W, int ldw,
                                  size_t w_stride_floats, int n_copies, float* C, int ldc, const float* bias, const float* R1, int ldr1,
                                  int M, int N, int K, int relu, int n_launch, void* stream) {
  for (int i = 0; i < n_launch; ++i)
    pfhip::launch_fused_ln_gemm(X, ldx, D, g, b, 1e-12f, W + (size_t)(i % n_copies) * w_stride_floats, ldw, C, ldc, bias, R1, ldr1, nullptr,
                                0, nullptr, 0, nullptr, M, N, K, relu != 0, S(stream));
  return done();
}
int pfhip_op_fused_gemv_1trip(const float* X, int ldx, const float* W, int ldw, float* C, int ldc, const float* bias,
                              const float* ln_colsum, float eps, const float* R1, int ldr1, const float* fsmn_v, int ldv,
                              const float* fsmn_w, int M, int N, int K, int relu, void* stream) {
  if (!pfhip::launch_fused_gemv_1trip(X, ldx, W, ldw, C, ldc, bias, ln_colsum, eps, R1, ldr1, fsmn_v, ldv, fsmn_w, M, N, K, relu != 0, S(stream)))
    return (int)hipErrorInvalidValue;
  return done();
}
// dev hook (tools/fused_gemv_bench.py): the one-trip form over the same ring of weight copies
int pfhip_dev_fused_gemv_1trip_bench(const float* X, int ldx, const float* W, int ldw, size_t w_stride_floats, int n_copies, float* C, int ldc,
                                     const float* bias, const float* ln_colsum, const float* R1, int ldr1, int M, int N, int K, int relu,
                                     int n_launch, void* stream) {
  for (int i = 0; i < n_launch; ++i)
    if (!pfhip::launch_fused_gemv_1trip(X, ldx, W + (size_t)(i % n_copies) * w_stride_floats, ldw, C, ldc, bias, ln_colsum, 1e-12f, R1, ldr1,
                                        nullptr, 0, nullptr, M, N, K, relu != 0, S(stream)))
      return (int)hipErrorInvalidValue;
  return done();
}
int pfhip_op_window_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int Lq, int Lk,
                              int H, float scale, void* stream) {
  if (!pfhip::launch_window_attention(Q, ldq, K, ldk, V, ldv, O, ldo, Lq, Lk, H, scale, S(stream))) return (int)hipErrorInvalidValue;
  return done();
}
int pfhip_op_fused_att_out(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int Lq, int Lk, int H, float scale,
                           const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* fsmn_v,
                           int ldfv, const float* fsmn_w, int N, void* stream) {
  if (!pfhip::launch_fused_att_out(Q, ldq, K, ldk, V, ldv, Lq, Lk, H, scale, W, ldw, C, ldc, bias, R1, ldr1, fsmn_v, ldfv, fsmn_w, N, S(stream)))
    return (int)hipErrorInvalidValue;
  return done();
}
int pfhip_op_gemm_f32_scaled(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                             int ldr1, const float* R2, int ldr2, int M, int N, int K, int relu, int guard, int kind, float w_scale,
                             void* stream) {
  if (K % pfhip::kTileK || kind < 0 || kind > 10 || kind == 6 || !(w_scale > 0.f)) return (int)hipErrorInvalidValue;
  pfhip::launch_gemm_f32_kind(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu != 0, guard != 0, kind, S(stream), w_scale);
  return (int)hipGetLastError();
}
float pfhip_op_best_w_scale(float max_abs) { return pfhip::best_w_scale(max_abs); }

size_t pfhip_op_plane_image_bytes(int rows, int K) { return pfhip::plane_image_bytes(rows, K); }
int pfhip_op_split_planes(const float* X, int ld, int rows_valid, int rows, int K, float scale, void* hi, void* lo, void* stream) {
  if (K < 16 || K % 16 || rows <= 0 || rows % 128 || rows_valid < 0 || rows_valid > rows || !hi || !lo || (rows_valid > 0 && (!X || ld < K)))
    return (int)hipErrorInvalidValue;
  pfhip::launch_split_planes(X, ld, rows_valid, rows, K, scale, hi, lo, S(stream));
  return (int)hipGetLastError();
}
int pfhip_op_gemm_p3(const void* Ah, const void* Al, int rows_a, const void* Wh, const void* Wl, int rows_w, float w_scale, float* C, int ldc,
                     void* Ph, void* Pl, int rows_p, const float* bias, const float* R1, int ldr1, int M, int N, int K, int relu,
                     const float* ln_stats, int ln_tiles, const float* ln_colsum, float* stats_out, int tile_rows, void* stream) {
  // everything the kernel and its grid assume, checked here: a wrong image size would be an out-of-bounds DMA on the device
  const int mp = (M + 127) / 128 * 128;
  if (M <= 0 || N <= 0 || K < 16 || K % 16 || N % 128 || rows_a % 128 || rows_w % 128 || rows_a < mp || rows_w < N || !Ah || !Al || !Wh ||
      !Wl || (!C && !Ph) || (Ph && (!Pl || rows_p % 128 || rows_p < mp)) || (C && ldc < N) || (R1 && ldr1 < N) || !(w_scale > 0.f) ||
      (ln_stats && (!ln_colsum || ln_tiles <= 0)) || (tile_rows != 0 && tile_rows != 64 && tile_rows != 128 && tile_rows != 256))
    return (int)hipErrorInvalidValue;
  pfhip::launch_gemm_p3(Ah, Al, rows_a, Wh, Wl, rows_w, w_scale, C, ldc, Ph, Pl, rows_p, bias, R1, ldr1, M, N, K, relu != 0, ln_stats, ln_tiles,
                        ln_colsum, stats_out, 4, S(stream), tile_rows);
  return (int)hipGetLastError();
}

int pfhip_op_layernorm(const float* x, int ldx, float* y, int ldy, const float* g, const float* b, int M, int D,
                       int Dout, float eps, void* stream) {
  if (D % 4 || Dout % 4 || Dout > 2048 || D > Dout) return (int)hipErrorInvalidValue;
  pfhip::launch_layernorm(x, ldx, y, ldy, g, b, M, D, Dout, eps, S(stream));
  return done();
}
int pfhip_op_fsmn(const float* v, int ldv, const float* w, const float* res, int ldres, float* out, int ldo,
                  const int* off, const int* len, int B, int max_len, int C, void* stream) {
  if (C % 4) return (int)hipErrorInvalidValue;
  pfhip::launch_fsmn(v, ldv, w, res, ldres, out, ldo, off, len, B, max_len, C, S(stream));
  return done();
}
int pfhip_op_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                       const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                       int max_q_len, float scale, void* stream) {
  pfhip::launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off, kv_len, B, H, max_q_len, scale, S(stream));
  return done();
}
int pfhip_op_attention_hd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                          const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                          int max_q_len, float scale, int head_dim, void* stream) {
  if (head_dim != 32 && head_dim != 128) return (int)hipErrorInvalidValue;
  pfhip::launch_attention_hd(Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off, kv_len, B, H, max_q_len, scale, head_dim,
                             S(stream));
  return done();
}
int pfhip_op_attention_planes(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, void* planes_hi, void* planes_lo,
                              int plane_rows, const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                              int max_q_len, int total_q_rows, float scale, void* stream) {
  // the image must hold every query row the launch writes (rows are global: q_off[b] + t), in whole 128-row tiles
  if (!planes_hi || !planes_lo || plane_rows % 128 || total_q_rows > plane_rows || B <= 0 || H <= 0) return (int)hipErrorInvalidValue;
  pfhip::launch_attention_x3(Q, ldq, K, ldk, V, ldv, nullptr, 0, q_off, q_len, kv_off, kv_len, B, H, max_q_len, scale, S(stream), nullptr,
                             nullptr, 0, false, planes_hi, planes_lo, plane_rows);
  return done();
}
int pfhip_op_split_rows(const float* X, int ld, int rows, int cols, void* hi, void* lo, int ldp, void* stream) {
  if (!X || !hi || !lo || rows <= 0 || cols <= 0 || cols % 8 || ld < cols || ldp < cols || ldp % 8) return (int)hipErrorInvalidValue;
  pfhip::launch_split_rows(X, ld, rows, cols, hi, lo, ldp, S(stream));
  return (int)hipGetLastError();
}
int pfhip_op_gemm_p3_qkv(const void* Ah, const void* Al, int rows_a, const void* Wh, const void* Wl, int rows_w, float w_scale, float* C, int ldc,
                         void* kv_hi, void* kv_lo, int ldkv, int q_cols, const float* bias, int M, int N, int K, const float* ln_stats,
                         int ln_tiles, const float* ln_colsum, int tile_rows, void* stream) {
  const int mp = (M + 127) / 128 * 128;
  if (M <= 0 || N <= 0 || K < 16 || K % 16 || N % 128 || rows_a % 128 || rows_w % 128 || rows_a < mp || rows_w < N || !Ah || !Al || !Wh ||
      !Wl || !C || !kv_hi || !kv_lo || q_cols <= 0 || q_cols % 128 || q_cols >= N || ldc < q_cols || ldkv < N - q_cols || ldkv % 8 ||
      !(w_scale > 0.f) || (ln_stats && (!ln_colsum || ln_tiles <= 0)) || (tile_rows != 0 && tile_rows != 64 && tile_rows != 128))
    return (int)hipErrorInvalidValue;
  pfhip::launch_gemm_p3(Ah, Al, rows_a, Wh, Wl, rows_w, w_scale, C, ldc, kv_hi, kv_lo, ldkv, bias, nullptr, 0, M, N, K, false, ln_stats, ln_tiles,
                        ln_colsum, nullptr, 4, S(stream), tile_rows, q_cols);
  return (int)hipGetLastError();
}
int pfhip_op_attention_kvplanes(const float* Q, int ldq, const void* kv_hi, const void* kv_lo, int ldkv, int v_col, int total_kv_rows, float* O,
                                int ldo, void* planes_hi, void* planes_lo, int plane_rows, const int* q_off, const int* q_len,
                                const int* kv_off, const int* kv_len, int B, int H, int max_q_len, int total_q_rows, float scale,
                                const float* fsmn_w, float* mem, int ldmem, int mem_accumulate, void* stream) {
  // shapes the kernel's DMA and stores assume: 16-byte chunks of whole heads inside a plane row, both operands inside one row
  if (!Q || !kv_hi || !kv_lo || B <= 0 || H <= 0 || ldkv % 8 || v_col % 8 || v_col < H * 128 || ldkv < v_col + H * 128 || total_kv_rows <= 0 ||
      (!O && !planes_hi) || (planes_hi && (!planes_lo || plane_rows % 128 || total_q_rows > plane_rows)) || (O && ldo < H * 128) ||
      (fsmn_w && (!mem || ldmem < H * 128)))
    return (int)hipErrorInvalidValue;
  pfhip::launch_attention_p3(Q, ldq, kv_hi, kv_lo, ldkv, v_col, O, ldo, q_off, q_len, kv_off, kv_len, B, H, max_q_len, scale, S(stream), fsmn_w,
                             mem, ldmem, mem_accumulate != 0, planes_hi, planes_lo, plane_rows);
  return done();
}
int pfhip_op_cif(const float* hidden, int ldh, const float* alphas, const int* row_off, const int* len, int B, int D,
                 float threshold, float tail, float* stage, int* n_fires, int* token_num, void* stream) {
  if (D > 1024) return (int)hipErrorInvalidValue;
  pfhip::launch_cif(hidden, ldh, alphas, row_off, len, B, D, threshold, tail, stage, n_fires, token_num, S(stream));
  return done();
}
int pfhip_op_logsoftmax_argmax(const float* logits, int ldl, int ML, int V, float* logp, int32_t* ids, void* stream) {
  pfhip::launch_logsoftmax_argmax(logits, ldl, ML, V, logp, ids, S(stream));
  return done();
}

}  // extern "C"
