// Row-wise HBM-bound kernels of the SAN-M blocks: LayerNorm, FSMN memory block (depthwise conv over
// time), predictor im2col / alpha head, log-softmax + argmax.  These stand in for the LayerNorm /
// Conv / Softmax / ArgMax nodes of the reference's ONNX graph (onnxruntime/src/paraformer.cpp:541) and
// for GreedySearch/FindMax (paraformer.cpp:386-395, util.cpp:63-74).  One wavefront (64 lanes) per
// row with 16-byte vector accesses; reductions are wave shuffles.
#include "kernels.h"

#include <math.h>

namespace pfhip {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// ---- LayerNorm ----------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx,
                                                        float* __restrict__ y, int ldy,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ b, int M, int D,
                                                        int Dout, float eps, int* range_flag) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = (c < D) ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  // NaN / Inf in the row: a later ReLU would swallow it (fmaxf(NaN, 0) = 0) and the head would see finite garbage — the
  // forward's range flag is raised here, where every residual row passes (kernels.h LaunchCtx)
  if (range_flag && lane == 0 && !(rstd < INFINITY)) atomicOr(range_flag, 1);
  float* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      const float4 gg = *reinterpret_cast<const float4*>(g + c);
      const float4 bb = *reinterpret_cast<const float4*>(b + c);
      float4 o;
      o.x = v[i].x * rstd * gg.x + bb.x;
      o.y = v[i].y * rstd * gg.y + bb.y;
      o.z = v[i].z * rstd * gg.z + bb.z;
      o.w = v[i].w * rstd * gg.w + bb.w;
      *reinterpret_cast<float4*>(yr + c) = o;
    } else if (c < Dout) {
      *reinterpret_cast<float4*>(yr + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// ---- FSMN memory block ----------------------------------------------------------------------------
// Each thread owns 4 channels and walks kTT consecutive frames of one utterance with an 11-row
// sliding window held in registers, so every input row is read once per tile (+10 halo rows).
constexpr int kFsmnK = 11;
constexpr int kTT = 16;

template <int LPAD>   // taps cover rows [t - LPAD, t + 10 - LPAD]
__global__ __launch_bounds__(128) void fsmn_kernel(const float* __restrict__ v, int ldv,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ res, int ldres,
                                                   float* __restrict__ out, int ldo,
                                                   const int* __restrict__ off,
                                                   const int* __restrict__ len, int C) {
  const int b = blockIdx.y;
  const int L = len[b];
  const int t0 = blockIdx.x * kTT;
  if (t0 >= L) return;
  const int c = (blockIdx.z * 128 + threadIdx.x) * 4;
  if (c >= C) return;
  const size_t base = (size_t)off[b];
  float wk[4][kFsmnK];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
#pragma unroll
    for (int j = 0; j < kFsmnK; ++j) wk[ch][j] = w[(size_t)(c + ch) * kFsmnK + j];

  constexpr int kHalf = LPAD;
  // all kTT + 10 input rows of this thread are requested up front (26 independent 16-byte loads in flight per lane; 8-row tiles with more blocks measured slower: the
  // kernel is HBM-latency-bound with only 8 waves per CU), then the 11-tap sums run from registers
  float4 rows[kTT + kFsmnK - 1];
#pragma unroll
  for (int j = 0; j < kTT + kFsmnK - 1; ++j) {
    const int t = t0 - kHalf + j;
    rows[j] = (t < 0 || t >= L) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(v + (base + t) * ldv + c);
  }
  float4 rr[kTT];
  if (res) {
#pragma unroll
    for (int s = 0; s < kTT; ++s)
      rr[s] = (t0 + s < L) ? *reinterpret_cast<const float4*>(res + (base + t0 + s) * ldres + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int s = 0; s < kTT; ++s) {
    const int t = t0 + s;
    if (t < L) {
      float4 o = rows[s + kHalf];
      if (res) { o.x += rr[s].x; o.y += rr[s].y; o.z += rr[s].z; o.w += rr[s].w; }
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < kFsmnK; ++j) {
        a.x += wk[0][j] * rows[s + j].x;
        a.y += wk[1][j] * rows[s + j].y;
        a.z += wk[2][j] * rows[s + j].z;
        a.w += wk[3][j] * rows[s + j].w;
      }
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      *reinterpret_cast<float4*>(out + (base + t) * ldo + c) = o;
    }
  }
}

// ---- predictor ------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void im2col3_kernel(const float* __restrict__ h, int ldh,
                                                      float* __restrict__ col, int ldc,
                                                      const int* __restrict__ row_pos,
                                                      const int* __restrict__ row_len, int M, int D) {
  const int row = blockIdx.x;
  if (row >= M) return;
  const int t = row_pos[row], L = row_len[row];
  for (int c4 = threadIdx.x; c4 < 3 * D / 4; c4 += blockDim.x) {
    const int c = c4 * 4;
    const int j = c / D, cc = c - j * D;
    const int tt = t + j - 1;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tt >= 0 && tt < L) val = *reinterpret_cast<const float4*>(h + (size_t)(row + j - 1) * ldh + cc);
    *reinterpret_cast<float4*>(col + (size_t)row * ldc + c) = val;
  }
}

__global__ __launch_bounds__(256) void alpha_kernel(const float* __restrict__ o, int ldo,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ b, float smooth,
                                                    float noise, float* __restrict__ alphas, int M,
                                                    int D) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const float* orow = o + (size_t)row * ldo;
  float s = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(orow + c);
    const float4 ww = *reinterpret_cast<const float4*>(w + c);
    s += (a.x * ww.x + a.y * ww.y) + (a.z * ww.z + a.w * ww.w);
  }
  s = wave_sum(s) + b[0];
  if (lane == 0) {
    float a = 1.0f / (1.0f + expf(-s));
    a = fmaxf(a * smooth - noise, 0.f);
    alphas[row] = a;
  }
}

// ---- head: log-softmax + argmax --------------------------------------------------------------------
__global__ __launch_bounds__(256) void logsoftmax_argmax_kernel(const float* __restrict__ logits,
                                                                int ldl, int ML, int V,
                                                                float* __restrict__ logp,
                                                                int32_t* __restrict__ ids, int* range_flag) {
  __shared__ float s_max[4];
  __shared__ int s_idx[4];
  __shared__ float s_sum[4];
  const int row = blockIdx.x;
  if (row >= ML) return;
  const float* lr = logits + (size_t)row * ldl;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float best = -INFINITY;
  int bidx = 0x7fffffff;
  int bad = 0;                                  // a NaN never wins a comparison: non-finite logits are looked for element by element
  if ((V & 3) == 0 && (ldl & 3) == 0) {         // 16-byte loads (every vocabulary of the model: 8404 = 4 x 2101 columns, padded rows)
    for (int c4 = threadIdx.x; 4 * c4 < V; c4 += 256) {
      const float4 x = *reinterpret_cast<const float4*>(lr + 4 * c4);
      const int c = 4 * c4;
      if (x.x > best) { best = x.x; bidx = c; }     // strict '>' in column order keeps the first maximum within a thread
      if (x.y > best) { best = x.y; bidx = c + 1; }
      if (x.z > best) { best = x.z; bidx = c + 2; }
      if (x.w > best) { best = x.w; bidx = c + 3; }
      bad |= (int)!(fabsf(x.x) < INFINITY) | (int)!(fabsf(x.y) < INFINITY) | (int)!(fabsf(x.z) < INFINITY) | (int)!(fabsf(x.w) < INFINITY);
    }
  } else {
    for (int c = threadIdx.x; c < V; c += 256) {
      const float x = lr[c];
      if (x > best) { best = x; bidx = c; }     // strict '>' keeps the first maximum within a thread
      bad |= (int)!(fabsf(x) < INFINITY);
    }
  }
  // wave arg-max, ties -> smaller index (util.cpp:63-74 scans left to right with strict '>')
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bidx, off);
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  }
  if (lane == 0) { s_max[wave] = best; s_idx[wave] = bidx; }
  __syncthreads();
  float m = s_max[0];
  int mi = s_idx[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (s_max[i] > m || (s_max[i] == m && s_idx[i] < mi)) { m = s_max[i]; mi = s_idx[i]; }
  if (!logp) {      // ids only (greedy search): no second pass over the row
    if (range_flag && __any(bad)) { if (lane == 0) atomicOr(range_flag, 1); }
    if (threadIdx.x == 0) ids[row] = mi;
    return;
  }
  float sum = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) sum += expf(lr[c] - m);
  sum = wave_sum(sum);
  if (lane == 0) s_sum[wave] = sum;
  __syncthreads();
  const float lse = logf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]));
  if (threadIdx.x == 0) {
    ids[row] = mi;
    if (range_flag && !(fabsf(lse) < INFINITY)) atomicOr(range_flag, 1);      // NaN / Inf reached the logits
  }
  if (logp) {
    float* pr = logp + (size_t)row * V;
    for (int c = threadIdx.x; c < V; c += 256) pr[c] = (lr[c] - m) - lse;
  }
}

__global__ __launch_bounds__(128) void compact_kernel(const float* __restrict__ stage,
                                                      float* __restrict__ emb,
                                                      const int* __restrict__ src_row, int ML,
                                                      int D) {
  const int row = blockIdx.x;
  if (row >= ML) return;
  const float* s = stage + (size_t)src_row[row] * D;
  float* d = emb + (size_t)row * D;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4)
    *reinterpret_cast<float4*>(d + c) = *reinterpret_cast<const float4*>(s + c);
}

__global__ __launch_bounds__(128) void gather_rows_kernel(const int32_t* __restrict__ ids,
                                                          const float* __restrict__ table, int D,
                                                          float* __restrict__ out, int R) {
  const int r = blockIdx.x;
  if (r >= R) return;
  const float* src = table + (size_t)ids[r] * D;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4)
    *reinterpret_cast<float4*>(out + (size_t)r * D + c) = *reinterpret_cast<const float4*>(src + c);
}

__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ G, float* __restrict__ c,
                                                        float* __restrict__ h, const int32_t* __restrict__ lens,
                                                        int t, float* __restrict__ sel, int H, int D) {
  const int j = blockIdx.x;
  if (j >= H) return;
  const float* g = G + (size_t)j * 4 * D;
  const bool pick = lens[j] - 1 == t;
  for (int k = threadIdx.x; k < D; k += blockDim.x) {
    const float ig = 1.0f / (1.0f + expf(-g[k]));
    const float fg = 1.0f / (1.0f + expf(-g[D + k]));
    const float gg = tanhf(g[2 * D + k]);
    const float og = 1.0f / (1.0f + expf(-g[3 * D + k]));
    const float cn = fg * c[(size_t)j * D + k] + ig * gg;
    const float hn = og * tanhf(cn);
    c[(size_t)j * D + k] = cn;
    h[(size_t)j * D + k] = hn;
    if (pick) sel[(size_t)j * D + k] = hn;
  }
}

}  // namespace

void launch_gather_rows(const int32_t* ids, const float* table, int D, float* out, int R, hipStream_t s) {
  if (R <= 0) return;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(R), dim3(128), 0, s, ids, table, D, out, R);
}

void launch_lstm_cell(const float* G, float* c, float* h, const int32_t* lens, int t, float* sel, int H, int D,
                      hipStream_t s) {
  if (H <= 0) return;
  hipLaunchKernelGGL(lstm_cell_kernel, dim3(H), dim3(256), 0, s, G, c, h, lens, t, sel, H, D);
}

void launch_layernorm(const float* x, int ldx, float* y, int ldy, const float* g, const float* b,
                      int M, int D, int Dout, float eps, hipStream_t s) {
  if (M <= 0) return;
  const dim3 grid((M + 3) / 4), block(256);
  const int nv = (Dout + 255) / 256;
  if (nv <= 2)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, ldx, y, ldy, g, b, M, D, Dout, eps, launch_ctx().range_flag);
  else if (nv <= 3)
    hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, x, ldx, y, ldy, g, b, M, D, Dout, eps, launch_ctx().range_flag);
  else
    hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, ldx, y, ldy, g, b, M, D, Dout, eps, launch_ctx().range_flag);
}

void launch_fsmn(const float* v, int ldv, const float* w, const float* res, int ldres, float* out,
                 int ldo, const int* off, const int* len, int B, int max_len, int C, hipStream_t s) {
  if (B <= 0 || max_len <= 0) return;
  const dim3 grid((max_len + kTT - 1) / kTT, B, (C + 511) / 512), block(128);
  hipLaunchKernelGGL(fsmn_kernel<5>, grid, block, 0, s, v, ldv, w, res, ldres, out, ldo, off, len, C);
}

void launch_fsmn_shift(const float* v, int ldv, const float* w, const float* res, int ldres, float* out, int ldo,
                       const int* off, const int* len, int B, int max_len, int C, int shift, hipStream_t s) {
  if (B <= 0 || max_len <= 0) return;
  const dim3 grid((max_len + kTT - 1) / kTT, B, (C + 511) / 512), block(128);
  if (shift == 5)
    hipLaunchKernelGGL(fsmn_kernel<10>, grid, block, 0, s, v, ldv, w, res, ldres, out, ldo, off, len, C);
  else
    hipLaunchKernelGGL(fsmn_kernel<5>, grid, block, 0, s, v, ldv, w, res, ldres, out, ldo, off, len, C);
}

void launch_im2col3(const float* h, int ldh, float* col, int ldc, const int* row_pos,
                    const int* row_len, int M, int D, hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(im2col3_kernel, dim3(M), dim3(128), 0, s, h, ldh, col, ldc, row_pos, row_len, M, D);
}

void launch_alpha(const float* o, int ldo, const float* w, const float* b, float smooth, float noise,
                  float* alphas, int M, int D, hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(alpha_kernel, dim3((M + 3) / 4), dim3(256), 0, s, o, ldo, w, b, smooth, noise,
                     alphas, M, D);
}

void launch_compact(const float* stage, float* emb, const int* tok_row_src, int ML, int D,
                    hipStream_t s) {
  if (ML <= 0) return;
  hipLaunchKernelGGL(compact_kernel, dim3(ML), dim3(128), 0, s, stage, emb, tok_row_src, ML, D);
}

void launch_logsoftmax_argmax(const float* logits, int ldl, int ML, int V, float* logp, int32_t* ids,
                              hipStream_t s, int* range_flag) {
  if (ML <= 0) return;
  hipLaunchKernelGGL(logsoftmax_argmax_kernel, dim3(ML), dim3(256), 0, s, logits, ldl, ML, V, logp, ids, range_flag);
}

}  // namespace pfhip
