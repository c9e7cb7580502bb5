// FSMN-VAD kernels: generic LFR+CMVN gather, the causal memory block (depthwise conv over
// the `lorder` most recent frames with a (lorder-1)-frame cache), row softmax.  All HBM-bound and tiny next to
// the ASR model: a 10-minute file is 60 000 frames x ~0.4 M MACs.
#include "kernels.h"

#include <math.h>

namespace pfhip {
namespace {

// LfrCmvn (onnxruntime/src/fsmn-vad.cpp:198-238 == paraformer.cpp:421-461) for any (m, n): row i = frames
// [i*n - (m-1)/2, ...) with the first / last frame replicated; out = (x + mean) * istd; pad columns zeroed.
// lp = left padding in frames: (m-1)/2 for the offline routine; 0 for OnlineLfrCmvn (fsmn-vad-online.cpp:90-133), whose
// caller keeps the (m-1)/2 history frames in front of fb itself (lfr_splice_cache_).
__global__ __launch_bounds__(128) void lfr_cmvn_kernel(const float* __restrict__ fb, int F, int T, int m, int n,
                                                       int n_mels, const float* __restrict__ mean,
                                                       const float* __restrict__ istd, float* __restrict__ out,
                                                       int ldo, int lp) {
  const int i = blockIdx.x;
  if (i >= T) return;
  const int D = m * n_mels;
  for (int c = threadIdx.x; c < ldo; c += blockDim.x) {
    float v = 0.f;
    if (c < D) {
      const int j = c / n_mels, bin = c - j * n_mels;
      int f = i * n + j - lp;
      f = f < 0 ? 0 : (f > F - 1 ? F - 1 : f);
      v = (fb[(size_t)f * n_mels + bin] + mean[c]) * istd[c];
    }
    out[(size_t)i * ldo + c] = v;
  }
}

// out[t][c] = p[t][c] + sum_{j<K} w[c][j] * xcat[t+j][c], xcat = [cache (K-1 rows); p (T rows)].
// Time is tiled (kTT rows per block) with a K-row register window per thread (4 channels each).
// blockIdx.z = connection (VadSeg): its rows [row_off, row_off + T) of the packed matrices, its own cache for this layer.
template <int K>
__global__ __launch_bounds__(64) void fsmn_causal_kernel(const float* __restrict__ p_all, int ldp,
                                                         const float* __restrict__ w,
                                                         const VadSeg* __restrict__ segs, int layer,
                                                         float* __restrict__ out_all, int ldo, int C) {
  constexpr int kTT = 32;
  const int c = (blockIdx.y * 64 + threadIdx.x) * 4;
  if (c >= C) return;
  const VadSeg sg = segs[blockIdx.z];
  const int T = sg.T;
  const int t0 = blockIdx.x * kTT;
  if (t0 >= T) return;
  const float* p = p_all + (size_t)sg.row_off * ldp;
  float* out = out_all + (size_t)sg.row_off * ldo;
  const float* cache = sg.cache_in + (size_t)layer * (K - 1) * C;
  float wk[4][K];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
#pragma unroll
    for (int j = 0; j < K; ++j) wk[ch][j] = w[(size_t)(c + ch) * K + j];
  auto load_row = [&](int t) -> float4 {          // t relative to p; negative -> cache row (K-1+t)
    if (t >= 0) return t < T ? *reinterpret_cast<const float4*>(p + (size_t)t * ldp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(cache + (size_t)(K - 1 + t) * C + c);
  };
  float4 win[K];
#pragma unroll
  for (int j = 0; j < K - 1; ++j) win[j + 1] = load_row(t0 - (K - 1) + j);
  for (int s = 0; s < kTT; ++s) {
    const int t = t0 + s;
    if (t >= T) break;
#pragma unroll
    for (int j = 0; j < K - 1; ++j) win[j] = win[j + 1];
    win[K - 1] = load_row(t);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      a.x += wk[0][j] * win[j].x; a.y += wk[1][j] * win[j].y;
      a.z += wk[2][j] * win[j].z; a.w += wk[3][j] * win[j].w;
    }
    float4 o = win[K - 1];
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    *reinterpret_cast<float4*>(out + (size_t)t * ldo + c) = o;
  }
}

// new cache = last K-1 rows of [cache; p]
__global__ __launch_bounds__(128) void fsmn_cache_update_kernel(const float* __restrict__ p_all, int ldp,
                                                                const VadSeg* __restrict__ segs, int layer, int C,
                                                                int K1) {
  const VadSeg sg = segs[blockIdx.y];
  if (!sg.cache_out) return;                      // final call: caches are not advanced (fsmn-vad.cpp:129-134)
  const float* p = p_all + (size_t)sg.row_off * ldp;
  const float* cache_in = sg.cache_in + (size_t)layer * K1 * C;
  float* cache_out = sg.cache_out + (size_t)layer * K1 * C;
  const int T = sg.T;
  const int j = blockIdx.x;                       // row of the new cache, 0..K1-1
  const int src = T - K1 + j;                     // row of p, negative -> old cache row K1 + src
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    cache_out[(size_t)j * C + c] = src >= 0 ? p[(size_t)src * ldp + c] : cache_in[(size_t)(K1 + src) * C + c];
}

// col0 (optional): the class-0 column alone, packed — the only score the end-point detector reads (e2e-vad.h:607-609)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int ldx, int M, int N,
                                                           float* __restrict__ y, float* __restrict__ col0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const float* xr = x + (size_t)row * ldx;
  float m = -INFINITY;
  for (int c = lane; c < N; c += 64) m = fmaxf(m, xr[c]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  float s = 0.f;
  for (int c = lane; c < N; c += 64) s += expf(xr[c] - m);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  for (int c = lane; c < N; c += 64) y[(size_t)row * N + c] = expf(xr[c] - m) / s;
  if (col0 && lane == 0) col0[row] = expf(xr[0] - m) / s;
}

}  // namespace

void launch_lfr_cmvn(const float* fb, int F, int T, int m, int n, int n_mels, const float* mean, const float* istd,
                     float* out, int ldo, hipStream_t s) {
  if (T <= 0) return;
  hipLaunchKernelGGL(lfr_cmvn_kernel, dim3(T), dim3(128), 0, s, fb, F, T, m, n, n_mels, mean, istd, out, ldo, (m - 1) / 2);
}

void launch_lfr_cmvn_online(const float* fb, int F, int T, int m, int n, int n_mels, const float* mean, const float* istd,
                            float* out, int ldo, hipStream_t s) {
  if (T <= 0) return;
  hipLaunchKernelGGL(lfr_cmvn_kernel, dim3(T), dim3(128), 0, s, fb, F, T, m, n, n_mels, mean, istd, out, ldo, 0);
}

void launch_fsmn_causal20(const float* p, int ldp, const float* w, const VadSeg* segs, int B, int max_T, int layer, float* out,
                          int ldo, int C, hipStream_t s) {
  if (B <= 0 || max_T <= 0) return;
  hipLaunchKernelGGL(fsmn_causal_kernel<20>, dim3((max_T + 31) / 32, (C + 255) / 256, B), dim3(64), 0, s, p, ldp, w, segs,
                     layer, out, ldo, C);
  hipLaunchKernelGGL(fsmn_cache_update_kernel, dim3(19, B), dim3(128), 0, s, p, ldp, segs, layer, C, 19);
}

// OnlineLfrCmvn rows of many connections in one launch: op i writes rows [row_off, row_off + n_out) of the packed matrix
__global__ __launch_bounds__(128) void lfr_cmvn_online_batch_kernel(const VadLfrOp* __restrict__ ops, int m, int n, int n_mels,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ istd, float* __restrict__ out,
                                                                    int ldo) {
  const VadLfrOp op = ops[blockIdx.y];
  const int i = blockIdx.x;
  if (i >= op.n_out) return;
  const int D = m * n_mels;
  for (int c = threadIdx.x; c < ldo; c += blockDim.x) {
    float v = 0.f;
    if (c < D) {
      const int j = c / n_mels, bin = c - j * n_mels;
      int f = i * n + j;
      f = f > op.Tin - 1 ? op.Tin - 1 : f;
      v = (op.fb[(size_t)f * n_mels + bin] + mean[c]) * istd[c];
    }
    out[(size_t)(op.row_off + i) * ldo + c] = v;
  }
}

void launch_lfr_cmvn_online_batch(const VadLfrOp* ops, int n_ops, int max_rows, int m, int n, int n_mels, const float* mean,
                                  const float* istd, float* out, int ldo, hipStream_t s) {
  if (n_ops <= 0 || max_rows <= 0) return;
  hipLaunchKernelGGL(lfr_cmvn_online_batch_kernel, dim3(max_rows, n_ops), dim3(128), 0, s, ops, m, n, n_mels, mean, istd, out, ldo);
}

void launch_softmax_rows(const float* x, int ldx, int M, int N, float* y, float* col0, hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, ldx, M, N, y, col0);
}

}  // namespace pfhip
