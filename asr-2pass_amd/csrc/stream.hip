// Small kernels of the chunk-streaming path (ParaformerOnline, onnxruntime/src/paraformer-online.cpp).
// The stream state (fbank splice cache, overlap-window feature cache, CIF carry, decoder FSMN caches)
// lives in HBM per connection; the host only keeps counters.  A chunk is 20 x 560 floats: these kernels
// are launch-latency-bound, the chunk's cost is the weight stream of the GEMMs.
#include "kernels.h"

#include <math.h>

namespace pfhip {
namespace {

constexpr int kMels = 80, kLfrM = 7, kLfrN = 6, kFeat = kMels * kLfrM;

__global__ __launch_bounds__(192) void stream_lfr_kernel(const float* __restrict__ fb, int T, int n_rows,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ istd, float scale,
                                                         const float* __restrict__ inv_ts, int pos0,
                                                         float* __restrict__ out, int ldo) {
  const int i = blockIdx.x;
  if (i >= n_rows) return;
  const float pos = (float)(pos0 + i + 1);
  for (int c = threadIdx.x; c < kFeat; c += blockDim.x) {
    const int j = c / kMels, bin = c - j * kMels;
    int f = i * kLfrN + j;
    if (f > T - 1) f = T - 1;                       // OnlineLfrCmvn :210-218: pad with the last frame
    float x = fb[(size_t)f * kMels + bin];
    x = (x + mean[c]) * istd[c];                    // :231-235
    x = x * scale;                                  // Forward :549-553
    const int half = kFeat / 2;
    const int k = c < half ? c : c - half;
    const float coe = inv_ts[k] * pos;              // GetPosEmb :251-259
    x = x + (c < half ? sinf(coe) : cosf(coe));
    out[(size_t)i * ldo + c] = x;
  }
}

__global__ __launch_bounds__(256) void rows_copy_kernel(float* __restrict__ dst, int ldd,
                                                        const float* __restrict__ src, int lds_, int nrows,
                                                        int ncols) {
  const int r = blockIdx.x;
  if (r >= nrows) return;
  for (int c = threadIdx.x; c < ldd; c += blockDim.x)
    dst[(size_t)r * ldd + c] = (src && c < ncols) ? src[(size_t)r * lds_ + c] : 0.f;
}

// One block per connection of the batch (blockIdx.x): the window of stream b is rows [row_off, row_off + n) of the packed
// encoder output, its fires go to emb_all[b * emb_rows ...], its carry lives in the stream's own buffer.
// Batched forms for many connections: blockIdx.y = operation, each with its own pointers (descriptor array in HBM).
__global__ __launch_bounds__(192) void stream_lfr_batch_kernel(const StreamLfrOp* __restrict__ ops, const float* __restrict__ mean,
                                                               const float* __restrict__ istd, float scale,
                                                               const float* __restrict__ inv_ts, int ldo) {
  const StreamLfrOp op = ops[blockIdx.y];
  const int i = blockIdx.x;
  if (i >= op.n_rows) return;
  const float pos = (float)(op.pos0 + i + 1);
  for (int c = threadIdx.x; c < kFeat; c += blockDim.x) {
    const int j = c / kMels, bin = c - j * kMels;
    int f = i * kLfrN + j;
    if (f > op.T - 1) f = op.T - 1;
    float x = op.fb[(size_t)f * kMels + bin];
    x = (x + mean[c]) * istd[c];
    x = x * scale;
    const int half = kFeat / 2;
    const int k = c < half ? c : c - half;
    const float coe = inv_ts[k] * pos;
    x = x + (c < half ? sinf(coe) : cosf(coe));
    op.out[(size_t)i * ldo + c] = x;
  }
}

__global__ __launch_bounds__(256) void rows_copy_batch_kernel(const RowsCopyOp* __restrict__ ops) {
  const RowsCopyOp op = ops[blockIdx.y];
  const int r = blockIdx.x;
  if (r >= op.nrows) return;
  for (int c = threadIdx.x; c < op.ldd; c += blockDim.x)
    op.dst[(size_t)r * op.ldd + c] = (op.src && c < op.ncols) ? op.src[(size_t)r * op.lds + c] : 0.f;
}

template <int NC>
__global__ __launch_bounds__(512) void cif_stream_kernel(const float* __restrict__ enc_all, int lde,
                                                         const float* __restrict__ alphas_all,
                                                         const StreamSeg* __restrict__ segs, float thr, float tail,
                                                         float* __restrict__ emb_all, int emb_rows,
                                                         int* __restrict__ n_fire_all, int D) {
  const StreamSeg sg = segs[blockIdx.x];
  const float* enc = enc_all + (size_t)sg.row_off * lde;
  const float* alphas = alphas_all + sg.row_off;
  const int n = sg.n, pre = sg.pre, suf = sg.suf, is_last = sg.is_last;
  float* carry_hidden = sg.carry;
  float* carry_alpha = sg.carry + D;
  float* emb = emb_all + (size_t)blockIdx.x * emb_rows * D;
  int* n_fire = n_fire_all + blockIdx.x;
  float frames[NC], hv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) frames[c] = 0.f;
  float integrate = 0.f;
  int nf = 0;
  const int len_time = 1 + n + (is_last ? 1 : 0);
  for (int i = 0; i < len_time; ++i) {
    float alpha;
    if (i == 0) {
      alpha = carry_alpha[0];
#pragma unroll
      for (int c = 0; c < NC; ++c) { const int ch = threadIdx.x + c * 512; hv[c] = ch < D ? carry_hidden[ch] : 0.f; }
    } else if (i <= n) {
      const int t = i - 1;
      alpha = (t < pre || t >= suf) ? 0.f : alphas[t];            // :279-286
#pragma unroll
      for (int c = 0; c < NC; ++c) { const int ch = threadIdx.x + c * 512; hv[c] = ch < D ? enc[(size_t)t * lde + ch] : 0.f; }
    } else {
      alpha = tail;                                                // :295-299
#pragma unroll
      for (int c = 0; c < NC; ++c) hv[c] = 0.f;
    }
    if (alpha + integrate < thr) {                                 // :306-327
      integrate += alpha;
#pragma unroll
      for (int c = 0; c < NC; ++c) frames[c] += alpha * hv[c];
    } else {
      const float w = thr - integrate;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        frames[c] += w * hv[c];
        const int ch = threadIdx.x + c * 512;
        if (ch < D && nf < emb_rows) emb[(size_t)nf * D + ch] = frames[c];
      }
      ++nf;
      integrate += alpha;
      integrate -= thr;
#pragma unroll
      for (int c = 0; c < NC; ++c) frames[c] = integrate * hv[c];
    }
  }
  __syncthreads();                                                 // all reads of the old carry are done
  // :330-340
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = threadIdx.x + c * 512;
    if (ch < D) carry_hidden[ch] = integrate > 0.f ? frames[c] / integrate : frames[c];
  }
  if (threadIdx.x == 0) { carry_alpha[0] = integrate; n_fire[0] = nf; }
}

constexpr int kFsmnK = 11;

// blockIdx.y = connection: its tokens are rows [tok_off, tok_off + n_tok) of the packed decoder matrices, its cache for
// this layer is dcache + layer * 10 * C.  A connection without tokens in this round keeps its cache untouched.
__global__ __launch_bounds__(128) void fsmn_cached_kernel(const float* __restrict__ t2_all,
                                                          const float* __restrict__ w, const float* res_all,
                                                          float* out_all, const StreamSeg* __restrict__ segs, int layer,
                                                          int C) {
  const int c = (blockIdx.x * 128 + threadIdx.x) * 4;
  if (c >= C) return;
  const StreamSeg sg = segs[blockIdx.y];
  const int N = sg.n_tok;
  if (N <= 0) return;
  const float* t2 = t2_all + (size_t)sg.tok_off * C;
  const float* res = res_all + (size_t)sg.tok_off * C;
  float* out = out_all + (size_t)sg.tok_off * C;
  float* cache = sg.dcache + (size_t)layer * (kFsmnK - 1) * C;
  float wk[4][kFsmnK];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
#pragma unroll
    for (int j = 0; j < kFsmnK; ++j) wk[ch][j] = w[(size_t)(c + ch) * kFsmnK + j];
  // window of the last k rows of xcat = [cache(10 rows); t2(N rows)]
  float4 win[kFsmnK];
#pragma unroll
  for (int j = 0; j < kFsmnK - 1; ++j) win[j + 1] = *reinterpret_cast<const float4*>(cache + (size_t)j * C + c);
  for (int n = 0; n < N; ++n) {
#pragma unroll
    for (int j = 0; j < kFsmnK - 1; ++j) win[j] = win[j + 1];
    win[kFsmnK - 1] = *reinterpret_cast<const float4*>(t2 + (size_t)n * C + c);
    float4 o = win[kFsmnK - 1];
    const float4 rr = *reinterpret_cast<const float4*>(res + (size_t)n * C + c);
    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < kFsmnK; ++j) {
      a.x += wk[0][j] * win[j].x; a.y += wk[1][j] * win[j].y;
      a.z += wk[2][j] * win[j].z; a.w += wk[3][j] * win[j].w;
    }
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    *reinterpret_cast<float4*>(out + (size_t)n * C + c) = o;
  }
  // new cache = last k-1 rows of xcat = win[1..k-1]
#pragma unroll
  for (int j = 0; j < kFsmnK - 1; ++j) *reinterpret_cast<float4*>(cache + (size_t)j * C + c) = win[j + 1];
}

}  // namespace

void launch_stream_lfr(const float* fb, int T, int n_rows, const float* mean, const float* istd, float scale,
                       const float* inv_ts, int pos0, float* out, int ldo, hipStream_t s) {
  if (n_rows <= 0) return;
  hipLaunchKernelGGL(stream_lfr_kernel, dim3(n_rows), dim3(192), 0, s, fb, T, n_rows, mean, istd, scale, inv_ts,
                     pos0, out, ldo);
}

void launch_rows_copy(float* dst, int ldd, const float* src, int lds_, int nrows, int ncols, hipStream_t s) {
  if (nrows <= 0) return;
  hipLaunchKernelGGL(rows_copy_kernel, dim3(nrows), dim3(256), 0, s, dst, ldd, src, lds_, nrows, ncols);
}

void launch_stream_lfr_batch(const StreamLfrOp* ops, int n_ops, int max_rows, const float* mean, const float* istd, float scale,
                             const float* inv_ts, int ldo, hipStream_t s) {
  if (n_ops <= 0 || max_rows <= 0) return;
  hipLaunchKernelGGL(stream_lfr_batch_kernel, dim3(max_rows, n_ops), dim3(192), 0, s, ops, mean, istd, scale, inv_ts, ldo);
}

void launch_rows_copy_batch(const RowsCopyOp* ops, int n_ops, int max_rows, hipStream_t s) {
  if (n_ops <= 0 || max_rows <= 0) return;
  hipLaunchKernelGGL(rows_copy_batch_kernel, dim3(max_rows, n_ops), dim3(256), 0, s, ops);
}

void launch_cif_stream(const float* enc, int lde, const float* alphas, const StreamSeg* segs, int B, float threshold, float tail,
                       float* emb_all, int emb_rows, int* n_fire, int D, hipStream_t s) {
  if (B <= 0) return;
  if (D <= 512)
    hipLaunchKernelGGL(cif_stream_kernel<1>, dim3(B), dim3(512), 0, s, enc, lde, alphas, segs, threshold, tail, emb_all, emb_rows,
                       n_fire, D);
  else
    hipLaunchKernelGGL(cif_stream_kernel<2>, dim3(B), dim3(512), 0, s, enc, lde, alphas, segs, threshold, tail, emb_all, emb_rows,
                       n_fire, D);
}

void launch_fsmn_cached(const float* t2, const float* w, const float* res, float* out, const StreamSeg* segs, int B, int layer,
                        int C, hipStream_t s) {
  if (B <= 0) return;
  hipLaunchKernelGGL(fsmn_cached_kernel, dim3((C + 511) / 512, B), dim3(128), 0, s, t2, w, res, out, segs, layer, C);
}

}  // namespace pfhip
