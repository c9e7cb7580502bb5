// CIF (continuous integrate-and-fire) per utterance — the scalar recurrence the reference states in
// ParaformerOnline::CifSearch (onnxruntime/src/paraformer-online.cpp:301-327), plus the offline tail
// slot (alpha = tail_threshold on a zero hidden frame).
//
// The fire decisions are threshold comparisons on a running float sum, so the integrate scalar is kept
// strictly sequential (every thread carries its own identical copy: no cross-lane traffic, no
// reordering); what is parallel is the weighted accumulation of the hidden vectors, one channel per
// thread.  Built with -ffp-contract=off: frames += alpha*h rounds like the reference's scalar loop.
// Latency-bound (T+1 dependent steps); rows are prefetched 8 deep to keep the HBM pipe busy.
#include "kernels.h"

#include <math.h>

namespace pfhip {
namespace {

constexpr int kPf = 8;

template <int NC>
__global__ __launch_bounds__(512) void cif_kernel(const float* __restrict__ hidden, int ldh,
                                                  const float* __restrict__ alphas,
                                                  const int* __restrict__ row_off,
                                                  const int* __restrict__ len, int D, float thr,
                                                  float tail, float* __restrict__ stage,
                                                  int* __restrict__ n_fires,
                                                  int* __restrict__ token_num) {
  const int b = blockIdx.x;
  const int T = len[b];
  if (T <= 0) {
    if (threadIdx.x == 0) { n_fires[b] = 0; token_num[b] = 0; }
    return;
  }
  const size_t base = (size_t)row_off[b];
  float* out = stage + (base + b) * D;
  const float* al = alphas + base;

  float frames[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) frames[c] = 0.f;
  float integrate = 0.f, asum = 0.f;
  int nf = 0;

  for (int i0 = 0; i0 <= T; i0 += kPf) {
    float a[kPf];
    float hv[kPf][NC];
#pragma unroll
    for (int u = 0; u < kPf; ++u) {
      const int i = i0 + u;
      a[u] = (i < T) ? al[i] : tail;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = threadIdx.x + c * 512;
        hv[u][c] = (i < T && ch < D) ? hidden[(base + i) * ldh + ch] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < kPf; ++u) {
      if (i0 + u > T) break;
      const float alpha = a[u];
      asum += alpha;
      if (alpha + integrate < thr) {
        integrate += alpha;
#pragma unroll
        for (int c = 0; c < NC; ++c) frames[c] += alpha * hv[u][c];
      } else {
        const float w = thr - integrate;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          frames[c] += w * hv[u][c];
          const int ch = threadIdx.x + c * 512;
          if (ch < D) out[(size_t)nf * D + ch] = frames[c];
        }
        ++nf;
        integrate += alpha;
        integrate -= thr;
#pragma unroll
        for (int c = 0; c < NC; ++c) frames[c] = integrate * hv[u][c];
      }
    }
  }
  if (threadIdx.x == 0) {
    n_fires[b] = nf;
    token_num[b] = (int)floorf(asum);
  }
}

}  // namespace

void launch_cif(const float* hidden, int ldh, const float* alphas, const int* row_off,
                const int* len, int B, int D, float threshold, float tail, float* stage,
                int* n_fires, int* token_num, hipStream_t s) {
  if (B <= 0) return;
  if (D <= 512)
    hipLaunchKernelGGL(cif_kernel<1>, dim3(B), dim3(512), 0, s, hidden, ldh, alphas, row_off, len, D,
                       threshold, tail, stage, n_fires, token_num);
  else
    hipLaunchKernelGGL(cif_kernel<2>, dim3(B), dim3(512), 0, s, hidden, ldh, alphas, row_off, len, D,
                       threshold, tail, stage, n_fires, token_num);
}

}  // namespace pfhip
