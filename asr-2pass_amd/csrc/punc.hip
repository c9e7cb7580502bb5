// CT-Transformer helper kernels: embedding gather fused with x*sqrt(d) + sinusoidal PE, and the
// first-maximum argmax over the first ncls classes (onnxruntime/src/ct-transformer.cpp:193-196).
#include "kernels.h"

#include <math.h>

namespace pfhip {
namespace {
__global__ __launch_bounds__(128) void embed_gather_kernel(const int32_t* __restrict__ ids,
                                                           const float* __restrict__ table, int vocab, int D,
                                                           float* __restrict__ out, int ldo, int N,
                                                           const float* __restrict__ inv_ts, float scale,
                                                           const int* __restrict__ pos_of_row) {
  const int row = blockIdx.x;
  if (row >= N) return;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float pos = (float)((pos_of_row ? pos_of_row[row] : row) + 1);      // position inside its own sequence
  const int half = D >> 1;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const int i = c < half ? c : c - half;
    const float coe = inv_ts[i] * pos;
    const float pe = c < half ? sinf(coe) : cosf(coe);
    out[(size_t)row * ldo + c] = table[(size_t)id * D + c] * scale + pe;
  }
}
__global__ __launch_bounds__(256) void argmax_first_kernel(const float* __restrict__ logits, int ldl, int N, int ncls,
                                                           int32_t* __restrict__ out) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= N) return;
  const float* r = logits + (size_t)row * ldl;
  float best = r[0];
  int bi = 0;
  for (int c = 1; c < ncls; ++c)
    if (r[c] > best) { best = r[c]; bi = c; }        // std::max_element: first maximum
  out[row] = bi;
}
}  // namespace

void launch_embed_gather(const int32_t* ids, const float* table, int vocab, int D, float* out, int ldo, int N,
                         const float* inv_ts, float scale, const int* pos_of_row, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(embed_gather_kernel, dim3(N), dim3(128), 0, s, ids, table, vocab, D, out, ldo, N, inv_ts, scale, pos_of_row);
}
void launch_argmax_first(const float* logits, int ldl, int N, int ncls, int32_t* out, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(argmax_first_kernel, dim3((N + 255) / 256), dim3(256), 0, s, logits, ldl, N, ncls, out);
}
}  // namespace pfhip
