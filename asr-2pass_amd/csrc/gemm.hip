// fp32 GEMM on the gfx950 matrix cores: C = A * W^T (+bias, +residuals, ReLU).
//
// This is what replaces the MatMul/Gemm nodes inside the reference's opaque `m_session_->Run`
// (onnxruntime/src/paraformer.cpp:541; ORT CPU MLAS SGEMM, 1 intra-op thread).  fp32 in / fp32
// accumulate on v_mfma_f32_32x32x2_f32 — bit-wise an fmaf chain, so the 1e-3 log-prob tolerance of
// BASELINE.json is met with margin.  MFMA-bound: peak 157.3 TFLOP/s (MI355X_MICROARCH.md).
//
// Tiling: 128x128 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of 32x32 (64 fp32
// accumulators per lane).  K-step 32.  Both operands are K-contiguous (activations [M][K], weights
// torch-style [N][K]), staged global -> registers -> LDS with a 36-float row stride (144 B = 9 slots
// of 16 B, so the 16-lane groups of ds_read_b128 hit 16 distinct slots: conflict-free), two LDS
// buffers, next tile's global loads in flight under the current tile's 64 MFMAs per wave.
// A lane reads 4 consecutive k of its row with one ds_read_b128; lane half h takes k = 8*kb+4*h+kk
// at MFMA step kk, for A and B alike, so the k permutation cancels.
#include "kernels.h"

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kLds = 36;  // padded row stride in floats

__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* __restrict__ C,
    int ldc, const float* __restrict__ bias, const float* __restrict__ R1, int ldr1,
    const float* __restrict__ R2, int ldr2, int M, int N, int K, int relu, int tiles_n, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float As[2][kTileM * kLds];
  __shared__ __attribute__((aligned(16))) float Bs[2][kTileN * kLds];

  // XCD-aware tile order (cdna_hip_programming.md T1, bijective form): blocks that share an XCD
  // (equal blockIdx % 8) walk a contiguous run of tiles, n fastest, so an A row-panel is fetched into
  // one L2 instead of eight.
  int bid = blockIdx.x;
  {
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * kTileM, n0 = tn * kTileN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int lrow = tid >> 3, lc4 = tid & 7;

  const float* Ag = A + (size_t)(m0 + lrow) * lda + 4 * lc4;
  const float* Wg = W + (size_t)(n0 + lrow) * ldw + 4 * lc4;

  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(Ag + (size_t)(32 * i) * lda + k0);
      rb[i] = *reinterpret_cast<const float4*>(Wg + (size_t)(32 * i) * ldw + k0);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&As[buf][(lrow + 32 * i) * kLds + 4 * lc4]) = ra[i];
      *reinterpret_cast<float4*>(&Bs[buf][(lrow + 32 * i) * kLds + 4 * lc4]) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / kTileK;
  gload(0);
  sstore(0);
  __syncthreads();

  const int a_off = (wr * 64 + r) * kLds + 4 * h;
  const int b_off = (wc * 64 + r) * kLds + 4 * h;

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * kTileK);
    const float* as = &As[cur][a_off];
    const float* bs = &Bs[cur][b_off];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 a0 = *reinterpret_cast<const float4*>(as + kb * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(as + 32 * kLds + kb * 8);
      const float4 b0 = *reinterpret_cast<const float4*>(bs + kb * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(bs + 32 * kLds + kb * 8);
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w};
      const float av1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv0[4] = {b0.x, b0.y, b0.z, b0.w};
      const float bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[kk], bv0[kk], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[kk], bv1[kk], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[kk], bv0[kk], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[kk], bv1[kk], acc[1][1], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + j * 32 + r;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < M) {
          float v = acc[i][j][e] + bv;
          if (R1) v += R1[(size_t)row * ldr1 + col];
          if (R2) v += R2[(size_t)row * ldr2 + col];
          if (relu) v = fmaxf(v, 0.f);
          C[(size_t)row * ldc + col] = v;
        }
      }
    }
  }
}

}  // namespace

void launch_gemm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc,
                     const float* bias, const float* R1, int ldr1, const float* R2, int ldr2, int M,
                     int N, int K, bool relu, hipStream_t s) {
  if (M <= 0 || N <= 0) return;
  const int tiles_m = (M + kTileM - 1) / kTileM;
  const int tiles_n = (N + kTileN - 1) / kTileN;
  const int n_tiles = tiles_m * tiles_n;
  hipLaunchKernelGGL(gemm_f32_mfma_kernel, dim3(n_tiles), dim3(256), 0, s, A, lda, W, ldw, C, ldc,
                     bias, R1, ldr1, R2, ldr2, M, N, K, relu ? 1 : 0, tiles_n, n_tiles);
}

}  // namespace pfhip
