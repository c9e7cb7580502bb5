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

template <bool GUARD, bool HAS_BIAS, bool HAS_R1, bool HAS_R2, bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C,
    int ldc, const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int tiles_n, int n_tiles) {
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

  // staging registers are named scalars (not arrays captured by lambdas): hipcc keeps them in VGPRs
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define PFHIP_GLOAD(k0)                                                               \
  do {                                                                                \
    ra0 = *reinterpret_cast<const float4*>(Ag + (k0));                                \
    ra1 = *reinterpret_cast<const float4*>(Ag + (size_t)32 * lda + (k0));             \
    ra2 = *reinterpret_cast<const float4*>(Ag + (size_t)64 * lda + (k0));             \
    ra3 = *reinterpret_cast<const float4*>(Ag + (size_t)96 * lda + (k0));             \
    rb0 = *reinterpret_cast<const float4*>(Wg + (k0));                                \
    rb1 = *reinterpret_cast<const float4*>(Wg + (size_t)32 * ldw + (k0));             \
    rb2 = *reinterpret_cast<const float4*>(Wg + (size_t)64 * ldw + (k0));             \
    rb3 = *reinterpret_cast<const float4*>(Wg + (size_t)96 * ldw + (k0));             \
  } while (0)
#define PFHIP_SSTORE(buf)                                                             \
  do {                                                                                \
    float* as_ = &As[buf][lrow * kLds + 4 * lc4];                                     \
    float* bs_ = &Bs[buf][lrow * kLds + 4 * lc4];                                     \
    *reinterpret_cast<float4*>(as_) = ra0;                                            \
    *reinterpret_cast<float4*>(as_ + 32 * kLds) = ra1;                                \
    *reinterpret_cast<float4*>(as_ + 64 * kLds) = ra2;                                \
    *reinterpret_cast<float4*>(as_ + 96 * kLds) = ra3;                                \
    *reinterpret_cast<float4*>(bs_) = rb0;                                            \
    *reinterpret_cast<float4*>(bs_ + 32 * kLds) = rb1;                                \
    *reinterpret_cast<float4*>(bs_ + 64 * kLds) = rb2;                                \
    *reinterpret_cast<float4*>(bs_ + 96 * kLds) = rb3;                                \
  } while (0)

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }

  const int nk = K / kTileK;
  PFHIP_GLOAD(0);
  PFHIP_SSTORE(0);
  __syncthreads();

  const int a_off = (wr * 64 + r) * kLds + 4 * h;
  const int b_off = (wc * 64 + r) * kLds + 4 * h;

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // unconditional (the last iteration re-fetches its own tile): keeps the loop body straight-line so
    // the loads stay in flight under the MFMAs instead of being waited for at once
    const int knext = (kt + 1 < nk ? kt + 1 : kt) * kTileK;
    PFHIP_GLOAD(knext);
    __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise sinks the loads to just above their ds_write
    const float* as = &As[cur][a_off];
    const float* bs = &Bs[cur][b_off];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 a0 = *reinterpret_cast<const float4*>(as + kb * 8);
      const float4 a1 = *reinterpret_cast<const float4*>(as + 32 * kLds + kb * 8);
      const float4 b0 = *reinterpret_cast<const float4*>(bs + kb * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(bs + 32 * kLds + kb * 8);
#define PFHIP_MFMA4(c)                                                              \
  acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b0.c, acc00, 0, 0, 0);         \
  acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b1.c, acc01, 0, 0, 0);         \
  acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b0.c, acc10, 0, 0, 0);         \
  acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b1.c, acc11, 0, 0, 0);
      PFHIP_MFMA4(x) PFHIP_MFMA4(y) PFHIP_MFMA4(z) PFHIP_MFMA4(w)
#undef PFHIP_MFMA4
    }
    __builtin_amdgcn_sched_barrier(0);
    PFHIP_SSTORE(cur ^ 1);
    __syncthreads();
  }
#undef PFHIP_GLOAD
#undef PFHIP_SSTORE

  // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  // Unguarded form: C / R1 / R2 have ceil(M/128)*128 rows and ceil(N/128)*128 columns allocated and
  // bias is readable up to the padded N (pad outputs are junk nobody reads) -> no per-element branches.
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wc * 64 + j * 32 + r;
      const bool col_ok = !GUARD || col < N;
      float bv = 0.f;
      if (HAS_BIAS && col_ok) bv = bias[col];
      const int rbase = m0 + wr * 64 + i * 32 + 4 * h;
      const f32x16& accv = (i == 0) ? (j == 0 ? acc00 : acc01) : (j == 0 ? acc10 : acc11);
      float r1v[16], r2v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        const bool ok = !GUARD || (col_ok && row < M);
        const int rr = (GUARD && !ok) ? 0 : row;
        const int cc = (GUARD && !ok) ? 0 : col;
        r1v[e] = HAS_R1 ? R1[(size_t)rr * ldr1 + cc] : 0.f;
        r2v[e] = HAS_R2 ? R2[(size_t)rr * ldr2 + cc] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        float v = accv[e] + bv;
        if (HAS_R1) v += r1v[e];
        if (HAS_R2) v += r2v[e];
        if (RELU) v = fmaxf(v, 0.f);
        if (!GUARD || (col_ok && row < M)) C[(size_t)row * ldc + col] = v;
      }
    }
  }
}

}  // namespace

template <bool GUARD>
static void launch_variant(const float* A, int lda, const float* W, int ldw, float* C, int ldc,
                           const float* bias, const float* R1, int ldr1, const float* R2, int ldr2,
                           int M, int N, int K, bool relu, hipStream_t s) {
  const int tiles_m = (M + kTileM - 1) / kTileM;
  const int tiles_n = (N + kTileN - 1) / kTileN;
  const int n_tiles = tiles_m * tiles_n;
  const dim3 grid(n_tiles), block(256);
#define PFHIP_GEMM(B_, R1_, R2_, RL_)                                                              \
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<GUARD, B_, R1_, R2_, RL_>), grid, block, 0, s, A, lda, W, \
                     ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles)
  const int key = (bias ? 8 : 0) | (R1 ? 4 : 0) | (R2 ? 2 : 0) | (relu ? 1 : 0);
  switch (key) {
    case 0: PFHIP_GEMM(false, false, false, false); break;   // decoder ffn2 (no bias)
    case 1: PFHIP_GEMM(false, false, false, true); break;
    case 2: PFHIP_GEMM(false, false, true, false); break;
    case 3: PFHIP_GEMM(false, false, true, true); break;
    case 4: PFHIP_GEMM(false, true, false, false); break;
    case 5: PFHIP_GEMM(false, true, false, true); break;
    case 6: PFHIP_GEMM(false, true, true, false); break;
    case 7: PFHIP_GEMM(false, true, true, true); break;
    case 8: PFHIP_GEMM(true, false, false, false); break;    // qkv, q, kv, vocab
    case 9: PFHIP_GEMM(true, false, false, true); break;     // ffn1, predictor conv
    case 10: PFHIP_GEMM(true, false, true, false); break;
    case 11: PFHIP_GEMM(true, false, true, true); break;
    case 12: PFHIP_GEMM(true, true, false, false); break;    // ffn2 + residual, first-layer out-proj
    case 13: PFHIP_GEMM(true, true, false, true); break;     // predictor conv + residual
    case 14: PFHIP_GEMM(true, true, true, false); break;     // out-proj + fsmn memory + residual
    default: PFHIP_GEMM(true, true, true, true); break;
  }
#undef PFHIP_GEMM
}

void launch_gemm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc,
                     const float* bias, const float* R1, int ldr1, const float* R2, int ldr2, int M,
                     int N, int K, bool relu, bool guard, hipStream_t s) {
  if (M <= 0 || N <= 0) return;
  if (guard) launch_variant<true>(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, s);
  else launch_variant<false>(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, s);
}

}  // namespace pfhip
