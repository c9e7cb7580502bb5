// PROBE (not built into libpfhip.so): second attempt at overlapping the epilogue — the 256 x 128 kernel as a continuous K-stream per CU
// with the finished tile parked in registers and written out element by element under the next tile's MFMAs.  Correct (GEMM parity
// tests as kind 6), but 114-126 TF on the K = 512 shapes against 150-165 TF of the same loop without it, and 184 vs 203 TF at K = 2048:
// the scalar residual loads / stores, their address arithmetic and the cursor logic inside the step cost more than the overlap returns.
// STREAMING variant of the BF16-split GEMM (gemm_x6.hip, 256 x 128 tile, same loop): one workgroup per CU walks its tiles as
// ONE continuous K-stream (loads, splits and fragment reads of the next tile's first K-steps are in flight while the current
// tile finishes) and a finished tile's accumulators are PARKED IN REGISTERS and written out straight from there — one C-layout
// element per accumulator per double K-step, residual loads issued one double-step ahead — so that prologue, epilogue and the
// workgroup hand-over of the one-tile-per-workgroup kernels overlap with MFMA work.  K >= 512 (16 double-steps drain the 64
// parked values per lane).
//
// (the scheme itself:) fp32 GEMM on the gfx950 BF16 matrix cores: C = A * W^T (+bias, +residuals, ReLU), with fp32-grade results.
//
// On CDNA4 the fp32 MFMA (`v_mfma_f32_32x32x2_f32`) runs at the vector rate, 157 TFLOP/s — 1/16 of the BF16 MFMA
// (MI355X_MICROARCH.md § Matrix cores).  An fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significand bits):
//     a = a1 + a2 + a3,  a1 = top 16 bits of a,  a2 = top 16 bits of (a - a1),  a3 = a - a1 - a2      (all subtractions exact)
// and a product of two bf16 numbers is exact in fp32, so
//     a*b = a1*b1 + (a1*b2 + a2*b1) + (a1*b3 + a3*b1 + a2*b2) + O(2^-24 |a*b|)
// costs six `v_mfma_f32_32x32x16_bf16` per 32x32x16 block instead of eight fp32 MFMAs of four times their duration: a ceiling of
// 2.5 PF / 6 = 417 TFLOP/s of fp32-equivalent work.  The three dropped terms are below fp32's own rounding of the product;
// accumulation is fp32 in the matrix core as before.  Measured against an fp64 reference the result is slightly CLOSER than
// the fp32 MFMA chain (rms 1.2e-7 vs 2.9e-7 relative at K = 512: the partial products carry no rounding of their own).
//
// Tiling: 256 x 128 block tile, 8 waves as 4 x 2 (two per SIMD), each 64 x 64 = 2 x 2 MFMA tiles.  K-step 16 = one MFMA depth.
// Operands are split while they are staged: global fp32 -> registers -> three bf16 planes in LDS (row stride 48 B: the 16-lane
// groups of ds_read_b128 hit 16 distinct 16-B slots), double-buffered, one barrier per K-step; the split of the NEXT step's
// operands (5.5 VALU ops per element) and its LDS writes sit between this step's MFMAs, and the second wave of the SIMD fills
// what is left.  Per K-step a wave issues 12 ds_read_b128 for 24 MFMAs.
// Epilogue as in gemm.hip: accumulators transposed through LDS (two 128-row halves), rows written with 16-byte accesses.
#include "kernels.h"

#include <algorithm>
#include <atomic>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kBM = 256, kBN = 128, kBK = 16;
constexpr int kRowB = 48;                                   // bytes per operand row in LDS (16 bf16 + pad)
constexpr int kPlaneA = kBM * kRowB, kPlaneW = kBN * kRowB; // bytes per plane
constexpr int kStageB = 3 * (kPlaneA + kPlaneW);            // 55,296 B
constexpr int kCs = kBN + 4;                                // padded C-tile row stride (floats)
constexpr int kLdsBytes = 2 * kStageB;                      // 110,592 B
static_assert(128 * kCs * 4 <= kLdsBytes, "half C tile must fit the operand buffers");

__device__ __forceinline__ unsigned top16_pair(float lo, float hi) {      // (bf16 trunc of hi) << 16 | (bf16 trunc of lo)
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float rest(float x) {                           // x - top16(x), exact
  return x - __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
}

// XCD-aware, column-group-major tile order (same scheme as gemm.hip's tile_of_block)
__device__ __forceinline__ void tile_of_block_x6(int bid, int n_tiles, int tiles_n, int gw, int& tm, int& tn) {
  {
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tiles_m = n_tiles / tiles_n, full = tiles_n / gw, span = tiles_m * gw;
  if (bid < full * span) {
    const int g = bid / span, j = bid - g * span;
    tm = j / gw; tn = g * gw + (j - tm * gw);
  } else {
    const int j = bid - full * span, w = tiles_n - full * gw;
    tm = j / w; tn = full * gw + (j - tm * w);
  }
}

__global__ __launch_bounds__(512, 1) void gemm_f32_bf16x6_stream_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C, int ldc,
    const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int tiles_n,
    int n_tiles, int gw, int relu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int nk = K / kBK;
  const int G = gridDim.x;
  const int my_tiles = (n_tiles - (int)blockIdx.x + G - 1) / G;

  const int srow = tid >> 2, sq = tid & 3;
  const int a_st = srow * kRowB + 8 * sq;
  const int w_st = 3 * kPlaneA + srow * kRowB + 8 * sq;
  const int a_fr = (wr * 64 + r) * kRowB + 16 * h;
  const int w_fr = 3 * kPlaneA + (wc * 64 + r) * kRowB + 16 * h;

  // load cursor: (tile of my sequence, K-step) of the next raw load; runs three K-steps ahead of the MFMAs
  int ld_ti = 0, ld_kt = 0;
  const float *Ag0, *Ag1, *Wg;
  auto set_tile_ptr = [&](int ti) {
    int tm, tn;
    tile_of_block_x6((int)blockIdx.x + min(ti, my_tiles - 1) * G, n_tiles, tiles_n, gw, tm, tn);
    const int m0 = tm * kBM, n0 = tn * kBN;
    Ag0 = A + (size_t)min(m0 + srow, M - 1) * lda + 4 * sq;
    Ag1 = A + (size_t)min(m0 + srow + 128, M - 1) * lda + 4 * sq;
    Wg = W + (size_t)min(n0 + srow, N - 1) * ldw + 4 * sq;
  };
  set_tile_ptr(0);
  float4 xa0, xa1, xw, ya0, ya1, yw;
#define PFHIP_LOAD_NEXT(RA0, RA1, RW)                                   \
  RA0 = *reinterpret_cast<const float4*>(Ag0 + ld_kt * kBK);            \
  RA1 = *reinterpret_cast<const float4*>(Ag1 + ld_kt * kBK);            \
  RW = *reinterpret_cast<const float4*>(Wg + ld_kt * kBK);              \
  if (++ld_kt == nk) { ld_kt = 0; set_tile_ptr(++ld_ti); }
  auto split3 = [&](const float4& v, unsigned char* base, int plane_bytes) {
    uint2 p;
    p.x = top16_pair(v.x, v.y); p.y = top16_pair(v.z, v.w);
    *reinterpret_cast<uint2*>(base) = p;
    float4 s = make_float4(rest(v.x), rest(v.y), rest(v.z), rest(v.w));
    p.x = top16_pair(s.x, s.y); p.y = top16_pair(s.z, s.w);
    *reinterpret_cast<uint2*>(base + plane_bytes) = p;
    s = make_float4(rest(s.x), rest(s.y), rest(s.z), rest(s.w));
    p.x = top16_pair(s.x, s.y); p.y = top16_pair(s.z, s.w);
    *reinterpret_cast<uint2*>(base + 2 * plane_bytes) = p;
  };
#define PFHIP_SPLIT_STORE(RA0, RA1, RW, stage)                                \
  split3(RA0, lds + (stage) * kStageB + a_st, kPlaneA);                       \
  split3(RA1, lds + (stage) * kStageB + a_st + 128 * kRowB, kPlaneA);         \
  split3(RW, lds + (stage) * kStageB + w_st, kPlaneW);

  f32x16 acc00, acc01, acc10, acc11, par00, par01, par10, par11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; par00[e] = 0.f; par01[e] = 0.f; par10[e] = 0.f; par11[e] = 0.f; }

  bf16x8 fa[3][2], fb[3][2], ga[3][2], gb[3][2];
#define PFHIP_FRAGS(FA, FB, stage)                                                                                  \
  _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                 \
      FA[p][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (stage) * kStageB + p * kPlaneA + a_fr + i * 32 * kRowB)); \
      FB[p][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (stage) * kStageB + p * kPlaneW + w_fr + i * 32 * kRowB)); \
    }                                                                                                               \
  }
#define PFHIP_X6(FA, FB, pa, pb)                                                              \
  acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[pa][0], FB[pb][0], acc00, 0, 0, 0);      \
  acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[pa][0], FB[pb][1], acc01, 0, 0, 0);      \
  acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[pa][1], FB[pb][0], acc10, 0, 0, 0);      \
  acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[pa][1], FB[pb][1], acc11, 0, 0, 0);
#define PFHIP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define PFHIP_STEP(FA, FB, GA, GB, RA0, RA1, RW, nxt)                                         \
  PFHIP_SPLIT_STORE(RA0, RA1, RW, nxt)                                                        \
  PFHIP_LOAD_NEXT(RA0, RA1, RW)                                                               \
  PFHIP_X6(FA, FB, 1, 1) PFHIP_X6(FA, FB, 0, 2) PFHIP_X6(FA, FB, 2, 0) PFHIP_X6(FA, FB, 0, 1)  \
  _Pragma("unroll") for (int q = 0; q < 9; ++q) {                                             \
    PFHIP_SGB(0x8, 1); PFHIP_SGB(0x2, 8); PFHIP_SGB(0x200, 1);                                \
  }                                                                                           \
  PFHIP_SGB(0x8, 1); PFHIP_SGB(0x20, 1); PFHIP_SGB(0x8, 1); PFHIP_SGB(0x20, 1);               \
  PFHIP_SGB(0x8, 1); PFHIP_SGB(0x20, 1); PFHIP_SGB(0x8, 4);                                   \
  __builtin_amdgcn_sched_barrier(0);                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                              \
  __builtin_amdgcn_sched_barrier(0);                                                          \
  PFHIP_FRAGS(GA, GB, nxt)                                                                    \
  PFHIP_X6(FA, FB, 1, 0) PFHIP_X6(FA, FB, 0, 0)                                               \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
    PFHIP_SGB(0x8, 1); PFHIP_SGB(0x100, 2); PFHIP_SGB(0x8, 1); PFHIP_SGB(0x100, 1);           \
  }                                                                                           \
  __builtin_amdgcn_sched_barrier(0);

  // ---- the parked tile: element e of the four accumulators is row (e&3)+8(e>>2)+4h (+32) x column r (+32) of the wave's
  //      64 x 64 patch; drained one e per double K-step: residuals loaded before the two steps, value stored after them -----
  int pm0 = 0, pn0 = 0;                       // origin of the parked tile
  bool parked = false;
  float pb0 = 0.f, pb1 = 0.f;                 // bias of this lane's two columns
  float t00, t01, t10, t11;                   // residual sums of the element in flight
  auto elem_rows = [&](int e, int& ra, int& rb2) { ra = pm0 + wr * 64 + (e & 3) + 8 * (e >> 2) + 4 * h; rb2 = ra + 32; };
  auto drain_load = [&](int e) {
    int ra, rb2;
    elem_rows(e, ra, rb2);
    const int c0 = pn0 + wc * 64 + r, c1 = c0 + 32;
    t00 = t01 = t10 = t11 = 0.f;
    if (R1) {
      if (ra < M && c0 < N) t00 += R1[(size_t)ra * ldr1 + c0];
      if (ra < M && c1 < N) t01 += R1[(size_t)ra * ldr1 + c1];
      if (rb2 < M && c0 < N) t10 += R1[(size_t)rb2 * ldr1 + c0];
      if (rb2 < M && c1 < N) t11 += R1[(size_t)rb2 * ldr1 + c1];
    }
    if (R2) {
      if (ra < M && c0 < N) t00 += R2[(size_t)ra * ldr2 + c0];
      if (ra < M && c1 < N) t01 += R2[(size_t)ra * ldr2 + c1];
      if (rb2 < M && c0 < N) t10 += R2[(size_t)rb2 * ldr2 + c0];
      if (rb2 < M && c1 < N) t11 += R2[(size_t)rb2 * ldr2 + c1];
    }
  };
  auto drain_store = [&](int e, float v00, float v01, float v10, float v11) {
    int ra, rb2;
    elem_rows(e, ra, rb2);
    const int c0 = pn0 + wc * 64 + r, c1 = c0 + 32;
    v00 += pb0 + t00; v01 += pb1 + t01; v10 += pb0 + t10; v11 += pb1 + t11;
    if (relu) { v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f); }
    if (ra < M && c0 < N) C[(size_t)ra * ldc + c0] = v00;
    if (ra < M && c1 < N) C[(size_t)ra * ldc + c1] = v01;
    if (rb2 < M && c0 < N) C[(size_t)rb2 * ldc + c0] = v10;
    if (rb2 < M && c1 < N) C[(size_t)rb2 * ldc + c1] = v11;
  };
#define PFHIP_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define PFHIP_STORE_CASE(e) case e: drain_store(e, par00[e], par01[e], par10[e], par11[e]); break;
  auto park = [&](int ti) {
    int tm, tn;
    tile_of_block_x6((int)blockIdx.x + ti * G, n_tiles, tiles_n, gw, tm, tn);
    pm0 = tm * kBM; pn0 = tn * kBN;
    const int c0 = pn0 + wc * 64 + r;
    pb0 = (bias && c0 < N) ? bias[c0] : 0.f;
    pb1 = (bias && c0 + 32 < N) ? bias[c0 + 32] : 0.f;
    par00 = acc00; par01 = acc01; par10 = acc10; par11 = acc11;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
    parked = true;
  };

  // pipeline fill
  PFHIP_LOAD_NEXT(xa0, xa1, xw)
  PFHIP_SPLIT_STORE(xa0, xa1, xw, 0)
  PFHIP_LOAD_NEXT(ya0, ya1, yw)
  PFHIP_LOAD_NEXT(xa0, xa1, xw)
  __syncthreads();
  PFHIP_FRAGS(fa, fb, 0)

  int cur_ti = 0, kt = 0;
  const int total = my_tiles * nk;            // nk is even: tile boundaries fall on even steps
  for (int g = 0; g < total; g += 2) {
    const int e = kt >> 1;                    // element of the parked tile handled in this double step (nk >= 32: all 16 get a turn)
    const bool drain = parked && e < 16;
    if (drain) drain_load(e);
    PFHIP_STEP(fa, fb, ga, gb, ya0, ya1, yw, 1)
    PFHIP_STEP(ga, gb, fa, fb, xa0, xa1, xw, 0)
    if (drain) {
      switch (e) { PFHIP_CASES(PFHIP_STORE_CASE) default: break; }
    }
    kt += 2;
    if (kt == nk) { park(cur_ti); kt = 0; ++cur_ti; }
  }
#undef PFHIP_STEP
#undef PFHIP_SPLIT_STORE
#undef PFHIP_LOAD_NEXT
#undef PFHIP_SGB
#undef PFHIP_X6
#undef PFHIP_FRAGS
  // the last tile was parked by the last double step: write it out
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    drain_load(e);
    drain_store(e, par00[e], par01[e], par10[e], par11[e]);
  }
#undef PFHIP_STORE_CASE
#undef PFHIP_CASES
}

}  // namespace

void launch_gemm_f32_bf16x6_stream(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                                   const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, int gw,
                                   hipStream_t s) {
  if (M <= 0 || N <= 0) return;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_done.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_bf16x6_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsBytes);
    attr_done.fetch_or(1ull << (dev & 63));
  }
  const int tiles_m = (M + kBM - 1) / kBM, tiles_n = (N + kBN - 1) / kBN, n_tiles = tiles_m * tiles_n;
  gw = std::max(1, std::min(gw, tiles_n));
  hipLaunchKernelGGL(gemm_f32_bf16x6_stream_kernel, dim3(std::min(n_tiles, 256)), dim3(512), kLdsBytes, s, A, lda, W, ldw, C, ldc, bias,
                     R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0);
}

}  // namespace pfhip
