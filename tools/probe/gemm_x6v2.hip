// fp32 GEMM on the gfx950 BF16 matrix cores, second form: the same exact three-way operand split as gemm_x6.hip
// (a = a1 + a2 + a3 in bf16, six plane products per block, fp32 accumulation), on `v_mfma_f32_16x16x32_bf16`.
//
// Why another form.  The 32x32x16 loop of gemm_x6.hip runs at the clock the chip holds under it (1.7-1.9 GHz): the K = 2048
// launches sit at the practical ceiling of that instruction (≈1.2 PF of executed bf16 MFMA), and the K = 512 launches lose
// another 15-30 % to per-tile prologue / epilogue time that nothing overlaps.  MI355X_MICROARCH.md (DVFS give-back, item 7)
// measures the 16x16x32 form at equal cycles per FLOP and 1.12-1.15 x the FLOP/s, because the chip holds a higher clock
// under it.  This kernel is built around that instruction:
//   * K-concatenated operands: a 16x16x32 MFMA sums over 32 k, the LDS stage holds 16 k of each plane — so the two halves of
//     the MFMA's k range read TWO DIFFERENT PLANES (lanes 0-31: k-groups 0,1 of the first, lanes 32-63: k-groups 0,1 of the
//     second), and one instruction adds two of the six plane products:
//         [a1|a2].[b1|b1] = a1b1 + a2b1      [a1|a2].[b2|b2] = a1b2 + a2b2      [a1|a3].[b3|b1] = a1b3 + a3b1
//     3 MFMAs of 16 cycles per 16x16x16 block = the cycles of the 32x32x16 form (6 x 32 per 32x32x16).
//   * Unpadded 32-byte LDS rows: with this fragment shape (lane = 16 rows x 4 k-groups) a row stride of 32 B is conflict-free
//     for ds_read_b128 (the four 16-lane service groups each cover rows {0-3,12-15} at one k-group and rows {4-11} at the
//     other: 16 distinct 16-B slots of 256 B), so a 128 x 128 tile stage is 24 KB instead of 36 KB.
//   * 128 x 128 tile, FOUR waves (2 x 2, each 64 x 64 = 4 x 4 MFMA tiles), two or three blocks per CU: each SIMD hosts waves
//     of DIFFERENT blocks, which drift out of phase — one block's prologue / epilogue runs under the other's MFMAs.
// Staging, the split, the pipeline (global loads two K-steps ahead, one barrier per K-step) and the epilogue (accumulators
// transposed through LDS, 512-byte row segments) follow gemm_x6.hip.
#include "kernels.h"

#include <algorithm>
#include <atomic>

namespace pfhip {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kT = 128;                    // tile rows = tile columns
constexpr int kBK = 16;                    // k per LDS stage
constexpr int kRowB = 32;                  // bytes per operand row in LDS: 16 bf16, no padding
constexpr int kPlane = kT * kRowB;         // 4,096 B
constexpr int kStageB = 6 * kPlane;        // 24,576 B: A1 A2 A3 W1 W2 W3
constexpr int kStages = 2;
constexpr int kCs = kT + 4;                // padded C-tile row stride (floats)
constexpr int kLdsBytes = kStages * kStageB;      // 49,152 B
static_assert(64 * kCs * 4 <= kLdsBytes, "half a C tile must fit the operand buffers");

__device__ __forceinline__ unsigned top16_pair(float lo, float hi) {      // (bf16 trunc of hi) << 16 | (bf16 trunc of lo)
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float rest(float x) {                           // x - top16(x), exact
  return x - __uint_as_float(__float_as_uint(x) & 0xFFFF0000u);
}

// XCD-aware, column-group-major tile order (gemm_x6.hip)
__device__ __forceinline__ void tile_of_block(int bid, int n_tiles, int tiles_n, int gw, int& tm, int& tn) {
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

__global__ __launch_bounds__(256, 2) void gemm_f32_bf16x6_v2_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C, int ldc,
    const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int tiles_n,
    int n_tiles, int gw, int relu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int tm, tn;
  tile_of_block(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kT, n0 = tn * kT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // staging: thread t holds 4 consecutive k (one 16-B quarter of a row's 64-B K-step) of A rows t/4, t/4 + 64 and of W rows
  // t/4, t/4 + 64
  const int srow = tid >> 2, sq = tid & 3;
  const float* Ag0 = A + (size_t)min(m0 + srow, M - 1) * lda + 4 * sq;
  const float* Ag1 = A + (size_t)min(m0 + srow + 64, M - 1) * lda + 4 * sq;
  const float* Wg0 = W + (size_t)min(n0 + srow, N - 1) * ldw + 4 * sq;
  const float* Wg1 = W + (size_t)min(n0 + srow + 64, N - 1) * ldw + 4 * sq;
  const int st_off = srow * kRowB + 8 * sq;

  // fragment addresses: lane = (row l16 of a 16-row tile, k-group g); g & 1 picks the 16-B half of the row, g >> 1 the plane
  const int l16 = lane & 15, g = lane >> 4, hi = g >> 1;
  const int fr_a = (wr * 64 + l16) * kRowB + (g & 1) * 16;
  const int fr_w = (wc * 64 + l16) * kRowB + (g & 1) * 16;
  const int ax_off = (hi ? 1 : 0) * kPlane + fr_a;          // [a1|a2]
  const int ay_off = (hi ? 2 : 0) * kPlane + fr_a;          // [a1|a3]
  const int bp_off = 3 * kPlane + fr_w;                     // [b1|b1]
  const int bq_off = 4 * kPlane + fr_w;                     // [b2|b2]
  const int br_off = (hi ? 3 : 5) * kPlane + fr_w;          // [b3|b1]

  float4 xa0, xa1, xw0, xw1, ya0, ya1, yw0, yw1;
#define PFHIP_LOAD_RAW(RA0, RA1, RW0, RW1, k0)                 \
  RA0 = *reinterpret_cast<const float4*>(Ag0 + (k0));          \
  RA1 = *reinterpret_cast<const float4*>(Ag1 + (k0));          \
  RW0 = *reinterpret_cast<const float4*>(Wg0 + (k0));          \
  RW1 = *reinterpret_cast<const float4*>(Wg1 + (k0));
  auto split3 = [&](const float4& v, unsigned char* base) {
    uint2 p;
    p.x = top16_pair(v.x, v.y); p.y = top16_pair(v.z, v.w);
    *reinterpret_cast<uint2*>(base) = p;
    float4 s = make_float4(rest(v.x), rest(v.y), rest(v.z), rest(v.w));
    p.x = top16_pair(s.x, s.y); p.y = top16_pair(s.z, s.w);
    *reinterpret_cast<uint2*>(base + kPlane) = p;
    s = make_float4(rest(s.x), rest(s.y), rest(s.z), rest(s.w));
    p.x = top16_pair(s.x, s.y); p.y = top16_pair(s.z, s.w);
    *reinterpret_cast<uint2*>(base + 2 * kPlane) = p;
  };
#define PFHIP_SPLIT_STORE(RA0, RA1, RW0, RW1, stage)                              \
  split3(RA0, lds + (stage) * kStageB + st_off);                                  \
  split3(RA1, lds + (stage) * kStageB + st_off + 64 * kRowB);                     \
  split3(RW0, lds + (stage) * kStageB + 3 * kPlane + st_off);                     \
  split3(RW1, lds + (stage) * kStageB + 3 * kPlane + st_off + 64 * kRowB);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define PFHIP_FRAG(off) __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + (off)))
  // one K-step from the stage at `sb`: 8 A fragments stay in registers, the 12 W fragments pass through three at a time
#define PFHIP_COMPUTE(stage)                                                                              \
  {                                                                                                       \
    const unsigned char* sb = lds + (stage) * kStageB;                                                    \
    bf16x8 ax[4], ay[4];                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
      ax[i] = PFHIP_FRAG(ax_off + i * 16 * kRowB);                                                        \
      ay[i] = PFHIP_FRAG(ay_off + i * 16 * kRowB);                                                        \
    }                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                       \
      const bf16x8 bp = PFHIP_FRAG(bp_off + j * 16 * kRowB);                                              \
      const bf16x8 bq = PFHIP_FRAG(bq_off + j * 16 * kRowB);                                              \
      const bf16x8 br = PFHIP_FRAG(br_off + j * 16 * kRowB);                                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[i], bq, acc[i][j], 0, 0, 0);               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ay[i], br, acc[i][j], 0, 0, 0);               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[i], bp, acc[i][j], 0, 0, 0);               \
      }                                                                                                   \
    }                                                                                                     \
  }
  // One K-step, one barrier: split + store the next step's operands into the other stage, issue the loads of the step after,
  // then this step's fragments and MFMAs.  The barrier is `s_waitcnt lgkmcnt(0); s_barrier` by hand: __syncthreads() would
  // also drain the global loads just issued.
#define PFHIP_STEP(RA0, RA1, RW0, RW1, cur, knext)                                                        \
  PFHIP_SPLIT_STORE(RA0, RA1, RW0, RW1, (cur) ^ 1)                                                        \
  PFHIP_LOAD_RAW(RA0, RA1, RW0, RW1, knext)                                                               \
  PFHIP_COMPUTE(cur)                                                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  const int nk = K / kBK;
  auto kclamp = [&](int t) { return (t < nk ? t : nk - 1) * kBK; };
  PFHIP_LOAD_RAW(xa0, xa1, xw0, xw1, 0)
  PFHIP_SPLIT_STORE(xa0, xa1, xw0, xw1, 0)
  PFHIP_LOAD_RAW(ya0, ya1, yw0, yw1, kclamp(1))              // K-step 1 -> y, K-step 2 -> x: step kt splits set (kt + 1) & 1
  PFHIP_LOAD_RAW(xa0, xa1, xw0, xw1, kclamp(2))
  __syncthreads();

  // splits / loads past the last K-step redo the last one (never used): keeps the bodies straight-line
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    { const int knext = kclamp(kt + 3); PFHIP_STEP(ya0, ya1, yw0, yw1, 0, knext) }
    { const int knext = kclamp(kt + 4); PFHIP_STEP(xa0, xa1, xw0, xw1, 1, knext) }
  }
  if (kt < nk) { const int knext = kclamp(nk); PFHIP_STEP(ya0, ya1, yw0, yw1, 0, knext) }
#undef PFHIP_STEP
#undef PFHIP_COMPUTE
#undef PFHIP_FRAG
#undef PFHIP_SPLIT_STORE
#undef PFHIP_LOAD_RAW
  __syncthreads();

  // ---- epilogue: two 64-row halves through LDS.  C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + e ------
  float* const Cs = reinterpret_cast<float*>(lds);
  const int c4 = tid & 31, rsub = tid >> 5;
  const int gcol = n0 + 4 * c4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) {
    if (gcol + 3 < N) bv = *reinterpret_cast<const float4*>(bias + gcol);
    else {
      if (gcol < N) bv.x = bias[gcol];
      if (gcol + 1 < N) bv.y = bias[gcol + 1];
      if (gcol + 2 < N) bv.z = bias[gcol + 2];
    }
  }
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (wr == half) {
      float* cw = Cs + (4 * g) * kCs + wc * 64 + l16;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) cw[(16 * i + e) * kCs + 16 * j] = acc[i][j][e];
    }
    __syncthreads();
#pragma unroll 2
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 8 + rsub;
      const int grow = m0 + half * 64 + row;
      float4 v = *reinterpret_cast<const float4*>(Cs + row * kCs + 4 * c4);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      if (grow < M && gcol + 3 < N) {
        if (R1) {
          const float4 t = *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol);
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (R2) {
          const float4 t = *reinterpret_cast<const float4*>(R2 + (size_t)grow * ldr2 + gcol);
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(C + (size_t)grow * ldc + gcol) = v;
      } else if (grow < M && gcol < N) {      // ragged right edge: element-wise
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int q = 0; q < 4 && gcol + q < N; ++q) {
          float o = vv[q];
          if (R1) o += R1[(size_t)grow * ldr1 + gcol + q];
          if (R2) o += R2[(size_t)grow * ldr2 + gcol + q];
          if (relu) o = fmaxf(o, 0.f);
          C[(size_t)grow * ldc + gcol + q] = o;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace

void launch_gemm_f32_bf16x6_v2(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                               int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, int gw, hipStream_t s) {
  if (M <= 0 || N <= 0) return;
  const int tiles_m = (M + kT - 1) / kT, tiles_n = (N + kT - 1) / kT, n_tiles = tiles_m * tiles_n;
  gw = std::max(1, std::min(gw, tiles_n));
  hipLaunchKernelGGL(gemm_f32_bf16x6_v2_kernel, dim3(n_tiles), dim3(256), kLdsBytes, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2,
                     ldr2, M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0);
}

}  // namespace pfhip
