// fp32 GEMM on the gfx950 FP16 matrix cores with THREE products per block: C = A * W^T (+bias, +residuals, ReLU), fp32-grade.
//
// gemm_x6.hip splits an fp32 operand EXACTLY into three bf16 planes (8 + 8 + 8 significand bits) and needs six bf16 MFMAs per
// 32x32x16 block (ceiling 2.5 PF / 6 = 417 TFLOP/s).  fp16 has 11 significand bits, so TWO planes carry 22-23 of fp32's 24:
//     x * S = x1 + x2 + e,   x1 = fp16_rtz(x * S),   x2 = fp16_rtz(x * S - x1)   (the subtraction is exact in fp32),
//     |e| <= max(2^-22 |x1|, 2^-24)     (S a power of two; the second bound is fp16's subnormal spacing — the matrix cores keep
//                                        subnormal fp16 operands, tools/probe/f16_split_probe.hip)
// and a * b = a1 b1 + (a1 b2 + a2 b1) + [a2 b2 ~ 2^-22 |ab|, dropped]: THREE `v_mfma_f32_32x32x16_f16` per block, fp32
// accumulation in the matrix core, a ceiling of 2.5 PF / 3 = 833 TFLOP/s of fp32-equivalent work.  The representation error of
// the operands (rms ~2^-24 relative, unbiased to first order across a row) is BELOW the rounding an fp32 accumulation chain of
// the same length commits: against an fp64 reference the result is as close as the fp32 MFMA chain's
// (tests/test_gpu_ops.py::test_gemm_f16_split_is_fp32_grade) — unlike the usual "bf16x3" (three largest bf16 products, 16 bits:
// 360 x worse), which moved the model's log-probs by 3.5e-3 and was rejected in round 2.
//
// Scaling.  fp16's exponent range is what this form pays with.  Weights are staged as w * S_w (a power of two: exact) and the
// accumulator is multiplied by 1 / S_w in the epilogue; activations are staged as they are (S_a = 1):
//   * overflow: an operand with |a| >= 65504 (or +-Inf) saturates the high plane and overflows the low one: every output
//     of its row is Inf / NaN — a loud failure, as the bf16 form has for Inf.  S_w is chosen per weight matrix at load time
//     from its largest magnitude (kernels.h best_w_scale: max |w| S_w <= 32768, never overflows); S_a = 1: |a| < 65504.
//     Inputs beyond that belong on the bf16 form (PFHIP_GEMM_X3=0), whose range is fp32's;
//   * small values: below 2^-3 / S the low plane is subnormal and the ABSOLUTE error per element is 2^-25 / S.  For the weights
//     S_w makes that irrelevant (error / norm as the bf16 form: tools/gemm_x6_probe.py).  For activations (S_a = 1) it is
//     3e-8 per element: a row's error is bounded as that of an O(1) row — relative to ITS norm only when the norm is >= ~1,
//     which rows of 512-2048 LayerNorm-ed, residual or ReLU values are (22 for a LayerNorm output).
// NaN stays NaN.
//
// Tiling, staging, pipelining and epilogues are those of gemm_x6.hip (three tile shapes: 256 x 128, 128 x 128, 64 x 128; K-step 16;
// operands split while staged, LDS double-buffered, one hand-made barrier per K-step; LayerNorm folded in through the row
// statistics hand-off) with two planes instead of three: per K-step and wave 12 / 6 / 3 MFMAs, 8 / 6 / 4 fragment reads.
#include "kernels.h"

// Timing-only ablations of the K-step (tools/x3_ablate.sh builds one library per value; results are WRONG for any value but 0):
//   1 no s_barrier   2 no split / LDS writes   3 no fragment reads   4 no global loads   5 MFMAs only (2 + 3 + 4, barrier kept)
//   6 staging as a plain copy (one 16-byte LDS write per float4, no split arithmetic): the cost model of operands that arrive
//     pre-split from their producers
#ifndef PFHIP_X3_ABLATE
#define PFHIP_X3_ABLATE 0
#endif
#if PFHIP_X3_ABLATE == 1
#define PFHIP_X3_BARRIER "s_waitcnt lgkmcnt(0)"
#else
#define PFHIP_X3_BARRIER "s_waitcnt lgkmcnt(0)\n\ts_barrier"
#endif
#define PFHIP_X3_DO_SPLIT (PFHIP_X3_ABLATE != 2 && PFHIP_X3_ABLATE != 5)
#define PFHIP_X3_DO_READ (PFHIP_X3_ABLATE != 3 && PFHIP_X3_ABLATE != 5)
#define PFHIP_X3_DO_LOAD (PFHIP_X3_ABLATE != 4 && PFHIP_X3_ABLATE != 5)

#include <algorithm>
#include <atomic>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using float2v = __attribute__((ext_vector_type(2))) float;

constexpr int kBM = 256, kBN = 128, kBK = 16;
constexpr int kRowB = 48;                                   // bytes per operand row in LDS (16 bf16 + pad)
constexpr int kPlaneA = kBM * kRowB, kPlaneW = kBN * kRowB; // bytes per plane
constexpr int kStageB = 2 * (kPlaneA + kPlaneW);            // 36,864 B
constexpr int kCs = kBN + 4;                                // padded C-tile row stride (floats)
constexpr int kRing = 3;                                    // LDS stages: the one being read, the next one (read ahead), the one being written
constexpr int kLdsBytes = kRing * kStageB;                  // 110,592 B
static_assert(128 * kCs * 4 <= kLdsBytes, "half C tile must fit the operand buffers");

// x - (float)h for the low / high half of a packed fp16 pair, ONE instruction each (v_fma_mix_f32 reads an fp16 source in
// place: fma(h, -1.0, x)); exact, because h has at most 11 of x's 24 significant bits and the same exponent or the one below
__device__ __forceinline__ float sub_lo(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float sub_hi(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
// two fp16 planes of v (* scale) for four consecutive k: hi = rtz(x), lo = rn(x - hi); the round-toward-zero conversion of hi
// saturates instead of producing Inf.  8 vector instructions per float4 (12 scaled) against 22 for the three bf16 planes of gemm_x6.hip.
template <bool SC>
__device__ __forceinline__ void split2(const float4& v, float scale, unsigned char* base, int plane_bytes) {
  const float x0 = SC ? v.x * scale : v.x, x1 = SC ? v.y * scale : v.y, x2 = SC ? v.z * scale : v.z, x3 = SC ? v.w * scale : v.w;
  const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x2, x3));
  *reinterpret_cast<uint2*>(base) = make_uint2(h01, h23);
  // the low plane rounds to nearest (v_cvt_pk_f16_f32, one instruction per pair on gfx950): half the error of truncation
  // and no bias; it cannot overflow (|x - hi| < 2^-10 |hi|)
  const float2v r01 = {sub_lo(x0, h01), sub_hi(x1, h01)}, r23 = {sub_lo(x2, h23), sub_hi(x3, h23)};
  const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, half2v));
  const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, half2v));
  *reinterpret_cast<uint2*>(base + plane_bytes) = make_uint2(l01, l23);
}

// The same split in two pieces that can be placed between MFMAs by hand: HI writes the high plane of a float4 and keeps the four
// residuals (and, scaled, the products), LO rounds and writes the low plane.
struct SplitTmp { float r0, r1, r2, r3; };
template <bool SC>
__device__ __forceinline__ void split_hi(const float4& v, float scale, unsigned char* dst, SplitTmp& t) {
#if PFHIP_X3_ABLATE == 6
  *reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(reinterpret_cast<uintptr_t>(dst) & ~(uintptr_t)15)) = v;
  t.r0 = t.r1 = t.r2 = t.r3 = 0.f;
  return;
#endif
  const float x0 = SC ? v.x * scale : v.x, x1 = SC ? v.y * scale : v.y, x2 = SC ? v.z * scale : v.z, x3 = SC ? v.w * scale : v.w;
  const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x2, x3));
  *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
  t.r0 = sub_lo(x0, h01); t.r1 = sub_hi(x1, h01); t.r2 = sub_lo(x2, h23); t.r3 = sub_hi(x3, h23);
}
__device__ __forceinline__ void split_lo(unsigned char* dst, const SplitTmp& t) {
#if PFHIP_X3_ABLATE == 6
  return;
#endif
  const float2v r01 = {t.r0, t.r1}, r23 = {t.r2, t.r3};
  *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(r01, half2v)),
                                              __builtin_bit_cast(unsigned, __builtin_convertvector(r23, half2v)));
}

// XCD-aware, column-group-major tile order (same scheme as gemm.hip's tile_of_block)
__device__ __forceinline__ void tile_of_block_x3(int bid, int n_tiles, int tiles_n, int gw, int& tm, int& tn) {
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

// LayerNorm statistics of the rows this tile just finished, for the GEMM that consumes them (LN-on-load below): the 32 lanes
// that hold one row's 128 columns reduce (mean of the tile's columns, M2 = sum of squared deviations from THAT mean) and lane 0
// writes the pair to stats[row][tile column][2].  The consumer merges the tiles_n pairs of a row with Chan's formula — as
// accurate as a two-pass LayerNorm, no atomics, no ordering between tiles.
// sum over the 32 lanes of a half wave, result in every lane: four DPP steps inside the 16-lane rows (quad swaps, half-row
// mirror, row mirror — vector-ALU speed) and ONE cross-row shuffle; five ds_bpermute round trips per sum cost the epilogue
// ~2 us per tile
__device__ __forceinline__ float half_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  v += __shfl_xor(v, 16);
  return v;
}
__device__ __forceinline__ void tile_row_stats(const float4& v, int grow, int M, int tn, int tiles_n, int c4, float* __restrict__ stats) {
  const float sum = half_wave_sum((v.x + v.y) + (v.z + v.w));
  const float mean = sum * (1.0f / kBN);
  const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
  const float q = half_wave_sum((a * a + b * b) + (c * c + d * d));
  if (c4 == 0 && grow < M) *reinterpret_cast<float2*>(stats + ((size_t)grow * tiles_n + tn) * 2) = make_float2(mean, q);
}

// the consumer's half.  Row statistics merged from the producer's per-tile pairs (Chan: n = 128 per tile) — one thread per row,
// at kernel start, parked in registers under the K-loop and published through LDS for the epilogue passes ...
// A row whose rms lies outside [2^-8, 2^12] (or is not finite) is outside the domain the two fp16 planes of the raw residual
// stream cover at fp32 grade: the forward's range flag is raised (kernels.h LaunchCtx) and the host redoes the batch on the
// bf16 three-plane kernels.
__device__ __forceinline__ float2 ln_row_stats(const float* __restrict__ stats, int tiles, float eps, int row, int* range_flag) {
  const float* sp = stats + (size_t)row * tiles * 2;
  float msum = 0.f, m2 = 0.f;
  for (int t = 0; t < tiles; ++t) msum += sp[2 * t];
  const float mean = msum / (float)tiles;
  for (int t = 0; t < tiles; ++t) { const float dm = sp[2 * t] - mean; m2 += sp[2 * t + 1] + (float)kBN * dm * dm; }
  const float rstd = 1.0f / sqrtf(m2 / (float)(tiles * kBN) + eps);
  if (range_flag && !(rstd > kLnRstdMin && rstd < kLnRstdMax)) atomicOr(range_flag, 2);
  return make_float2(mean, rstd);
}
// ... where v (four columns of x W'^T) becomes rstd * (v - mean * colsum)
__device__ __forceinline__ void ln_finish(float4& v, const float4& cs, float2 mr) {
  v.x = mr.y * (v.x - mr.x * cs.x); v.y = mr.y * (v.y - mr.x * cs.y);
  v.z = mr.y * (v.z - mr.x * cs.z); v.w = mr.y * (v.w - mr.x * cs.w);
}

template <bool SC>
__global__ __launch_bounds__(512, 1) void gemm_f32_f16x3_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C, int ldc,
    const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int tiles_n,
    int n_tiles, int gw, int relu, float* __restrict__ stats_out, float sw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int tm, tn;
  tile_of_block_x3(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kBM, n0 = tn * kBN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // staging map: thread t holds 4 consecutive k (one of the four 16-B pieces of a row's 64-B K-step) of A rows t/4 and
  // t/4 + 128 and of W row t/4: a wave's load instruction covers 16 rows x 64 contiguous bytes
  const int srow = tid >> 2, sq = tid & 3;
  const float* Ag0 = A + (size_t)min(m0 + srow, M - 1) * lda + 4 * sq;
  const float* Ag1 = A + (size_t)min(m0 + srow + 128, M - 1) * lda + 4 * sq;
  const float* Wg = W + (size_t)min(n0 + srow, N - 1) * ldw + 4 * sq;
  const int a_st = srow * kRowB + 8 * sq;                   // byte offset inside an A plane (second row: + 128 rows)
  const int w_st = 2 * kPlaneA + srow * kRowB + 8 * sq;     // byte offset of the W planes inside a stage
  const int a_fr = (wr * 64 + r) * kRowB + 16 * h;
  const int w_fr = 2 * kPlaneA + (wc * 64 + r) * kRowB + 16 * h;

  // two raw register sets: the global loads run two K-steps ahead of the split that consumes them
  float4 xa0, xa1, xw, ya0, ya1, yw;
#define PFHIP_LOAD_RAW(RA0, RA1, RW, k0)                       \
  RA0 = *reinterpret_cast<const float4*>(Ag0 + (k0));          \
  RA1 = *reinterpret_cast<const float4*>(Ag1 + (k0));          \
  RW = *reinterpret_cast<const float4*>(Wg + (k0));
#define PFHIP_SPLIT_STORE(RA0, RA1, RW, stage)                                \
  split2<false>(RA0, 1.0f, lds + (stage) * kStageB + a_st, kPlaneA);                   \
  split2<false>(RA1, 1.0f, lds + (stage) * kStageB + a_st + 128 * kRowB, kPlaneA);     \
  split2<SC>(RW, sw, lds + (stage) * kStageB + w_st, kPlaneW);

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }

  // operand fragments: even K-steps in f*, odd in g* — [plane][tile]
  half8 fa[2][2], fb[2][2], ga[2][2], gb[2][2];
#define PFHIP_FRAGS(FA, FB, stage)                                                                                  \
  _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                 \
      FA[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (stage) * kStageB + p * kPlaneA + a_fr + i * 32 * kRowB)); \
      FB[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (stage) * kStageB + p * kPlaneW + w_fr + i * 32 * kRowB)); \
    }                                                                                                               \
  }
#define PFHIP_X6(FA, FB, pa, pb)                                                              \
  acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa][0], FB[pb][0], acc00, 0, 0, 0);      \
  acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa][0], FB[pb][1], acc01, 0, 0, 0);      \
  acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa][1], FB[pb][0], acc10, 0, 0, 0);      \
  acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa][1], FB[pb][1], acc11, 0, 0, 0);
#define PFHIP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
  // One K-step, one barrier (`s_waitcnt lgkmcnt(0); s_barrier` by hand: __syncthreads() would also wait for vmcnt(0), i.e. for the
  // global loads issued a few instructions earlier).  The LDS stages form a RING OF THREE: step k multiplies the fragments it
  // already holds (stage k), reads the fragments of stage k + 1 — written during step k - 1, visible since the barrier that
  // ended it — and writes the split of K-step k + 2 into stage k + 2 (last read during step k - 2).  So neither the fragment reads
  // nor the split's writes sit between a barrier and the MFMAs that need them: each has the whole step's 12 MFMAs as cover.
  // (With two stages the eight reads of all eight waves — 64 KB — landed in a burst right behind the barrier with four MFMAs
  // to hide them.)
  // The step is placed BY HAND, every statement pinned with a scheduling barrier: left to the compiler — sched_group_barrier
  // patterns included — the 30-odd vector instructions of the split all ran before the first MFMA, in both waves of a SIMD at
  // once (they meet at the barrier every step).  Between two MFMAs sit at most a pair of fragment reads or one piece of the split.
  // Timing-only builds (PFHIP_X3_ABLATE, K = 2048, 256 x 128 tile): the real kernel 110 us; without the step barrier 107; without
  // split + LDS writes 64; without fragment reads 83; without global loads 84; MFMAs alone 60 (562 TF: the rate the chip holds
  // under fp16 MFMA loops) — the vector and LDS work of a step still does not hide behind its MFMAs.  Measured and dropped: a
  // two-phase step with waves 4-7 one phase behind waves 0-3 (one group in its MFMA phase while the other stages: 134 us), a
  // staging map whose LDS writes are bank-conflict-free (no change).
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_M(acc, A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc, 0, 0, 0); PFHIP_SB;
#define PFHIP_RA(GA, rst, p, i) if (PFHIP_X3_DO_READ) GA[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (rst) * kStageB + (p) * kPlaneA + a_fr + (i) * 32 * kRowB));
#define PFHIP_RB(GB, rst, p, i) if (PFHIP_X3_DO_READ) GB[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (rst) * kStageB + (p) * kPlaneW + w_fr + (i) * 32 * kRowB));
#define PFHIP_STEP(FA, FB, GA, GB, RA0, RA1, RW, wst, rst, knext)                                                              \
  {                                                                                                                            \
    SplitTmp t_;                                                                                                               \
    unsigned char* const wa_ = lds + (wst) * kStageB + a_st;                                                                   \
    unsigned char* const ww_ = lds + (wst) * kStageB + w_st;                                                                   \
    PFHIP_M(acc00, FA[0][0], FB[1][0]) PFHIP_RA(GA, rst, 0, 0) PFHIP_RB(GB, rst, 0, 0) PFHIP_SB;                               \
    PFHIP_M(acc01, FA[0][0], FB[1][1]) if (PFHIP_X3_DO_SPLIT) split_hi<false>(RA0, 1.0f, wa_, t_); PFHIP_SB;                   \
    PFHIP_M(acc10, FA[0][1], FB[1][0]) PFHIP_RA(GA, rst, 1, 0) PFHIP_RB(GB, rst, 1, 0) PFHIP_SB;                               \
    PFHIP_M(acc11, FA[0][1], FB[1][1]) if (PFHIP_X3_DO_SPLIT) split_lo(wa_ + kPlaneA, t_); if (PFHIP_X3_DO_LOAD) RA0 = *reinterpret_cast<const float4*>(Ag0 + (knext)); PFHIP_SB; \
    PFHIP_M(acc00, FA[1][0], FB[0][0]) PFHIP_RA(GA, rst, 0, 1) PFHIP_RB(GB, rst, 0, 1) PFHIP_SB;                               \
    PFHIP_M(acc01, FA[1][0], FB[0][1]) if (PFHIP_X3_DO_SPLIT) split_hi<false>(RA1, 1.0f, wa_ + 128 * kRowB, t_); PFHIP_SB;     \
    PFHIP_M(acc10, FA[1][1], FB[0][0]) PFHIP_RA(GA, rst, 1, 1) PFHIP_RB(GB, rst, 1, 1) PFHIP_SB;                               \
    PFHIP_M(acc11, FA[1][1], FB[0][1]) if (PFHIP_X3_DO_SPLIT) split_lo(wa_ + 128 * kRowB + kPlaneA, t_); if (PFHIP_X3_DO_LOAD) RA1 = *reinterpret_cast<const float4*>(Ag1 + (knext)); PFHIP_SB; \
    PFHIP_M(acc00, FA[0][0], FB[0][0]) if (PFHIP_X3_DO_SPLIT) split_hi<SC>(RW, sw, ww_, t_); PFHIP_SB;                         \
    PFHIP_M(acc01, FA[0][0], FB[0][1]) if (PFHIP_X3_DO_SPLIT) split_lo(ww_ + kPlaneW, t_); if (PFHIP_X3_DO_LOAD) RW = *reinterpret_cast<const float4*>(Wg + (knext)); PFHIP_SB; \
    PFHIP_M(acc10, FA[0][1], FB[0][0])                                                                                         \
    PFHIP_M(acc11, FA[0][1], FB[0][1])                                                                                         \
    asm volatile(PFHIP_X3_BARRIER ::: "memory");                                                                               \
    PFHIP_SB;                                                                                                                  \
  }

  const int nk = K / kBK;
  auto kclamp = [&](int t) { return (t < nk ? t : nk - 1) * kBK; };
  PFHIP_LOAD_RAW(xa0, xa1, xw, 0)
  PFHIP_LOAD_RAW(ya0, ya1, yw, kclamp(1))
  PFHIP_SPLIT_STORE(xa0, xa1, xw, 0)
  PFHIP_SPLIT_STORE(ya0, ya1, yw, 1)
  PFHIP_LOAD_RAW(xa0, xa1, xw, kclamp(2))                  // K-steps 2 and 3 wait in registers: step k splits K-step k + 2
  PFHIP_LOAD_RAW(ya0, ya1, yw, kclamp(3))
  __syncthreads();
  PFHIP_FRAGS(fa, fb, 0)

  // step k: fragments f / g by parity, raw set x / y by parity, writes stage (k + 2) % 3, reads stage (k + 1) % 3: a period of
  // six steps.  Splits / loads / reads past the last K-step redo the last one (never used): keeps the bodies straight-line.
#define PFHIP_S0(kt) { const int knext = kclamp((kt) + 4); PFHIP_STEP(fa, fb, ga, gb, xa0, xa1, xw, 2, 1, knext) }
#define PFHIP_S1(kt) { const int knext = kclamp((kt) + 5); PFHIP_STEP(ga, gb, fa, fb, ya0, ya1, yw, 0, 2, knext) }
#define PFHIP_S2(kt) { const int knext = kclamp((kt) + 6); PFHIP_STEP(fa, fb, ga, gb, xa0, xa1, xw, 1, 0, knext) }
#define PFHIP_S3(kt) { const int knext = kclamp((kt) + 7); PFHIP_STEP(ga, gb, fa, fb, ya0, ya1, yw, 2, 1, knext) }
#define PFHIP_S4(kt) { const int knext = kclamp((kt) + 8); PFHIP_STEP(fa, fb, ga, gb, xa0, xa1, xw, 0, 2, knext) }
#define PFHIP_S5(kt) { const int knext = kclamp((kt) + 9); PFHIP_STEP(ga, gb, fa, fb, ya0, ya1, yw, 1, 0, knext) }
  int kt = 0;
  for (; kt + 5 < nk; kt += 6) { PFHIP_S0(kt) PFHIP_S1(kt) PFHIP_S2(kt) PFHIP_S3(kt) PFHIP_S4(kt) PFHIP_S5(kt) }
  if (kt < nk) PFHIP_S0(kt)
  if (kt + 1 < nk) PFHIP_S1(kt)
  if (kt + 2 < nk) PFHIP_S2(kt)
  if (kt + 3 < nk) PFHIP_S3(kt)
  if (kt + 4 < nk) PFHIP_S4(kt)
#undef PFHIP_S0
#undef PFHIP_S1
#undef PFHIP_S2
#undef PFHIP_S3
#undef PFHIP_S4
#undef PFHIP_S5
#undef PFHIP_STEP
#ifdef PFHIP_M
#undef PFHIP_M
#undef PFHIP_RA
#undef PFHIP_RB
#undef PFHIP_SB
#endif
#undef PFHIP_SPLIT_STORE
#undef PFHIP_LOAD_RAW
#undef PFHIP_SGB
#undef PFHIP_X6
#undef PFHIP_FRAGS
  // residual rows of the whole tile requested before the accumulators go through LDS: one memory latency for the epilogue
  // instead of one per pass
  const int c4 = tid & 31, rsub = tid >> 5;
  const int gcol = n0 + 4 * c4;
  float4 r1v[16];
#pragma unroll
  for (int pass = 0; pass < 16; ++pass) {
    const int grow = m0 + pass * 16 + rsub;
    r1v[pass] = (R1 && grow < M && gcol + 3 < N) ? *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();                        // every wave has finished reading operand fragments

  // ---- epilogue: two 128-row halves through LDS (C/D map: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5)) ------------------
  float* const Cs = reinterpret_cast<float*>(lds);
  const float inv = SC ? 1.0f / sw : 1.0f;                  // a power of two: exact
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) {
    if (gcol + 3 < N) bv = *reinterpret_cast<const float4*>(bias + gcol);
    else {
      if (gcol < N) bv.x = bias[gcol];
      if (gcol + 1 < N) bv.y = bias[gcol + 1];
      if (gcol + 2 < N) bv.z = bias[gcol + 2];
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if ((wr >> 1) == half) {
      float* cw = Cs + ((wr & 1) * 64 + 4 * h) * kCs + wc * 64 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ro = ((e & 3) + 8 * (e >> 2)) * kCs;
        cw[ro] = acc00[e];
        cw[ro + 32] = acc01[e];
        cw[ro + 32 * kCs] = acc10[e];
        cw[ro + 32 * kCs + 32] = acc11[e];
      }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 16 + rsub;
      const int grow = m0 + half * 128 + row;
      float4 v = *reinterpret_cast<const float4*>(Cs + row * kCs + 4 * c4);
      v.x = v.x * inv + bv.x; v.y = v.y * inv + bv.y; v.z = v.z * inv + bv.z; v.w = v.w * inv + bv.w;
      if (grow < M && gcol + 3 < N) {
        if (R1) {
          const float4 t = r1v[half * 8 + pass];
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
      if (stats_out) tile_row_stats(v, grow, M, tn, tiles_n, c4, stats_out);
    }
    __syncthreads();
  }
}

// ---- 128 x 128 sibling for grids that would not fill half a round of 256 x 128 tiles (decoder-side GEMMs over a few thousand
// token rows, batches of 4-8 utterances, rounds of streaming connections): 8 waves as 2 x 4, each 64 x 32 = 2 x 1 MFMA tiles,
// LDS 2 x 36,864 B so two blocks share a CU; same staging / split / pipelining scheme (per K-step and wave: 12 MFMAs, 44 VALU
// ops of splitting, 6 LDS writes, 2 global loads, 9 fragment reads).
constexpr int kSM = 128;
constexpr int kSPlane = kSM * kRowB;                        // 6,144 B per plane (A and W tiles have 128 rows each)
constexpr int kSStageB = 4 * kSPlane;                       // 24,576 B
constexpr int kSLdsBytes = kRing * kSStageB;                // 73,728 B (>= the C tile + row statistics of the epilogue, 68,608 B)
static_assert(kSM * kCs * 4 <= kSLdsBytes, "C tile must fit the operand buffers");

template <bool LN, bool SC>
__global__ __launch_bounds__(512, 4) void gemm_f32_f16x3_128_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C, int ldc,
    const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int tiles_n,
    int n_tiles, int gw, int relu, const float* __restrict__ ln_stats, int ln_tiles, float ln_eps, float* __restrict__ stats_out,
    const float* __restrict__ ln_colsum, float sw, int* range_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int tm, tn;
  tile_of_block_x3(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kSM, n0 = tn * kBN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 31, h = lane >> 5;

  const int srow = tid >> 2, sq = tid & 3;
  const float* Ag = A + (size_t)min(m0 + srow, M - 1) * lda + 4 * sq;
  const float* Wg = W + (size_t)min(n0 + srow, N - 1) * ldw + 4 * sq;
  const int a_st = srow * kRowB + 8 * sq;
  const int w_st = 2 * kSPlane + a_st;
  const int a_fr = (wr * 64 + r) * kRowB + 16 * h;
  const int w_fr = 2 * kSPlane + (wc * 32 + r) * kRowB + 16 * h;

  // LayerNorm folded in (LN): the A operand is the RAW residual stream x; with gamma folded into the weights (W' = W gamma)
  //     LN(x) W^T = rstd_i * (x W'^T - mean_i * colsum(W')_n) + (bias + W beta)_n,
  // so the K-loop is the plain one and the normalisation is two FMAs per output element in the epilogue (mean_i / rstd_i merged
  // from the per-tile pairs the producing GEMM left, Chan's formula; colsum and the folded bias prepared at load).  The
  // subtraction amplifies the accumulation error by about sqrt(mean^2 + var) / std of the row — a small factor for a residual
  // stream — where normalising on load did not, but on-load cost the 4-waves-per-SIMD loop 4 % (8 VALU ops per K-step).
  float2 ln_mr = make_float2(0.f, 1.f);
  if (LN && tid < kSM) ln_mr = ln_row_stats(ln_stats, ln_tiles, ln_eps, min(m0 + tid, M - 1), range_flag);
  float4 xa, xw, ya, yw;
#define PFHIP_LOAD_RAW(RA, RW, k0)                            \
  RA = *reinterpret_cast<const float4*>(Ag + (k0));           \
  RW = *reinterpret_cast<const float4*>(Wg + (k0));
#define PFHIP_SPLIT_STORE(RA, RW, stage)                      \
  split2<false>(RA, 1.0f, lds + (stage) * kSStageB + a_st, kSPlane);   \
  split2<SC>(RW, sw, lds + (stage) * kSStageB + w_st, kSPlane);

  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }

  half8 fa[2][2], fb[2], ga[2][2], gb[2];
#define PFHIP_FRAGS(FA, FB, stage)                                                                                  \
  _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                                   \
    FA[p][0] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (stage) * kSStageB + p * kSPlane + a_fr));               \
    FA[p][1] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (stage) * kSStageB + p * kSPlane + a_fr + 32 * kRowB));  \
    FB[p] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (stage) * kSStageB + p * kSPlane + w_fr));                  \
  }
#define PFHIP_X6(FA, FB, pa, pb)                                                          \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa][0], FB[pb], acc0, 0, 0, 0);       \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[pa][1], FB[pb], acc1, 0, 0, 0);
#define PFHIP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
  // ring of three LDS stages, one hand-placed region per K-step (see the 256 x 128 kernel)
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_M(acc, A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc, 0, 0, 0); PFHIP_SB;
#define PFHIP_RA(GA, rst, p, i) if (PFHIP_X3_DO_READ) GA[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (rst) * kSStageB + (p) * kSPlane + a_fr + (i) * 32 * kRowB));
#define PFHIP_RB(GB, rst, p) if (PFHIP_X3_DO_READ) GB[p] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (rst) * kSStageB + (p) * kSPlane + w_fr));
#define PFHIP_STEP(FA, FB, GA, GB, RA, RW, wst, rst, knext)                                                                    \
  {                                                                                                                            \
    SplitTmp t_;                                                                                                               \
    unsigned char* const wa_ = lds + (wst) * kSStageB + a_st;                                                                  \
    unsigned char* const ww_ = lds + (wst) * kSStageB + w_st;                                                                  \
    PFHIP_M(acc0, FA[0][0], FB[1]) PFHIP_RA(GA, rst, 0, 0) PFHIP_RB(GB, rst, 0) PFHIP_SB;                                      \
    PFHIP_M(acc1, FA[0][1], FB[1]) if (PFHIP_X3_DO_SPLIT) split_hi<false>(RA, 1.0f, wa_, t_); PFHIP_SB;                        \
    PFHIP_M(acc0, FA[1][0], FB[0]) PFHIP_RA(GA, rst, 1, 0) PFHIP_RA(GA, rst, 0, 1) if (PFHIP_X3_DO_SPLIT) split_lo(wa_ + kSPlane, t_); if (PFHIP_X3_DO_LOAD) RA = *reinterpret_cast<const float4*>(Ag + (knext)); PFHIP_SB; \
    PFHIP_M(acc1, FA[1][1], FB[0]) if (PFHIP_X3_DO_SPLIT) split_hi<SC>(RW, sw, ww_, t_); PFHIP_SB;                             \
    PFHIP_M(acc0, FA[0][0], FB[0]) PFHIP_RA(GA, rst, 1, 1) PFHIP_RB(GB, rst, 1) if (PFHIP_X3_DO_SPLIT) split_lo(ww_ + kSPlane, t_); if (PFHIP_X3_DO_LOAD) RW = *reinterpret_cast<const float4*>(Wg + (knext)); PFHIP_SB; \
    PFHIP_M(acc1, FA[0][1], FB[0])                                                                                             \
    asm volatile(PFHIP_X3_BARRIER ::: "memory");                                                                               \
    PFHIP_SB;                                                                                                                  \
  }

  const int nk = K / kBK;
  auto kclamp = [&](int t) { return (t < nk ? t : nk - 1) * kBK; };
  PFHIP_LOAD_RAW(xa, xw, 0)
  PFHIP_LOAD_RAW(ya, yw, kclamp(1))
  PFHIP_SPLIT_STORE(xa, xw, 0)
  PFHIP_SPLIT_STORE(ya, yw, 1)
  PFHIP_LOAD_RAW(xa, xw, kclamp(2))
  PFHIP_LOAD_RAW(ya, yw, kclamp(3))
  __syncthreads();
  PFHIP_FRAGS(fa, fb, 0)
#define PFHIP_S0(kt) { const int knext = kclamp((kt) + 4); PFHIP_STEP(fa, fb, ga, gb, xa, xw, 2, 1, knext) }
#define PFHIP_S1(kt) { const int knext = kclamp((kt) + 5); PFHIP_STEP(ga, gb, fa, fb, ya, yw, 0, 2, knext) }
#define PFHIP_S2(kt) { const int knext = kclamp((kt) + 6); PFHIP_STEP(fa, fb, ga, gb, xa, xw, 1, 0, knext) }
#define PFHIP_S3(kt) { const int knext = kclamp((kt) + 7); PFHIP_STEP(ga, gb, fa, fb, ya, yw, 2, 1, knext) }
#define PFHIP_S4(kt) { const int knext = kclamp((kt) + 8); PFHIP_STEP(fa, fb, ga, gb, xa, xw, 0, 2, knext) }
#define PFHIP_S5(kt) { const int knext = kclamp((kt) + 9); PFHIP_STEP(ga, gb, fa, fb, ya, yw, 1, 0, knext) }
  int kt = 0;
  for (; kt + 5 < nk; kt += 6) { PFHIP_S0(kt) PFHIP_S1(kt) PFHIP_S2(kt) PFHIP_S3(kt) PFHIP_S4(kt) PFHIP_S5(kt) }
  if (kt < nk) PFHIP_S0(kt)
  if (kt + 1 < nk) PFHIP_S1(kt)
  if (kt + 2 < nk) PFHIP_S2(kt)
  if (kt + 3 < nk) PFHIP_S3(kt)
  if (kt + 4 < nk) PFHIP_S4(kt)
#undef PFHIP_S0
#undef PFHIP_S1
#undef PFHIP_S2
#undef PFHIP_S3
#undef PFHIP_S4
#undef PFHIP_S5
#undef PFHIP_STEP
#ifdef PFHIP_M
#undef PFHIP_M
#undef PFHIP_RA
#undef PFHIP_RB
#undef PFHIP_SB
#endif
#undef PFHIP_SPLIT_STORE
#undef PFHIP_LOAD_RAW
#undef PFHIP_SGB
#undef PFHIP_X6
#undef PFHIP_FRAGS
  // residual rows of the whole tile requested before the accumulators go through LDS: one memory latency for the epilogue
  // instead of one per pass
  const int c4 = tid & 31, rsub = tid >> 5;
  const int gcol = n0 + 4 * c4;
  float4 r1v[8];
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int grow = m0 + pass * 16 + rsub;
    r1v[pass] = (R1 && grow < M && gcol + 3 < N) ? *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  float* const Cs = reinterpret_cast<float*>(lds);
  {
    float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ro = ((e & 3) + 8 * (e >> 2)) * kCs;
      cw[ro] = acc0[e];
      cw[ro + 32 * kCs] = acc1[e];
    }
  }
  float2* const s_mr = reinterpret_cast<float2*>(lds + kSM * kCs * 4);          // behind the C tile: 128 x (mean, rstd)
  static_assert(kSM * kCs * 4 + kSM * 8 <= kSLdsBytes, "row statistics must fit behind the C tile");
  if (LN && tid < kSM) s_mr[tid] = ln_mr;
  __syncthreads();
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) {
    if (gcol + 3 < N) bv = *reinterpret_cast<const float4*>(bias + gcol);
    else {
      if (gcol < N) bv.x = bias[gcol];
      if (gcol + 1 < N) bv.y = bias[gcol + 1];
      if (gcol + 2 < N) bv.z = bias[gcol + 2];
    }
  }
  float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (LN && gcol + 3 < N) cs4 = *reinterpret_cast<const float4*>(ln_colsum + gcol);
  const float inv = SC ? 1.0f / sw : 1.0f;                  // a power of two: exact
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 16 + rsub;
    const int grow = m0 + row;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * kCs + 4 * c4);
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    if (LN) ln_finish(v, cs4, s_mr[row]);
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    if (grow < M && gcol + 3 < N) {
      if (R1) {
        const float4 t = r1v[pass];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (R2) {
        const float4 t = *reinterpret_cast<const float4*>(R2 + (size_t)grow * ldr2 + gcol);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(C + (size_t)grow * ldc + gcol) = v;
    } else if (grow < M && gcol < N) {
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int q = 0; q < 4 && gcol + q < N; ++q) {
        float o = vv[q];
        if (R1) o += R1[(size_t)grow * ldr1 + gcol + q];
        if (R2) o += R2[(size_t)grow * ldr2 + gcol + q];
        if (relu) o = fmaxf(o, 0.f);
        C[(size_t)grow * ldc + gcol + q] = o;
      }
    }
    if (stats_out) tile_row_stats(v, grow, M, tn, tiles_n, c4, stats_out);
  }
}


// ---- 64 x 128 sibling for grids that would leave most of a round of 128 x 128 tiles empty (at most 256 of these tiles: rounds of
// streaming connections — 128 x 20 rows = 160 tiles at N = 512 —, offline batches of a few utterances): ONE workgroup per CU, so
// nothing hides a step's fixed costs (the barrier, the LDS write -> read turn-around, the issue of the split) but the step's own
// MFMAs, and a wave owns a single 32 x 32 tile: three MFMAs per 16-deep step.  This kernel therefore takes K-steps of 32 — two
// MFMA depths, six MFMAs per wave between barriers, 128-byte row pieces per global load (full cache lines; the 16-deep kernels
// load 64-byte pieces) — in a ring of three stages of 30,720 B.  8 waves as 2 x 4; every thread stages one float4 of A (64 rows)
// and two of W (128 rows).
constexpr int kHM = 64, kHK = 32;
constexpr int kHRowB = 80;                                  // bytes per operand row: 32 fp16 + 16 pad (conflict-free ds_read_b128)
constexpr int kHPlaneA = kHM * kHRowB;                      // 5,120 B
constexpr int kHPlaneW = kBN * kHRowB;                      // 10,240 B
constexpr int kHStageB = 2 * (kHPlaneA + kHPlaneW);         // 30,720 B
constexpr int kHLdsBytes = kRing * kHStageB;                // 92,160 B
static_assert(kHM * kCs * 4 + kHM * 8 <= kHLdsBytes, "C tile + row statistics must fit the operand buffers");

template <bool LN, bool SC>
__global__ __launch_bounds__(512, 2) void gemm_f32_f16x3_64_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C, int ldc,
    const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int tiles_n,
    int n_tiles, int gw, int relu, const float* __restrict__ ln_stats, int ln_tiles, float ln_eps, float* __restrict__ stats_out,
    const float* __restrict__ ln_colsum, float sw, int* range_flag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int tm, tn;
  tile_of_block_x3(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kHM, n0 = tn * kBN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 31, h = lane >> 5;

  // staging map: thread t holds 4 consecutive k (one of the eight 16-B pieces of a row's 128-B K-step) of A row t/8 and of W
  // rows t/8 and t/8 + 64: a wave's load instruction covers 8 rows x 128 contiguous bytes
  const int srow = tid >> 3, sq = tid & 7;
  const float* Ag = A + (size_t)min(m0 + srow, M - 1) * lda + 4 * sq;
  const float* Wg0 = W + (size_t)min(n0 + srow, N - 1) * ldw + 4 * sq;
  const float* Wg1 = W + (size_t)min(n0 + srow + 64, N - 1) * ldw + 4 * sq;
  const int a_st = srow * kHRowB + 8 * sq;
  const int w_st = 2 * kHPlaneA + srow * kHRowB + 8 * sq;
  const int a_fr = (wr * 32 + r) * kHRowB + 16 * h;               // + 32 per 16-deep half of the step
  const int w_fr = 2 * kHPlaneA + (wc * 32 + r) * kHRowB + 16 * h;

  float2 ln_mr = make_float2(0.f, 1.f);
  if (LN && tid < kHM) ln_mr = ln_row_stats(ln_stats, ln_tiles, ln_eps, min(m0 + tid, M - 1), range_flag);
  float4 xa, xw0, xw1, ya, yw0, yw1;
#define PFHIP_LOAD_RAW(RA, RW0, RW1, k0)                      \
  RA = *reinterpret_cast<const float4*>(Ag + (k0));           \
  RW0 = *reinterpret_cast<const float4*>(Wg0 + (k0));         \
  RW1 = *reinterpret_cast<const float4*>(Wg1 + (k0));
#define PFHIP_SPLIT_STORE(RA, RW0, RW1, stage)                                                    \
  split2<false>(RA, 1.0f, lds + (stage) * kHStageB + a_st, kHPlaneA);                             \
  split2<SC>(RW0, sw, lds + (stage) * kHStageB + w_st, kHPlaneW);                                 \
  split2<SC>(RW1, sw, lds + (stage) * kHStageB + w_st + 64 * kHRowB, kHPlaneW);

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;

  // operand fragments [plane][16-deep half of the step]: even K-steps in f*, odd in g*
  half8 fa[2][2], fb[2][2], ga[2][2], gb[2][2];
#define PFHIP_RA(GA, rst, p, ks) if (PFHIP_X3_DO_READ) GA[p][ks] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (rst) * kHStageB + (p) * kHPlaneA + a_fr + 32 * (ks)));
#define PFHIP_RB(GB, rst, p, ks) if (PFHIP_X3_DO_READ) GB[p][ks] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (rst) * kHStageB + (p) * kHPlaneW + w_fr + 32 * (ks)));
#define PFHIP_FRAGS(FA, FB, stage)                                                                                  \
  _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { PFHIP_RA(FA, stage, p, ks) PFHIP_RB(FB, stage, p, ks) }       \
  }
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_M(A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc, 0, 0, 0); PFHIP_SB;
  // ring of three LDS stages, one hand-placed region per K-step of 32 (see the 256 x 128 kernel): per 16-deep half the products
  // a_hi w_lo, a_lo w_hi, a_hi w_hi, in that order
#define PFHIP_STEP(FA, FB, GA, GB, RA, RW0, RW1, wst, rst, knext)                                                              \
  {                                                                                                                            \
    SplitTmp t_;                                                                                                               \
    unsigned char* const wa_ = lds + (wst) * kHStageB + a_st;                                                                  \
    unsigned char* const ww_ = lds + (wst) * kHStageB + w_st;                                                                  \
    PFHIP_M(FA[0][0], FB[1][0]) PFHIP_RA(GA, rst, 0, 0) PFHIP_RB(GB, rst, 0, 0) PFHIP_SB;                                      \
    PFHIP_M(FA[1][0], FB[0][0]) if (PFHIP_X3_DO_SPLIT) split_hi<false>(RA, 1.0f, wa_, t_); PFHIP_SB;                           \
    PFHIP_M(FA[0][0], FB[0][0]) PFHIP_RA(GA, rst, 1, 0) PFHIP_RB(GB, rst, 1, 0) if (PFHIP_X3_DO_SPLIT) split_lo(wa_ + kHPlaneA, t_); if (PFHIP_X3_DO_LOAD) RA = *reinterpret_cast<const float4*>(Ag + (knext)); PFHIP_SB; \
    PFHIP_M(FA[0][1], FB[1][1]) if (PFHIP_X3_DO_SPLIT) split_hi<SC>(RW0, sw, ww_, t_); PFHIP_SB;                               \
    PFHIP_M(FA[1][1], FB[0][1]) PFHIP_RA(GA, rst, 0, 1) PFHIP_RB(GB, rst, 0, 1) if (PFHIP_X3_DO_SPLIT) split_lo(ww_ + kHPlaneW, t_); if (PFHIP_X3_DO_LOAD) RW0 = *reinterpret_cast<const float4*>(Wg0 + (knext)); PFHIP_SB; \
    PFHIP_M(FA[0][1], FB[0][1]) if (PFHIP_X3_DO_SPLIT) { split_hi<SC>(RW1, sw, ww_ + 64 * kHRowB, t_); split_lo(ww_ + 64 * kHRowB + kHPlaneW, t_); } \
    if (PFHIP_X3_DO_LOAD) RW1 = *reinterpret_cast<const float4*>(Wg1 + (knext)); PFHIP_RA(GA, rst, 1, 1) PFHIP_RB(GB, rst, 1, 1) PFHIP_SB; \
    asm volatile(PFHIP_X3_BARRIER ::: "memory");                                                                               \
    PFHIP_SB;                                                                                                                  \
  }

  const int nk = K / kHK;
  auto kclamp = [&](int t) { return (t < nk ? t : nk - 1) * kHK; };
  PFHIP_LOAD_RAW(xa, xw0, xw1, 0)
  PFHIP_LOAD_RAW(ya, yw0, yw1, kclamp(1))
  PFHIP_SPLIT_STORE(xa, xw0, xw1, 0)
  PFHIP_SPLIT_STORE(ya, yw0, yw1, 1)
  PFHIP_LOAD_RAW(xa, xw0, xw1, kclamp(2))
  PFHIP_LOAD_RAW(ya, yw0, yw1, kclamp(3))
  __syncthreads();
  PFHIP_FRAGS(fa, fb, 0)
#define PFHIP_S0(kt) { const int knext = kclamp((kt) + 4); PFHIP_STEP(fa, fb, ga, gb, xa, xw0, xw1, 2, 1, knext) }
#define PFHIP_S1(kt) { const int knext = kclamp((kt) + 5); PFHIP_STEP(ga, gb, fa, fb, ya, yw0, yw1, 0, 2, knext) }
#define PFHIP_S2(kt) { const int knext = kclamp((kt) + 6); PFHIP_STEP(fa, fb, ga, gb, xa, xw0, xw1, 1, 0, knext) }
#define PFHIP_S3(kt) { const int knext = kclamp((kt) + 7); PFHIP_STEP(ga, gb, fa, fb, ya, yw0, yw1, 2, 1, knext) }
#define PFHIP_S4(kt) { const int knext = kclamp((kt) + 8); PFHIP_STEP(fa, fb, ga, gb, xa, xw0, xw1, 0, 2, knext) }
#define PFHIP_S5(kt) { const int knext = kclamp((kt) + 9); PFHIP_STEP(ga, gb, fa, fb, ya, yw0, yw1, 1, 0, knext) }
  int kt = 0;
  for (; kt + 5 < nk; kt += 6) { PFHIP_S0(kt) PFHIP_S1(kt) PFHIP_S2(kt) PFHIP_S3(kt) PFHIP_S4(kt) PFHIP_S5(kt) }
  if (kt < nk) PFHIP_S0(kt)
  if (kt + 1 < nk) PFHIP_S1(kt)
  if (kt + 2 < nk) PFHIP_S2(kt)
  if (kt + 3 < nk) PFHIP_S3(kt)
  if (kt + 4 < nk) PFHIP_S4(kt)
#undef PFHIP_S0
#undef PFHIP_S1
#undef PFHIP_S2
#undef PFHIP_S3
#undef PFHIP_S4
#undef PFHIP_S5
#undef PFHIP_STEP
#undef PFHIP_M
#undef PFHIP_RA
#undef PFHIP_RB
#undef PFHIP_SB
#undef PFHIP_SPLIT_STORE
#undef PFHIP_LOAD_RAW
#undef PFHIP_FRAGS
  // residual rows of the whole tile requested before the accumulators go through LDS: one memory latency for the epilogue
  // instead of one per pass
  const int c4 = tid & 31, rsub = tid >> 5;
  const int gcol = n0 + 4 * c4;
  float4 r1v[4];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int grow = m0 + pass * 16 + rsub;
    r1v[pass] = (R1 && grow < M && gcol + 3 < N) ? *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();

  float* const Cs = reinterpret_cast<float*>(lds);
  {
    float* cw = Cs + (wr * 32 + 4 * h) * kCs + wc * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) cw[((e & 3) + 8 * (e >> 2)) * kCs] = acc[e];
  }
  float2* const s_mr = reinterpret_cast<float2*>(lds + kHM * kCs * 4);
  static_assert(kHM * kCs * 4 + kHM * 8 <= kHLdsBytes, "row statistics must fit behind the C tile");
  if (LN && tid < kHM) s_mr[tid] = ln_mr;
  __syncthreads();
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) {
    if (gcol + 3 < N) bv = *reinterpret_cast<const float4*>(bias + gcol);
    else {
      if (gcol < N) bv.x = bias[gcol];
      if (gcol + 1 < N) bv.y = bias[gcol + 1];
      if (gcol + 2 < N) bv.z = bias[gcol + 2];
    }
  }
  float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (LN && gcol + 3 < N) cs4 = *reinterpret_cast<const float4*>(ln_colsum + gcol);
  const float inv = SC ? 1.0f / sw : 1.0f;                  // a power of two: exact
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = pass * 16 + rsub;
    const int grow = m0 + row;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * kCs + 4 * c4);
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    if (LN) ln_finish(v, cs4, s_mr[row]);
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    if (grow < M && gcol + 3 < N) {
      if (R1) {
        const float4 t = r1v[pass];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (R2) {
        const float4 t = *reinterpret_cast<const float4*>(R2 + (size_t)grow * ldr2 + gcol);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(C + (size_t)grow * ldc + gcol) = v;
    } else if (grow < M && gcol < N) {
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int q = 0; q < 4 && gcol + q < N; ++q) {
        float o = vv[q];
        if (R1) o += R1[(size_t)grow * ldr1 + gcol + q];
        if (R2) o += R2[(size_t)grow * ldr2 + gcol + q];
        if (relu) o = fmaxf(o, 0.f);
        C[(size_t)grow * ldc + gcol + q] = o;
      }
    }
    if (stats_out) tile_row_stats(v, grow, M, tn, tiles_n, c4, stats_out);
  }
}

}  // namespace

namespace {
template <auto kern, class... Args>
void launch_with_lds(int n_tiles, int lds_bytes, hipStream_t s, Args... args) {
  // > 64 KB of dynamic LDS needs the opt-in once per kernel (= per instantiation of this function) and device
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_done.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_done.fetch_or(1ull << (dev & 63));
  }
  hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(512), lds_bytes, s, args...);
}
}  // namespace

void launch_gemm_f32_f16x3(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                           int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, int gw, hipStream_t s, bool small_tile,
                           const float* ln_stats, int ln_tiles, float* stats_out, bool half_tile, const float* ln_colsum, float sw) {
  if (M <= 0 || N <= 0) return;
  if (ln_stats) small_tile = true;              // LayerNorm-on-load lives in the 128 / 64-row kernels (its consumers have K = 512)
  const bool sc = sw != 1.0f;
  const int rl = relu ? 1 : 0;
  const float eps = 1e-12f;
  const int tiles_n = (N + kBN - 1) / kBN;
  gw = std::max(1, std::min(gw, tiles_n));
#define PFHIP_X3_LAUNCH_LN(KERN, TM, LDS)                                                                                        \
  {                                                                                                                              \
    const int n_tiles = ((M + (TM) - 1) / (TM)) * tiles_n;                                                                       \
    if (ln_stats && sc) launch_with_lds<KERN<true, true>>(n_tiles, LDS, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw, rl, ln_stats, ln_tiles, eps, stats_out, ln_colsum, sw, launch_ctx().range_flag);      \
    else if (ln_stats) launch_with_lds<KERN<true, false>>(n_tiles, LDS, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw, rl, ln_stats, ln_tiles, eps, stats_out, ln_colsum, sw, launch_ctx().range_flag);       \
    else if (sc) launch_with_lds<KERN<false, true>>(n_tiles, LDS, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw, rl, ln_stats, ln_tiles, eps, stats_out, ln_colsum, sw, launch_ctx().range_flag);            \
    else launch_with_lds<KERN<false, false>>(n_tiles, LDS, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw, rl, ln_stats, ln_tiles, eps, stats_out, ln_colsum, sw, launch_ctx().range_flag);                    \
  }
  if (small_tile && half_tile) {
    PFHIP_X3_LAUNCH_LN(gemm_f32_f16x3_64_kernel, kHM, kHLdsBytes)
    return;
  }
  if (small_tile) {
    PFHIP_X3_LAUNCH_LN(gemm_f32_f16x3_128_kernel, kSM, kSLdsBytes)
    return;
  }
#undef PFHIP_X3_LAUNCH_LN
  const int n_tiles = ((M + kBM - 1) / kBM) * tiles_n;
  if (sc) launch_with_lds<gemm_f32_f16x3_kernel<true>>(n_tiles, kLdsBytes, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw, rl, stats_out, sw);
  else launch_with_lds<gemm_f32_f16x3_kernel<false>>(n_tiles, kLdsBytes, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw, rl, stats_out, sw);
}

}  // namespace pfhip
