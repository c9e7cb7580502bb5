// fp32-grade GEMM on PRE-SPLIT operands: the three-product fp16 scheme of gemm_x3.hip with the split taken out of the K-loop.
//
// gemm_x3.hip stages fp32 operands through registers and splits them into two fp16 planes on the way into LDS.  With three MFMAs per
// block that staging — not the matrix cores — bounds the kernel: timing-only builds of its 256 x 128 tile at K = 2048 run 110 us as
// built, 64 us without the split and its LDS writes, 60 us with MFMAs alone; replacing the split by a plain 16-byte register copy
// changes nothing (so it is the global -> VGPR -> LDS traffic, not the arithmetic).  Here both operands ARRIVE as plane images:
//   * weights are split once at load (launch_split_planes), pre-multiplied by their power-of-two scale;
//   * activations are written as planes by the kernel that produces them (this kernel's epilogue, the attention's) — 2 + 2 bytes
//     per element, the bytes of the fp32 value they replace;
// and the K-loop is LDS-DMA (`global_load_lds_dwordx4`: global -> LDS, no VGPR, no ds_write, no vector instruction), fragment
// reads and MFMAs.
//
// Plane image of a matrix X[rows, K] (rows padded to 128): for each plane, [K / 16][rows][16] fp16 — the 32 bytes of one row's
// K-step are contiguous, the rows of a K-step are contiguous, so the 128 rows x 16 k of a tile's K-step are ONE 4-KB run per plane
// that four wave-instructions copy into LDS verbatim.  The two 16-byte halves of a row are swapped where row bit 3 is set: the LDS
// image is then conflict-free for `ds_read_b128` on unpadded 32-byte rows (16 lanes = 16 rows read 16 distinct 16-byte slots of
// the 256-byte bank row), and the swizzle costs nothing because it is baked into the image.
//
// Tile 128 x 128, 4 waves as 2 x 2 (each 64 x 64 = four MFMA tiles, twelve MFMAs per K-step), two workgroups per CU, ring of FOUR
// 16-KB stages: step k multiplies fragments it holds, reads the fragments of stage k + 1, and issues the DMA of K-step k + 4 into the
// stage it has just finished reading; `s_waitcnt vmcnt(8)` + the step barrier retire K-step k + 2's DMA two full steps after issue.
// Epilogues: fp32 C (bias, LayerNorm-fold finish, residual, ReLU, row statistics — as gemm_x3.hip), plane images of C for the
// next GEMM, or both.
// In the model (pfhip.cpp, batches of 3500+ rows): QKV' <LN, fp32>, out-projection <fp32 + planes + statistics> on the planes the
// attention writes, FFN1' <LN, planes>, FFN2 <fp32 + planes + statistics>.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <atomic>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using float2v = __attribute__((ext_vector_type(2))) float;

constexpr int kPM = 128, kPN = 128, kPK = 16;
constexpr int kPRowB = 32;                              // bytes of one (row, K-step) of one plane
constexpr int kPPlane = 128 * kPRowB;                   // 4,096 B: one plane of one operand of one stage
constexpr int kPStage = 4 * kPPlane;                    // A hi | A lo | W hi | W lo = 16,384 B
constexpr int kPRing = 4;
constexpr int kPCs = kPN + 4;                           // padded C-tile row stride (floats)
constexpr int kPLds = kPM * kPCs * 4 + kPM * 8;         // 68,608 B: the C tile + row statistics (> 4 stages = 65,536 B)
static_assert(kPRing * kPStage <= kPLds, "the ring must fit");

__device__ __forceinline__ float sub_lo(float x, unsigned h) {      // x - (float)low half of h: one instruction, exact
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float sub_hi(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
// eight consecutive values -> 16 bytes of the hi plane and 16 bytes of the lo plane (hi = rtz, lo = rn(x - hi))
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
  unsigned* hp = &hi.x;
  unsigned* lp = &lo.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hp[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2 * i], v[2 * i + 1]));
    const float2v r = {sub_lo(v[2 * i], hp[i]), sub_hi(v[2 * i + 1], hp[i])};
    lp[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, half2v));
  }
}
// byte offset of (row, 8-k piece) inside a plane image with `rows` rows per K-step
__device__ __forceinline__ size_t image_off(int kstep, int row, int piece, int rows) {
  return ((size_t)kstep * rows + row) * kPRowB + (size_t)((piece ^ ((row >> 3) & 1)) << 4);
}

// X[rows_valid, K] fp32 (row stride ld) -> plane images; rows >= rows_valid are written as zeros.  One thread per (row, 8-k piece).
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ X, int ld, int rows_valid, int rows, int K, float scale,
                                                           unsigned char* __restrict__ hi, unsigned char* __restrict__ lo) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int pieces = K / 8;
  if (idx >= (size_t)rows * pieces) return;
  // consecutive threads walk the rows of one piece: the image writes of a wave are contiguous in pairs of 16 B
  const int row = (int)(idx % rows), pc = (int)(idx / rows);
  float v[8];
  if (row < rows_valid) {
    const float4 a = *reinterpret_cast<const float4*>(X + (size_t)row * ld + 8 * pc);
    const float4 b = *reinterpret_cast<const float4*>(X + (size_t)row * ld + 8 * pc + 4);
    v[0] = a.x * scale; v[1] = a.y * scale; v[2] = a.z * scale; v[3] = a.w * scale;
    v[4] = b.x * scale; v[5] = b.y * scale; v[6] = b.z * scale; v[7] = b.w * scale;
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
  }
  uint4 h, l;
  split8(v, h, l);
  const size_t off = image_off(pc >> 1, row, pc & 1, rows);
  *reinterpret_cast<uint4*>(hi + off) = h;
  *reinterpret_cast<uint4*>(lo + off) = l;
}

__device__ __forceinline__ void tile_of_block_p3(int bid, int n_tiles, int tiles_n, int gw, int& tm, int& tn) {
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

__device__ __forceinline__ float half_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  v += __shfl_xor(v, 16);
  return v;
}
__device__ __forceinline__ void tile_row_stats(const float4& v, int grow, int M, int tn, int tiles_n, int c4, float* __restrict__ stats) {
  const float sum = half_wave_sum((v.x + v.y) + (v.z + v.w));
  const float mean = sum * (1.0f / kPN);
  const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
  const float q = half_wave_sum((a * a + b * b) + (c * c + d * d));
  if (c4 == 0 && grow < M) *reinterpret_cast<float2*>(stats + ((size_t)grow * tiles_n + tn) * 2) = make_float2(mean, q);
}
// (a row whose rms leaves [2^-8, 2^12] raises the forward's range flag: gemm_x3.hip, kernels.h LaunchCtx)
__device__ __forceinline__ float2 ln_row_stats(const float* __restrict__ stats, int tiles, float eps, int row, int* range_flag) {
  const float* sp = stats + (size_t)row * tiles * 2;
  float msum = 0.f, m2 = 0.f;
  for (int t = 0; t < tiles; ++t) msum += sp[2 * t];
  const float mean = msum / (float)tiles;
  for (int t = 0; t < tiles; ++t) { const float dm = sp[2 * t] - mean; m2 += sp[2 * t + 1] + (float)kPN * dm * dm; }
  const float rstd = 1.0f / sqrtf(m2 / (float)(tiles * kPN) + eps);
  if (range_flag && !(rstd > kLnRstdMin && rstd < kLnRstdMax)) atomicOr(range_flag, 2);
  return make_float2(mean, rstd);
}
// The same for T tiles (4: d_model = 512; 16: the decoder's LayerNorm over the 2048 hidden channels) from the row's 8 T bytes fetched as
// T / 2 16-byte loads — requested at kernel entry and first used behind the prologue's DMA issue.  The loop above fetches one value per
// trip with a wait in each: 2 T dependent round trips, 4,000 cycles (T = 4) in front of the first DMA of a 56,000-cycle workgroup
// (in-kernel stamps, tools/p3_stamps.py).  Same operations in the same order: bit-identical.
template <int T>
struct LnRaw { float4 v[T / 2]; };
template <int T>
__device__ __forceinline__ void ln_raw_load(LnRaw<T>& r, const float* __restrict__ stats, int row) {
  const float4* sp = reinterpret_cast<const float4*>(stats + (size_t)row * T * 2);
#pragma unroll
  for (int i = 0; i < T / 2; ++i) r.v[i] = sp[i];
}
template <int T>
__device__ __forceinline__ float2 ln_row_stats_raw(const LnRaw<T>& r, float eps, int* range_flag) {
  float msum = 0.f, m2 = 0.f;
#pragma unroll
  for (int i = 0; i < T / 2; ++i) { msum += r.v[i].x; msum += r.v[i].z; }
  const float mean = msum / (float)T;
#pragma unroll
  for (int i = 0; i < T / 2; ++i) {
    { const float dm = r.v[i].x - mean; m2 += r.v[i].y + (float)kPN * dm * dm; }
    { const float dm = r.v[i].z - mean; m2 += r.v[i].w + (float)kPN * dm * dm; }
  }
  const float rstd = 1.0f / sqrtf(m2 / (float)(T * kPN) + eps);
  if (range_flag && !(rstd > kLnRstdMin && rstd < kLnRstdMax)) atomicOr(range_flag, 2);
  return make_float2(mean, rstd);
}

// Epilogue of a tile that leaves as ROW-MAJOR planes (OUT & 4: the K | V columns of the QKV projection, read by attention_p3.hip): thread =
// (row, 8 columns), 16 lanes x 16 B = the tile's 256 bytes of a row per plane.  Cs = the accumulator tile in LDS.
template <bool LN, int ROWS, int THREADS>
__device__ __forceinline__ void row_planes_tile(const float* Cs, const float2* s_mr, int m0, int n0, int M, int N, float inv_scale,
                                                const float* __restrict__ bias, const float* __restrict__ ln_colsum, unsigned char* __restrict__ Ph,
                                                unsigned char* __restrict__ Pl, int ldp, int split_col, int tid, bool do_store) {
  const int c8 = tid & 15, rsub = tid >> 4;
  const int gcol = n0 + 8 * c8;
  float bb[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bb[e] = bias ? bias[gcol + e] : 0.f; ss[e] = LN ? ln_colsum[gcol + e] : 0.f; }
  constexpr int RPP = THREADS / 16;
#pragma unroll
  for (int pass = 0; pass < ROWS / RPP; ++pass) {
    const int row = pass * RPP + rsub, grow = m0 + row;
    const float4 a = *reinterpret_cast<const float4*>(Cs + row * kPCs + 8 * c8);
    const float4 b = *reinterpret_cast<const float4*>(Cs + row * kPCs + 8 * c8 + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const float2 mr = LN ? s_mr[row] : make_float2(0.f, 1.f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] * inv_scale;
      if (LN) x = mr.y * (x - mr.x * ss[e]);
      v[e] = x + bb[e];
    }
    uint4 hh, ll;
    split8(v, hh, ll);
    if (do_store && grow < M) {
      const size_t off = ((size_t)grow * ldp + (gcol - split_col)) * 2;
      *reinterpret_cast<uint4*>(Ph + off) = hh;
      *reinterpret_cast<uint4*>(Pl + off) = ll;
    }
  }
}

// OUT: 1 = fp32 C, 2 = plane images of C, 3 = both, 5 = fp32 C for columns < split_col and row-major planes for the rest.  A / W images: rows_a / rows_w rows per K-step.  inv_scale = 1 / (weight scale).
// FOUR waves (2 x 2, each 64 x 64 = four MFMA tiles, twelve MFMAs and eight fragment reads per K-step: 0.67 KB of LDS reads per
// MFMA; the first form of this kernel — eight waves of 64 x 32, 1 KB per MFMA — ran 8-20 % over gemm_x3.hip and was bound by the
// LDS reads), two workgroups per CU.
constexpr int kPThreads = 256;
template <bool LN, int OUT>
__global__ __launch_bounds__(kPThreads, 2) void gemm_p3_128_kernel(
    const unsigned char* __restrict__ Ah, const unsigned char* __restrict__ Al, int rows_a, const unsigned char* __restrict__ Wh,
    const unsigned char* __restrict__ Wl, int rows_w, float* C, int ldc, unsigned char* __restrict__ Ph, unsigned char* __restrict__ Pl,
    int rows_p, const float* __restrict__ bias, const float* R1, int ldr1, int M, int N, int K, int tiles_n, int n_tiles, int gw, int relu,
    const float* __restrict__ ln_stats, int ln_tiles, float ln_eps, const float* __restrict__ ln_colsum, float* __restrict__ stats_out,
    float inv_scale, int* range_flag, int split_col) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int tm, tn;
  tile_of_block_p3(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kPM, n0 = tn * kPN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
#ifdef PFHIP_P3_STAMPS      // dev build (tools/p3_stamps.py): s_memtime at the boundaries inside tile (0, 0) and one mid-grid tile, into their C rows
  const bool stamp_wg = C != nullptr && tn == 0 && (tm == 0 || tm == (M / kPM) / 2);
  // 64 stamps per C row (the tile's own 128 columns), rows m0 + 32 wave ..
  auto dbg_at = [&](int i) { return reinterpret_cast<unsigned long long*>(C + (size_t)(m0 + 32 * wave + (i >> 6)) * ldc) + (i & 63); };
  const bool stamping = stamp_wg && lane == 0;
  int n_stamp = 0;
#define PFHIP_STAMP { asm volatile("s_nop 0" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (stamping) *dbg_at(n_stamp) = t_; ++n_stamp; }
  PFHIP_STAMP
#else
#define PFHIP_STAMP
#endif

  // DMA map: per K-step the block copies 16 chunks of 1 KB (4 regions x 4 chunks of 32 rows); wave w moves rows 32 w .. 32 w + 31
  // of all four regions (A hi, A lo, W hi, W lo)
  const size_t ka = (size_t)rows_a * kPRowB, kw = (size_t)rows_w * kPRowB;      // bytes per K-step of an image
  const unsigned char* const gah = Ah + ((size_t)(m0 + 32 * wave)) * kPRowB + lane * 16;
  const unsigned char* const gal = Al + ((size_t)(m0 + 32 * wave)) * kPRowB + lane * 16;
  const unsigned char* const gwh = Wh + ((size_t)(n0 + 32 * wave)) * kPRowB + lane * 16;
  const unsigned char* const gwl = Wl + ((size_t)(n0 + 32 * wave)) * kPRowB + lane * 16;
  const int lds_c = wave * 1024;
#ifndef PFHIP_P3_ABLATE
#define PFHIP_P3_ABLATE 0      // timing-only builds (tools/x3_variant.sh): 1 no step barrier, 2 no DMA, 3 no fragment reads, 4 MFMAs only
#endif
#define PFHIP_DMA1(src, stage, region)                                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                              \
                                   (__attribute__((address_space(3))) void*)(lds + (stage) * kPStage + (region) * kPPlane + lds_c), 16, 0, 0);
  // timing-only: 5 every DMA re-reads K-step 0 (always an L2 hit), 6 only the A operand does, 7 only the W operand does
#define PFHIP_KA(ks) ((PFHIP_P3_ABLATE == 5 || PFHIP_P3_ABLATE == 6) ? 0 : (ks))
#define PFHIP_KW(ks) ((PFHIP_P3_ABLATE == 5 || PFHIP_P3_ABLATE == 7) ? 0 : (ks))
  // 8: only the high planes are copied (half the bytes; the vmcnt accounting below still counts four pieces, so waits come late)
#define PFHIP_DMA(stage, ks)                                                                                          \
  PFHIP_DMA1(gah + (size_t)PFHIP_KA(ks) * ka, stage, 0) if (PFHIP_P3_ABLATE != 8) { PFHIP_DMA1(gal + (size_t)PFHIP_KA(ks) * ka, stage, 1) } \
  PFHIP_DMA1(gwh + (size_t)PFHIP_KW(ks) * kw, stage, 2) if (PFHIP_P3_ABLATE != 8) { PFHIP_DMA1(gwl + (size_t)PFHIP_KW(ks) * kw, stage, 3) }

  // fragment addresses (row R of the tile, 8-k piece h, swizzled like the image): rows R and R + 32 share bit 3
  const int ra = wr * 64 + r, rb = wc * 64 + r;
  const int a_fr = ra * kPRowB + ((h ^ ((ra >> 3) & 1)) << 4);
  const int w_fr = 2 * kPPlane + rb * kPRowB + ((h ^ ((rb >> 3) & 1)) << 4);

  // The epilogue's operands in the layout of its fp32 pass (thread = 4 columns x 16 rows: bias, LayerNorm column sums, the residual
  // rows), requested NOW: fetched in the epilogue they were a chain of exposed latencies behind the loop (in-kernel stamps: 4,300
  // cycles for the bias of a mid-grid out-projection tile, 13,000 for its residual pass); here they land under the prologue's DMAs.
  const int e_c4 = tid & 31, e_rsub = tid >> 5, e_gcol = n0 + 4 * e_c4;
  const bool e_cols = e_gcol + 3 < N;
  const bool kv_tile = (OUT & 4) && n0 >= split_col;
  float4 e_bv = make_float4(0.f, 0.f, 0.f, 0.f), e_cs = e_bv, e_r1[16];
#pragma unroll
  for (int pass = 0; pass < 16; ++pass) e_r1[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!kv_tile && (OUT & 1)) {
    if (bias && e_cols) e_bv = *reinterpret_cast<const float4*>(bias + e_gcol);
    if (LN && e_cols) e_cs = *reinterpret_cast<const float4*>(ln_colsum + e_gcol);
    if (R1 && e_cols) {
#pragma unroll
      for (int pass = 0; pass < 16; ++pass) {
        const int grow = m0 + pass * 8 + e_rsub;
        if (grow < M) e_r1[pass] = *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + e_gcol);
      }
    }
  }
  float2 ln_mr = make_float2(0.f, 1.f);
  LnRaw<4> ls4;      // the row's statistics (four tiles, or sixteen), requested now, used behind the prologue's DMAs
  LnRaw<16> ls16;
  if (LN && tid < kPM && ln_tiles == 4) ln_raw_load(ls4, ln_stats, min(m0 + tid, M - 1));
  if (LN && tid < kPM && ln_tiles == 16) ln_raw_load(ls16, ln_stats, min(m0 + tid, M - 1));

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  half8 fa[2][2], fb[2][2], ga_[2][2], gb_[2][2];          // [plane][tile]
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_M(acc, A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc, 0, 0, 0); PFHIP_SB;
#define PFHIP_RA(G, st, p, i) if (PFHIP_P3_ABLATE < 3) G[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (st) * kPStage + (p) * kPPlane + a_fr + (i) * 32 * kPRowB));
#define PFHIP_RB(G, st, p, i) if (PFHIP_P3_ABLATE < 3) G[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (st) * kPStage + (p) * kPPlane + w_fr + (i) * 32 * kPRowB));
  // step: fragments FA / FB hold K-step k; read K-step k + 1 from stage `rst` into GA / GB; DMA K-step `kdma` into stage `wst`.
  // Per accumulator: a_hi w_lo, a_lo w_hi, a_hi w_hi (the order of gemm_x3.hip).
#ifndef PFHIP_P3_SPREAD
#define PFHIP_P3_SPREAD 1      // the four DMA pieces of a K-step between its MFMAs (0: all four at the top of the step)
#endif
#define PFHIP_D1(i, wst, kdma)                                                                                        \
  if (PFHIP_P3_SPREAD && PFHIP_P3_ABLATE != 2 && PFHIP_P3_ABLATE != 4) {                                              \
    if (i == 0) { PFHIP_DMA1(gah + (size_t)PFHIP_KA(kdma) * ka, wst, 0) }                                             \
    if (i == 1 && PFHIP_P3_ABLATE != 8) { PFHIP_DMA1(gal + (size_t)PFHIP_KA(kdma) * ka, wst, 1) }                     \
    if (i == 2) { PFHIP_DMA1(gwh + (size_t)PFHIP_KW(kdma) * kw, wst, 2) }                                             \
    if (i == 3 && PFHIP_P3_ABLATE != 8) { PFHIP_DMA1(gwl + (size_t)PFHIP_KW(kdma) * kw, wst, 3) }                     \
    PFHIP_SB;                                                                                                         \
  }
#define PFHIP_STEP(FA, FB, GA, GB, wst, rst, kdma)                                                                    \
  {                                                                                                                   \
    PFHIP_STAMP                                                                                                       \
    if (!PFHIP_P3_SPREAD && PFHIP_P3_ABLATE != 2 && PFHIP_P3_ABLATE != 4) { PFHIP_DMA(wst, kdma) } PFHIP_SB;          \
    PFHIP_STAMP                                                                                                       \
    PFHIP_M(acc00, FA[0][0], FB[1][0]) PFHIP_RA(GA, rst, 0, 0) PFHIP_SB;                                              \
    PFHIP_D1(0, wst, kdma)                                                                                            \
    PFHIP_M(acc01, FA[0][0], FB[1][1]) PFHIP_RB(GB, rst, 0, 0) PFHIP_SB;                                              \
    PFHIP_M(acc10, FA[0][1], FB[1][0]) PFHIP_RA(GA, rst, 0, 1) PFHIP_SB;                                              \
    PFHIP_M(acc11, FA[0][1], FB[1][1]) PFHIP_RB(GB, rst, 0, 1) PFHIP_SB;                                              \
    PFHIP_D1(1, wst, kdma)                                                                                            \
    PFHIP_M(acc00, FA[1][0], FB[0][0]) PFHIP_RA(GA, rst, 1, 0) PFHIP_SB;                                              \
    PFHIP_M(acc01, FA[1][0], FB[0][1]) PFHIP_RB(GB, rst, 1, 0) PFHIP_SB;                                              \
    PFHIP_M(acc10, FA[1][1], FB[0][0]) PFHIP_RA(GA, rst, 1, 1) PFHIP_SB;                                              \
    PFHIP_D1(2, wst, kdma)                                                                                            \
    PFHIP_M(acc11, FA[1][1], FB[0][1]) PFHIP_RB(GB, rst, 1, 1) PFHIP_SB;                                              \
    PFHIP_M(acc00, FA[0][0], FB[0][0])                                                                                \
    PFHIP_M(acc01, FA[0][0], FB[0][1])                                                                                \
    PFHIP_D1(3, wst, kdma)                                                                                            \
    PFHIP_M(acc10, FA[0][1], FB[0][0])                                                                                \
    PFHIP_M(acc11, FA[0][1], FB[0][1])                                                                                \
    PFHIP_STAMP                                                                                                       \
    if (PFHIP_P3_ABLATE == 1 || PFHIP_P3_ABLATE == 4) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");     \
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                     \
    PFHIP_STAMP                                                                                                       \
    PFHIP_SB;                                                                                                         \
  }

  const int nk = K / kPK;
  auto kclamp = [&](int t) { return t < nk ? t : nk - 1; };
  PFHIP_STAMP
  PFHIP_DMA(0, 0)
  PFHIP_DMA(1, kclamp(1))
  PFHIP_DMA(2, kclamp(2))
  PFHIP_DMA(3, kclamp(3))
  PFHIP_STAMP
  if (LN && tid < kPM) ln_mr = ln_tiles == 4 ? ln_row_stats_raw(ls4, ln_eps, range_flag) : ln_tiles == 16 ? ln_row_stats_raw(ls16, ln_eps, range_flag) : ln_row_stats(ln_stats, ln_tiles, ln_eps, min(m0 + tid, M - 1), range_flag);
  asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");          // K-steps 0 and 1 have landed, for every wave
  PFHIP_STAMP
  PFHIP_RA(fa, 0, 0, 0) PFHIP_RA(fa, 0, 0, 1) PFHIP_RA(fa, 0, 1, 0) PFHIP_RA(fa, 0, 1, 1)
  PFHIP_RB(fb, 0, 0, 0) PFHIP_RB(fb, 0, 0, 1) PFHIP_RB(fb, 0, 1, 0) PFHIP_RB(fb, 0, 1, 1)
  if (PFHIP_P3_ABLATE >= 3) {          // timing-only builds: defined register contents
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + p * kPPlane + a_fr + i * 32 * kPRowB));
        fb[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + p * kPPlane + w_fr + i * 32 * kPRowB));
        ga_[p][i] = fa[p][i]; gb_[p][i] = fb[p][i];
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // stage 0 is free for the DMA of K-step 4
  PFHIP_STAMP      // (stamp 4: the loop starts; four stamps per K-step follow)
  // step k: fragments f / g by parity; DMA K-step k + 4 into stage k % 4 (whose fragments this step holds in registers); read stage
  // (k + 1) % 4.  A DMA has TWO full steps to land (memory latency under load is longer than one step: the three-stage form of this
  // loop waited on vmcnt every step).
#define PFHIP_S0(kt) PFHIP_STEP(fa, fb, ga_, gb_, 0, 1, kclamp((kt) + 4))
#define PFHIP_S1(kt) PFHIP_STEP(ga_, gb_, fa, fb, 1, 2, kclamp((kt) + 5))
#define PFHIP_S2(kt) PFHIP_STEP(fa, fb, ga_, gb_, 2, 3, kclamp((kt) + 6))
#define PFHIP_S3(kt) PFHIP_STEP(ga_, gb_, fa, fb, 3, 0, kclamp((kt) + 7))
  int kt = 0;
  for (; kt + 3 < nk; kt += 4) { PFHIP_S0(kt) PFHIP_S1(kt) PFHIP_S2(kt) PFHIP_S3(kt) }
  if (kt < nk) PFHIP_S0(kt)
  if (kt + 1 < nk) PFHIP_S1(kt)
  if (kt + 2 < nk) PFHIP_S2(kt)
#undef PFHIP_S0
#undef PFHIP_S1
#undef PFHIP_S2
#undef PFHIP_S3
#undef PFHIP_STEP
#undef PFHIP_D1
#undef PFHIP_M
#undef PFHIP_RA
#undef PFHIP_RB
#undef PFHIP_DMA
#undef PFHIP_DMA1
#undef PFHIP_SB
  PFHIP_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the last (redundant) DMAs must not land in the C tile
  __syncthreads();
  PFHIP_STAMP
#ifndef PFHIP_P3_EPI
#define PFHIP_P3_EPI 0      // timing-only builds: 1 epilogue without its global stores, 2 no epilogue at all
#endif
  if (PFHIP_P3_EPI == 2) {      // keep the accumulators alive behind a condition the host never makes true
    if (M < 0) C[tid] = acc00[0] + acc01[1] + acc10[2] + acc11[3];
    return;
  }
#ifndef PFHIP_P3_STAMPS
  const bool do_store = PFHIP_P3_EPI != 1 || M < 0;
#endif

  // ---- epilogue ------------------------------------------------------------------------------------------------------------------
  float* const Cs = reinterpret_cast<float*>(lds);
  {
    float* cw = Cs + (wr * 64 + 4 * h) * kPCs + wc * 64 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ro = ((e & 3) + 8 * (e >> 2)) * kPCs;
      cw[ro] = acc00[e];
      cw[ro + 32] = acc01[e];
      cw[ro + 32 * kPCs] = acc10[e];
      cw[ro + 32 * kPCs + 32] = acc11[e];
    }
  }
  float2* const s_mr = reinterpret_cast<float2*>(lds + kPM * kPCs * 4);
  if (LN && tid < kPM) s_mr[tid] = ln_mr;
  __syncthreads();
  PFHIP_STAMP
#ifdef PFHIP_P3_STAMPS
  const bool do_store = !stamp_wg;      // its C rows hold the stamps
#define PFHIP_P3_STAMP_END if (stamping) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); *dbg_at(n_stamp) = __builtin_amdgcn_s_memtime(); *dbg_at(n_stamp + 1) = __builtin_amdgcn_s_memrealtime(); }
#else
#define PFHIP_P3_STAMP_END
#endif

  if (kv_tile) {      // a K | V tile of the QKV projection: row-major planes, 16 lanes x 16 B per row and plane
    row_planes_tile<LN, kPM, kPThreads>(Cs, s_mr, m0, n0, M, N, inv_scale, bias, ln_colsum, Ph, Pl, rows_p, split_col, tid, do_store);
  } else if (OUT & 1) {
    // fp32 pass: 32 lanes x 16 B per row, the tile's values read from LDS in one go; the final values go to C and, when planes follow,
    // back into the LDS tile
    const int c4 = e_c4, rsub = e_rsub, gcol = e_gcol;
    float4 cv[16];
    float2 mrv[16];
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
      const int row = pass * 8 + rsub;
      cv[pass] = *reinterpret_cast<const float4*>(Cs + row * kPCs + 4 * c4);
      mrv[pass] = LN ? s_mr[row] : make_float2(0.f, 1.f);
    }
#ifdef PFHIP_P3_STAMPS
    { float keep = cv[15].x + mrv[15].x; asm volatile("" : "+v"(keep)); }
    PFHIP_STAMP
#endif
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
      const int row = pass * 8 + rsub, grow = m0 + row;
      float4 v = cv[pass];
      v.x *= inv_scale; v.y *= inv_scale; v.z *= inv_scale; v.w *= inv_scale;
      if (LN) {
        const float2 mr = mrv[pass];
        v.x = mr.y * (v.x - mr.x * e_cs.x); v.y = mr.y * (v.y - mr.x * e_cs.y); v.z = mr.y * (v.z - mr.x * e_cs.z); v.w = mr.y * (v.w - mr.x * e_cs.w);
      }
      v.x += e_bv.x + e_r1[pass].x; v.y += e_bv.y + e_r1[pass].y; v.z += e_bv.z + e_r1[pass].z; v.w += e_bv.w + e_r1[pass].w;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (do_store && grow < M && e_cols) *reinterpret_cast<float4*>(C + (size_t)grow * ldc + gcol) = v;
      if (OUT & 2) *reinterpret_cast<float4*>(Cs + row * kPCs + 4 * c4) = v;
      if (stats_out) tile_row_stats(v, grow, M, tn, tiles_n, c4, stats_out);
    }
#ifdef PFHIP_P3_STAMPS
    PFHIP_STAMP
#endif
    if (OUT & 2) __syncthreads();
  }
  if (OUT & 2) {      // plane images of C: thread = (row, K-step of the consumer): 64 lanes = 64 consecutive rows = 2 KB per plane, contiguous
    const int row = tid & 127, grow = m0 + row;
    const float2 mr = (LN && OUT == 2) ? s_mr[row] : make_float2(0.f, 1.f);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = (tid >> 7) + 2 * jj;                    // 16-column group of the tile
      float v[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(Cs + row * kPCs + 16 * j + 4 * c);
        v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
      }
      if (OUT == 2) {     // planes only: the epilogue arithmetic happens here (the fp32 pass did not run)
        const int gc = n0 + 16 * j;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + gc + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 s4 = LN ? *reinterpret_cast<const float4*>(ln_colsum + gc + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = v[4 * c + e] * inv_scale;
            if (LN) x = mr.y * (x - mr.x * ss[e]);
            x += bb[e];
            if (relu) x = fmaxf(x, 0.f);
            v[4 * c + e] = x;
          }
        }
      }
      const int ksp = (n0 >> 4) + j;                        // K-step of the consumer this column group is
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const float w8[8] = {v[8 * pc], v[8 * pc + 1], v[8 * pc + 2], v[8 * pc + 3], v[8 * pc + 4], v[8 * pc + 5], v[8 * pc + 6], v[8 * pc + 7]};
        uint4 hh, ll;
        split8(w8, hh, ll);
        const size_t off = image_off(ksp, grow, pc, rows_p);
        if (do_store) {
          *reinterpret_cast<uint4*>(Ph + off) = hh;
          *reinterpret_cast<uint4*>(Pl + off) = ll;
        } else if (hh.x == 0x12345678u && ll.y == 0x9abcdef0u) {
          Ph[0] = 1;
        }
      }
    }
  }
  PFHIP_P3_STAMP_END
}
#undef PFHIP_P3_STAMP_END
#undef PFHIP_STAMP

// ---- the 256 x 128 tile (round 4): one workgroup of EIGHT waves per CU ---------------------------------------------------------------
// Timing-only builds of the 128 x 128 kernel above (tools/p3_probe.py, 16000 rows, us per launch QKV / out-projection / FFN1 / FFN2)
// say what its loop is bound by: as built 97.6 / 49.1 / 127.1 / 109.6; without the epilogue 81.0 / 24.8 / 95.8 / 91.8; MFMAs alone
// 60.1 / 19.3 / 72.1 / 64.8; every DMA re-reading K-step 0 (always an L2 hit) 72.9 / 25.3 / 91.4 / 82.5; only the HIGH planes copied
// (half the bytes) 66.6 / 20.9 / 82.2 / 74.8.  The time above the MFMAs is linear in the bytes the LDS-DMAs move and does not care
// where they come from: the loop is bound by the rate a CU takes operands INTO its LDS (MI355X_MICROARCH.md prices one 1-KB piece at
// 60-185 cycles of issue; a 128 x 128 tile asks for 16 KB per 48 MFMAs).  Dedicated loader waves (a 4 + 2-wave workgroup: the MFMA
// waves issue no memory instruction at all) were built and measured: +5..10 % — the same bytes through the same path.  What helps
// is fewer bytes per MFMA: this tile moves 24 KB per 96 MFMAs (0.75 x), the two workgroups of a CU become one of eight waves
// (wave = 64 x 64 as before: same fragment traffic, same MFMA order per accumulator, bit-identical results), and a launch needs 256
// tiles to fill the chip instead of 512 (the N = 512 GEMMs of a 32 x 30 s batch: 252).
constexpr int kQM = 256;
constexpr int kQThreads = 512;
constexpr int kQAPlane = kQM * kPRowB;                  // 8,192 B: one plane of the A operand of one stage
constexpr int kQWOff = 2 * kQAPlane;                    // the W planes (4 KB each) start here
constexpr int kQStage = kQWOff + 2 * kPPlane;           // 24,576 B
constexpr int kQLds = kPRing * kQStage;                 // 98,304 B (a 128-row half of the C tile + 256 rows of statistics need 69,632)
static_assert(kPM * kPCs * 4 + kQM * 8 <= kQLds, "half of the C tile and the row statistics must fit the ring");
template <bool LN, int OUT>
__global__ __launch_bounds__(kQThreads, 2) void gemm_p3_256_kernel(
    const unsigned char* __restrict__ Ah, const unsigned char* __restrict__ Al, int rows_a, const unsigned char* __restrict__ Wh,
    const unsigned char* __restrict__ Wl, int rows_w, float* C, int ldc, unsigned char* __restrict__ Ph, unsigned char* __restrict__ Pl,
    int rows_p, const float* __restrict__ bias, const float* R1, int ldr1, int M, int N, int K, int tiles_n, int n_tiles, int gw, int relu,
    const float* __restrict__ ln_stats, int ln_tiles, float ln_eps, const float* __restrict__ ln_colsum, float* __restrict__ stats_out,
    float inv_scale, int* range_flag, int split_col) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int tm, tn;
  tile_of_block_p3(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kQM, n0 = tn * kPN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;                 // 4 x 2 waves of 64 x 64
  const int r = lane & 31, h = lane >> 5;

  // DMA map: per K-step 24 pieces of 1 KB; wave w moves rows 32 w .. 32 w + 31 of A hi and of A lo, and chunk (w & 3) of W plane
  // (w >> 2).  The last row panel of a matrix whose row count is an odd multiple of 128 reaches past the A image: its source rows
  // are clamped into the image (those tile rows are computed on repeated data and never stored).
  const size_t ka = (size_t)rows_a * kPRowB, kw = (size_t)rows_w * kPRowB;
  const int arow = min(m0 + 32 * wave, rows_a - 32);
  const unsigned char* const gah = Ah + (size_t)arow * kPRowB + lane * 16;
  const unsigned char* const gal = Al + (size_t)arow * kPRowB + lane * 16;
  const unsigned char* const gww = (wave < 4 ? Wh : Wl) + ((size_t)(n0 + 32 * (wave & 3))) * kPRowB + lane * 16;
  const int lds_a = wave * 1024, lds_w = kQWOff + (wave >> 2) * kPPlane + (wave & 3) * 1024;
#define PFHIP_QDMA1(src, off)                                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                              \
                                   (__attribute__((address_space(3))) void*)(lds + (off)), 16, 0, 0);
#define PFHIP_QDMA(stage, ks)                                                                                         \
  PFHIP_QDMA1(gah + (size_t)(ks) * ka, (stage) * kQStage + lds_a)                                                     \
  PFHIP_QDMA1(gal + (size_t)(ks) * ka, (stage) * kQStage + kQAPlane + lds_a)                                          \
  PFHIP_QDMA1(gww + (size_t)(ks) * kw, (stage) * kQStage + lds_w)

  const int ra = wr * 64 + r, rb = wc * 64 + r;
  const int a_fr = ra * kPRowB + ((h ^ ((ra >> 3) & 1)) << 4);
  const int w_fr = kQWOff + rb * kPRowB + ((h ^ ((rb >> 3) & 1)) << 4);

  float2 ln_mr = make_float2(0.f, 1.f);
  if (LN && tid < kQM) ln_mr = ln_row_stats(ln_stats, ln_tiles, ln_eps, min(m0 + tid, M - 1), range_flag);

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  half8 fa[2][2], fb[2][2], ga_[2][2], gb_[2][2];          // [plane][tile]
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_M(acc, A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc, 0, 0, 0); PFHIP_SB;
#define PFHIP_RA(G, st, p, i) G[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (st) * kQStage + (p) * kQAPlane + a_fr + (i) * 32 * kPRowB));
#define PFHIP_RB(G, st, p, i) G[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (st) * kQStage + (p) * kPPlane + w_fr + (i) * 32 * kPRowB));
  // the step of the 128 x 128 kernel: a_hi w_lo, a_lo w_hi, a_hi w_hi per accumulator; three pieces per wave, so `vmcnt(6)` leaves
  // the two newest K-steps in flight
#define PFHIP_STEP(FA, FB, GA, GB, wst, rst, kdma)                                                                    \
  {                                                                                                                   \
    PFHIP_M(acc00, FA[0][0], FB[1][0]) PFHIP_RA(GA, rst, 0, 0) PFHIP_SB;                                              \
    PFHIP_QDMA1(gah + (size_t)(kdma) * ka, (wst) * kQStage + lds_a) PFHIP_SB;                                         \
    PFHIP_M(acc01, FA[0][0], FB[1][1]) PFHIP_RB(GB, rst, 0, 0) PFHIP_SB;                                              \
    PFHIP_M(acc10, FA[0][1], FB[1][0]) PFHIP_RA(GA, rst, 0, 1) PFHIP_SB;                                              \
    PFHIP_M(acc11, FA[0][1], FB[1][1]) PFHIP_RB(GB, rst, 0, 1) PFHIP_SB;                                              \
    PFHIP_M(acc00, FA[1][0], FB[0][0]) PFHIP_RA(GA, rst, 1, 0) PFHIP_SB;                                              \
    PFHIP_QDMA1(gal + (size_t)(kdma) * ka, (wst) * kQStage + kQAPlane + lds_a) PFHIP_SB;                              \
    PFHIP_M(acc01, FA[1][0], FB[0][1]) PFHIP_RB(GB, rst, 1, 0) PFHIP_SB;                                              \
    PFHIP_M(acc10, FA[1][1], FB[0][0]) PFHIP_RA(GA, rst, 1, 1) PFHIP_SB;                                              \
    PFHIP_M(acc11, FA[1][1], FB[0][1]) PFHIP_RB(GB, rst, 1, 1) PFHIP_SB;                                              \
    PFHIP_M(acc00, FA[0][0], FB[0][0])                                                                                \
    PFHIP_QDMA1(gww + (size_t)(kdma) * kw, (wst) * kQStage + lds_w) PFHIP_SB;                                         \
    PFHIP_M(acc01, FA[0][0], FB[0][1])                                                                                \
    PFHIP_M(acc10, FA[0][1], FB[0][0])                                                                                \
    PFHIP_M(acc11, FA[0][1], FB[0][1])                                                                                \
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
    PFHIP_SB;                                                                                                         \
  }

  const int nk = K / kPK;
  auto kclamp = [&](int t) { return t < nk ? t : nk - 1; };
  PFHIP_QDMA(0, 0)
  PFHIP_QDMA(1, kclamp(1))
  PFHIP_QDMA(2, kclamp(2))
  PFHIP_QDMA(3, kclamp(3))
  asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");          // K-steps 0 and 1 have landed, for every wave
  PFHIP_RA(fa, 0, 0, 0) PFHIP_RA(fa, 0, 0, 1) PFHIP_RA(fa, 0, 1, 0) PFHIP_RA(fa, 0, 1, 1)
  PFHIP_RB(fb, 0, 0, 0) PFHIP_RB(fb, 0, 0, 1) PFHIP_RB(fb, 0, 1, 0) PFHIP_RB(fb, 0, 1, 1)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // stage 0 is free for the DMA of K-step 4
#define PFHIP_S0(kt) PFHIP_STEP(fa, fb, ga_, gb_, 0, 1, kclamp((kt) + 4))
#define PFHIP_S1(kt) PFHIP_STEP(ga_, gb_, fa, fb, 1, 2, kclamp((kt) + 5))
#define PFHIP_S2(kt) PFHIP_STEP(fa, fb, ga_, gb_, 2, 3, kclamp((kt) + 6))
#define PFHIP_S3(kt) PFHIP_STEP(ga_, gb_, fa, fb, 3, 0, kclamp((kt) + 7))
  int kt = 0;
  for (; kt + 3 < nk; kt += 4) { PFHIP_S0(kt) PFHIP_S1(kt) PFHIP_S2(kt) PFHIP_S3(kt) }
  if (kt < nk) PFHIP_S0(kt)
  if (kt + 1 < nk) PFHIP_S1(kt)
  if (kt + 2 < nk) PFHIP_S2(kt)
#undef PFHIP_S0
#undef PFHIP_S1
#undef PFHIP_S2
#undef PFHIP_S3
#undef PFHIP_STEP
#undef PFHIP_M
#undef PFHIP_RA
#undef PFHIP_RB
#undef PFHIP_QDMA
#undef PFHIP_QDMA1
#undef PFHIP_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the last (redundant) DMAs must not land in the C tile
  __syncthreads();

  // ---- epilogue: the tile leaves in two halves of 128 rows through the LDS image of the 128 x 128 kernel (waves 0-3, then 4-7) ----
  float* const Cs = reinterpret_cast<float*>(lds);
  float2* const s_mr = reinterpret_cast<float2*>(lds + kPM * kPCs * 4);
  if (LN && tid < kQM) s_mr[tid] = ln_mr;
#pragma unroll 1
  for (int hf = 0; hf < 2; ++hf) {
    if ((wr >> 1) == hf) {
      float* cw = Cs + ((wr & 1) * 64 + 4 * h) * kPCs + wc * 64 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ro = ((e & 3) + 8 * (e >> 2)) * kPCs;
        cw[ro] = acc00[e];
        cw[ro + 32] = acc01[e];
        cw[ro + 32 * kPCs] = acc10[e];
        cw[ro + 32 * kPCs + 32] = acc11[e];
      }
    }
    __syncthreads();
    const int mh = m0 + 128 * hf;
    if (OUT & 1) {      // fp32 rows: 32 lanes x 16 B per row, 16 rows per pass
      const int c4 = tid & 31, rsub = tid >> 5;
      const int gcol = n0 + 4 * c4;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cs4 = bv;
      if (bias && gcol + 3 < N) bv = *reinterpret_cast<const float4*>(bias + gcol);
      if (LN && gcol + 3 < N) cs4 = *reinterpret_cast<const float4*>(ln_colsum + gcol);
      float4 r1v[8];
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int grow = mh + pass * 16 + rsub;
        r1v[pass] = (R1 && grow < M && gcol + 3 < N) ? *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 16 + rsub, grow = mh + row;
        float4 v = *reinterpret_cast<const float4*>(Cs + row * kPCs + 4 * c4);
        v.x *= inv_scale; v.y *= inv_scale; v.z *= inv_scale; v.w *= inv_scale;
        if (LN) {
          const float2 mr = s_mr[128 * hf + row];
          v.x = mr.y * (v.x - mr.x * cs4.x); v.y = mr.y * (v.y - mr.x * cs4.y); v.z = mr.y * (v.z - mr.x * cs4.z); v.w = mr.y * (v.w - mr.x * cs4.w);
        }
        v.x += bv.x + r1v[pass].x; v.y += bv.y + r1v[pass].y; v.z += bv.z + r1v[pass].z; v.w += bv.w + r1v[pass].w;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (grow < M && gcol + 3 < N) *reinterpret_cast<float4*>(C + (size_t)grow * ldc + gcol) = v;
        if (OUT & 2) *reinterpret_cast<float4*>(Cs + row * kPCs + 4 * c4) = v;
        if (stats_out) tile_row_stats(v, grow, M, tn, tiles_n, c4, stats_out);
      }
      if (OUT & 2) __syncthreads();
    }
    if (OUT & 2) {      // plane images of C: thread = (row, 16-column group): 64 lanes = 64 consecutive rows = 2 KB per plane, contiguous
      const int row = tid & 127, grow = mh + row;
      const float2 mr = (LN && OUT == 2) ? s_mr[128 * hf + row] : make_float2(0.f, 1.f);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = (tid >> 7) + 4 * jj;                  // 16-column group of the tile
        float v[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 t = *reinterpret_cast<const float4*>(Cs + row * kPCs + 16 * j + 4 * c);
          v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
        }
        if (OUT == 2) {     // planes only: the epilogue arithmetic happens here (the fp32 pass did not run)
          const int gc = n0 + 16 * j;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + gc + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 s4 = LN ? *reinterpret_cast<const float4*>(ln_colsum + gc + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = v[4 * c + e] * inv_scale;
              if (LN) x = mr.y * (x - mr.x * ss[e]);
              x += bb[e];
              if (relu) x = fmaxf(x, 0.f);
              v[4 * c + e] = x;
            }
          }
        }
        const int ksp = (n0 >> 4) + j;                      // K-step of the consumer this column group is
        if (grow < rows_p) {
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) {
            const float w8[8] = {v[8 * pc], v[8 * pc + 1], v[8 * pc + 2], v[8 * pc + 3], v[8 * pc + 4], v[8 * pc + 5], v[8 * pc + 6], v[8 * pc + 7]};
            uint4 hh, ll;
            split8(w8, hh, ll);
            const size_t off = image_off(ksp, grow, pc, rows_p);
            *reinterpret_cast<uint4*>(Ph + off) = hh;
            *reinterpret_cast<uint4*>(Pl + off) = ll;
          }
        }
      }
    }
    __syncthreads();                                        // the second half overwrites the image
  }
}

// The epilogue of 64 tile rows that sit in the LDS image Cs (row stride kPCs) with their LayerNorm (mean, rstd) in s_mr: K | V planes,
// fp32 rows, plane images, row statistics — the whole epilogue of the 64-row kernel, one half of the three-stage 128-row kernel's.
template <bool LN, int OUT>
__device__ __forceinline__ void epilogue_rows64(const int tid, float* const Cs, const float2* const s_mr, const int m0, const int n0, const int tn,
                                                const int tiles_n, float* C, int ldc, unsigned char* __restrict__ Ph, unsigned char* __restrict__ Pl,
                                                int rows_p, const float* __restrict__ bias, const float* R1, int ldr1, int M, int N, int relu,
                                                const float* __restrict__ ln_colsum, float* __restrict__ stats_out, float inv_scale, int split_col) {
  if ((OUT & 4) && n0 >= split_col) {
    row_planes_tile<LN, 64, kPThreads>(Cs, s_mr, m0, n0, M, N, inv_scale, bias, ln_colsum, Ph, Pl, rows_p, split_col, tid, true);
  } else if (OUT & 1) {
    const int c4 = tid & 31, rsub = tid >> 5;
    const int gcol = n0 + 4 * c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), cs4 = bv;
    if (bias && gcol + 3 < N) bv = *reinterpret_cast<const float4*>(bias + gcol);
    if (LN && gcol + 3 < N) cs4 = *reinterpret_cast<const float4*>(ln_colsum + gcol);
    float4 r1v[8];
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int grow = m0 + pass * 8 + rsub;
      r1v[pass] = (R1 && grow < M && gcol + 3 < N) ? *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 8 + rsub, grow = m0 + row;
      float4 v = *reinterpret_cast<const float4*>(Cs + row * kPCs + 4 * c4);
      v.x *= inv_scale; v.y *= inv_scale; v.z *= inv_scale; v.w *= inv_scale;
      if (LN) {
        const float2 mr = s_mr[row];
        v.x = mr.y * (v.x - mr.x * cs4.x); v.y = mr.y * (v.y - mr.x * cs4.y); v.z = mr.y * (v.z - mr.x * cs4.z); v.w = mr.y * (v.w - mr.x * cs4.w);
      }
      v.x += bv.x + r1v[pass].x; v.y += bv.y + r1v[pass].y; v.z += bv.z + r1v[pass].z; v.w += bv.w + r1v[pass].w;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (grow < M && gcol + 3 < N) *reinterpret_cast<float4*>(C + (size_t)grow * ldc + gcol) = v;
      if (OUT & 2) *reinterpret_cast<float4*>(Cs + row * kPCs + 4 * c4) = v;
      if (stats_out) tile_row_stats(v, grow, M, tn, tiles_n, c4, stats_out);
    }
    if (OUT & 2) __syncthreads();
  }
  if (OUT & 2) {      // plane images of C: 64 lanes = the tile's 64 rows = 2 KB per plane, contiguous
    const int row = tid & 63, grow = m0 + row;
    const float2 mr = (LN && OUT == 2) ? s_mr[row] : make_float2(0.f, 1.f);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = (tid >> 6) + 4 * jj;                    // 16-column group of the tile
      float v[16];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(Cs + row * kPCs + 16 * j + 4 * c);
        v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
      }
      if (OUT == 2) {
        const int gc = n0 + 16 * j;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + gc + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 s4 = LN ? *reinterpret_cast<const float4*>(ln_colsum + gc + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = v[4 * c + e] * inv_scale;
            if (LN) x = mr.y * (x - mr.x * ss[e]);
            x += bb[e];
            if (relu) x = fmaxf(x, 0.f);
            v[4 * c + e] = x;
          }
        }
      }
      const int ksp = (n0 >> 4) + j;
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const float w8[8] = {v[8 * pc], v[8 * pc + 1], v[8 * pc + 2], v[8 * pc + 3], v[8 * pc + 4], v[8 * pc + 5], v[8 * pc + 6], v[8 * pc + 7]};
        uint4 hh, ll;
        split8(w8, hh, ll);
        const size_t off = image_off(ksp, grow, pc, rows_p);
        *reinterpret_cast<uint4*>(Ph + off) = hh;
        *reinterpret_cast<uint4*>(Pl + off) = ll;
      }
    }
  }
}

// ---- the 64 x 128 tile: grids that leave most of a round of 128 x 128 tiles empty (a few utterances, the long-audio flow's packed
// forwards, streaming rounds).  Same loop, half the rows: 4 waves as 2 x 2 (each 32 x 64 = two MFMA tiles, six MFMAs and six
// fragment reads per K-step), a stage of 12 KB (A hi | A lo | W hi | W lo), ring of four = 48 KB: THREE workgroups per CU.
constexpr int kHM = 64;
constexpr int kHAPlane = kHM * kPRowB;                  // 2,048 B: one plane of the A operand of one stage
constexpr int kHWOff = 2 * kHAPlane;                    // the W planes start here
constexpr int kHStage = kHWOff + 2 * kPPlane;           // 12,288 B
constexpr int kHLds = kPRing * kHStage;                 // 49,152 B (the 64 x 132 C tile + row statistics need 34,304)
static_assert(kHM * kPCs * 4 + kHM * 8 <= kHLds, "the C tile must fit the ring");
template <bool LN, int OUT>
__global__ __launch_bounds__(kPThreads, 3) void gemm_p3_64_kernel(
    const unsigned char* __restrict__ Ah, const unsigned char* __restrict__ Al, int rows_a, const unsigned char* __restrict__ Wh,
    const unsigned char* __restrict__ Wl, int rows_w, float* C, int ldc, unsigned char* __restrict__ Ph, unsigned char* __restrict__ Pl,
    int rows_p, const float* __restrict__ bias, const float* R1, int ldr1, int M, int N, int K, int tiles_n, int n_tiles, int gw, int relu,
    const float* __restrict__ ln_stats, int ln_tiles, float ln_eps, const float* __restrict__ ln_colsum, float* __restrict__ stats_out,
    float inv_scale, int* range_flag, int split_col) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int tm, tn;
  tile_of_block_p3(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kHM, n0 = tn * kPN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // DMA map: 12 chunks of 1 KB per K-step, three per wave: one of the four A chunks (plane wave >> 1, rows 32 (wave & 1) ..), and
  // rows 32 wave .. of both W planes
  const size_t ka = (size_t)rows_a * kPRowB, kw = (size_t)rows_w * kPRowB;
  const unsigned char* const ga = ((wave >> 1) ? Al : Ah) + ((size_t)(m0 + 32 * (wave & 1))) * kPRowB + lane * 16;
  const unsigned char* const gwh = Wh + ((size_t)(n0 + 32 * wave)) * kPRowB + lane * 16;
  const unsigned char* const gwl = Wl + ((size_t)(n0 + 32 * wave)) * kPRowB + lane * 16;
  const int lds_a = (wave >> 1) * kHAPlane + (wave & 1) * 1024, lds_w = kHWOff + wave * 1024;
#define PFHIP_DMA1(src, off)                                                                                           \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                              \
                                   (__attribute__((address_space(3))) void*)(lds + (off)), 16, 0, 0);
#define PFHIP_DMA(stage, ks)                                                                                          \
  PFHIP_DMA1(ga + (size_t)(ks) * ka, (stage) * kHStage + lds_a) PFHIP_DMA1(gwh + (size_t)(ks) * kw, (stage) * kHStage + lds_w)     \
  PFHIP_DMA1(gwl + (size_t)(ks) * kw, (stage) * kHStage + lds_w + kPPlane)

  const int ra = wr * 32 + r, rb = wc * 64 + r;
  const int a_fr = ra * kPRowB + ((h ^ ((ra >> 3) & 1)) << 4);
  const int w_fr = kHWOff + rb * kPRowB + ((h ^ ((rb >> 3) & 1)) << 4);

  float2 ln_mr = make_float2(0.f, 1.f);
  LnRaw<4> ls4;
  LnRaw<16> ls16;
  if (LN && tid < kHM && ln_tiles == 4) ln_raw_load(ls4, ln_stats, min(m0 + tid, M - 1));
  if (LN && tid < kHM && ln_tiles == 16) ln_raw_load(ls16, ln_stats, min(m0 + tid, M - 1));

  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  half8 fa[2], fb[2][2], ga_[2], gb_[2][2];          // A [plane]; W [plane][tile]
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_M(acc, A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc, 0, 0, 0); PFHIP_SB;
#define PFHIP_RA(G, st, p) G[p] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (st) * kHStage + (p) * kHAPlane + a_fr));
#define PFHIP_RB(G, st, p, i) G[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (st) * kHStage + (p) * kPPlane + w_fr + (i) * 32 * kPRowB));
  // per accumulator: a_hi w_lo, a_lo w_hi, a_hi w_hi (the order of gemm_x3.hip and of the 128-row kernel: bit-identical results)
#define PFHIP_STEP(FA, FB, GA, GB, wst, rst, kdma)                                                                    \
  {                                                                                                                   \
    PFHIP_M(acc0, FA[0], FB[1][0]) PFHIP_RA(GA, rst, 0) PFHIP_SB;                                                     \
    PFHIP_DMA1(ga + (size_t)(kdma) * ka, (wst) * kHStage + lds_a) PFHIP_SB;                                           \
    PFHIP_M(acc1, FA[0], FB[1][1]) PFHIP_RB(GB, rst, 0, 0) PFHIP_SB;                                                  \
    PFHIP_M(acc0, FA[1], FB[0][0]) PFHIP_RB(GB, rst, 0, 1) PFHIP_SB;                                                  \
    PFHIP_DMA1(gwh + (size_t)(kdma) * kw, (wst) * kHStage + lds_w) PFHIP_SB;                                          \
    PFHIP_M(acc1, FA[1], FB[0][1]) PFHIP_RA(GA, rst, 1) PFHIP_SB;                                                     \
    PFHIP_M(acc0, FA[0], FB[0][0]) PFHIP_RB(GB, rst, 1, 0) PFHIP_SB;                                                  \
    PFHIP_DMA1(gwl + (size_t)(kdma) * kw, (wst) * kHStage + lds_w + kPPlane) PFHIP_SB;                                \
    PFHIP_M(acc1, FA[0], FB[0][1]) PFHIP_RB(GB, rst, 1, 1) PFHIP_SB;                                                  \
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
    PFHIP_SB;                                                                                                         \
  }

  const int nk = K / kPK;
  auto kclamp = [&](int t) { return t < nk ? t : nk - 1; };
  PFHIP_DMA(0, 0)
  PFHIP_DMA(1, kclamp(1))
  PFHIP_DMA(2, kclamp(2))
  PFHIP_DMA(3, kclamp(3))
  if (LN && tid < kHM) ln_mr = ln_tiles == 4 ? ln_row_stats_raw(ls4, ln_eps, range_flag) : ln_tiles == 16 ? ln_row_stats_raw(ls16, ln_eps, range_flag) : ln_row_stats(ln_stats, ln_tiles, ln_eps, min(m0 + tid, M - 1), range_flag);
  asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");          // K-steps 0 and 1 have landed, for every wave
  PFHIP_RA(fa, 0, 0) PFHIP_RA(fa, 0, 1)
  PFHIP_RB(fb, 0, 0, 0) PFHIP_RB(fb, 0, 0, 1) PFHIP_RB(fb, 0, 1, 0) PFHIP_RB(fb, 0, 1, 1)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // stage 0 is free for the DMA of K-step 4
#define PFHIP_S0(kt) PFHIP_STEP(fa, fb, ga_, gb_, 0, 1, kclamp((kt) + 4))
#define PFHIP_S1(kt) PFHIP_STEP(ga_, gb_, fa, fb, 1, 2, kclamp((kt) + 5))
#define PFHIP_S2(kt) PFHIP_STEP(fa, fb, ga_, gb_, 2, 3, kclamp((kt) + 6))
#define PFHIP_S3(kt) PFHIP_STEP(ga_, gb_, fa, fb, 3, 0, kclamp((kt) + 7))
  int kt = 0;
  for (; kt + 3 < nk; kt += 4) { PFHIP_S0(kt) PFHIP_S1(kt) PFHIP_S2(kt) PFHIP_S3(kt) }
  if (kt < nk) PFHIP_S0(kt)
  if (kt + 1 < nk) PFHIP_S1(kt)
  if (kt + 2 < nk) PFHIP_S2(kt)
#undef PFHIP_S0
#undef PFHIP_S1
#undef PFHIP_S2
#undef PFHIP_S3
#undef PFHIP_STEP
#undef PFHIP_M
#undef PFHIP_RA
#undef PFHIP_RB
#undef PFHIP_DMA
#undef PFHIP_DMA1
#undef PFHIP_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the last (redundant) DMAs must not land in the C tile
  __syncthreads();

  // ---- epilogue (as the 128-row kernel, half the rows) -------------------------------------------------------------------------------
  float* const Cs = reinterpret_cast<float*>(lds);
  {
    float* cw = Cs + (wr * 32 + 4 * h) * kPCs + wc * 64 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ro = ((e & 3) + 8 * (e >> 2)) * kPCs;
      cw[ro] = acc0[e];
      cw[ro + 32] = acc1[e];
    }
  }
  float2* const s_mr = reinterpret_cast<float2*>(lds + kHM * kPCs * 4);
  if (LN && tid < kHM) s_mr[tid] = ln_mr;
  __syncthreads();

  epilogue_rows64<LN, OUT>(tid, Cs, s_mr, m0, n0, tn, tiles_n, C, ldc, Ph, Pl, rows_p, bias, R1, ldr1, M, N, relu, ln_colsum, stats_out, inv_scale, split_col);
}

// ---- the 128 x 128 tile on a ring of THREE stages: three workgroups per CU (round 4, second half) ----------------------------------------
// In-kernel timelines of the four-stage kernel: a workgroup spends ~30 % of its life outside its K-loop (prologue, drain, C
// tile to LDS, epilogue arithmetic and stores) and a lone workgroup in the loop keeps the matrix pipe ~55 % busy — so with two workgroups
// per CU the pipe idles whenever one of them is not looping.  Here the ring is 3 x 16 KB and the C tile leaves in two 64-row halves
// through a 34-KB image (the epilogue of the 64-row kernel, twice): 48 KB of LDS, <= 168 registers, THREE workgroups per CU = three
// waves per SIMD.  A DMA has one full K-step to land instead of two (`vmcnt(4)`); same MFMA order per accumulator as the four-stage
// kernel: bit-identical results.
constexpr int kP3Lds = 3 * kPStage;      // 49,152 B (a 64-row half of the C tile + 128 rows of statistics need 34,816)
static_assert(64 * kPCs * 4 + kPM * 8 <= kP3Lds, "half of the C tile and the row statistics must fit the ring");
template <bool LN, int OUT>
__global__ __launch_bounds__(kPThreads, 3) void gemm_p3_128r3_kernel(
    const unsigned char* __restrict__ Ah, const unsigned char* __restrict__ Al, int rows_a, const unsigned char* __restrict__ Wh,
    const unsigned char* __restrict__ Wl, int rows_w, float* C, int ldc, unsigned char* __restrict__ Ph, unsigned char* __restrict__ Pl,
    int rows_p, const float* __restrict__ bias, const float* R1, int ldr1, int M, int N, int K, int tiles_n, int n_tiles, int gw, int relu,
    const float* __restrict__ ln_stats, int ln_tiles, float ln_eps, const float* __restrict__ ln_colsum, float* __restrict__ stats_out,
    float inv_scale, int* range_flag, int split_col) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  int tm, tn;
  tile_of_block_p3(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kPM, n0 = tn * kPN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // DMA map of the four-stage kernel: wave w moves rows 32 w .. 32 w + 31 of all four regions (A hi, A lo, W hi, W lo)
  const size_t ka = (size_t)rows_a * kPRowB, kw = (size_t)rows_w * kPRowB;
  const unsigned char* const gah = Ah + ((size_t)(m0 + 32 * wave)) * kPRowB + lane * 16;
  const unsigned char* const gal = Al + ((size_t)(m0 + 32 * wave)) * kPRowB + lane * 16;
  const unsigned char* const gwh = Wh + ((size_t)(n0 + 32 * wave)) * kPRowB + lane * 16;
  const unsigned char* const gwl = Wl + ((size_t)(n0 + 32 * wave)) * kPRowB + lane * 16;
  const int lds_c = wave * 1024;
#define PFHIP_DMA1(src, stage, region)                                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                              \
                                   (__attribute__((address_space(3))) void*)(lds + (stage) * kPStage + (region) * kPPlane + lds_c), 16, 0, 0);
#define PFHIP_DMA(stage, ks)                                                                                          \
  PFHIP_DMA1(gah + (size_t)(ks) * ka, stage, 0) PFHIP_DMA1(gal + (size_t)(ks) * ka, stage, 1)                         \
  PFHIP_DMA1(gwh + (size_t)(ks) * kw, stage, 2) PFHIP_DMA1(gwl + (size_t)(ks) * kw, stage, 3)

  const int ra = wr * 64 + r, rb = wc * 64 + r;
  const int a_fr = ra * kPRowB + ((h ^ ((ra >> 3) & 1)) << 4);
  const int w_fr = 2 * kPPlane + rb * kPRowB + ((h ^ ((rb >> 3) & 1)) << 4);

  float2 ln_mr = make_float2(0.f, 1.f);
  LnRaw<4> ls4;      // the row's statistics, requested now, used behind the prologue's DMAs
  LnRaw<16> ls16;
  if (LN && tid < kPM && ln_tiles == 4) ln_raw_load(ls4, ln_stats, min(m0 + tid, M - 1));
  if (LN && tid < kPM && ln_tiles == 16) ln_raw_load(ls16, ln_stats, min(m0 + tid, M - 1));

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  half8 fa[2][2], fb[2][2], ga_[2][2], gb_[2][2];          // [plane][tile]
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_M(acc, A_, B_) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, acc, 0, 0, 0); PFHIP_SB;
#define PFHIP_RA(G, st, p, i) G[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (st) * kPStage + (p) * kPPlane + a_fr + (i) * 32 * kPRowB));
#define PFHIP_RB(G, st, p, i) G[p][i] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (st) * kPStage + (p) * kPPlane + w_fr + (i) * 32 * kPRowB));
  // step k: fragments FA / FB hold K-step k; read K-step k + 1 from stage `rst` into GA / GB; DMA K-step `kdma` (= k + 3) into stage
  // `wst` (= k % 3, whose fragments this step holds in registers), one piece behind every third MFMA.  Per accumulator: a_hi w_lo,
  // a_lo w_hi, a_hi w_hi.
#define PFHIP_STEP(FA, FB, GA, GB, wst, rst, kdma)                                                                    \
  {                                                                                                                   \
    PFHIP_M(acc00, FA[0][0], FB[1][0]) PFHIP_RA(GA, rst, 0, 0) PFHIP_SB;                                              \
    PFHIP_DMA1(gah + (size_t)(kdma) * ka, wst, 0) PFHIP_SB;                                                           \
    PFHIP_M(acc01, FA[0][0], FB[1][1]) PFHIP_RB(GB, rst, 0, 0) PFHIP_SB;                                              \
    PFHIP_M(acc10, FA[0][1], FB[1][0]) PFHIP_RA(GA, rst, 0, 1) PFHIP_SB;                                              \
    PFHIP_M(acc11, FA[0][1], FB[1][1]) PFHIP_RB(GB, rst, 0, 1) PFHIP_SB;                                              \
    PFHIP_DMA1(gal + (size_t)(kdma) * ka, wst, 1) PFHIP_SB;                                                           \
    PFHIP_M(acc00, FA[1][0], FB[0][0]) PFHIP_RA(GA, rst, 1, 0) PFHIP_SB;                                              \
    PFHIP_M(acc01, FA[1][0], FB[0][1]) PFHIP_RB(GB, rst, 1, 0) PFHIP_SB;                                              \
    PFHIP_M(acc10, FA[1][1], FB[0][0]) PFHIP_RA(GA, rst, 1, 1) PFHIP_SB;                                              \
    PFHIP_DMA1(gwh + (size_t)(kdma) * kw, wst, 2) PFHIP_SB;                                                           \
    PFHIP_M(acc11, FA[1][1], FB[0][1]) PFHIP_RB(GB, rst, 1, 1) PFHIP_SB;                                              \
    PFHIP_M(acc00, FA[0][0], FB[0][0])                                                                                \
    PFHIP_M(acc01, FA[0][0], FB[0][1])                                                                                \
    PFHIP_DMA1(gwl + (size_t)(kdma) * kw, wst, 3) PFHIP_SB;                                                           \
    PFHIP_M(acc10, FA[0][1], FB[0][0])                                                                                \
    PFHIP_M(acc11, FA[0][1], FB[0][1])                                                                                \
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                          \
    PFHIP_SB;                                                                                                         \
  }

  const int nk = K / kPK;
  auto kclamp = [&](int t) { return t < nk ? t : nk - 1; };
  PFHIP_DMA(0, 0)
  PFHIP_DMA(1, kclamp(1))
  PFHIP_DMA(2, kclamp(2))
  if (LN && tid < kPM) ln_mr = ln_tiles == 4 ? ln_row_stats_raw(ls4, ln_eps, range_flag) : ln_tiles == 16 ? ln_row_stats_raw(ls16, ln_eps, range_flag) : ln_row_stats(ln_stats, ln_tiles, ln_eps, min(m0 + tid, M - 1), range_flag);
  asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");          // K-steps 0 and 1 have landed, for every wave
  PFHIP_RA(fa, 0, 0, 0) PFHIP_RA(fa, 0, 0, 1) PFHIP_RA(fa, 0, 1, 0) PFHIP_RA(fa, 0, 1, 1)
  PFHIP_RB(fb, 0, 0, 0) PFHIP_RB(fb, 0, 0, 1) PFHIP_RB(fb, 0, 1, 0) PFHIP_RB(fb, 0, 1, 1)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // stage 0 is free for the DMA of K-step 3
  // fragment sets alternate and stages cycle mod 3: the pattern repeats every six steps
#define PFHIP_T0(kt) PFHIP_STEP(fa, fb, ga_, gb_, 0, 1, kclamp((kt) + 3))
#define PFHIP_T1(kt) PFHIP_STEP(ga_, gb_, fa, fb, 1, 2, kclamp((kt) + 4))
#define PFHIP_T2(kt) PFHIP_STEP(fa, fb, ga_, gb_, 2, 0, kclamp((kt) + 5))
#define PFHIP_T3(kt) PFHIP_STEP(ga_, gb_, fa, fb, 0, 1, kclamp((kt) + 6))
#define PFHIP_T4(kt) PFHIP_STEP(fa, fb, ga_, gb_, 1, 2, kclamp((kt) + 7))
#define PFHIP_T5(kt) PFHIP_STEP(ga_, gb_, fa, fb, 2, 0, kclamp((kt) + 8))
  int kt = 0;
  for (; kt + 5 < nk; kt += 6) { PFHIP_T0(kt) PFHIP_T1(kt) PFHIP_T2(kt) PFHIP_T3(kt) PFHIP_T4(kt) PFHIP_T5(kt) }
  if (kt < nk) PFHIP_T0(kt)
  if (kt + 1 < nk) PFHIP_T1(kt)
  if (kt + 2 < nk) PFHIP_T2(kt)
  if (kt + 3 < nk) PFHIP_T3(kt)
  if (kt + 4 < nk) PFHIP_T4(kt)
#undef PFHIP_T0
#undef PFHIP_T1
#undef PFHIP_T2
#undef PFHIP_T3
#undef PFHIP_T4
#undef PFHIP_T5
#undef PFHIP_STEP
#undef PFHIP_M
#undef PFHIP_RA
#undef PFHIP_RB
#undef PFHIP_DMA
#undef PFHIP_DMA1
#undef PFHIP_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the last (redundant) DMAs must not land in the C tile
  __syncthreads();

  // ---- epilogue: the tile leaves in two halves of 64 rows (waves 0-1, then 2-3) through the 64-row image ------------------------------
  float* const Cs = reinterpret_cast<float*>(lds);
  float2* const s_mr = reinterpret_cast<float2*>(lds + 64 * kPCs * 4);
  if (LN && tid < kPM) s_mr[tid] = ln_mr;
#pragma unroll 1
  for (int hf = 0; hf < 2; ++hf) {
    if (wr == hf) {
      float* cw = Cs + (4 * h) * kPCs + wc * 64 + r;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ro = ((e & 3) + 8 * (e >> 2)) * kPCs;
        cw[ro] = acc00[e];
        cw[ro + 32] = acc01[e];
        cw[ro + 32 * kPCs] = acc10[e];
        cw[ro + 32 * kPCs + 32] = acc11[e];
      }
    }
    __syncthreads();
    epilogue_rows64<LN, OUT>(tid, Cs, s_mr + 64 * hf, m0 + 64 * hf, n0, tn, tiles_n, C, ldc, Ph, Pl, rows_p, bias, R1, ldr1, M, N, relu, ln_colsum,
                             stats_out, inv_scale, split_col);
    __syncthreads();                                        // the second half overwrites the image
  }
}

template <auto kern, int threads = kPThreads, class... Args>
void launch_with_lds(int n_tiles, int lds_bytes, hipStream_t s, Args... args) {
  static std::atomic<unsigned long long> attr_done{0};      // > 64 KB of dynamic LDS needs the opt-in once per kernel and device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_done.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_done.fetch_or(1ull << (dev & 63));
  }
  hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(threads), lds_bytes, s, args...);
}

}  // namespace

size_t plane_image_bytes(int rows, int K) { return (size_t)((rows + 127) / 128 * 128) * (size_t)K * 2; }

void launch_split_planes(const float* X, int ld, int rows_valid, int rows, int K, float scale, void* hi, void* lo, hipStream_t s) {
  if (rows <= 0 || K <= 0) return;
  const size_t n = (size_t)rows * (K / 8);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, ld, rows_valid, rows, K, scale,
                     static_cast<unsigned char*>(hi), static_cast<unsigned char*>(lo));
}

#ifndef PFHIP_P3_LDS_PAD
#define PFHIP_P3_LDS_PAD 0      // timing-only builds: extra LDS per workgroup of the 128-row kernel (32768: ONE workgroup per CU)
#endif
void launch_gemm_p3(const void* Ah, const void* Al, int rows_a, const void* Wh, const void* Wl, int rows_w, float w_scale, float* C, int ldc,
                    void* Ph, void* Pl, int rows_p, const float* bias, const float* R1, int ldr1, int M, int N, int K, bool relu,
                    const float* ln_stats, int ln_tiles, const float* ln_colsum, float* stats_out, int gw, hipStream_t s, int tile_rows,
                    int row_planes_from) {
  if (M <= 0 || N <= 0) return;
  const int tiles_n = (N + kPN - 1) / kPN;
  // 64-row tiles (three workgroups per CU) when 128-row tiles would leave most of a round of 512 slots empty
  static const int half_env = [] { const char* e = getenv("PFHIP_P3_HALF_TILES"); return e && *e ? atoi(e) : 400; }();
  const bool half = tile_rows == kHM || (tile_rows != kPM && tile_rows != kQM && ((M + kPM - 1) / kPM) * tiles_n <= half_env);
  // 256-row tiles (one eight-wave workgroup per CU) where they fill the chip: at least 85 % of the CU slots of their rounds
  // (32 x 30 s: 252 / 756 / 1008 tiles for N = 512 / 1536 / 2048); PFHIP_P3_TILE256=0 keeps the 128-row kernel
  static const bool q_env = [] { const char* e = getenv("PFHIP_P3_TILE256"); return !(e && e[0] == '0'); }();
  const int nq = ((M + kQM - 1) / kQM) * tiles_n;
  // Measured (tools/p3_probe.py, 16000 rows, two runs, 256-row tile against 128-row tile): FFN2 (K = 2048) 108.4 / 108.6 against
  // 110.7 / 111.3 us, QKV 98.0 / 96.3 against 97.9 / 100.3, FFN1 129.7 / 129.8 against 126.6 / 127.3, out-projection 51.8 / 50.3
  // against 47.5 / 47.1: what the loop gains from 0.75 x the bytes the single workgroup loses in its epilogue, which no second
  // workgroup covers.  Taken where the loop is long (K >= 1024); PFHIP_P3_TILE256=2 takes it wherever it fills the chip.
  static const bool q_all = [] { const char* e = getenv("PFHIP_P3_TILE256"); return e && e[0] == '2'; }();
  const bool quad = row_planes_from <= 0 && (tile_rows == kQM || (tile_rows == 0 && q_env && !half && rows_a >= kQM && (K >= 1024 || q_all) &&
                                         100 * nq >= 85 * 256 * ((nq + 255) / 256)));
  // the three-stage 128-row kernel (three workgroups per CU) for grids of more than two rounds of its 768 slots.  Measured
  // (tools/p3_probe.py, 16000 rows, two same-session pairs): FFN1' (2000 tiles) 119.7 / 121.5 us against 123.8 / 122.6 on the four-stage
  // kernel, QKV' (1500 tiles) 100.2 / 99.0 against 98.3 / 96.3, out-projection (500) level — a third workgroup per CU does not lift the
  // loop (it is the CU's operand path that is busy), it only smooths the rounds of a long grid.  PFHIP_P3_R3=0 / 1:
  // never / wherever the 128-row tile is taken.
  const int r3_env = [] { const char* e = getenv("PFHIP_P3_R3"); return e && *e ? atoi(e) : -1; }();      // (read per launch: tests switch it)
  const bool ring3 = !quad && !half && (r3_env == 1 || (r3_env != 0 && ((M + kPM - 1) / kPM) * tiles_n > 1536));
  const int tmr = quad ? kQM : (half ? kHM : kPM);
  const int n_tiles = ((M + tmr - 1) / tmr) * tiles_n;
  static const int gw_env = [] { const char* e = getenv("PFHIP_P3_GW"); return e && *e ? atoi(e) : 0; }();      // experiments

  if (gw_env > 0) gw = gw_env;
  gw = std::max(1, std::min(gw, tiles_n));
  const int out = (C ? 1 : 0) | (Ph ? 2 : 0);
  const float inv = 1.0f / w_scale;
  const unsigned char *ah = static_cast<const unsigned char*>(Ah), *al = static_cast<const unsigned char*>(Al);
  const unsigned char *wh = static_cast<const unsigned char*>(Wh), *wl = static_cast<const unsigned char*>(Wl);
  unsigned char *ph = static_cast<unsigned char*>(Ph), *pl = static_cast<unsigned char*>(Pl);
#define PFHIP_P3(LNF, OUTM)                                                                                                     \
  {                                                                                                                             \
    if (quad)                                                                                                                   \
      launch_with_lds<gemm_p3_256_kernel<LNF, OUTM>, kQThreads>(n_tiles, kQLds, s, ah, al, rows_a, wh, wl, rows_w, C, ldc, ph, pl, rows_p, bias, R1, ldr1, \
                                                    M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0, ln_stats, ln_tiles, 1e-12f, ln_colsum, stats_out, inv, launch_ctx().range_flag, row_planes_from); \
    else if (half)                                                                                                              \
      launch_with_lds<gemm_p3_64_kernel<LNF, OUTM>>(n_tiles, kHLds, s, ah, al, rows_a, wh, wl, rows_w, C, ldc, ph, pl, rows_p, bias, R1, ldr1, \
                                                    M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0, ln_stats, ln_tiles, 1e-12f, ln_colsum, stats_out, inv, launch_ctx().range_flag, row_planes_from); \
    else if (ring3)                                                                                                             \
      launch_with_lds<gemm_p3_128r3_kernel<LNF, OUTM>>(n_tiles, kP3Lds, s, ah, al, rows_a, wh, wl, rows_w, C, ldc, ph, pl, rows_p, bias, R1, ldr1, \
                                                     M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0, ln_stats, ln_tiles, 1e-12f, ln_colsum, stats_out, inv, launch_ctx().range_flag, row_planes_from); \
    else                                                                                                                        \
      launch_with_lds<gemm_p3_128_kernel<LNF, OUTM>>(n_tiles, kPLds + PFHIP_P3_LDS_PAD, s, ah, al, rows_a, wh, wl, rows_w, C, ldc, ph, pl, rows_p, bias, R1, ldr1, \
                                                     M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0, ln_stats, ln_tiles, 1e-12f, ln_colsum, stats_out, inv, launch_ctx().range_flag, row_planes_from); \
  }
  if (row_planes_from > 0) {      // the QKV projection: fp32 Q | row-major K, V planes (C and Ph both given)
#define PFHIP_P3S(LNF)                                                                                                          \
  {                                                                                                                             \
    if (half)                                                                                                                   \
      launch_with_lds<gemm_p3_64_kernel<LNF, 5>>(n_tiles, kHLds, s, ah, al, rows_a, wh, wl, rows_w, C, ldc, ph, pl, rows_p, bias, R1, ldr1, \
                                                 M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0, ln_stats, ln_tiles, 1e-12f, ln_colsum, stats_out, inv, launch_ctx().range_flag, row_planes_from); \
    else                                                                                                                        \
      launch_with_lds<gemm_p3_128_kernel<LNF, 5>>(n_tiles, kPLds, s, ah, al, rows_a, wh, wl, rows_w, C, ldc, ph, pl, rows_p, bias, R1, ldr1, \
                                                  M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0, ln_stats, ln_tiles, 1e-12f, ln_colsum, stats_out, inv, launch_ctx().range_flag, row_planes_from); \
  }
    if (ln_stats) PFHIP_P3S(true) else PFHIP_P3S(false)
#undef PFHIP_P3S
    return;
  }
  if (ln_stats) {
    if (out == 1) PFHIP_P3(true, 1) else if (out == 2) PFHIP_P3(true, 2) else PFHIP_P3(true, 3)
  } else {
    if (out == 1) PFHIP_P3(false, 1) else if (out == 2) PFHIP_P3(false, 2) else PFHIP_P3(false, 3)
  }
#undef PFHIP_P3
}

}  // namespace pfhip
