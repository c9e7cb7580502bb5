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
// buffers, next tile's global loads in flight under the current tile's 64 MFMAs per wave, every memory
// instruction issued singly between two MFMAs (see the loop).
// A lane reads 4 consecutive k of its row with one ds_read_b128; lane half h takes k = 8*kb+4*h+kk
// at MFMA step kk, for A and B alike, so the k permutation cancels.
//
// Epilogue: the 32x32 accumulator layout has ONE column per lane, so storing from registers means 64
// four-byte stores (and 64+64 four-byte residual loads) per lane — store-issue-bound, measured at
// ~25 % of the kernel.  The tile is instead transposed through the (now idle) LDS and written as
// full 512-byte rows with 16-byte accesses; bias / residual / ReLU are applied on the row pass.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kLds = 36;                       // padded operand row stride (floats)
constexpr int kStage = kTileM * kLds;          // floats per operand per buffer
constexpr int kCs = kTileN + 4;                // padded C-tile row stride (floats)
constexpr int kLdsFloats = 4 * kStage;         // 73,728 B: A0 A1 B0 B1; the C tile (128 x 132) reuses it
static_assert(kTileM * kCs <= kLdsFloats, "C tile must fit the operand buffers");

// XCD-aware tile order (cdna_hip_programming.md T1, bijective form): blocks that share an XCD (equal blockIdx % 8) walk a
// contiguous run of a linear tile order.  That order is column-GROUP major: group g = column tiles [g*gw, (g+1)*gw), inside
// a group row panel by row panel, n fastest — so the gw weight tiles of a group (<= 2 MiB, the host picks gw) stay in the
// XCD's 4 MB L2 while its row panels stream through, instead of the whole weight matrix being re-fetched for every
// handful of row panels.  Same time, 35-70 % less L2->fabric traffic on the wide GEMMs (tools/probe/gemm_sched.hip + PMC).
__device__ __forceinline__ void tile_of_block(int bid, int n_tiles, int tiles_n, int gw, int& tm, int& tn) {
  {
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tiles_m = n_tiles / tiles_n, full = tiles_n / gw, span = tiles_m * gw;
  if (bid < full * span) {
    const int g = bid / span, j = bid - g * span;
    tm = j / gw; tn = g * gw + (j - tm * gw);
  } else {                                        // the last, narrower group
    const int j = bid - full * span, w = tiles_n - full * gw;
    tm = j / w; tn = full * gw + (j - tm * w);
  }
}

template <bool GUARD, bool HAS_BIAS, bool HAS_R1, bool HAS_R2, bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C, int ldc,
    const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N,
    int K, int tiles_n, int n_tiles, int gw) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  float* const As = lds;                  // [2][128][36]
  float* const Bs = lds + 2 * kStage;     // [2][128][36]

  int tm, tn;
  tile_of_block(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kTileM, n0 = tn * kTileN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int lrow = tid >> 3, lc4 = tid & 7;

  const float* Ag = A + (size_t)(m0 + lrow) * lda + 4 * lc4;
  const float* Wg = W + (size_t)(n0 + lrow) * ldw + 4 * lc4;

  // staging registers are named scalars (not arrays captured by lambdas): hipcc keeps them in VGPRs.
  // Two sets (r*, s*): the global loads run two K-tiles ahead of the MFMAs.
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
  float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;          // operand fragments: even k-blocks in f, odd in g
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_GL(reg, base, ld, j, k0) reg = *reinterpret_cast<const float4*>(base + (size_t)(32 * (j)) * ld + (k0))
#define PFHIP_SW(reg, base, buf, j) \
  *reinterpret_cast<float4*>(base + (buf) * kStage + lrow * kLds + 4 * lc4 + 32 * (j) * kLds) = reg
#define PFHIP_FR(reg, base, off, buf, kb, j) \
  reg = *reinterpret_cast<const float4*>(base + (buf) * kStage + off + (kb) * 8 + 32 * (j) * kLds)
#define PFHIP_MM(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

  // Software pipeline, one barrier per K-tile.  Per wave a K-tile is 64 MFMAs (4 k-blocks x 16) plus 8 global loads
  // (the tile two ahead), 16 ds_read_b128 (fragments of the next k-block) and 8 ds_write_b128 (the next tile, loaded
  // one iteration earlier, into the other LDS buffer).  The memory instructions are issued ONE AT A TIME between
  // single MFMAs, every statement pinned by a scheduling barrier: issued in bursts between groups of 16 MFMAs (the
  // obvious layout) the same kernel is 15 % slower — each ds_read/global_load holds the wave's issue port long enough
  // that a burst of 4-8 lets the matrix pipe run dry.  Prefetch distance 2 plus LDS writes as late as the dependences
  // allow (barrier in the middle of k-block 3) are worth another +3..12 % (interleaved A/B runs, warm clocks).
  // The two bodies are generated (tools/probe/gen_gemm_loop.py, schedule "H"; tools/probe/gemm_sched.hip compares
  // schedules): _0 computes from LDS buffer 0 (even K-tiles), loads into r*, stores s*; _1 the mirror image.
  //   k-block 0 | global loads, reads of k-block 1;  k-block 1 | reads of k-block 2;
  //   k-block 2 | reads of k-block 3, first half of the LDS writes;
  //   k-block 3 | second half of the LDS writes, barrier, reads of k-block 0 of the next tile.
#define PFHIP_BODY_0 \
  PFHIP_MM(acc00, fa0.x, fb0.x); PFHIP_SB; PFHIP_GL(ra0, Ag, lda, 0, knext); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.x, fb1.x); PFHIP_SB; PFHIP_GL(ra1, Ag, lda, 1, knext); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.x, fb0.x); PFHIP_SB; PFHIP_GL(ra2, Ag, lda, 2, knext); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.x, fb1.x); PFHIP_SB; PFHIP_GL(ra3, Ag, lda, 3, knext); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.y, fb0.y); PFHIP_SB; PFHIP_GL(rb0, Wg, ldw, 0, knext); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.y, fb1.y); PFHIP_SB; PFHIP_GL(rb1, Wg, ldw, 1, knext); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.y, fb0.y); PFHIP_SB; PFHIP_GL(rb2, Wg, ldw, 2, knext); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.y, fb1.y); PFHIP_SB; PFHIP_GL(rb3, Wg, ldw, 3, knext); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.z, fb0.z); PFHIP_SB; PFHIP_FR(ga0, As, a_off, 0, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.z, fb1.z); PFHIP_SB; PFHIP_FR(ga1, As, a_off, 0, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.z, fb0.z); PFHIP_SB; PFHIP_FR(gb0, Bs, b_off, 0, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.z, fb1.z); PFHIP_SB; PFHIP_FR(gb1, Bs, b_off, 0, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.w, fb0.w); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.w, fb1.w); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.w, fb0.w); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.w, fb1.w); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.x, gb0.x); PFHIP_SB; PFHIP_FR(fa0, As, a_off, 0, 2, 0); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.x, gb1.x); PFHIP_SB; PFHIP_FR(fa1, As, a_off, 0, 2, 1); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.x, gb0.x); PFHIP_SB; PFHIP_FR(fb0, Bs, b_off, 0, 2, 0); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.x, gb1.x); PFHIP_SB; PFHIP_FR(fb1, Bs, b_off, 0, 2, 1); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.y, gb0.y); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.y, gb1.y); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.y, gb0.y); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.y, gb1.y); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.z, gb0.z); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.z, gb1.z); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.z, gb0.z); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.z, gb1.z); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.w, gb1.w); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.w, gb1.w); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.x, fb0.x); PFHIP_SB; PFHIP_FR(ga0, As, a_off, 0, 3, 0); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.x, fb1.x); PFHIP_SB; PFHIP_FR(ga1, As, a_off, 0, 3, 1); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.x, fb0.x); PFHIP_SB; PFHIP_FR(gb0, Bs, b_off, 0, 3, 0); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.x, fb1.x); PFHIP_SB; PFHIP_FR(gb1, Bs, b_off, 0, 3, 1); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.y, fb0.y); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.y, fb1.y); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.y, fb0.y); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.y, fb1.y); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.z, fb0.z); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.z, fb1.z); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.z, fb0.z); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.z, fb1.z); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.w, fb0.w); PFHIP_SB; PFHIP_SW(sa0, As, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.w, fb1.w); PFHIP_SB; PFHIP_SW(sa1, As, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.w, fb0.w); PFHIP_SB; PFHIP_SW(sa2, As, 1, 2); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.w, fb1.w); PFHIP_SB; PFHIP_SW(sa3, As, 1, 3); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.x, gb0.x); PFHIP_SB; PFHIP_SW(sb0, Bs, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.x, gb1.x); PFHIP_SB; PFHIP_SW(sb1, Bs, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.x, gb0.x); PFHIP_SB; PFHIP_SW(sb2, Bs, 1, 2); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.x, gb1.x); PFHIP_SB; PFHIP_SW(sb3, Bs, 1, 3); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.y, gb0.y); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.y, gb1.y); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.y, gb0.y); PFHIP_SB; __syncthreads(); \
  PFHIP_MM(acc11, ga1.y, gb1.y); PFHIP_SB; PFHIP_FR(fa0, As, a_off, 1, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.z, gb0.z); PFHIP_SB; PFHIP_FR(fa1, As, a_off, 1, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.z, gb1.z); PFHIP_SB; PFHIP_FR(fb0, Bs, b_off, 1, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.z, gb0.z); PFHIP_SB; PFHIP_FR(fb1, Bs, b_off, 1, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.z, gb1.z); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.w, gb1.w); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.w, gb1.w); PFHIP_SB;
#define PFHIP_BODY_1 \
  PFHIP_MM(acc00, fa0.x, fb0.x); PFHIP_SB; PFHIP_GL(sa0, Ag, lda, 0, knext); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.x, fb1.x); PFHIP_SB; PFHIP_GL(sa1, Ag, lda, 1, knext); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.x, fb0.x); PFHIP_SB; PFHIP_GL(sa2, Ag, lda, 2, knext); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.x, fb1.x); PFHIP_SB; PFHIP_GL(sa3, Ag, lda, 3, knext); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.y, fb0.y); PFHIP_SB; PFHIP_GL(sb0, Wg, ldw, 0, knext); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.y, fb1.y); PFHIP_SB; PFHIP_GL(sb1, Wg, ldw, 1, knext); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.y, fb0.y); PFHIP_SB; PFHIP_GL(sb2, Wg, ldw, 2, knext); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.y, fb1.y); PFHIP_SB; PFHIP_GL(sb3, Wg, ldw, 3, knext); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.z, fb0.z); PFHIP_SB; PFHIP_FR(ga0, As, a_off, 1, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.z, fb1.z); PFHIP_SB; PFHIP_FR(ga1, As, a_off, 1, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.z, fb0.z); PFHIP_SB; PFHIP_FR(gb0, Bs, b_off, 1, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.z, fb1.z); PFHIP_SB; PFHIP_FR(gb1, Bs, b_off, 1, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.w, fb0.w); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.w, fb1.w); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.w, fb0.w); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.w, fb1.w); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.x, gb0.x); PFHIP_SB; PFHIP_FR(fa0, As, a_off, 1, 2, 0); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.x, gb1.x); PFHIP_SB; PFHIP_FR(fa1, As, a_off, 1, 2, 1); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.x, gb0.x); PFHIP_SB; PFHIP_FR(fb0, Bs, b_off, 1, 2, 0); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.x, gb1.x); PFHIP_SB; PFHIP_FR(fb1, Bs, b_off, 1, 2, 1); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.y, gb0.y); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.y, gb1.y); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.y, gb0.y); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.y, gb1.y); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.z, gb0.z); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.z, gb1.z); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.z, gb0.z); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.z, gb1.z); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.w, gb1.w); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.w, gb1.w); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.x, fb0.x); PFHIP_SB; PFHIP_FR(ga0, As, a_off, 1, 3, 0); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.x, fb1.x); PFHIP_SB; PFHIP_FR(ga1, As, a_off, 1, 3, 1); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.x, fb0.x); PFHIP_SB; PFHIP_FR(gb0, Bs, b_off, 1, 3, 0); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.x, fb1.x); PFHIP_SB; PFHIP_FR(gb1, Bs, b_off, 1, 3, 1); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.y, fb0.y); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.y, fb1.y); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.y, fb0.y); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.y, fb1.y); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.z, fb0.z); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.z, fb1.z); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.z, fb0.z); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.z, fb1.z); PFHIP_SB; \
  PFHIP_MM(acc00, fa0.w, fb0.w); PFHIP_SB; PFHIP_SW(ra0, As, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc01, fa0.w, fb1.w); PFHIP_SB; PFHIP_SW(ra1, As, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc10, fa1.w, fb0.w); PFHIP_SB; PFHIP_SW(ra2, As, 0, 2); PFHIP_SB; \
  PFHIP_MM(acc11, fa1.w, fb1.w); PFHIP_SB; PFHIP_SW(ra3, As, 0, 3); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.x, gb0.x); PFHIP_SB; PFHIP_SW(rb0, Bs, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.x, gb1.x); PFHIP_SB; PFHIP_SW(rb1, Bs, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.x, gb0.x); PFHIP_SB; PFHIP_SW(rb2, Bs, 0, 2); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.x, gb1.x); PFHIP_SB; PFHIP_SW(rb3, Bs, 0, 3); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.y, gb0.y); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.y, gb1.y); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.y, gb0.y); PFHIP_SB; __syncthreads(); \
  PFHIP_MM(acc11, ga1.y, gb1.y); PFHIP_SB; PFHIP_FR(fa0, As, a_off, 0, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.z, gb0.z); PFHIP_SB; PFHIP_FR(fa1, As, a_off, 0, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.z, gb1.z); PFHIP_SB; PFHIP_FR(fb0, Bs, b_off, 0, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.z, gb0.z); PFHIP_SB; PFHIP_FR(fb1, Bs, b_off, 0, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.z, gb1.z); PFHIP_SB; \
  PFHIP_MM(acc00, ga0.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc01, ga0.w, gb1.w); PFHIP_SB; \
  PFHIP_MM(acc10, ga1.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc11, ga1.w, gb1.w); PFHIP_SB;

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }

  const int nk = K / kTileK;
  {
    int knext = 0;
    PFHIP_GL(ra0, Ag, lda, 0, knext); PFHIP_GL(ra1, Ag, lda, 1, knext); PFHIP_GL(ra2, Ag, lda, 2, knext); PFHIP_GL(ra3, Ag, lda, 3, knext);
    PFHIP_GL(rb0, Wg, ldw, 0, knext); PFHIP_GL(rb1, Wg, ldw, 1, knext); PFHIP_GL(rb2, Wg, ldw, 2, knext); PFHIP_GL(rb3, Wg, ldw, 3, knext);
    PFHIP_SW(ra0, As, 0, 0); PFHIP_SW(ra1, As, 0, 1); PFHIP_SW(ra2, As, 0, 2); PFHIP_SW(ra3, As, 0, 3);
    PFHIP_SW(rb0, Bs, 0, 0); PFHIP_SW(rb1, Bs, 0, 1); PFHIP_SW(rb2, Bs, 0, 2); PFHIP_SW(rb3, Bs, 0, 3);
    knext = nk > 1 ? kTileK : 0;
    PFHIP_GL(sa0, Ag, lda, 0, knext); PFHIP_GL(sa1, Ag, lda, 1, knext); PFHIP_GL(sa2, Ag, lda, 2, knext); PFHIP_GL(sa3, Ag, lda, 3, knext);
    PFHIP_GL(sb0, Wg, ldw, 0, knext); PFHIP_GL(sb1, Wg, ldw, 1, knext); PFHIP_GL(sb2, Wg, ldw, 2, knext); PFHIP_GL(sb3, Wg, ldw, 3, knext);
  }
  __syncthreads();

  const int a_off = (wr * 64 + r) * kLds + 4 * h;
  const int b_off = (wc * 64 + r) * kLds + 4 * h;
  PFHIP_FR(fa0, As, a_off, 0, 0, 0); PFHIP_FR(fa1, As, a_off, 0, 0, 1);
  PFHIP_FR(fb0, Bs, b_off, 0, 0, 0); PFHIP_FR(fb1, Bs, b_off, 0, 0, 1);

  // loads past the last K-tile re-fetch it (never used): keeps the bodies straight-line
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    { const int knext = (kt + 2 < nk ? kt + 2 : nk - 1) * kTileK; PFHIP_BODY_0 }
    { const int knext = (kt + 3 < nk ? kt + 3 : nk - 1) * kTileK; PFHIP_BODY_1 }
  }
  if (kt < nk) { const int knext = (nk - 1) * kTileK; PFHIP_BODY_0 }
#undef PFHIP_BODY_0
#undef PFHIP_BODY_1
#undef PFHIP_SB
#undef PFHIP_GL
#undef PFHIP_SW
#undef PFHIP_FR
#undef PFHIP_MM

  // ---- epilogue: accumulators -> LDS (C/D map: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5)) ----
  __syncthreads();                        // every wave has finished reading operand fragments
  float* const Cs = lds;
  {
    float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 64 + r;
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
  // ---- row pass: 8 rows x 512 B per step, 16-byte accesses ---------------------------------------
  const int c4 = tid & 31, rsub = tid >> 5;
  const int gcol = n0 + 4 * c4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (HAS_BIAS) {
    if (!GUARD || gcol + 3 < N) bv = *reinterpret_cast<const float4*>(bias + gcol);
    else {
      if (gcol < N) bv.x = bias[gcol];
      if (gcol + 1 < N) bv.y = bias[gcol + 1];
      if (gcol + 2 < N) bv.z = bias[gcol + 2];
    }
  }
#pragma unroll 4
  for (int pass = 0; pass < 16; ++pass) {
    const int row = pass * 8 + rsub;
    const int grow = m0 + row;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * kCs + 4 * c4);
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    if (!GUARD || (grow < M && gcol + 3 < N)) {
      if (HAS_R1) {
        const float4 t = *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (HAS_R2) {
        const float4 t = *reinterpret_cast<const float4*>(R2 + (size_t)grow * ldr2 + gcol);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(C + (size_t)grow * ldc + gcol) = v;
    } else if (grow < M && gcol < N) {      // ragged right edge (GUARD only): element-wise
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int q = 0; q < 4 && gcol + q < N; ++q) {
        float o = vv[q];
        if (HAS_R1) o += R1[(size_t)grow * ldr1 + gcol + q];
        if (HAS_R2) o += R2[(size_t)grow * ldr2 + gcol + q];
        if (RELU) o = fmaxf(o, 0.f);
        C[(size_t)grow * ldc + gcol + q] = o;
      }
    }
  }
}

// ---- 64 x 128 block tile for launches with too few 128 x 128 tiles to fill the chip (decoder-side GEMMs over a few thousand
// token rows, batches of 4-16 utterances): twice the blocks, half the serial K walk per block.  4 waves as 2 x 2, each
// 32 x 64 = 1 x 2 MFMA tiles; four accumulators (two output tiles x even / odd k pairs) so an accumulator is reused only
// every 4th MFMA; same staging / pipelining scheme as above (prefetch distance 2, memory instructions issued singly, body
// generated by tools/probe/gen_gemm_loop.py emit64).
constexpr int kTileM64 = 64;
constexpr int kStageA64 = kTileM64 * kLds;      // floats per buffer
constexpr int kStageB64 = kTileN * kLds;
constexpr int kLdsFloats64 = 2 * (kStageA64 + kStageB64);      // 55,296 B; the 64 x 132 C tile reuses it
static_assert(kTileM64 * kCs <= kLdsFloats64, "C tile must fit the operand buffers");

template <bool GUARD, bool HAS_BIAS, bool HAS_R1, bool HAS_R2, bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma64_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C, int ldc,
    const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N,
    int K, int tiles_n, int n_tiles, int gw) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats64];
  float* const As = lds;                      // [2][64][36]
  float* const Bs = lds + 2 * kStageA64;      // [2][128][36]
  int tm, tn;
  tile_of_block(blockIdx.x, n_tiles, tiles_n, gw, tm, tn);
  const int m0 = tm * kTileM64, n0 = tn * kTileN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int lrow = tid >> 3, lc4 = tid & 7;     // 32 rows x 8 float4 per staging pass

  const float* Ag = A + (size_t)(m0 + lrow) * lda + 4 * lc4;
  const float* Wg = W + (size_t)(n0 + lrow) * ldw + 4 * lc4;
  float4 ra0, ra1, rb0, rb1, rb2, rb3, sa0, sa1, sb0, sb1, sb2, sb3;
  float4 fa, fb0, fb1, ga, gb0, gb1;
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_GL(reg, base, ld, j, k0) reg = *reinterpret_cast<const float4*>(base + (size_t)(32 * (j)) * ld + (k0))
#define PFHIP_SWA(reg, base, buf, j) \
  *reinterpret_cast<float4*>(base + (buf) * kStageA64 + lrow * kLds + 4 * lc4 + 32 * (j) * kLds) = reg
#define PFHIP_SWB(reg, base, buf, j) \
  *reinterpret_cast<float4*>(base + (buf) * kStageB64 + lrow * kLds + 4 * lc4 + 32 * (j) * kLds) = reg
#define PFHIP_FRA(reg, buf, kb) reg = *reinterpret_cast<const float4*>(As + (buf) * kStageA64 + a_off + (kb) * 8)
#define PFHIP_FRB(reg, buf, kb, j) \
  reg = *reinterpret_cast<const float4*>(Bs + (buf) * kStageB64 + b_off + (kb) * 8 + 32 * (j) * kLds)
#define PFHIP_MM(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
#define PFHIP_BODY_0 \
  PFHIP_MM(acc0e, fa.x, fb0.x); PFHIP_SB; PFHIP_FRA(ga, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc1e, fa.x, fb1.x); PFHIP_SB; PFHIP_FRB(gb0, 0, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc0o, fa.y, fb0.y); PFHIP_SB; PFHIP_FRB(gb1, 0, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc1o, fa.y, fb1.y); PFHIP_SB; PFHIP_GL(ra0, Ag, lda, 0, knext); PFHIP_SB; \
  PFHIP_MM(acc0e, fa.z, fb0.z); PFHIP_SB; PFHIP_GL(ra1, Ag, lda, 1, knext); PFHIP_SB; \
  PFHIP_MM(acc1e, fa.z, fb1.z); PFHIP_SB; PFHIP_GL(rb0, Wg, ldw, 0, knext); PFHIP_SB; \
  PFHIP_MM(acc0o, fa.w, fb0.w); PFHIP_SB; PFHIP_GL(rb1, Wg, ldw, 1, knext); PFHIP_SB; \
  PFHIP_MM(acc1o, fa.w, fb1.w); PFHIP_SB; PFHIP_GL(rb2, Wg, ldw, 2, knext); PFHIP_SB; \
  PFHIP_MM(acc0e, ga.x, gb0.x); PFHIP_SB; PFHIP_FRA(fa, 0, 2); PFHIP_SB; \
  PFHIP_MM(acc1e, ga.x, gb1.x); PFHIP_SB; PFHIP_FRB(fb0, 0, 2, 0); PFHIP_SB; \
  PFHIP_MM(acc0o, ga.y, gb0.y); PFHIP_SB; PFHIP_FRB(fb1, 0, 2, 1); PFHIP_SB; \
  PFHIP_MM(acc1o, ga.y, gb1.y); PFHIP_SB; PFHIP_GL(rb3, Wg, ldw, 3, knext); PFHIP_SB; \
  PFHIP_MM(acc0e, ga.z, gb0.z); PFHIP_SB; \
  PFHIP_MM(acc1e, ga.z, gb1.z); PFHIP_SB; \
  PFHIP_MM(acc0o, ga.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc1o, ga.w, gb1.w); PFHIP_SB; \
  PFHIP_MM(acc0e, fa.x, fb0.x); PFHIP_SB; PFHIP_FRA(ga, 0, 3); PFHIP_SB; \
  PFHIP_MM(acc1e, fa.x, fb1.x); PFHIP_SB; PFHIP_FRB(gb0, 0, 3, 0); PFHIP_SB; \
  PFHIP_MM(acc0o, fa.y, fb0.y); PFHIP_SB; PFHIP_FRB(gb1, 0, 3, 1); PFHIP_SB; \
  PFHIP_MM(acc1o, fa.y, fb1.y); PFHIP_SB; \
  PFHIP_MM(acc0e, fa.z, fb0.z); PFHIP_SB; PFHIP_SWA(sa0, As, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc1e, fa.z, fb1.z); PFHIP_SB; PFHIP_SWA(sa1, As, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc0o, fa.w, fb0.w); PFHIP_SB; PFHIP_SWB(sb0, Bs, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc1o, fa.w, fb1.w); PFHIP_SB; PFHIP_SWB(sb1, Bs, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc0e, ga.x, gb0.x); PFHIP_SB; PFHIP_SWB(sb2, Bs, 1, 2); PFHIP_SB; \
  PFHIP_MM(acc1e, ga.x, gb1.x); PFHIP_SB; PFHIP_SWB(sb3, Bs, 1, 3); PFHIP_SB; \
  PFHIP_MM(acc0o, ga.y, gb0.y); PFHIP_SB; __syncthreads(); \
  PFHIP_MM(acc1o, ga.y, gb1.y); PFHIP_SB; PFHIP_FRA(fa, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc0e, ga.z, gb0.z); PFHIP_SB; PFHIP_FRB(fb0, 1, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc1e, ga.z, gb1.z); PFHIP_SB; PFHIP_FRB(fb1, 1, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc0o, ga.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc1o, ga.w, gb1.w); PFHIP_SB;
#define PFHIP_BODY_1 \
  PFHIP_MM(acc0e, fa.x, fb0.x); PFHIP_SB; PFHIP_FRA(ga, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc1e, fa.x, fb1.x); PFHIP_SB; PFHIP_FRB(gb0, 1, 1, 0); PFHIP_SB; \
  PFHIP_MM(acc0o, fa.y, fb0.y); PFHIP_SB; PFHIP_FRB(gb1, 1, 1, 1); PFHIP_SB; \
  PFHIP_MM(acc1o, fa.y, fb1.y); PFHIP_SB; PFHIP_GL(sa0, Ag, lda, 0, knext); PFHIP_SB; \
  PFHIP_MM(acc0e, fa.z, fb0.z); PFHIP_SB; PFHIP_GL(sa1, Ag, lda, 1, knext); PFHIP_SB; \
  PFHIP_MM(acc1e, fa.z, fb1.z); PFHIP_SB; PFHIP_GL(sb0, Wg, ldw, 0, knext); PFHIP_SB; \
  PFHIP_MM(acc0o, fa.w, fb0.w); PFHIP_SB; PFHIP_GL(sb1, Wg, ldw, 1, knext); PFHIP_SB; \
  PFHIP_MM(acc1o, fa.w, fb1.w); PFHIP_SB; PFHIP_GL(sb2, Wg, ldw, 2, knext); PFHIP_SB; \
  PFHIP_MM(acc0e, ga.x, gb0.x); PFHIP_SB; PFHIP_FRA(fa, 1, 2); PFHIP_SB; \
  PFHIP_MM(acc1e, ga.x, gb1.x); PFHIP_SB; PFHIP_FRB(fb0, 1, 2, 0); PFHIP_SB; \
  PFHIP_MM(acc0o, ga.y, gb0.y); PFHIP_SB; PFHIP_FRB(fb1, 1, 2, 1); PFHIP_SB; \
  PFHIP_MM(acc1o, ga.y, gb1.y); PFHIP_SB; PFHIP_GL(sb3, Wg, ldw, 3, knext); PFHIP_SB; \
  PFHIP_MM(acc0e, ga.z, gb0.z); PFHIP_SB; \
  PFHIP_MM(acc1e, ga.z, gb1.z); PFHIP_SB; \
  PFHIP_MM(acc0o, ga.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc1o, ga.w, gb1.w); PFHIP_SB; \
  PFHIP_MM(acc0e, fa.x, fb0.x); PFHIP_SB; PFHIP_FRA(ga, 1, 3); PFHIP_SB; \
  PFHIP_MM(acc1e, fa.x, fb1.x); PFHIP_SB; PFHIP_FRB(gb0, 1, 3, 0); PFHIP_SB; \
  PFHIP_MM(acc0o, fa.y, fb0.y); PFHIP_SB; PFHIP_FRB(gb1, 1, 3, 1); PFHIP_SB; \
  PFHIP_MM(acc1o, fa.y, fb1.y); PFHIP_SB; \
  PFHIP_MM(acc0e, fa.z, fb0.z); PFHIP_SB; PFHIP_SWA(ra0, As, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc1e, fa.z, fb1.z); PFHIP_SB; PFHIP_SWA(ra1, As, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc0o, fa.w, fb0.w); PFHIP_SB; PFHIP_SWB(rb0, Bs, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc1o, fa.w, fb1.w); PFHIP_SB; PFHIP_SWB(rb1, Bs, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc0e, ga.x, gb0.x); PFHIP_SB; PFHIP_SWB(rb2, Bs, 0, 2); PFHIP_SB; \
  PFHIP_MM(acc1e, ga.x, gb1.x); PFHIP_SB; PFHIP_SWB(rb3, Bs, 0, 3); PFHIP_SB; \
  PFHIP_MM(acc0o, ga.y, gb0.y); PFHIP_SB; __syncthreads(); \
  PFHIP_MM(acc1o, ga.y, gb1.y); PFHIP_SB; PFHIP_FRA(fa, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc0e, ga.z, gb0.z); PFHIP_SB; PFHIP_FRB(fb0, 0, 0, 0); PFHIP_SB; \
  PFHIP_MM(acc1e, ga.z, gb1.z); PFHIP_SB; PFHIP_FRB(fb1, 0, 0, 1); PFHIP_SB; \
  PFHIP_MM(acc0o, ga.w, gb0.w); PFHIP_SB; \
  PFHIP_MM(acc1o, ga.w, gb1.w); PFHIP_SB;

  f32x16 acc0e, acc0o, acc1e, acc1o;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0e[e] = 0.f; acc0o[e] = 0.f; acc1e[e] = 0.f; acc1o[e] = 0.f; }
  const int nk = K / kTileK;
  {
    int knext = 0;
    PFHIP_GL(ra0, Ag, lda, 0, knext); PFHIP_GL(ra1, Ag, lda, 1, knext);
    PFHIP_GL(rb0, Wg, ldw, 0, knext); PFHIP_GL(rb1, Wg, ldw, 1, knext); PFHIP_GL(rb2, Wg, ldw, 2, knext); PFHIP_GL(rb3, Wg, ldw, 3, knext);
    PFHIP_SWA(ra0, As, 0, 0); PFHIP_SWA(ra1, As, 0, 1);
    PFHIP_SWB(rb0, Bs, 0, 0); PFHIP_SWB(rb1, Bs, 0, 1); PFHIP_SWB(rb2, Bs, 0, 2); PFHIP_SWB(rb3, Bs, 0, 3);
    knext = nk > 1 ? kTileK : 0;
    PFHIP_GL(sa0, Ag, lda, 0, knext); PFHIP_GL(sa1, Ag, lda, 1, knext);
    PFHIP_GL(sb0, Wg, ldw, 0, knext); PFHIP_GL(sb1, Wg, ldw, 1, knext); PFHIP_GL(sb2, Wg, ldw, 2, knext); PFHIP_GL(sb3, Wg, ldw, 3, knext);
  }
  __syncthreads();
  const int a_off = (wr * 32 + r) * kLds + 4 * h;
  const int b_off = (wc * 64 + r) * kLds + 4 * h;
  PFHIP_FRA(fa, 0, 0); PFHIP_FRB(fb0, 0, 0, 0); PFHIP_FRB(fb1, 0, 0, 1);
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    { const int knext = (kt + 2 < nk ? kt + 2 : nk - 1) * kTileK; PFHIP_BODY_0 }
    { const int knext = (kt + 3 < nk ? kt + 3 : nk - 1) * kTileK; PFHIP_BODY_1 }
  }
  if (kt < nk) { const int knext = (nk - 1) * kTileK; PFHIP_BODY_0 }
#undef PFHIP_BODY_0
#undef PFHIP_BODY_1
#undef PFHIP_SB
#undef PFHIP_GL
#undef PFHIP_SWA
#undef PFHIP_SWB
#undef PFHIP_FRA
#undef PFHIP_FRB
#undef PFHIP_MM

  // ---- epilogue: accumulators -> LDS -> 512-byte rows (as the 128-row kernel) ----
  __syncthreads();
  float* const Cs = lds;
  {
    float* cw = Cs + (wr * 32 + 4 * h) * kCs + wc * 64 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ro = ((e & 3) + 8 * (e >> 2)) * kCs;
      cw[ro] = acc0e[e] + acc0o[e];
      cw[ro + 32] = acc1e[e] + acc1o[e];
    }
  }
  __syncthreads();
  const int c4 = tid & 31, rsub = tid >> 5;
  const int gcol = n0 + 4 * c4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (HAS_BIAS) {
    if (!GUARD || gcol + 3 < N) bv = *reinterpret_cast<const float4*>(bias + gcol);
    else {
      if (gcol < N) bv.x = bias[gcol];
      if (gcol + 1 < N) bv.y = bias[gcol + 1];
      if (gcol + 2 < N) bv.z = bias[gcol + 2];
    }
  }
#pragma unroll 4
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 8 + rsub;
    const int grow = m0 + row;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * kCs + 4 * c4);
    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    if (!GUARD || (grow < M && gcol + 3 < N)) {
      if (HAS_R1) {
        const float4 t = *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (HAS_R2) {
        const float4 t = *reinterpret_cast<const float4*>(R2 + (size_t)grow * ldr2 + gcol);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(C + (size_t)grow * ldc + gcol) = v;
    } else if (grow < M && gcol < N) {
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int q = 0; q < 4 && gcol + q < N; ++q) {
        float o = vv[q];
        if (HAS_R1) o += R1[(size_t)grow * ldr1 + gcol + q];
        if (HAS_R2) o += R2[(size_t)grow * ldr2 + gcol + q];
        if (RELU) o = fmaxf(o, 0.f);
        C[(size_t)grow * ldc + gcol + q] = o;
      }
    }
  }
}

// ---- skinny GEMM for the chunk-streaming path (M <= 64 rows: one 20-row encoder window, a few tokens) ----
// Weight-streaming-bound: every weight is read once per chunk, so the job is to keep many independent
// 16-byte loads in flight, not to tile for reuse.  One block = one 32-column strip of W over the full K,
// its 16 waves split K sixteen ways (each lane streams W[n][k..k+3] straight into VGPRs, deep unrolled, no LDS
// staging: cdna_hip_programming.md "GEMV / M <= 16" row), partial 32x32 tiles are summed through LDS and
// written with the same fused epilogue.  Activations (<= 64 x K, L2-resident) are read the same way.
constexpr int kSkinnyWaves = 16;

__global__ __launch_bounds__(1024) void gemm_f32_skinny_kernel(const float* __restrict__ A, int lda,
                                                               const float* __restrict__ W, int ldw, float* C,
                                                               int ldc, const float* __restrict__ bias,
                                                               const float* R1, int ldr1, const float* R2,
                                                               int ldr2, int M, int N, int K, int relu) {
  __shared__ float red[kSkinnyWaves][32 * 33];
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // K is split 16 ways in units of 8 (one MFMA k-block): K = 512 -> 4 loads per wave, all in flight at once
  const int nkb = K >> 3;
  const int per = (nkb + kSkinnyWaves - 1) / kSkinnyWaves;
  const int kb0 = wave * per;
  const int kb1 = kb0 + per < nkb ? kb0 + per : nkb;
  const float* ap = A + (size_t)(m0 + r) * lda + 4 * h;      // rows up to the 128-row allocation exist
  const float* wp = W + (size_t)(n0 + r) * ldw + 4 * h;      // rows up to the 128-row padding are readable
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  int kb = kb0;
  for (; kb + 8 <= kb1; kb += 8) {
    float4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = *reinterpret_cast<const float4*>(ap + 8 * (kb + u));
      b[u] = *reinterpret_cast<const float4*>(wp + 8 * (kb + u));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
  }
  if (kb + 4 <= kb1) {
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const float4*>(ap + 8 * (kb + u));
      b[u] = *reinterpret_cast<const float4*>(wp + 8 * (kb + u));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
    }
    kb += 4;
  }
  for (; kb < kb1; ++kb) {
    const float4 a = *reinterpret_cast<const float4*>(ap + 8 * kb);
    const float4 b = *reinterpret_cast<const float4*>(wp + 8 * kb);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
  // D[i = activation row][j = weight column]: col j = lane&31, row i = (e&3) + 8*(e>>2) + 4*h
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[e];
  __syncthreads();
  const int row = tid >> 5, col = tid & 31;          // 1024 threads = the 32 x 32 tile
  const int grow = m0 + row, gcol = n0 + col;
  if (grow >= M || gcol >= N) return;
  const int o = row * 33 + col;
  float v = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < kSkinnyWaves; ++w2) v += red[w2][o];
  if (bias) v += bias[gcol];
  if (R1) v += R1[(size_t)grow * ldr1 + gcol];
  if (R2) v += R2[(size_t)grow * ldr2 + gcol];
  if (relu) v = fmaxf(v, 0.f);
  C[(size_t)grow * ldc + gcol] = v;
}

// column-group width of the tile order (see tile_of_block), from a small traffic model that reproduces the PMC numbers
// (profiles/r01/pmc_hbm_traffic.json) within ~30 %: per XCD, `rows_conc` row panels are in flight at once (64 block slots);
//   all columns in one group: A is fetched once, the weights once per XCD per PASS over its row panels
//                             (passes = row panels of the XCD / rows_conc) when they do not fit the L2 share, else once;
//   groups of g column tiles (<= 2 MiB of weights, resident): A once per group, the weights once per XCD.
inline int column_group_width(int M, int K, int tiles_n) {
  constexpr double kL2Share = 2.0 * 1024 * 1024;
  const double wt = (double)kTileN * K * 4, w_total = wt * tiles_n, a_total = (double)M * K * 4;
  if (w_total <= kL2Share) return tiles_n;
  const int g = std::max(1, (int)(kL2Share / wt));
  if (g >= tiles_n) return tiles_n;
  const double tiles_m = (M + kTileM - 1) / kTileM;
  const double rows_conc = std::max(1.0, 64.0 / tiles_n), passes = std::max(1.0, tiles_m / 8.0 / rows_conc);
  const int groups = (tiles_n + g - 1) / g;
  const double est_all = a_total + w_total * 8 * passes, est_grouped = a_total * groups + w_total * 8;
  return est_grouped < est_all ? g : tiles_n;
}

template <bool GUARD>
void launch_variant64(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                      const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu,
                      hipStream_t s) {
  const int tiles_m = (M + kTileM64 - 1) / kTileM64;
  const int tiles_n = (N + kTileN - 1) / kTileN;
  const int n_tiles = tiles_m * tiles_n;
  const dim3 grid(n_tiles), block(256);
  const int gw = column_group_width(M, K, tiles_n);
#define PFHIP_GEMM64(B_, R1_, R2_, RL_)                                                              \
  hipLaunchKernelGGL((gemm_f32_mfma64_kernel<GUARD, B_, R1_, R2_, RL_>), grid, block, 0, s, A, lda, W, \
                     ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw)
  const int key = (bias ? 8 : 0) | (R1 ? 4 : 0) | (R2 ? 2 : 0) | (relu ? 1 : 0);
  switch (key) {
    case 0: PFHIP_GEMM64(false, false, false, false); break;
    case 1: PFHIP_GEMM64(false, false, false, true); break;
    case 2: PFHIP_GEMM64(false, false, true, false); break;
    case 3: PFHIP_GEMM64(false, false, true, true); break;
    case 4: PFHIP_GEMM64(false, true, false, false); break;
    case 5: PFHIP_GEMM64(false, true, false, true); break;
    case 6: PFHIP_GEMM64(false, true, true, false); break;
    case 7: PFHIP_GEMM64(false, true, true, true); break;
    case 8: PFHIP_GEMM64(true, false, false, false); break;
    case 9: PFHIP_GEMM64(true, false, false, true); break;
    case 10: PFHIP_GEMM64(true, false, true, false); break;
    case 11: PFHIP_GEMM64(true, false, true, true); break;
    case 12: PFHIP_GEMM64(true, true, false, false); break;
    case 13: PFHIP_GEMM64(true, true, false, true); break;
    case 14: PFHIP_GEMM64(true, true, true, false); break;
    default: PFHIP_GEMM64(true, true, true, true); break;
  }
#undef PFHIP_GEMM64
}

template <bool GUARD>
void launch_variant(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                    const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu,
                    hipStream_t s) {
  const int tiles_m = (M + kTileM - 1) / kTileM;
  const int tiles_n = (N + kTileN - 1) / kTileN;
  const int n_tiles = tiles_m * tiles_n;
  const dim3 grid(n_tiles), block(256);
  const int gw = column_group_width(M, K, tiles_n);
#define PFHIP_GEMM(B_, R1_, R2_, RL_)                                                              \
  hipLaunchKernelGGL((gemm_f32_mfma_kernel<GUARD, B_, R1_, R2_, RL_>), grid, block, 0, s, A, lda, W, \
                     ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, tiles_n, n_tiles, gw)
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

}  // namespace

// Which kernel: measured on the model's (N, K) over M = 32 ... 16000 (tools/gemm_sweep.py).  A launch runs in ROUNDS of one
// block per CU (a second co-resident block does not shorten a round), a 128 x 128 tile costs ~34 us per round at K = 512
// (~120 us at K = 2048), a 64 x 128 tile 0.55 of that, and the weight-streaming kernel (32 x 32 patches, 16 waves splitting K)
// ~9 us per round of patches.  Hence: below 64 tiles the streaming kernel (one 30-s utterance, streaming windows: 2-7x faster
// than a lone 128 x 128 tile walking its K range serially); up to 128 tiles the 64-row kernel (one round instead of a
// half-empty one: -40 %); above that whichever of the two tilings needs less round time (the 64-row one wins where the
// 128-row grid would leave most of its last round empty, e.g. 257-384 tiles: -14 %).
constexpr int kCUs = 256;
constexpr int kStreamingBelowTiles = 64;
inline bool prefer_half_tile(int tiles128) {
  if (tiles128 <= 128) return true;
  const int r128 = (tiles128 + kCUs - 1) / kCUs, r64 = (2 * tiles128 + kCUs - 1) / kCUs;
  return 0.55 * r64 < 0.98 * r128;
}

// The split-operand kernels: fp16 two-plane / three-product form (gemm_x3.hip, default) or bf16 three-plane / six-product form
// (gemm_x6.hip: fp32's exponent range; PFHIP_GEMM_X3=0).
float best_w_scale(float max_abs) {
  if (!(max_abs > 0.f) || !std::isfinite(max_abs)) return 1.0f;
  int e = 0;
  (void)frexpf(32768.0f / max_abs, &e);          // 32768 / max = f * 2^e, f in [0.5, 1): 2^(e-1) <= 32768 / max
  e = std::max(-8, std::min(20, e - 1));
  return ldexpf(1.0f, e);                          // max_abs * scale <= 32768 < 65504
}
LaunchCtx& launch_ctx() {
  static thread_local LaunchCtx ctx;
  return ctx;
}
static bool x3_enabled() {
  static const bool on = [] { const char* e = getenv("PFHIP_GEMM_X3"); return !(e && e[0] == '0'); }();
  return on && !launch_ctx().exact;
}
static void launch_split_gemm(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                              int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, int gw, hipStream_t s, bool small_tile,
                              const float* ln_stats, int ln_tiles, float* stats_out, bool half_tile, const float* ln_colsum, int form,
                              float w_scale) {
  if (form == 3 || (form == 0 && x3_enabled())) {
    // PFHIP_X3_SW: override of the per-tensor weight scale the caller passes (log2)
    static const float sw_env = [] { const char* e = getenv("PFHIP_X3_SW"); return e ? ldexpf(1.f, atoi(e)) : 0.f; }();
    launch_gemm_f32_f16x3(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, gw, s, small_tile, ln_stats, ln_tiles, stats_out,
                          half_tile, ln_colsum, sw_env > 0.f ? sw_env : w_scale);
    return;
  }
  launch_gemm_f32_bf16x6(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, gw, s, small_tile, ln_stats, ln_tiles, stats_out,
                         half_tile, ln_colsum);
}

void launch_gemm_f32_kind(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                          int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, bool guard, int kind,
                          hipStream_t s, float w_scale) {
  if (M <= 0 || N <= 0) return;
  const int tiles = ((M + kTileM - 1) / kTileM) * ((N + kTileN - 1) / kTileN);
  // Launches of at least half a round of tiles go to the BF16 matrix cores (exact three-way split, six MFMAs per block:
  // gemm_x6.hip) — 1.3-1.6 x the fp32 MFMA kernels, results at least as close to fp64 as the fp32 chain.  Its 128 x 128 tile
  // (two blocks per CU) wins on the K = 512 shapes and on every under-filled grid; the 256 x 128 tile on long-K launches that
  // fill the chip (FFN2 at full batch: 201 vs 194 TF).  PFHIP_GEMM_X6=0 turns the path off.
  static const bool x6_on = [] { const char* e = getenv("PFHIP_GEMM_X6"); return !(e && e[0] == '0'); }();
  const int tiles256 = ((M + 255) / 256) * ((N + kTileN - 1) / kTileN);
  // (tools/gemm_small_probe.py: between 48 and 128 tiles of 128 x 128 the half-height BF16-split kernel beats both the fp32-MFMA
  // 64-row kernel and the 128 x 128 BF16-split kernel by 15-25 % — 1-5 utterances of 30 s, rounds of 25-128 streaming connections)
  // kinds 4 / 5 / 7: the bf16 six-product form (256 x 128, 128 x 128, 64 x 128 tile); 8 / 9 / 10: the fp16 three-product form
  if (kind == 4 || kind == 5 || kind == 7 || kind == 8 || kind == 9 || kind == 10 || (kind == 0 && x6_on && tiles >= 48)) {
    const int form = kind == 0 ? 0 : (kind >= 8 ? 3 : 6);
    if (kind >= 8) kind = kind == 8 ? 4 : (kind == 9 ? 5 : 7);
    const bool small_tile = kind == 5 || kind == 7 || (kind == 0 && !(K >= 1024 && tiles256 >= 180));
    // measured (tools/gemm_mid_probe.py, kinds 5 vs 7, N = 512): the half-height tile wins while its own grid still fits one
    // workgroup per CU (<= 128 tiles of 128 x 128 = 256 half tiles: 19.0 vs 22.1 us at M = 4000, K = 512; 61 vs 66 at K = 2048);
    // at 160 tiles (316 half tiles: a second, mostly empty round) the taller tile is ahead again (23.2 vs 27.4, 72 vs 93 us at
    // M = 5000), as at 220 (28.3 vs 31.9 at M = 7015)
    const bool half_tile = kind == 7 || (kind == 0 && small_tile && tiles <= 128);
    launch_split_gemm(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu,
                      column_group_width(M, K, (N + kTileN - 1) / kTileN), s, small_tile, nullptr, 0, nullptr, half_tile, nullptr, form,
                      w_scale);
    return;
  }
  const bool skinny = kind == 2 || (kind == 0 && tiles < kStreamingBelowTiles);
  const bool half = kind == 3 || (kind == 0 && prefer_half_tile(tiles));
  if (skinny) {   // always bounds-checked
    const dim3 grid((N + 31) / 32, (M + 31) / 32), block(1024);
    hipLaunchKernelGGL(gemm_f32_skinny_kernel, grid, block, 0, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N,
                       K, relu ? 1 : 0);
    return;
  }
  if (half) {
    if (guard) launch_variant64<true>(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, s);
    else launch_variant64<false>(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, s);
    return;
  }
  if (guard) launch_variant<true>(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, s);
  else launch_variant<false>(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, s);
}

static bool x6_enabled() {
  static const bool on = [] { const char* e = getenv("PFHIP_GEMM_X6"); return !(e && e[0] == '0'); }();
  return on;
}
// whether the large GEMMs run the fp16 two-plane form (gemm_x3.hip / gemm_p3.hip) — PFHIP_GEMM_X3=0 asks for the bf16 three-plane one
bool gemm_f16_planes_form() { return x6_enabled() && x3_enabled(); }
bool gemm_x6_ln_ok(int M) {
  static const bool ln_on = [] { const char* e = getenv("PFHIP_GEMM_LN"); return !(e && e[0] == '0'); }();
  // from the batch size at which the N = 512 launches (four column tiles per row panel) go to the BF16-split kernels at all
  return ln_on && x6_enabled() && ((M + kTileM - 1) / kTileM) * 4 >= 48;
}
void launch_gemm_f32_x6_ln(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1,
                           int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, const float* ln_stats, int ln_tiles,
                           const float* ln_colsum, float* stats_out, hipStream_t s, float w_scale) {
  if (M <= 0 || N <= 0) return;
  const int tiles256 = ((M + 255) / 256) * ((N + kTileN - 1) / kTileN);
  const bool small_tile = !(K >= 1024 && tiles256 >= 180);
  const int tiles128 = ((M + kTileM - 1) / kTileM) * ((N + kTileN - 1) / kTileN);
  const bool half_tile = small_tile && tiles128 <= 128;       // as the default dispatch: 64-row tiles where 128-row ones leave CUs idle
  launch_split_gemm(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu,
                    column_group_width(M, K, (N + kTileN - 1) / kTileN), s, small_tile, ln_stats, ln_tiles, stats_out, half_tile,
                    ln_colsum, 0, w_scale);
}

void launch_gemm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc,
                     const float* bias, const float* R1, int ldr1, const float* R2, int ldr2, int M,
                     int N, int K, bool relu, bool guard, hipStream_t s, float w_scale) {
  launch_gemm_f32_kind(A, lda, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, M, N, K, relu, guard, 0, s, w_scale);
}

}  // namespace pfhip
