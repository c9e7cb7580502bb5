// Fused kernels for ONE streaming window (`ForwardChunk`'s encoder and decoder Runs,
// onnxruntime/src/paraformer-online.cpp:426-515, at M = 20 rows / a handful of tokens).
//
// A lone 600-ms chunk is neither compute- nor bandwidth-bound: it reads 0.88 GB of weights behind several hundred DEPENDENT
// launches.  A dependent launch costs its boundary (1.5-2 us) plus a kernel that cannot be shorter than one memory latency, and
// grid-wide barriers inside one launch cost more than a boundary (MI355X_MICROARCH.md price list: barrier-xcd 4-5 us) — so the
// fusion respects the data dependences between whole matrices, and what is left is (a) the number of launches and (b) the
// number of dependent round trips inside each.
//   (a) a row-wise operator in front of a GEMM is RECOMPUTED by every workgroup of that GEMM (20 x 512 elements: nothing):
//       LayerNorm -> GEMM is one launch; the SAN-M FSMN memory is an epilogue term of the output projection; the 16 decoder
//       layers' K/V projections of the (fixed) window are one GEMM over the concatenated weights (stream.cpp).
//   (b) three generations of the LN -> GEMM (+bias, residual, FSMN, ReLU) launch, newest last:
//       fused_ln_gemm_kernel   fp32 MFMA, K split over 16 waves (kept for the vocabulary projection and odd shapes),
//       fused_ln_gemv_kernel   the same on the vector ALUs with LayerNorm on load (fallback: K = 576 first layer, M > 20),
//       fused_gemv1t_kernel    every operand of a lane requested at kernel start, LayerNorm applied algebraically;
//       and window_attention_kernel for the 20 x 20 attention between them.
// All kernels take M <= 32 rows (one window of the [5,10,5] chunking is 20).
#include "kernels.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kWaves = 16;
constexpr int kThreads = kWaves * 64;
constexpr int kRows = 32;                  // activation rows a workgroup handles (one MFMA tile)

// mean / rstd of rows [0, M) of X[., 0..D) into LDS, two-pass over values held in registers (one trip to memory; D <= 2048;
// rows >= M get 0 / 0).  1024 threads: thread = (row = tid >> 5, 32 threads per row).  Ends with __syncthreads().
__device__ __forceinline__ void row_stats(const float* __restrict__ X, int ldx, int M, int D, float eps, float* s_mean, float* s_rstd) {
  const int tid = threadIdx.x, row = tid >> 5, c = tid & 31;
  float mean = 0.f, rstd = 0.f;
  if (row < M) {
    const float* xr = X + (size_t)row * ldx;
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = 4 * c + 128 * u;
      v[u] = k < D ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (4 * c + 128 * u < D) {
        const float a = v[u].x - mean, b = v[u].y - mean, cc = v[u].z - mean, d = v[u].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
    rstd = 1.0f / sqrtf(q / (float)D + eps);
  }
  if (c == 0) { s_mean[row] = mean; s_rstd[row] = rstd; }
  __syncthreads();
}

// C[M<=32, N] = act(LN?(X) W^T + bias) (+R1) (+R2) (+FSMN memory of V), one workgroup per 32 output columns.
//   LN:   X rows are normalised on the way into the MFMA operand: (x - mean) * rstd * g[k] + b[k] for k < D, 0 for k >= D
//         (the first encoder layer's K is padded 560 -> 576).
//   FSMN: + V[row][col] + sum_j fw[col][j] * V[row + j - 5][col] over rows inside [0, M) — the SAN-M memory block on the value
//         projection of ONE window (UPSTREAM encoder layer; kernels.h launch_fsmn), added like a residual.
// CW = output columns per workgroup (32, 8 or 4).  The job is weight STREAMING: a GEMM with N = 512 on 32-column patches keeps
// 16 of the 256 CUs busy and takes 7-23 us (measured, K = 512 / 2048); narrower patches put 64-128 workgroups on it.  The MFMA
// tile stays 32 wide (lanes beyond CW feed zeros): matrix-core work is not what this kernel waits for.
template <bool LN, int CW>
__global__ __launch_bounds__(kThreads) void fused_ln_gemm_kernel(
    const float* __restrict__ X, int ldx, int D, const float* __restrict__ g, const float* __restrict__ b, float eps,
    const float* __restrict__ W, int ldw, float* C, int ldc, const float* __restrict__ bias, const float* R1, int ldr1,
    const float* R2, int ldr2, const float* __restrict__ V, int ldv, const float* __restrict__ fw, int M, int N, int K, int relu) {
  __shared__ float red[kWaves][32 * (CW + 1)];
  __shared__ float s_mean[kRows], s_rstd[kRows];
  const int n0 = blockIdx.x * CW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (LN) row_stats(X, ldx, M, D, eps, s_mean, s_rstd);
  const float mean = LN ? s_mean[r] : 0.f, rstd = LN ? s_rstd[r] : 0.f;
  const int nkb = K >> 3;
  const int per = (nkb + kWaves - 1) / kWaves;
  const int kb0 = wave * per;
  const int kb1 = kb0 + per < nkb ? kb0 + per : nkb;
  const float* ap = X + (size_t)(r < M ? r : M - 1) * ldx + 4 * h;
  const bool wlive = r < CW;
  const float* wp = W + (size_t)(n0 + (wlive ? r : 0)) * ldw + 4 * h;      // weight rows up to the 128-row padding are readable
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  auto norm = [&](float4 a, int k) {
    if (!LN) return a;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < D) {               // D % 4 == 0: a float4 is inside or outside as a whole
      const float4 gg = *reinterpret_cast<const float4*>(g + k);
      const float4 bb = *reinterpret_cast<const float4*>(b + k);
      o.x = (a.x - mean) * rstd * gg.x + bb.x; o.y = (a.y - mean) * rstd * gg.y + bb.y;
      o.z = (a.z - mean) * rstd * gg.z + bb.z; o.w = (a.w - mean) * rstd * gg.w + bb.w;
    }
    return o;
  };
  int kb = kb0;
  for (; kb + 4 <= kb1; kb += 4) {
    float4 a[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const float4*>(ap + 8 * (kb + u));
      w[u] = wlive ? *reinterpret_cast<const float4*>(wp + 8 * (kb + u)) : zero4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 an = norm(a[u], 8 * (kb + u) + 4 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.x, w[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.y, w[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.z, w[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.w, w[u].w, acc, 0, 0, 0);
    }
  }
  for (; kb < kb1; ++kb) {
    const float4 an = norm(*reinterpret_cast<const float4*>(ap + 8 * kb), 8 * kb + 4 * h);
    const float4 w = wlive ? *reinterpret_cast<const float4*>(wp + 8 * kb) : zero4;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.x, w.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.y, w.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.z, w.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.w, w.w, acc, 0, 0, 0);
  }
  // D[i = activation row][j = weight column]: col j = lane & 31, row i = (e & 3) + 8 * (e >> 2) + 4 * h
  if (wlive) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * h) * (CW + 1) + r] = acc[e];
  }
  __syncthreads();
  if (tid >= 32 * CW) return;
  const int row = tid / CW, col = tid % CW;          // 32 x CW outputs
  const int gcol = n0 + col;
  if (row >= M || gcol >= N) return;
  const int o = row * (CW + 1) + col;
  float v = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < kWaves; ++w2) v += red[w2][o];
  if (bias) v += bias[gcol];
  if (R1) v += R1[(size_t)row * ldr1 + gcol];
  if (R2) v += R2[(size_t)row * ldr2 + gcol];
  if (V) {
    float mem = V[(size_t)row * ldv + gcol];
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const int t = row + j - 5;
      if (t >= 0 && t < M) mem += fw[gcol * 11 + j] * V[(size_t)t * ldv + gcol];
    }
    v += mem;
  }
  if (relu) v = fmaxf(v, 0.f);
  C[(size_t)row * ldc + gcol] = v;
}


// ---- the same operator on the vector ALUs -----------------------------------------------------------------------------------
// Measured on one window (profiles/r02/stream_one_by_kernel_a.txt): the MFMA form above spends 7 us (K = 512) to 23 us
// (K = 2048) per launch whatever the patch width, because a 32 x 32 x K tile on `v_mfma_f32_32x32x2_f32` costs K/2 x 64 cycles
// spread over 4 SIMDs — 7.8 us at K = 2048 — and only 20 x CW of its 1024 outputs are wanted.  The fp32 MFMA runs at the fp32
// VECTOR rate, so nothing is lost by computing exactly the wanted M x CW outputs with v_fma: lane = (output column c, k-subset),
// every lane keeps one accumulator per activation row, reads float4s of its weight row and of every activation row (the same
// address in the CW lanes of a k-subset: one fetch), and the k-subsets are summed with log2(64 / CW) xor-shuffles, the waves
// through LDS in wave order (deterministic).  Per workgroup: 20 x CW x K FMAs (a few hundred per thread), CW x K weights
// from HBM, M x K activations from L2.  Measured in isolation on cold weights (tools/fused_gemv_bench.py): 6.7 us at
// N = K = 512, 11 us for QKV / FFN1, 14-18 us at K = 2048 (the MFMA form: 7 / 10 / 20-26); the vocabulary projection
// (N = 8404) stays on the MFMA form (20 vs 42 us).  Tried and dropped: staging the activation slab through LDS with every
// independent load issued up front (one trip to memory instead of three): 9.6 us at N = K = 512 and 33-56 us with LayerNorm;
// prefetching the first row group's raw values across the statistics barrier (30-50 spilled registers at the 128 a 1024-thread
// workgroup leaves: 23-27 us).
template <bool LN, int CW, int MR>
__global__ __launch_bounds__(1024) void fused_ln_gemv_kernel(
    const float* __restrict__ X, int ldx, int D, const float* __restrict__ g, const float* __restrict__ b, float eps,
    const float* __restrict__ W, int ldw, float* C, int ldc, const float* __restrict__ bias, const float* R1, int ldr1,
    const float* R2, int ldr2, const float* __restrict__ V, int ldv, const float* __restrict__ fw, int M, int N, int K, int relu) {
  constexpr int KS = 64 / CW;            // k-subsets per wave
  constexpr int KI = 4 * KS;             // k per wave and iteration
  __shared__ float red[kWaves][MR * CW];
  __shared__ float s_mean[kRows], s_rstd[kRows];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const int c = lane % CW, ks = lane / CW;
  const int n0 = blockIdx.x * CW;
  const float* wrow = W + (size_t)(n0 + c) * ldw;           // weight rows up to the 128-row padding are readable
  // the lane's first weight float4 does not depend on anything: in flight while the row statistics are computed
  const int n_blocks = K / KI;
  float4 w_first = make_float4(0.f, 0.f, 0.f, 0.f);
  if (wave < n_blocks) w_first = *reinterpret_cast<const float4*>(wrow + wave * KI + 4 * ks);
  if (LN) {
    // row statistics, two passes over values held in registers (one trip to memory): 32 threads per row, rows in rounds
    for (int row = tid >> 5; row < kRows; row += blockDim.x >> 5) {
      const int cc = tid & 31;
      float mean = 0.f, rstd = 0.f;
      if (row < M) {
        const float* xr = X + (size_t)row * ldx;
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int k = 4 * cc + 128 * u;
          v[u] = k < D ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) sum += (v[u].x + v[u].y) + (v[u].z + v[u].w);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        mean = sum / (float)D;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (4 * cc + 128 * u < D) {
            const float a0 = v[u].x - mean, a1 = v[u].y - mean, a2 = v[u].z - mean, a3 = v[u].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
          }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
        rstd = 1.0f / sqrtf(q / (float)D + eps);
      }
      if (cc == 0) { s_mean[row] = mean; s_rstd[row] = rstd; }
    }
    __syncthreads();
  }
  float acc[MR];
#pragma unroll
  for (int i = 0; i < MR; ++i) acc[i] = 0.f;
  for (int kbk = wave; kbk < n_blocks; kbk += n_waves) {
    const int k = kbk * KI + 4 * ks;
    const float4 w4 = kbk == wave ? w_first : *reinterpret_cast<const float4*>(wrow + k);
    float4 gg = make_float4(0.f, 0.f, 0.f, 0.f), bb = gg;
    const bool live = !LN || k < D;
    if (LN && live) { gg = *reinterpret_cast<const float4*>(g + k); bb = *reinterpret_cast<const float4*>(b + k); }
    // rows in groups of 8: eight 16-byte loads in flight per lane (x 16 waves) cover the latency; more would spill at the
    // 128 registers a 1024-thread workgroup leaves each lane
#pragma unroll
    for (int i0 = 0; i0 < MR; i0 += 8) {
      float4 a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const float4*>(X + (size_t)(i0 + i < M ? i0 + i : M - 1) * ldx + k);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float4 v = a[i];
        if (LN) {
          const float mu = s_mean[i0 + i], rs = s_rstd[i0 + i];
          v.x = (v.x - mu) * rs * gg.x + bb.x; v.y = (v.y - mu) * rs * gg.y + bb.y;
          v.z = (v.z - mu) * rs * gg.z + bb.z; v.w = (v.w - mu) * rs * gg.w + bb.w;
          if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        acc[i0 + i] = fmaf(v.x, w4.x, acc[i0 + i]); acc[i0 + i] = fmaf(v.y, w4.y, acc[i0 + i]);
        acc[i0 + i] = fmaf(v.z, w4.z, acc[i0 + i]); acc[i0 + i] = fmaf(v.w, w4.w, acc[i0 + i]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // sum the k-subsets of the wave (lanes c, c + CW, ...), then the waves in order
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    float v = acc[i];
#pragma unroll
    for (int o = 32; o >= CW; o >>= 1) v += __shfl_xor(v, o);
    acc[i] = v;
  }
  if (ks == 0) {
#pragma unroll
    for (int i = 0; i < MR; ++i) red[wave][i * CW + c] = acc[i];
  }
  __syncthreads();
  if (tid >= MR * CW) return;
  const int row = tid / CW, col = tid % CW, gcol = n0 + col;
  if (row >= M || gcol >= N) return;
  float v = 0.f;
  for (int w2 = 0; w2 < n_waves; ++w2) v += red[w2][tid];
  if (bias) v += bias[gcol];
  if (R1) v += R1[(size_t)row * ldr1 + gcol];
  if (R2) v += R2[(size_t)row * ldr2 + gcol];
  if (V) {
    float mem = V[(size_t)row * ldv + gcol];
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const int t = row + j - 5;
      if (t >= 0 && t < M) mem += fw[gcol * 11 + j] * V[(size_t)t * ldv + gcol];
    }
    v += mem;
  }
  if (relu) v = fmaxf(v, 0.f);
  C[(size_t)row * ldc + gcol] = v;
}


// ---- second form: ONE trip to memory ------------------------------------------------------------------------------------------
// What the kernel above still waits for is not bytes but dependent round trips: row statistics (one), then the activation rows in
// three groups of eight (three), then the epilogue's operands (one); and 60-180 cross-lane shuffles per wave.  Here
//   * lane = (output column c of CW, row group rg of RG, k-subset ks of KS), CW * RG * KS = 64: a lane owns RPL rows and NF float4s
//     of k, a wave owns ONE k-block (K = KS * 4 NF * waves: every launch shape of a window), so a lane's whole input — NF weight
//     float4s and RPL x NF activation float4s — is requested at once, and so are bias / residual / column sums by the lanes that
//     will finish the outputs; splitting the rows over lanes instead of the k range leaves log2(KS) = 1-3 shuffle steps on RPL values;
//   * LayerNorm is applied algebraically, as in the offline GEMMs (gemm_x6.hip): with gamma folded into the weights at load
//     (W' = W gamma, b' = b + W beta, s_n = sum_k W'_nk)      LN(x) W^T = rstd_i * (x W'^T - mean_i * s_n) + b'_n,
//     and mean_i / rstd_i come out of the SAME loaded values: every lane forms (mean, M2) of its 4 NF elements of a row (two passes
//     over registers), the k-subsets of a wave and then the waves are merged with Chan's formula (equal counts) next to the dot
//     products — as accurate as the two-pass form, no second pass over x.
template <bool LN, bool FS, int CW, int CPL, int RG, int RPL, int NF>
__global__ __launch_bounds__(1024) void fused_gemv1t_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw, float* C, int ldc, const float* __restrict__ bias,
    const float* __restrict__ colsum, float eps, const float* R1, int ldr1, const float* __restrict__ V, int ldv,
    const float* __restrict__ fw, int M, int N, int K, int relu) {
  constexpr int CG = CW / CPL;           // column groups: a lane owns CPL adjacent output columns
  constexpr int KS = 64 / (CG * RG);     // k-subsets per wave
  constexpr int KL = 4 * NF;             // k per lane
  constexpr int KI = KS * KL;            // k per wave
  constexpr int MR = RG * RPL;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  __shared__ float red[kWaves][MR * CW];
  __shared__ float2 st[kWaves][MR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const int cg = lane % CG, rg = (lane / CG) % RG, ks = lane / (CG * RG);
  const int n0 = blockIdx.x * CW;
  const int k = wave * KI + ks * KL;
  // everything this lane will ever read, in flight together
  f32x4 w[CPL][NF], a[RPL][NF];
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc)
#pragma unroll
    for (int j = 0; j < NF; ++j)         // weights are read once per window: non-temporal
      w[cc][j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + (size_t)(n0 + cg * CPL + cc) * ldw + k + 4 * j));
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    const int row = rg * RPL + i;
    const float* xr = X + (size_t)(row < M ? row : M - 1) * ldx + k;
#pragma unroll
    for (int j = 0; j < NF; ++j) a[i][j] = *reinterpret_cast<const f32x4*>(xr + 4 * j);
  }
  const int erow = tid / CW, ecol = tid % CW, gcol = n0 + ecol;
  const bool fin = tid < MR * CW && erow < M && gcol < N;
  float e_bias = 0.f, e_cs = 0.f, e_r1 = 0.f;
  if (fin) {
    if (bias) e_bias = bias[gcol];
    if (LN) e_cs = colsum[gcol];
    if (R1) e_r1 = R1[(size_t)erow * ldr1 + gcol];
  }
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    float d[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
      d[cc] = 0.f;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        d[cc] = fmaf(a[i][j][0], w[cc][j][0], d[cc]); d[cc] = fmaf(a[i][j][1], w[cc][j][1], d[cc]);
        d[cc] = fmaf(a[i][j][2], w[cc][j][2], d[cc]); d[cc] = fmaf(a[i][j][3], w[cc][j][3], d[cc]);
      }
    }
    float mu = 0.f, q = 0.f;
    if (LN) {
#pragma unroll
      for (int j = 0; j < NF; ++j) mu += (a[i][j][0] + a[i][j][1]) + (a[i][j][2] + a[i][j][3]);
      mu *= 1.0f / (float)KL;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const float d0 = a[i][j][0] - mu, d1 = a[i][j][1] - mu, d2 = a[i][j][2] - mu, d3 = a[i][j][3] - mu;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
    float half_n = 0.5f * (float)KL;     // half the element count of each side of a merge
#pragma unroll
    for (int o = CG * RG; o < 64; o <<= 1) {
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) d[cc] += __shfl_xor(d[cc], o);
      if (LN) {
        const float mo = __shfl_xor(mu, o), qo = __shfl_xor(q, o);
        const float dm = mo - mu;
        q = (q + qo) + dm * dm * half_n;
        mu = 0.5f * (mu + mo);
        half_n *= 2.0f;
      }
    }
    if (ks == 0) {
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) red[wave][(rg * RPL + i) * CW + cg * CPL + cc] = d[cc];
    }
    if (LN && ks == 0 && cg == 0) st[wave][rg * RPL + i] = make_float2(mu, q);
  }
  // the FSMN memory's operands (output projection of the encoder): requested before the barrier, consumed after it
  float fv[11], fk[11];
  if (FS) {
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const int t = erow + j - 5;
      const bool in = fin && t >= 0 && t < M;
      fv[j] = in ? V[(size_t)t * ldv + gcol] : 0.f;
      fk[j] = fin ? fw[gcol * 11 + j] : 0.f;
    }
  }
  __syncthreads();
  if (!fin) return;
  float v = 0.f;
  for (int w2 = 0; w2 < n_waves; ++w2) v += red[w2][tid];
  if (LN) {
    float msum = 0.f, m2 = 0.f;
    for (int w2 = 0; w2 < n_waves; ++w2) msum += st[w2][erow].x;
    const float mean = msum / (float)n_waves;
    for (int w2 = 0; w2 < n_waves; ++w2) { const float dm = st[w2][erow].x - mean; m2 += st[w2][erow].y + (float)KI * dm * dm; }
    const float rstd = 1.0f / sqrtf(m2 / (float)K + eps);
    v = rstd * (v - mean * e_cs);
  }
  v += e_bias;
  v += e_r1;
  if (FS) {
    float mem = fv[5];
#pragma unroll
    for (int j = 0; j < 11; ++j) mem += fk[j] * fv[j];
    v += mem;
  }
  if (relu) v = fmaxf(v, 0.f);
  C[(size_t)erow * ldc + gcol] = v;
}


// ---- attention over ONE window ----------------------------------------------------------------------------------------------
// O[q, h*128:(h+1)*128] = softmax(scale * Q_h K_h^T) V_h for Lq <= 32 queries and Lk <= 32 keys (the encoder's self-attention over a
// 20-row window; the decoder's few tokens against it), d_k = 128.  The general kernel (attention.hip) is built for hundreds of
// keys — query blocks, K/V tiles through LDS, online softmax — and takes 8.6 us here; this one is one workgroup per head, four
// waves: every operand is requested at kernel start (Q / K fragments of the wave's 32-wide slice of d, the V rows of its 32-wide
// slice of the output), scores = four partial 32 x 32 tiles on the fp32 MFMA summed through LDS, one softmax pass with 32 lanes per
// query, P V as ceil(Lk / 2) MFMAs per wave.  Same arithmetic type as the general kernel (fp32 MFMA = fmaf chains, exp2).
// Grid (head, window): with segment arrays (device: row offsets and lengths per window, as launch_attention takes them) a round of
// connections is one launch of H x B small workgroups.
__global__ __launch_bounds__(256) void window_attention_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                               const float* __restrict__ V, int ldv, float* __restrict__ O, int ldo,
                                                               int Lq, int Lk, float scale_log2e, const int* __restrict__ q_off,
                                                               const int* __restrict__ q_len, const int* __restrict__ kv_off,
                                                               const int* __restrict__ kv_len, const float* __restrict__ fsmn_w,
                                                               float* __restrict__ mem, int ldmem) {
  __shared__ float sp[4][32][33];
  __shared__ float P[32][33];
  if (q_off) {
    const int b = blockIdx.y;
    Lq = q_len[b]; Lk = kv_len[b];
    if (Lq <= 0 || Lk <= 0) return;
    Q += (size_t)q_off[b] * ldq; O += (size_t)q_off[b] * ldo;
    K += (size_t)kv_off[b] * ldk; V += (size_t)kv_off[b] * ldv;
    if (mem) mem += (size_t)kv_off[b] * ldmem;
  }
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 qf[4], kf[4];
  const float* qp = Q + (size_t)(r < Lq ? r : 0) * ldq + h * 128 + w * 32 + 4 * hh;
  const float* kp = K + (size_t)(r < Lk ? r : 0) * ldk + h * 128 + w * 32 + 4 * hh;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    qf[u] = r < Lq ? *reinterpret_cast<const float4*>(qp + 8 * u) : zero4;
    kf[u] = r < Lk ? *reinterpret_cast<const float4*>(kp + 8 * u) : zero4;
  }
  float vf[16];
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    const int key = 2 * st + hh;
    vf[st] = key < Lk ? V[(size_t)key * ldv + h * 128 + w * 32 + r] : 0.f;
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u].x, kf[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u].y, kf[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u].z, kf[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u].w, kf[u].w, acc, 0, 0, 0);
  }
  // D[i = query][j = key]: j = lane & 31, i = (e & 3) + 8 * (e >> 2) + 4 * hh
#pragma unroll
  for (int e = 0; e < 16; ++e) sp[w][(e & 3) + 8 * (e >> 2) + 4 * hh][r] = acc[e];
  // SAN-M memory block of the window (self-attention only: the V rows are the window's own): mem[t][c] = v[t][c] + sum_j
  // w[c][j] v[t + j - 5][c], rows outside [0, Lk) zero — launch_fsmn's operation order.  This lane holds column c = h*128 + w*32 + r
  // of the rows with its parity (vf[st] = row 2 st + hh); the other parity comes from lane ^ 32.
  if (fsmn_w) {
    const int c = h * 128 + w * 32 + r;
    float wk[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) wk[j] = fsmn_w[(size_t)c * 11 + j];
    float col[32];
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const float other = __shfl_xor(vf[st], 32);
      col[2 * st] = hh ? other : vf[st];
      col[2 * st + 1] = hh ? vf[st] : other;
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int t = 2 * st + par;                        // both lanes walk all rows; each stores its own parity
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 11; ++j) {
          const int u = t + j - 5;
          if (u >= 0 && u < 32) a += wk[j] * col[u];       // rows >= Lk were loaded as zero
        }
        if (par == hh && t < Lk) mem[(size_t)t * ldmem + c] = col[t] + a;
      }
    }
  }
  __syncthreads();
  for (int row = tid >> 5; row < Lq; row += 8) {          // 32 lanes per query
    const int k = tid & 31;
    float sc = ((sp[0][row][k] + sp[1][row][k]) + (sp[2][row][k] + sp[3][row][k])) * scale_log2e;
    if (k >= Lk) sc = -INFINITY;
    float mx = sc;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float ex = k < Lk ? exp2f(sc - mx) : 0.f;
    float sum = ex;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    P[row][k] = ex / sum;
  }
  __syncthreads();
  f32x16 o2;
#pragma unroll
  for (int e = 0; e < 16; ++e) o2[e] = 0.f;
  const int steps = (Lk + 1) >> 1;
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    if (st < steps) {
      const float a = r < Lq ? P[r][2 * st + hh] : 0.f;   // keys >= Lk carry probability 0 (P's row was written up to column 31)
      o2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vf[st], o2, 0, 0, 0);
    }
  }
  // D[i = query][j = d]: a wave stores 32 consecutive floats per (e, hh)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int q = (e & 3) + 8 * (e >> 2) + 4 * hh;
    if (q < Lq) O[(size_t)q * ldo + h * 128 + w * 32 + r] = o2[e];
  }
}


// ---- attention of one window + output projection (+ FSMN memory, + residual) in ONE launch -----------------------------------------
// The projection's operand is the whole context matrix (all four heads), so every workgroup of the projection redoes the window's
// attention (4 heads x 32 x 32 x 128 x 2 products: 1 MFLOP padded, ~3 us of fp32 MFMA on one CU) and keeps the result in LDS — against
// a launch boundary, a wave ramp and a second trip to memory for a separate attention kernel.  16 waves = 4 heads x 4 (d-quarters for
// the scores, 32-wide output slices for P V) exactly as window_attention_kernel; the GEMV part is fused_gemv1t_kernel's
// <CW 4, 2 columns per lane, 4 row groups, 8 k-subsets> shape with the activation float4s read from LDS.  d_model = 4 x 128.
template <bool FS, int RPL>
__global__ __launch_bounds__(1024) void fused_att_out_kernel(
    const float* __restrict__ Q, int ldq, const float* __restrict__ Kx, int ldk, const float* __restrict__ Vx, int ldv, int Lq, int Lk,
    float scale_log2e, const float* __restrict__ W, int ldw, float* C, int ldc, const float* __restrict__ bias, const float* R1, int ldr1,
    const float* __restrict__ FV, int ldfv, const float* __restrict__ fw, int N) {
  constexpr int CW = 4, CPL = 2, CG = 2, RG = 4, MR = RG * RPL, D = 512, CS = D + 4;      // 8 k-subsets of 4 k per wave
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float* const ctx = reinterpret_cast<float*>(lds_raw);                         // [32][CS]; the score partials live here first
  float (*sp)[4][32][33] = reinterpret_cast<float (*)[4][32][33]>(lds_raw);     // [head][d-quarter][query][key]
  float (*P)[32][33] = reinterpret_cast<float (*)[32][33]>(lds_raw + 4 * 4 * 32 * 33 * 4);   // [head][query][key]
  float (*red)[MR * CW] = reinterpret_cast<float (*)[MR * CW]>(lds_raw + 4 * 4 * 32 * 33 * 4 + 4 * 32 * 33 * 4);
  static_assert(32 * CS * 4 <= 4 * 4 * 32 * 33 * 4, "context tile must fit where the score partials were");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = wave >> 2, w = wave & 3, r = lane & 31, hh = lane >> 5;
  const int cg = lane % CG, rg = (lane / CG) % RG, ks = lane / (CG * RG);
  const int n0 = blockIdx.x * CW;
  const int k = wave * 32 + ks * 4;
  // every operand of the launch, requested together
  f32x4 wv[CPL];
#pragma unroll
  for (int cc = 0; cc < CPL; ++cc)
    wv[cc] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + (size_t)(n0 + cg * CPL + cc) * ldw + k));
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 qf[4], kf[4];
  const float* qp = Q + (size_t)(r < Lq ? r : 0) * ldq + h * 128 + w * 32 + 4 * hh;
  const float* kp = Kx + (size_t)(r < Lk ? r : 0) * ldk + h * 128 + w * 32 + 4 * hh;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    qf[u] = r < Lq ? *reinterpret_cast<const float4*>(qp + 8 * u) : zero4;
    kf[u] = r < Lk ? *reinterpret_cast<const float4*>(kp + 8 * u) : zero4;
  }
  float vf[16];
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    const int key = 2 * st + hh;
    vf[st] = key < Lk ? Vx[(size_t)key * ldv + h * 128 + w * 32 + r] : 0.f;
  }
  const int erow = tid / CW, ecol = tid % CW, gcol = n0 + ecol;
  const bool fin = tid < MR * CW && erow < Lq && gcol < N;
  float e_bias = 0.f, e_r1 = 0.f;
  float fv[11], fk[11];
  if (fin) {
    if (bias) e_bias = bias[gcol];
    if (R1) e_r1 = R1[(size_t)erow * ldr1 + gcol];
  }
  if (FS) {
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const int t = erow + j - 5;
      const bool in = fin && t >= 0 && t < Lq;
      fv[j] = in ? FV[(size_t)t * ldfv + gcol] : 0.f;
      fk[j] = fin ? fw[gcol * 11 + j] : 0.f;
    }
  }
  // ---- scores: four partial tiles per head
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u].x, kf[u].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u].y, kf[u].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u].z, kf[u].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[u].w, kf[u].w, acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) sp[h][w][(e & 3) + 8 * (e >> 2) + 4 * hh][r] = acc[e];
  __syncthreads();
  // ---- softmax: 32 lanes per (head, query); 4 x Lq rows over 32 half-waves
  for (int row = tid >> 5; row < 4 * Lq; row += 32) {
    const int hd = row / Lq, q = row - hd * Lq, kk = tid & 31;
    float sc = ((sp[hd][0][q][kk] + sp[hd][1][q][kk]) + (sp[hd][2][q][kk] + sp[hd][3][q][kk])) * scale_log2e;
    if (kk >= Lk) sc = -INFINITY;
    float mx = sc;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float ex = kk < Lk ? exp2f(sc - mx) : 0.f;
    float sum = ex;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    P[hd][q][kk] = ex / sum;
  }
  __syncthreads();                       // probabilities complete; the score partials are dead: ctx may overwrite them
  f32x16 o2;
#pragma unroll
  for (int e = 0; e < 16; ++e) o2[e] = 0.f;
  const int steps = (Lk + 1) >> 1;
#pragma unroll
  for (int st = 0; st < 16; ++st) {
    if (st < steps) {
      const float a = r < Lq ? P[h][r][2 * st + hh] : 0.f;
      o2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vf[st], o2, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) ctx[((e & 3) + 8 * (e >> 2) + 4 * hh) * CS + h * 128 + w * 32 + r] = o2[e];
  __syncthreads();
  // ---- projection: this lane's RPL rows x CPL columns over its 4 k
#pragma unroll
  for (int i = 0; i < RPL; ++i) {
    const int row = rg * RPL + i;
    const f32x4 a = *reinterpret_cast<const f32x4*>(ctx + (row < Lq ? row : Lq - 1) * CS + k);
    float d[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
      d[cc] = a[0] * wv[cc][0];
      d[cc] = fmaf(a[1], wv[cc][1], d[cc]); d[cc] = fmaf(a[2], wv[cc][2], d[cc]); d[cc] = fmaf(a[3], wv[cc][3], d[cc]);
    }
#pragma unroll
    for (int o = CG * RG; o < 64; o <<= 1) {
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) d[cc] += __shfl_xor(d[cc], o);
    }
    if (ks == 0) {
#pragma unroll
      for (int cc = 0; cc < CPL; ++cc) red[wave][row * CW + cg * CPL + cc] = d[cc];
    }
  }
  __syncthreads();
  if (!fin) return;
  float v = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < kWaves; ++w2) v += red[w2][tid];
  v += e_bias;
  v += e_r1;
  if (FS) {
    float mem = fv[5];
#pragma unroll
    for (int j = 0; j < 11; ++j) mem += fk[j] * fv[j];
    v += mem;
  }
  C[(size_t)erow * ldc + gcol] = v;
}

}  // namespace

template <bool LN, int CW>
static void launch_gemv(const float* X, int ldx, int D, const float* g, const float* b, float eps, const float* W, int ldw, float* C,
                        int ldc, const float* bias, const float* R1, int ldr1, const float* R2, int ldr2, const float* fsmn_v, int ldv,
                        const float* fsmn_w, int M, int N, int K, bool relu, hipStream_t s) {
  constexpr int KI = 4 * (64 / CW);
  // LayerNorm needs 32 threads per row for its statistics: all 16 waves; without it, one wave per k-block is enough
  const int waves = g ? kWaves : std::max(1, std::min(kWaves, K / KI));
  const dim3 grid((N + CW - 1) / CW), block(64 * waves);
#define PFHIP_LAUNCH(MR_)                                                                                                            \
  hipLaunchKernelGGL((fused_ln_gemv_kernel<LN, CW, MR_>), grid, block, 0, s, X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, \
                     ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu ? 1 : 0)
  if (M <= 8) PFHIP_LAUNCH(8); else if (M <= 16) PFHIP_LAUNCH(16); else if (M <= 24) PFHIP_LAUNCH(24); else PFHIP_LAUNCH(32);
#undef PFHIP_LAUNCH
}

void launch_fused_ln_gemm(const float* X, int ldx, int D, const float* g, const float* b, float eps, const float* W, int ldw,
                          float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* R2, int ldr2,
                          const float* fsmn_v, int ldv, const float* fsmn_w, int M, int N, int K, bool relu, hipStream_t s) {
  if (M <= 0 || N <= 0) return;
  // patch width: at least ~128 workgroups per launch (N = 512 -> 4 columns, 1024-2048 -> 8, the vocabulary -> 32)
  // (N = 512 at K = 2048: 2 columns -> 256 workgroups and ONE k-block per wave instead of two: three dependent round trips to L2
  // per lane instead of six)
  const int cw = N <= 640 ? (K >= 1024 ? 2 : 4) : (N <= 4096 ? 8 : 32);
  static const bool use_mfma = [] { const char* e = getenv("PFHIP_STREAM_FUSED_MFMA"); return e && e[0] == '1'; }();
  if (!use_mfma && K % 64 == 0 && N <= 4096) {
    if (g) {
      if (cw == 2) launch_gemv<true, 2>(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu, s);
      else if (cw == 4) launch_gemv<true, 4>(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu, s);
      else if (cw == 8) launch_gemv<true, 8>(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu, s);
      else launch_gemv<true, 32>(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu, s);
    } else {
      if (cw == 2) launch_gemv<false, 2>(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu, s);
      else if (cw == 4) launch_gemv<false, 4>(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu, s);
      else if (cw == 8) launch_gemv<false, 8>(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu, s);
      else launch_gemv<false, 32>(X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu, s);
    }
    return;
  }
  const int cwm = cw == 2 ? 4 : cw;
  const dim3 grid((N + cwm - 1) / cwm), block(kThreads);
#define PFHIP_LAUNCH(LN_, CW_)                                                                                                    \
  hipLaunchKernelGGL((fused_ln_gemm_kernel<LN_, CW_>), grid, block, 0, s, X, ldx, D, g, b, eps, W, ldw, C, ldc, bias, R1, ldr1, R2, \
                     ldr2, fsmn_v, ldv, fsmn_w, M, N, K, relu ? 1 : 0)
  if (g) {
    if (cwm == 4) PFHIP_LAUNCH(true, 4); else if (cwm == 8) PFHIP_LAUNCH(true, 8); else PFHIP_LAUNCH(true, 32);
  } else {
    if (cwm == 4) PFHIP_LAUNCH(false, 4); else if (cwm == 8) PFHIP_LAUNCH(false, 8); else PFHIP_LAUNCH(false, 32);
  }
#undef PFHIP_LAUNCH
}


// One-trip form (fused_gemv1t_kernel).  W / bias are the LayerNorm-folded ones when ln_colsum is given (the caller's LayerNorm is
// over exactly the K operand columns).  False when the shape is outside what the kernel takes — the caller falls back to
// launch_fused_ln_gemm with the plain weights.  PFHIP_STREAM_1TRIP=0 turns it off.
bool launch_fused_gemv_1trip(const float* X, int ldx, const float* W, int ldw, float* C, int ldc, const float* bias, const float* ln_colsum,
                             float eps, const float* R1, int ldr1, const float* fsmn_v, int ldv, const float* fsmn_w, int M, int N, int K,
                             bool relu, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("PFHIP_STREAM_1TRIP"); return !(e && e[0] == '0'); }();
  if (!on || M <= 0 || M > 20 || N <= 0 || (fsmn_v && ln_colsum)) return false;
  // patch width as in launch_fused_ln_gemm; rows over 4 lane groups; a wave per k-block: K = waves * KS * 4 NF
  static const int cpl = [] { const char* e = getenv("PFHIP_1T_CPL"); return e ? atoi(e) : 2; }();     // columns per lane (measured: 2 beats 1 and 4 on every window shape)
  const int cw = N <= 640 ? (K >= 2048 ? 2 : 4) : 8;
  if (cpl > cw || (cpl != 1 && cpl != 2 && cpl != 4)) return false;
  const int ks = 64 / (cw / cpl * 4);
  int nf = K / (kWaves * ks * 4) >= 1 ? K / (kWaves * ks * 4) : 1;
  // N = 512, K = 512 (4-column patches): 8 waves with twice the k per lane beat 16 (4.45-4.66 vs 4.85 us: half the partial sums)
  if (cw == 4 && 2 * nf <= 4 && K % (ks * 4 * 2 * nf) == 0 && 64 * (K / (ks * 4 * 2 * nf)) >= 20 * cw) nf *= 2;
  const int ki = ks * 4 * nf;
  if (K % ki || K / ki > kWaves || N % cw || nf < 1 || nf > 4) return false;
  if (64 * (K / ki) < (M <= 8 ? 8 : 20) * cw) return false;       // the lanes that finish the outputs must exist (tiny K)
  const dim3 grid(N / cw), block(64 * (K / ki));
#define PFHIP_LAUNCH1T(LN_, FS_, CW_, CPL_, RPL_, NF_)                                                                                      \
  hipLaunchKernelGGL((fused_gemv1t_kernel<LN_, FS_, CW_, CPL_, 4, RPL_, NF_>), grid, block, 0, s, X, ldx, W, ldw, C, ldc, bias, ln_colsum, eps, \
                     R1, ldr1, fsmn_v, ldv, fsmn_w, M, N, K, relu ? 1 : 0)
#define PFHIP_PICK_MR(LN_, FS_, CW_, CPL_, NF_)                                   \
  do { if (M <= 8) PFHIP_LAUNCH1T(LN_, FS_, CW_, CPL_, 2, NF_); else PFHIP_LAUNCH1T(LN_, FS_, CW_, CPL_, 5, NF_); } while (0)
#define PFHIP_PICK_NF(LN_, FS_, CW_, CPL_)                                        \
  do { if (nf == 1) PFHIP_PICK_MR(LN_, FS_, CW_, CPL_, 1); else if (nf == 2) PFHIP_PICK_MR(LN_, FS_, CW_, CPL_, 2); else if (nf == 3) PFHIP_PICK_MR(LN_, FS_, CW_, CPL_, 3); else PFHIP_PICK_MR(LN_, FS_, CW_, CPL_, 4); } while (0)
#define PFHIP_PICK_CW(LN_, FS_)                                                                       \
  do {                                                                                                \
    if (cpl == 1) { if (cw == 2) PFHIP_PICK_NF(LN_, FS_, 2, 1); else if (cw == 4) PFHIP_PICK_NF(LN_, FS_, 4, 1); else PFHIP_PICK_NF(LN_, FS_, 8, 1); } \
    else if (cpl == 2) { if (cw == 2) PFHIP_PICK_NF(LN_, FS_, 2, 2); else if (cw == 4) PFHIP_PICK_NF(LN_, FS_, 4, 2); else PFHIP_PICK_NF(LN_, FS_, 8, 2); } \
    else { if (cw == 4) PFHIP_PICK_NF(LN_, FS_, 4, 4); else PFHIP_PICK_NF(LN_, FS_, 8, 4); }                                                              \
  } while (0)
  if (ln_colsum) PFHIP_PICK_CW(true, false);
  else if (fsmn_v) PFHIP_PICK_CW(false, true);
  else PFHIP_PICK_CW(false, false);
#undef PFHIP_PICK_CW
#undef PFHIP_PICK_NF
#undef PFHIP_PICK_MR
#undef PFHIP_LAUNCH1T
  return true;
}

// window_attention_kernel: false when the shape is outside what it takes (the caller uses launch_attention).
// PFHIP_STREAM_WATT=0 turns it off.
bool launch_window_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, int Lq, int Lk,
                             int H, float scale, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("PFHIP_STREAM_WATT"); return !(e && e[0] == '0'); }();
  if (!on || Lq < 1 || Lq > 32 || Lk < 1 || Lk > 32 || H < 1) return false;
  hipLaunchKernelGGL(window_attention_kernel, dim3(H), dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, Lq, Lk,
                     scale * 1.4426950408889634f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  return true;
}
// B windows in one launch: segment arrays on the device, max_q_len / max_kv_len their host-side maxima (both <= 32 or false).
// fsmn_w != nullptr (self-attention: q segments == kv segments, V = H * 128 channels wide): the launch also writes the encoder
// layer's FSMN memory of V into mem (what launch_fsmn computes) — one launch per layer less in a round of connections.
bool launch_window_attention_segments(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                                      const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                                      int max_q_len, int max_kv_len, float scale, hipStream_t s, const float* fsmn_w, float* mem,
                                      int ldmem) {
  static const bool on = [] { const char* e = getenv("PFHIP_STREAM_WATT"); return !(e && e[0] == '0'); }();
  if (!on || B < 1 || H < 1 || max_q_len < 1 || max_q_len > 32 || max_kv_len < 1 || max_kv_len > 32) return false;
  hipLaunchKernelGGL(window_attention_kernel, dim3(H, B), dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, 0, 0,
                     scale * 1.4426950408889634f, q_off, q_len, kv_off, kv_len, fsmn_w, mem, ldmem);
  return true;
}

// fused_att_out_kernel: attention of one window (Lq <= 20 queries, Lk <= 32 keys, 4 heads of 128) + projection by W [N, 512]
// (+bias, +R1, + FSMN memory of fsmn_v over the Lq rows).  False when the shape is outside what it takes, and in the streaming
// path unless PFHIP_STREAM_ATT_OUT=1: measured, the fused launch costs ~17 us where window_attention_kernel (7.1) + boundary (1.5) +
// projection (5.4) cost 14 — 128 workgroups x 16 waves each redoing 3 us of fp32 MFMA and 24 operand loads per lane is more than
// the launch it saves (2.92 against 2.70 ms per chunk on the same box).  Kept, tested, as the record of that.
bool launch_fused_att_out(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int Lq, int Lk, int H, float scale,
                          const float* W, int ldw, float* C, int ldc, const float* bias, const float* R1, int ldr1, const float* fsmn_v,
                          int ldfv, const float* fsmn_w, int N, hipStream_t s) {
  if (H != 4 || Lq < 1 || Lq > 20 || Lk < 1 || Lk > 32 || N < 4 || N % 4) return false;
  const size_t lds = (size_t)4 * 4 * 32 * 33 * 4 + 4 * 32 * 33 * 4 + (size_t)kWaves * 20 * 4 * 4;       // 89,600 B
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_done.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_att_out_kernel<true, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_att_out_kernel<false, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_att_out_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fused_att_out_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done.fetch_or(1ull << (dev & 63));
  }
  const float sl = scale * 1.4426950408889634f;
  const dim3 grid(N / 4), block(1024);
#define PFHIP_LAUNCH_AO(FS_, RPL_)                                                                                                    \
  hipLaunchKernelGGL((fused_att_out_kernel<FS_, RPL_>), grid, block, lds, s, Q, ldq, K, ldk, V, ldv, Lq, Lk, sl, W, ldw, C, ldc, bias, R1, \
                     ldr1, fsmn_v, ldfv, fsmn_w, N)
  if (fsmn_v) { if (Lq <= 8) PFHIP_LAUNCH_AO(true, 2); else PFHIP_LAUNCH_AO(true, 5); }
  else { if (Lq <= 8) PFHIP_LAUNCH_AO(false, 2); else PFHIP_LAUNCH_AO(false, 5); }
#undef PFHIP_LAUNCH_AO
  return true;
}

}  // namespace pfhip
