// PROTOTYPE, not part of the library: measured 6.7 ms per chunk against 2.46 for the launch path — kept as the record.
// It was wired in through a per-layer pointer table built at load and a call in stream.cpp's one-connection path (layers 1 .. L-1;
// flag read back after the launch, fallback to the launches on a time-out); ids matched the streaming oracle on the small and the
// full-size model.  In situ: barrier 1.5 us, the L1 invalidate (`buffer_inv sc1`, once per CU; once per WAVE it was ~50 us) another
// 2 us, and the phases with this simple MFMA body 25 (LN+QKV) / 1.3 (attention) / 12.7 (projection) / 26 (LN+FFN1) / 33 (FFN2) us per
// layer: a 32-column tile on the fp32 MFMA with K split over 16 waves pays its load trips one after the other (four at K = 2048)
// and 4 waves per SIMD queue on the matrix pipe.  Even with ideal phases (13 us of streaming per layer) the 245 barriers cost
// 0.4-0.9 ms per chunk, i.e. ~1.6-2.1 ms per chunk at best.
//
// The streaming encoder of ONE window as ONE persistent launch confined to ONE XCD (`ForwardChunk`'s encoder Run,
// onnxruntime/src/paraformer-online.cpp:426-466, at M <= 32 rows).
//
// The launch-per-operator path (stream_fused.hip) spends a quarter of a chunk in launch boundaries and starts every kernel cold.
// Inside one XCD the L2 is coherent, so 32 workgroups (one per CU of the XCD) can hand a 40-KB activation matrix to each other with
// plain stores, `s_waitcnt vmcnt(0)`, an L2 atomic and a bounded spin — 1.0 us per barrier (tools/probe/xcd_chunk_probe.hip), and one
// XCD streams weights at 1.03 TB/s.  Here a layer is five phases — LN1 + QKV | window attention (4 workgroups, one per head) |
// output projection + FSMN memory + residual | LN2 + FFN1 | FFN2 + residual — over the layers [l0, l1) without leaving the kernel.
// The arithmetic per phase is what stream_fused.hip's first generation does per launch: 32-column tiles on the fp32 MFMA, K split over
// the 16 waves of a workgroup, partial tiles summed through LDS, LayerNorm two-pass on load (the plain weights: no folding needed).
//
// Safety: placement is by `blockIdx.x % 8` (round-robin dispatch puts those 32 workgroups on one XCD) and VERIFIED on the device
// through XCC_ID; every spin is bounded; a time-out or a misplacement raises a flag, the kernel still runs to its end, and the host
// then redoes the encoder with the launch path and stops using this kernel for the model (stream.cpp).
#include "kernels.h"

#include <math.h>

#include <atomic>
#include <cstdlib>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kXW = 16;                   // waves per workgroup
constexpr int kXT = kXW * 64;             // 1024 threads
constexpr int kXGroup = 32;               // workgroups that take part = CUs of one XCD
constexpr int kRedFloats = kXW * 32 * 33; // partial tiles of the 16 waves
constexpr int kXcdLds = kRedFloats * 4 + 2 * 32 * 4;

struct Ctx {
  float* red;        // [16][32 * 33]
  float* s_mean;     // [32]
  float* s_rstd;     // [32]
};

// all threads of every participating workgroup call this the same number of times
__device__ __forceinline__ void xcd_barrier(unsigned* bar, unsigned& want, unsigned* flag, int dbg = 0) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's stores have reached the L2
  __syncthreads();
  if (threadIdx.x == 0) {
    want += kXGroup;
    atomicAdd(bar, 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      if (++spins > 40000u) { *flag = 1u; break; }           // ~4 ms: a lost partner must not hang the GPU
      __builtin_amdgcn_s_sleep(1);
    }
    if (!(dbg & 1)) asm volatile("buffer_inv sc1" ::: "memory");            // this CU's L1 may hold lines the other CUs have rewritten: ONE
  }                                                         // invalidate per CU (every wave issuing its own cost ~50 us per barrier)
  __syncthreads();
}

// mean / rstd of rows [0, M) of X[., 0..D): 32 threads per row (1024 threads = 32 rows); ends with __syncthreads()
__device__ __noinline__ void row_stats_x(const float* X, int ldx, int M, int D, float eps, const Ctx& c) {
  const int tid = threadIdx.x, row = tid >> 5, cc = tid & 31;
  float mean = 0.f, rstd = 0.f;
  if (row < M) {
    const float* xr = X + (size_t)row * ldx;
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = 4 * cc + 128 * u;
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
      if (4 * cc + 128 * u < D) {
        const float a = v[u].x - mean, b = v[u].y - mean, c2 = v[u].z - mean, d = v[u].w - mean;
        q += (a * a + b * b) + (c2 * c2 + d * d);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
    rstd = 1.0f / sqrtf(q / (float)D + eps);
  }
  if (cc == 0) { c.s_mean[row] = mean; c.s_rstd[row] = rstd; }
  __syncthreads();
}

// C[M <= 32, n0 .. n0 + 32) = act(LN?(X) W^T + bias) (+R1) (+FSMN memory of V): one 32-column tile by the whole workgroup
// (fused_ln_gemm_kernel's arithmetic: fp32 MFMA 32x32x2, K split over the 16 waves, partial tiles through LDS).  Statistics for LN
// must be in c.s_mean / c.s_rstd.  Every thread reaches the trailing __syncthreads().
template <bool LN>
__device__ __noinline__ void mfma_tile(const float* X, int ldx, const float* __restrict__ g, const float* __restrict__ b,
                                          const float* __restrict__ W, int ldw, float* C, int ldc, const float* __restrict__ bias,
                                          const float* R1, int ldr1, const float* V, int ldv, const float* __restrict__ fw, int M, int N,
                                          int K, bool relu, int n0, const Ctx& c) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const float mean = LN ? c.s_mean[r] : 0.f, rstd = LN ? c.s_rstd[r] : 0.f;
  const int nkb = K >> 3, per = (nkb + kXW - 1) / kXW;
  const int kb0 = wave * per, kb1 = kb0 + per < nkb ? kb0 + per : nkb;
  const float* ap = X + (size_t)(r < M ? r : M - 1) * ldx + 4 * h;
  const bool wlive = n0 + r < N;
  const float* wp = W + (size_t)(wlive ? n0 + r : 0) * ldw + 4 * h;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int kb = kb0; kb < kb1; kb += 4) {
    float4 a[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool in = kb + u < kb1;
      a[u] = in ? *reinterpret_cast<const float4*>(ap + 8 * (kb + u)) : zero4;
      if (in && wlive) {
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + 8 * (kb + u)));       // weights: read once
        w[u] = make_float4(t[0], t[1], t[2], t[3]);
      } else {
        w[u] = zero4;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 an = a[u];
      if (LN && kb + u < kb1) {
        const int k = 8 * (kb + u) + 4 * h;
        const float4 gg = *reinterpret_cast<const float4*>(g + k);
        const float4 bb = *reinterpret_cast<const float4*>(b + k);
        an.x = (an.x - mean) * rstd * gg.x + bb.x; an.y = (an.y - mean) * rstd * gg.y + bb.y;
        an.z = (an.z - mean) * rstd * gg.z + bb.z; an.w = (an.w - mean) * rstd * gg.w + bb.w;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.x, w[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.y, w[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.z, w[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(an.w, w[u].w, acc, 0, 0, 0);
    }
  }
  // D[i = activation row][j = weight column]: j = lane & 31, i = (e & 3) + 8 * (e >> 2) + 4 * h
  float* red = c.red + wave * (32 * 33);
#pragma unroll
  for (int e = 0; e < 16; ++e) red[((e & 3) + 8 * (e >> 2) + 4 * h) * 33 + r] = acc[e];
  __syncthreads();
  {
    const int row = tid >> 5, col = tid & 31, gcol = n0 + col;
    if (row < M && gcol < N) {
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < kXW; ++w2) v += c.red[w2 * (32 * 33) + row * 33 + col];
      if (bias) v += bias[gcol];
      if (R1) v += R1[(size_t)row * ldr1 + gcol];
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
  }
  __syncthreads();                       // red reusable
}

// window attention of head `head` by waves 0-3 of the workgroup (window_attention_kernel's arithmetic); sp / P live where the
// partial tiles do.  Every thread reaches every __syncthreads().
__device__ __noinline__ void window_attention_head(const float* Q, int ldq, const float* Kx, int ldk, const float* Vx, int ldv, float* O,
                                                      int ldo, int L, int head, float scale_log2e, const Ctx& c) {
  float (*sp)[32][33] = reinterpret_cast<float (*)[32][33]>(c.red);            // [4][32][33]
  float (*P)[33] = reinterpret_cast<float (*)[33]>(c.red + 4 * 32 * 33);       // [32][33]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const bool act = w < 4;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 qf[4], kf[4];
  float vf[16];
  if (act) {
    const float* qp = Q + (size_t)(r < L ? r : 0) * ldq + head * 128 + w * 32 + 4 * hh;
    const float* kp = Kx + (size_t)(r < L ? r : 0) * ldk + head * 128 + w * 32 + 4 * hh;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qf[u] = r < L ? *reinterpret_cast<const float4*>(qp + 8 * u) : zero4;
      kf[u] = r < L ? *reinterpret_cast<const float4*>(kp + 8 * u) : zero4;
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int key = 2 * st + hh;
      vf[st] = key < L ? Vx[(size_t)key * ldv + head * 128 + w * 32 + r] : 0.f;
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
#pragma unroll
    for (int e = 0; e < 16; ++e) sp[w][(e & 3) + 8 * (e >> 2) + 4 * hh][r] = acc[e];
  }
  __syncthreads();
  {
    const int row = tid >> 5, kk = tid & 31;                 // 32 lanes per query, all 32 rows at once
    if (row < L) {
      float sc = ((sp[0][row][kk] + sp[1][row][kk]) + (sp[2][row][kk] + sp[3][row][kk])) * scale_log2e;
      if (kk >= L) sc = -INFINITY;
      float mx = sc;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float ex = kk < L ? exp2f(sc - mx) : 0.f;
      float sum = ex;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
      P[row][kk] = ex / sum;
    }
  }
  __syncthreads();
  if (act) {
    f32x16 o2;
#pragma unroll
    for (int e = 0; e < 16; ++e) o2[e] = 0.f;
    const int steps = (L + 1) >> 1;
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      if (st < steps) {
        const float a = r < L ? P[r][2 * st + hh] : 0.f;
        o2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vf[st], o2, 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (q < L) O[(size_t)q * ldo + head * 128 + w * 32 + r] = o2[e];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kXT, 1) void xcd_encoder_kernel(const XcdLayer* __restrict__ layers, int l0, int l1, float* x, float* qkv, float* ctx,
                                                             float* hbuf, int M, int d, int ffn, int n_head, float scale_log2e, float eps,
                                                             unsigned* bar, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  if (blockIdx.x % 8 != 0) return;                       // round-robin dispatch: the other seven XCDs' workgroups have nothing to do
  const int j = blockIdx.x / 8;
  if (j >= kXGroup) return;
  Ctx c;
  c.red = reinterpret_cast<float*>(lds_raw);
  c.s_mean = c.red + kRedFloats;
  c.s_rstd = c.s_mean + 32;
  unsigned* const flag = bar + 1;
  const unsigned my_xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
  if (threadIdx.x == 0) {
    if (j == 0) bar[2] = my_xcc + 1u;
  }
  unsigned want = 0;
  xcd_barrier(bar, want, flag, dbg);                          // everybody is here (and bar[2] is written)
  if (threadIdx.x == 0 && __hip_atomic_load(&bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_xcc + 1u) *flag = 2u;   // not one XCD
  for (int li = l0; li < l1; ++li) {
    const XcdLayer L = layers[li];
    // ---- LN1 + QKV: 3d / 32 tiles over the 32 workgroups
    if (!(dbg & 2)) row_stats_x(x, d, M, d, eps, c);
    for (int t = j; t < 3 * d / 32 && !(dbg & 2); t += kXGroup)
      mfma_tile<true>(x, d, L.n1g, L.n1b, L.qkvw, d, qkv, 3 * d, L.qkvb, nullptr, 0, nullptr, 0, nullptr, M, 3 * d, d, false, 32 * t, c);
    xcd_barrier(bar, want, flag, dbg);
    // ---- window attention: one workgroup per head
    if (j < n_head && !(dbg & 4)) window_attention_head(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, ctx, d, M, j, scale_log2e, c);
    xcd_barrier(bar, want, flag, dbg);
    // ---- output projection + FSMN memory of V + residual (x += ...)
    for (int t = j; t < d / 32 && !(dbg & 8); t += kXGroup)
      mfma_tile<false>(ctx, d, nullptr, nullptr, L.outw, d, x, d, L.outb, x, d, qkv + 2 * d, 3 * d, L.fsmnw, M, d, d, false, 32 * t, c);
    xcd_barrier(bar, want, flag, dbg);
    // ---- LN2 + FFN1 (ReLU)
    if (!(dbg & 16)) row_stats_x(x, d, M, d, eps, c);
    for (int t = j; t < ffn / 32 && !(dbg & 16); t += kXGroup)
      mfma_tile<true>(x, d, L.n2g, L.n2b, L.f1w, d, hbuf, ffn, L.f1b, nullptr, 0, nullptr, 0, nullptr, M, ffn, d, true, 32 * t, c);
    xcd_barrier(bar, want, flag, dbg);
    // ---- FFN2 + residual
    for (int t = j; t < d / 32 && !(dbg & 32); t += kXGroup)
      mfma_tile<false>(hbuf, ffn, nullptr, nullptr, L.f2w, ffn, x, d, L.f2b, x, d, nullptr, 0, nullptr, M, d, ffn, false, 32 * t, c);
    xcd_barrier(bar, want, flag, dbg);
  }
}

}  // namespace

// Layers [l0, l1) of the streaming encoder on the window x [M, d] (in place), one persistent launch on one XCD.  bar: 4 zeroed words
// on the device (the launch zeroes them itself); bar[1] != 0 afterwards = a barrier timed out (1) or the workgroups were not on one
// XCD (2): the results are then unusable and the caller must redo the layers another way.  Needs d % 32 == 0, d <= 2048, M <= 32,
// d == 128 * n_head, n_head <= 32.  Returns false (nothing launched) for other shapes.
bool launch_xcd_encoder(const XcdLayer* d_layers, int l0, int l1, float* x, float* qkv, float* ctx, float* hbuf, int M, int d, int ffn,
                        int n_head, float scale, unsigned* bar, hipStream_t s) {
  if (l1 <= l0 || M < 1 || M > 32 || d % 32 || d > 2048 || ffn % 32 || d != 128 * n_head || n_head > kXGroup) return false;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_done.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xcd_encoder_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kXcdLds);
    attr_done.fetch_or(1ull << (dev & 63));
  }
  (void)hipMemsetAsync(bar, 0, 16, s);
  hipLaunchKernelGGL(xcd_encoder_kernel, dim3(8 * kXGroup), dim3(kXT), kXcdLds, s, d_layers, l0, l1, x, qkv, ctx, hbuf, M, d, ffn, n_head,
                     scale * 1.4426950408889634f, 1e-12f, bar, getenv("PFHIP_XCD_DBG") ? atoi(getenv("PFHIP_XCD_DBG")) : 0);
  return true;
}

}  // namespace pfhip
