// PROBE (not built into libpfhip.so): measured 122-130 TF on the K = 512 shapes against 170-176 TF of the one-tile-per-workgroup
// 128 x 128 kernel (and 168 vs 193 TF at K = 2048): with the C buffer in LDS only one 16-wave workgroup fits a CU, a wave has 6 MFMAs
// per barrier and 2 MFMAs to hide 6 fragment reads behind — the loop loses more than the overlapped epilogue gains.  Kept as
// the record of that experiment (it passed the GEMM parity tests as kind 6).
// Persistent variant of the BF16-split GEMM (gemm_x6.hip) for the K = 512 shapes, whose tiles are short: one 16-wave workgroup
// per CU walks its tiles as ONE continuous K-stream — the operand loads, splits and fragment reads of the next tile's first
// K-steps are already in flight while the current tile finishes — and a finished tile's accumulators are parked in a separate
// LDS buffer and written out (bias / residuals / ReLU) in four row passes spread over the next tile's K-steps.  What the
// one-tile-per-workgroup kernels lose per tile (workgroup launch, prologue latency, an epilogue during which the matrix pipe
// of the CU idles because co-resident workgroups run in phase) is overlapped here.
// Tile 128 x 128, 16 waves as 4 x 4 (four per SIMD), each one 32 x 32 MFMA tile; K-step 16; LDS: two operand stages of
// 36,864 B + the 67,584-B C buffer.  Threads 0-511 stage A, 512-1023 stage W (one float4 each per K-step).
#include "kernels.h"

#include <algorithm>
#include <atomic>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int kPM = 128, kPN = 128, kPK = 16;
constexpr int kRowB = 48;
constexpr int kPlane = kPM * kRowB;                         // 6,144 B (A and W tiles have 128 rows each)
constexpr int kStageB = 6 * kPlane;                         // 36,864 B
constexpr int kCs = kPN + 4;
constexpr int kCBytes = kPM * kCs * 4;                      // 67,584 B
constexpr int kLdsBytes = 2 * kStageB + kCBytes;            // 141,312 B

__device__ __forceinline__ unsigned top16_pair(float lo, float hi) {
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }

__device__ __forceinline__ void tile_of_id(int bid, int n_tiles, int tiles_n, int gw, int& tm, int& tn) {
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

__global__ __launch_bounds__(1024, 1) void gemm_f32_bf16x6_persistent_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* C, int ldc,
    const float* __restrict__ bias, const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, int tiles_n,
    int n_tiles, int gw, int relu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* const Cs = reinterpret_cast<float*>(lds + 2 * kStageB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int nk = K / kPK;
  const int G = gridDim.x;
  const int my_tiles = (n_tiles - (int)blockIdx.x + G - 1) / G;

  // staging: threads 0-511 hold 4 consecutive k of A row t/4, threads 512-1023 of W row (t-512)/4
  const bool is_a = tid < 512;
  const int srow = (tid & 511) >> 2, sq = tid & 3;
  const int st_off = (is_a ? 0 : 3 * kPlane) + srow * kRowB + 8 * sq;
  const int a_fr = (wr * 32 + r) * kRowB + 16 * h;
  const int w_fr = 3 * kPlane + (wc * 32 + r) * kRowB + 16 * h;

  // load cursor: (tile index in my sequence, K-step) of the next raw load; runs three K-steps ahead of the MFMAs
  int ld_ti = 0, ld_kt = 0;
  const float* gp;
  auto set_tile_ptr = [&](int ti) {
    int tm, tn;
    tile_of_id((int)blockIdx.x + min(ti, my_tiles - 1) * G, n_tiles, tiles_n, gw, tm, tn);
    gp = is_a ? A + (size_t)min(tm * kPM + srow, M - 1) * lda + 4 * sq : W + (size_t)min(tn * kPN + srow, N - 1) * ldw + 4 * sq;
  };
  set_tile_ptr(0);
  float4 xr, yr;
#define PFHIP_LOAD_NEXT(R)                                     \
  R = *reinterpret_cast<const float4*>(gp + ld_kt * kPK);      \
  if (++ld_kt == nk) { ld_kt = 0; set_tile_ptr(++ld_ti); }
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

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  bf16x8 fa[3], fb[3], ga[3], gb[3];
#define PFHIP_FRAGS(FA, FB, stage)                                                                                        \
  _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                                         \
    FA[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (stage) * kStageB + p * kPlane + a_fr));     \
    FB[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + (stage) * kStageB + p * kPlane + w_fr));     \
  }
#define PFHIP_MM(FA, FB, pa, pb) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[pa], FB[pb], acc, 0, 0, 0);
#define PFHIP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

  // the tile being computed, the tile parked in the C buffer
  int cur_ti = 0, kt = 0;
  int cm0 = 0, cn0 = 0;
  bool parked = false;
  const int c4 = tid & 31, rsub = tid >> 5;
  const int dstride = nk >> 2;

  auto drain_pass = [&](int p) {          // rows 32p .. 32p+31 of the parked tile: bias / residuals / ReLU, 16-byte accesses
    const int row = p * 32 + rsub;
    const int grow = cm0 + row, gcol = cn0 + 4 * c4;
    float4 v = *reinterpret_cast<const float4*>(Cs + row * kCs + 4 * c4);
    if (bias) {
      if (gcol + 3 < N) { const float4 b = *reinterpret_cast<const float4*>(bias + gcol); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      else {
        if (gcol < N) v.x += bias[gcol];
        if (gcol + 1 < N) v.y += bias[gcol + 1];
        if (gcol + 2 < N) v.z += bias[gcol + 2];
      }
    }
    if (grow < M && gcol + 3 < N) {
      if (R1) { const float4 t = *reinterpret_cast<const float4*>(R1 + (size_t)grow * ldr1 + gcol); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      if (R2) { const float4 t = *reinterpret_cast<const float4*>(R2 + (size_t)grow * ldr2 + gcol); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
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
  };
  auto park = [&]() {                      // accumulators -> C buffer (C/D map: col = lane&31, row = (e&3)+8*(e>>2)+4*(lane>>5))
    int tm, tn;
    tile_of_id((int)blockIdx.x + cur_ti * G, n_tiles, tiles_n, gw, tm, tn);
    cm0 = tm * kPM; cn0 = tn * kPN;
    float* cw = Cs + (wr * 32 + 4 * h) * kCs + wc * 32 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) { cw[((e & 3) + 8 * (e >> 2)) * kCs] = acc[e]; acc[e] = 0.f; }
    parked = true;
  };

  // One K-step of the stream.  Region 1: 4 MFMAs, between them the split of the next step's operand piece (22 VALU ops,
  // 3 LDS writes) and the global load three steps ahead; a row pass of the parked tile when one is due; barrier
  // (`s_waitcnt lgkmcnt(0); s_barrier`: __syncthreads() would also drain the global loads); region 2: 2 MFMAs and the 6
  // fragment reads of the next step; at a tile's last step the accumulators are parked.
#define PFHIP_STEP(FA, FB, GA, GB, R, nxt)                                                    \
  split3(R, lds + (nxt) * kStageB + st_off);                                                  \
  PFHIP_LOAD_NEXT(R)                                                                          \
  PFHIP_MM(FA, FB, 1, 1) PFHIP_MM(FA, FB, 0, 2) PFHIP_MM(FA, FB, 2, 0) PFHIP_MM(FA, FB, 0, 1)  \
  PFHIP_SGB(0x8, 1); PFHIP_SGB(0x2, 8); PFHIP_SGB(0x200, 1);                                  \
  PFHIP_SGB(0x8, 1); PFHIP_SGB(0x2, 8); PFHIP_SGB(0x200, 1);                                  \
  PFHIP_SGB(0x8, 1); PFHIP_SGB(0x2, 8); PFHIP_SGB(0x200, 1);                                  \
  PFHIP_SGB(0x8, 1); PFHIP_SGB(0x20, 1);                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                          \
  if (parked && kt >= 2 && ((kt - 2) % dstride) == 0 && (kt - 2) / dstride < 4) drain_pass((kt - 2) / dstride); \
  __builtin_amdgcn_sched_barrier(0);                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                              \
  __builtin_amdgcn_sched_barrier(0);                                                          \
  PFHIP_FRAGS(GA, GB, nxt)                                                                    \
  PFHIP_MM(FA, FB, 1, 0) PFHIP_MM(FA, FB, 0, 0)                                               \
  PFHIP_SGB(0x8, 1); PFHIP_SGB(0x100, 3); PFHIP_SGB(0x8, 1); PFHIP_SGB(0x100, 3);             \
  __builtin_amdgcn_sched_barrier(0);                                                          \
  if (++kt == nk) { park(); kt = 0; ++cur_ti; }

  // pipeline fill: step 0 split into stage 0, raw of steps 1 and 2 in flight (y, x)
  PFHIP_LOAD_NEXT(xr)
  split3(xr, lds + st_off);
  PFHIP_LOAD_NEXT(yr)
  PFHIP_LOAD_NEXT(xr)
  __syncthreads();
  PFHIP_FRAGS(fa, fb, 0)

  const int total = my_tiles * nk;          // nk is even (K is a multiple of 32): tile boundaries fall on even steps
  for (int g = 0; g < total; g += 2) {
    PFHIP_STEP(fa, fb, ga, gb, yr, 1)
    PFHIP_STEP(ga, gb, fa, fb, xr, 0)
  }
#undef PFHIP_STEP
#undef PFHIP_SGB
#undef PFHIP_MM
#undef PFHIP_FRAGS
#undef PFHIP_LOAD_NEXT
  // the last tile was parked by the last step: write it out
  __syncthreads();
#pragma unroll 1
  for (int p = 0; p < 4; ++p) drain_pass(p);
}

}  // namespace

void launch_gemm_f32_bf16x6_persistent(const float* A, int lda, const float* W, int ldw, float* C, int ldc, const float* bias,
                                       const float* R1, int ldr1, const float* R2, int ldr2, int M, int N, int K, bool relu, int gw,
                                       hipStream_t s) {
  if (M <= 0 || N <= 0) return;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_done.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_bf16x6_persistent_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_done.fetch_or(1ull << (dev & 63));
  }
  const int tiles_m = (M + kPM - 1) / kPM, tiles_n = (N + kPN - 1) / kPN, n_tiles = tiles_m * tiles_n;
  gw = std::max(1, std::min(gw, tiles_n));
  const int grid = std::min(n_tiles, 256);
  hipLaunchKernelGGL(gemm_f32_bf16x6_persistent_kernel, dim3(grid), dim3(1024), kLdsBytes, s, A, lda, W, ldw, C, ldc, bias, R1, ldr1,
                     R2, ldr2, M, N, K, tiles_n, n_tiles, gw, relu ? 1 : 0);
}

}  // namespace pfhip
