// Dev probe (GPU): persistent one-block-per-CU fp32 MFMA GEMM ("solo"): continuous distance-2 operand prefetch across
// output tiles, C tile staged in its own LDS region and written out one 8-row pass per K-tile of the NEXT tile.
//   python3 gen_gemm_loop.py > gemm_loop_gen.h && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off gemm_solo.hip -o gemm_solo
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_loop_gen.h"
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kLds = 36, kStage = 128 * kLds, kCs = 132;
constexpr int kSoloLdsBytes = (4 * kStage + 128 * kCs) * 4;
#define SB __builtin_amdgcn_sched_barrier(0)
#define GL(reg, base, ld, j, k0) reg = *(const float4*)(base + (size_t)(32 * (j)) * ld + base##_off)
#define SW(reg, base, buf, j) *(float4*)(base + (buf) * kStage + lrow * kLds + 4 * lc4 + 32 * (j) * kLds) = reg
#define FR(reg, base, off, buf, kb, j) reg = *(const float4*)(base + (buf) * kStage + off + (kb) * 8 + 32 * (j) * kLds)
#define MM(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

template <bool HAS_R>
__global__ __launch_bounds__(256, 1) void gemm_solo(const float* __restrict__ A, const float* __restrict__ W, float* C, const float* R,
                                                    int lda, int ldw, int ldc, int nk, int tiles_n, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const As = lds; float* const Bs = lds + 2 * kStage; float* const Cs = lds + 4 * kStage;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int lrow = tid >> 3, lc4 = tid & 7;
  const int c4 = tid & 31, rsub = tid >> 5;
  const int G = gridDim.x;
  const int q8 = n_tiles >> 3, r8 = n_tiles & 7;
  auto tile_of = [&](int t, int& m0, int& n0) {
    const int xcd = t & 7;
    const int b2 = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (t >> 3);
    const int tm = b2 / tiles_n; m0 = tm * 128; n0 = (b2 - tm * tiles_n) * 128;
  };
  const float* Ag = A + (size_t)lrow * lda + 4 * lc4;
  const float* Wg = W + (size_t)lrow * ldw + 4 * lc4;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
  float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
  const int a_off = (wr * 64 + r) * kLds + 4 * h, b_off = (wc * 64 + r) * kLds + 4 * h;

  // load cursor: the K-tile stream of this block runs through all of its output tiles without a break
  int lt = blockIdx.x, lk = 0, lm0, ln0;
  tile_of(lt, lm0, ln0);
  size_t Ag_off = (size_t)lm0 * lda, Wg_off = (size_t)ln0 * ldw;
  auto advance = [&]() {
    if (++lk == nk) {
      if (lt + G < n_tiles) { lt += G; lk = 0; tile_of(lt, lm0, ln0); } else lk = nk - 1;   // past the end: re-fetch, never used
    }
    Ag_off = (size_t)lm0 * lda + (size_t)lk * 32;
    Wg_off = (size_t)ln0 * ldw + (size_t)lk * 32;
  };
  const int knext = 0; (void)knext;
  GL(ra0, Ag, lda, 0, 0); GL(ra1, Ag, lda, 1, 0); GL(ra2, Ag, lda, 2, 0); GL(ra3, Ag, lda, 3, 0);
  GL(rb0, Wg, ldw, 0, 0); GL(rb1, Wg, ldw, 1, 0); GL(rb2, Wg, ldw, 2, 0); GL(rb3, Wg, ldw, 3, 0);
  SW(ra0, As, 0, 0); SW(ra1, As, 0, 1); SW(ra2, As, 0, 2); SW(ra3, As, 0, 3);
  SW(rb0, Bs, 0, 0); SW(rb1, Bs, 0, 1); SW(rb2, Bs, 0, 2); SW(rb3, Bs, 0, 3);
  advance();
  GL(sa0, Ag, lda, 0, 0); GL(sa1, Ag, lda, 1, 0); GL(sa2, Ag, lda, 2, 0); GL(sa3, Ag, lda, 3, 0);
  GL(sb0, Wg, ldw, 0, 0); GL(sb1, Wg, ldw, 1, 0); GL(sb2, Wg, ldw, 2, 0); GL(sb3, Wg, ldw, 3, 0);
  __syncthreads();
  FR(fa0, As, a_off, 0, 0, 0); FR(fa1, As, a_off, 0, 0, 1); FR(fb0, Bs, b_off, 0, 0, 0); FR(fb1, Bs, b_off, 0, 0, 1);

  f32x16 acc00, acc01, acc10, acc11;
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  bool have_prev = false;
  int pm0 = 0, pn0 = 0;
  float4 rres = make_float4(0.f, 0.f, 0.f, 0.f), cv = rres;
  int pass = 0;
#define EP0() do { if (HAS_R && have_prev && pass < 16) rres = *(const float4*)(R + (size_t)(pm0 + pass * 8 + rsub) * ldc + pn0 + 4 * c4); } while (0)
#define EP1() do { if (have_prev && pass < 16) cv = *(const float4*)(Cs + (pass * 8 + rsub) * kCs + 4 * c4); } while (0)
#define EP2() do { if (have_prev && pass < 16) { if (HAS_R) { cv.x += rres.x; cv.y += rres.y; cv.z += rres.z; cv.w += rres.w; } \
    *(float4*)(C + (size_t)(pm0 + pass * 8 + rsub) * ldc + pn0 + 4 * c4) = cv; } } while (0)
  for (int t = blockIdx.x;; t += G) {
    int m0, n0;
    tile_of(t, m0, n0);
    for (int kt = 0; kt < nk; kt += 2) {
      advance(); pass = kt;
      LOOP_BODY_S_0
      advance(); pass = kt + 1;
      LOOP_BODY_S_1
    }
    if (nk < 16) {
      if (have_prev) for (pass = nk; pass < 16; ++pass) { EP0(); EP1(); EP2(); }
      __syncthreads();
    }
    { float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 64 + r;
      for (int e = 0; e < 16; ++e) { const int ro = ((e & 3) + 8 * (e >> 2)) * kCs; cw[ro] = acc00[e]; cw[ro + 32] = acc01[e];
        cw[ro + 32 * kCs] = acc10[e]; cw[ro + 32 * kCs + 32] = acc11[e]; } }
    for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
    __syncthreads();
    pm0 = m0; pn0 = n0; have_prev = true;
    if (t + G >= n_tiles) break;
  }
  for (pass = 0; pass < 16; ++pass) { EP0(); EP1(); EP2(); }
}

// reference: the non-persistent schedule A kernel, two blocks per CU
__global__ __launch_bounds__(256, 2) void gemm_ref(const float* __restrict__ A, const float* __restrict__ W, float* C, const float* R,
                                                   int lda, int ldw, int ldc, int nk, int tiles_n, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kStage];
  float* const As = lds; float* const Bs = lds + 2 * kStage;
  int bid = blockIdx.x;
  { const int q = n_tiles >> 3, rr = n_tiles & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3); }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int lrow = tid >> 3, lc4 = tid & 7;
  const float* Ag = A + (size_t)(m0 + lrow) * lda + 4 * lc4;
  const float* Wg = W + (size_t)(n0 + lrow) * ldw + 4 * lc4;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  f32x16 acc00, acc01, acc10, acc11;
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  size_t Ag_off = 0, Wg_off = 0;
  const int knext = 0; (void)knext;
  GL(ra0, Ag, lda, 0, 0); GL(ra1, Ag, lda, 1, 0); GL(ra2, Ag, lda, 2, 0); GL(ra3, Ag, lda, 3, 0);
  GL(rb0, Wg, ldw, 0, 0); GL(rb1, Wg, ldw, 1, 0); GL(rb2, Wg, ldw, 2, 0); GL(rb3, Wg, ldw, 3, 0);
  SW(ra0, As, 0, 0); SW(ra1, As, 0, 1); SW(ra2, As, 0, 2); SW(ra3, As, 0, 3);
  SW(rb0, Bs, 0, 0); SW(rb1, Bs, 0, 1); SW(rb2, Bs, 0, 2); SW(rb3, Bs, 0, 3);
  __syncthreads();
  const int a_off = (wr * 64 + r) * kLds + 4 * h, b_off = (wc * 64 + r) * kLds + 4 * h;
  float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
  FR(fa0, As, a_off, 0, 0, 0); FR(fa1, As, a_off, 0, 0, 1); FR(fb0, Bs, b_off, 0, 0, 0); FR(fb1, Bs, b_off, 0, 0, 1);
  for (int kt = 0; kt < nk; kt += 2) {
    Ag_off = Wg_off = (size_t)(kt + 1 < nk ? kt + 1 : nk - 1) * 32;
    LOOP_BODY_A_0
    Ag_off = Wg_off = (size_t)(kt + 2 < nk ? kt + 2 : nk - 1) * 32;
    LOOP_BODY_A_1
  }
  __syncthreads();
  float* const Cs = lds;
  { float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 64 + r;
    for (int e = 0; e < 16; ++e) { const int ro = ((e & 3) + 8 * (e >> 2)) * kCs; cw[ro] = acc00[e]; cw[ro + 32] = acc01[e];
      cw[ro + 32 * kCs] = acc10[e]; cw[ro + 32 * kCs + 32] = acc11[e]; } }
  __syncthreads();
  const int c4 = tid & 31, rsub = tid >> 5;
#pragma unroll 4
  for (int pass = 0; pass < 16; ++pass) { const int row = pass * 8 + rsub;
    float4 v = *(const float4*)(Cs + row * kCs + 4 * c4);
    if (R) { const float4 t = *(const float4*)(R + (size_t)(m0 + row) * ldc + n0 + 4 * c4); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    *(float4*)(C + (size_t)(m0 + row) * ldc + n0 + 4 * c4) = v; }
}

typedef void (*kern_t)(const float*, const float*, float*, const float*, int, int, int, int, int, int);
float run(kern_t k, bool solo, const float* A, const float* W, float* C, const float* R, int M, int N, int K, int iters) {
  const int tiles_n = N / 128, blocks = (M / 128) * tiles_n;
  const int grid = solo ? (blocks < 256 ? blocks : 256) : blocks;
  const size_t dyn = solo ? kSoloLdsBytes : 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), dyn, 0, A, W, C, R, K, K, N, K / 32, tiles_n, blocks);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), dyn, 0, A, W, C, R, K, K, N, K / 32, tiles_n, blocks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main(int argc, char** argv) {
  if (argc > 1) {   // sustained mode: loop one shape for argv[1] seconds, print TF once per second (clock / power watch)
    hipFuncSetAttribute((const void*)gemm_solo<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSoloLdsBytes);
    const int M = 16000, N = 2048, K = 512; const bool solo = argc > 2;
    float *A, *W, *C; hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0x3c, (size_t)M * K * 4); hipMemset(W, 0x3c, (size_t)N * K * 4);
    const double secs = atof(argv[1]); double total = 0;
    while (total < secs) { const float ms = run(solo ? gemm_solo<false> : gemm_ref, solo, A, W, C, nullptr, M, N, K, 2000);
      total += ms * 2000 / 1e3; printf("%s %.1f s: %.1f us %.1f TF\n", solo ? "solo" : "ref", total, ms * 1e3, 2.0 * M * N * K / ms / 1e9); fflush(stdout); }
    return 0;
  }
  hipFuncSetAttribute((const void*)gemm_solo<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSoloLdsBytes);
  hipFuncSetAttribute((const void*)gemm_solo<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSoloLdsBytes);
  const int shapes[][4] = {{16000, 2048, 512, 0}, {16000, 512, 2048, 1}, {16000, 1536, 512, 0}, {16000, 512, 512, 1}, {7040, 2048, 512, 0}, {1024, 512, 512, 1}, {16000, 512, 256, 0}};
  for (auto& s : shapes) {
    const int M = s[0] / 128 * 128, N = s[1], K = s[2]; const bool res = s[3];
    float *A, *W, *C, *R;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&R, (size_t)M * N * 4);
    std::vector<float> h((size_t)(M > N ? M : N) * K), hr((size_t)M * N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    for (size_t i = 0; i < hr.size(); ++i) hr[i] = (float)((i * 40503u) % 777) / 777.f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(R, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    const double fl = 2.0 * M * N * K;
    std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
    hipMemset(C, 0, c0.size() * 4);
    run(gemm_ref, false, A, W, C, res ? R : nullptr, M, N, K, 1); hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
    hipMemset(C, 0, c0.size() * 4);
    run(res ? gemm_solo<true> : gemm_solo<false>, true, A, W, C, R, M, N, K, 1); hipMemcpy(c1.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
    double md = 0; for (size_t i = 0; i < c0.size(); ++i) md = fmax(md, fabs((double)c0[i] - c1[i]));
    float b0 = 1e9f, b1 = 1e9f;
    for (int rep = 0; rep < 3; ++rep) { b0 = fminf(b0, run(gemm_ref, false, A, W, C, res ? R : nullptr, M, N, K, 20));
      b1 = fminf(b1, run(res ? gemm_solo<true> : gemm_solo<false>, true, A, W, C, R, M, N, K, 20)); }
    printf("%5dx%4dx%4d%s  ref %6.1f us %5.1f TF | solo %6.1f us %5.1f TF | max diff %g\n", M, N, K, res ? "+R" : "  ", b0 * 1e3, fl / b0 / 1e9, b1 * 1e3, fl / b1 / 1e9, md);
    hipFree(A); hipFree(W); hipFree(C); hipFree(R);
  }
  return 0;
}
