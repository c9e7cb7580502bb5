// Dev probe (GPU): compares generated K-loop instruction schedules (gen_gemm_loop.py -> gemm_loop_gen.h).
//   python3 gen_gemm_loop.py > gemm_loop_gen.h && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off gemm_sched.hip -o gemm_sched
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "gemm_loop_gen.h"
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kLds = 36, kStage = 128 * kLds, kCs = 132;
#define SB __builtin_amdgcn_sched_barrier(0)
#define GL(reg, base, ld, j, k0) reg = *(const float4*)(base + (size_t)(32 * (j)) * ld + (k0))
#define SW(reg, base, buf, j) *(float4*)(base + (buf) * kStage + lrow * kLds + 4 * lc4 + 32 * (j) * kLds) = reg
#define FR(reg, base, off, buf, kb, j) reg = *(const float4*)(base + (buf) * kStage + off + (kb) * 8 + 32 * (j) * kLds)
#define MM(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)

#define KERNEL(NAME)                                                                                                          \
  __global__ __launch_bounds__(256, 2) void gemm_##NAME(const float* __restrict__ A, const float* __restrict__ W, float* C,   \
                                                        int lda, int ldw, int ldc, int nk, int tiles_n, int n_tiles) {        \
    __shared__ __attribute__((aligned(16))) float lds[4 * kStage];                                                            \
    float* const As = lds; float* const Bs = lds + 2 * kStage;                                                                \
    int bid = blockIdx.x;                                                                                                     \
    { const int q = n_tiles >> 3, rr = n_tiles & 7, xcd = bid & 7;                                                            \
      bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3); }                                        \
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;                                                                    \
    const int m0 = tm * 128, n0 = tn * 128;                                                                                   \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                                            \
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;                                                    \
    const int lrow = tid >> 3, lc4 = tid & 7;                                                                                 \
    const float* Ag = A + (size_t)(m0 + lrow) * lda + 4 * lc4;                                                                \
    const float* Wg = W + (size_t)(n0 + lrow) * ldw + 4 * lc4;                                                                \
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;                                                                            \
    f32x16 acc00, acc01, acc10, acc11;                                                                                        \
    for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }                          \
    { const int knext = 0; const int cur = 1;                                                                                 \
      GL(ra0, Ag, lda, 0, knext); GL(ra1, Ag, lda, 1, knext); GL(ra2, Ag, lda, 2, knext); GL(ra3, Ag, lda, 3, knext);         \
      GL(rb0, Wg, ldw, 0, knext); GL(rb1, Wg, ldw, 1, knext); GL(rb2, Wg, ldw, 2, knext); GL(rb3, Wg, ldw, 3, knext);         \
      SW(ra0, As, cur ^ 1, 0); SW(ra1, As, cur ^ 1, 1); SW(ra2, As, cur ^ 1, 2); SW(ra3, As, cur ^ 1, 3);                     \
      SW(rb0, Bs, cur ^ 1, 0); SW(rb1, Bs, cur ^ 1, 1); SW(rb2, Bs, cur ^ 1, 2); SW(rb3, Bs, cur ^ 1, 3); }                   \
    __syncthreads();                                                                                                          \
    const int a_off = (wr * 64 + r) * kLds + 4 * h, b_off = (wc * 64 + r) * kLds + 4 * h;                                     \
    float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;                                                                            \
    FR(fa0, As, a_off, 0, 0, 0); FR(fa1, As, a_off, 0, 0, 1); FR(fb0, Bs, b_off, 0, 0, 0); FR(fb1, Bs, b_off, 0, 0, 1);       \
    for (int kt = 0; kt < nk; ++kt) {                                                                                         \
      const int cur = kt & 1;                                                                                                 \
      const int knext = (kt + 1 < nk ? kt + 1 : kt) * 32;                                                                     \
      LOOP_BODY_##NAME                                                                                                        \
    }                                                                                                                         \
    __syncthreads();                                                                                                          \
    float* const Cs = lds;                                                                                                    \
    { float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 64 + r;                                                                 \
      for (int e = 0; e < 16; ++e) { const int ro = ((e & 3) + 8 * (e >> 2)) * kCs; cw[ro] = acc00[e]; cw[ro + 32] = acc01[e]; \
        cw[ro + 32 * kCs] = acc10[e]; cw[ro + 32 * kCs + 32] = acc11[e]; } }                                                  \
    __syncthreads();                                                                                                          \
    const int c4 = tid & 31, rsub = tid >> 5;                                                                                 \
    _Pragma("unroll 4") for (int pass = 0; pass < 16; ++pass) { const int row = pass * 8 + rsub;                              \
      *(float4*)(C + (size_t)(m0 + row) * ldc + n0 + 4 * c4) = *(const float4*)(Cs + row * kCs + 4 * c4); }                   \
  }

KERNEL(A) KERNEL(B) KERNEL(C) KERNEL(D) KERNEL(E)
typedef void (*kern_t)(const float*, const float*, float*, int, int, int, int, int, int);
static kern_t kernels[] = {gemm_A, gemm_B, gemm_C, gemm_D, gemm_E};
static const char* names[] = {"A", "B", "C", "D", "E"};

float run(kern_t k, const float* A, const float* W, float* C, int M, int N, int K, int iters) {
  const int tiles_n = N / 128, blocks = (M / 128) * tiles_n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, A, W, C, K, K, N, K / 32, tiles_n, blocks);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, A, W, C, K, K, N, K / 32, tiles_n, blocks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int shapes[4][3] = {{16000, 2048, 512}, {16000, 512, 2048}, {16000, 1536, 512}, {16000, 512, 512}};
  const int nv = sizeof(kernels) / sizeof(kernels[0]);
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    float *A, *W, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    const double fl = 2.0 * M * N * K;
    std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
    printf("%5dx%4dx%4d:", M, N, K);
    for (int v = 0; v < nv; ++v) {
      hipMemset(C, 0, c0.size() * 4);
      run(kernels[v], A, W, C, M, N, K, 1);
      hipMemcpy(v == 0 ? c0.data() : c1.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
      double md = 0; if (v) for (size_t i = 0; i < c0.size(); ++i) md = fmax(md, fabs((double)c0[i] - c1[i]));
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) best = fminf(best, run(kernels[v], A, W, C, M, N, K, 20));
      printf("  %s %6.1f us %5.1f TF%s", names[v], best * 1e3, fl / best / 1e9, md == 0 ? "" : " MISMATCH");
    }
    printf("\n");
    hipFree(A); hipFree(W); hipFree(C);
  }
  return 0;
}
