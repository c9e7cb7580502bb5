// Dev probe (GPU): main-loop instruction interleaving for the fp32 MFMA GEMM.
//   V=0  production order (memory instructions in bursts between groups of 16 MFMAs)
//   V=1  memory instructions spread one at a time between single MFMAs
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off gemm_il.hip -o gemm_il && ./gemm_il
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kLds = 36, kStage = 128 * kLds, kCs = 132;
#define SB __builtin_amdgcn_sched_barrier(0)

template <int V>
__global__ __launch_bounds__(256, 2) void gemm_il(const float* __restrict__ A, const float* __restrict__ W, float* C,
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
#define GL(reg, base, ld, j, k0) reg = *(const float4*)(base + (size_t)(32 * (j)) * ld + (k0))
#define GLOAD(k0) do { GL(ra0, Ag, lda, 0, k0); GL(ra1, Ag, lda, 1, k0); GL(ra2, Ag, lda, 2, k0); GL(ra3, Ag, lda, 3, k0); \
  GL(rb0, Wg, ldw, 0, k0); GL(rb1, Wg, ldw, 1, k0); GL(rb2, Wg, ldw, 2, k0); GL(rb3, Wg, ldw, 3, k0); } while (0)
#define SW(reg, base, buf, j) *(float4*)(base + (buf) * kStage + lrow * kLds + 4 * lc4 + 32 * (j) * kLds) = reg
#define SSTORE(buf) do { SW(ra0, As, buf, 0); SW(ra1, As, buf, 1); SW(ra2, As, buf, 2); SW(ra3, As, buf, 3); \
  SW(rb0, Bs, buf, 0); SW(rb1, Bs, buf, 1); SW(rb2, Bs, buf, 2); SW(rb3, Bs, buf, 3); } while (0)
  f32x16 acc00, acc01, acc10, acc11;
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  GLOAD(0);
  SSTORE(0);
  __syncthreads();
  const int a_off = (wr * 64 + r) * kLds + 4 * h, b_off = (wc * 64 + r) * kLds + 4 * h;
  float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define FR(reg, base, off, buf, kb, j) reg = *(const float4*)(base + (buf) * kStage + off + (kb) * 8 + 32 * (j) * kLds)
#define FRAG(A0, A1, B0, B1, buf, kb) do { FR(A0, As, a_off, buf, kb, 0); FR(A1, As, a_off, buf, kb, 1); FR(B0, Bs, b_off, buf, kb, 0); FR(B1, Bs, b_off, buf, kb, 1); } while (0)
#define MM(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
#define M4(A0, A1, B0, B1, c) MM(acc00, A0.c, B0.c); MM(acc01, A0.c, B1.c); MM(acc10, A1.c, B0.c); MM(acc11, A1.c, B1.c);
#define M16(A0, A1, B0, B1) M4(A0, A1, B0, B1, x) M4(A0, A1, B0, B1, y) M4(A0, A1, B0, B1, z) M4(A0, A1, B0, B1, w)
  FRAG(fa0, fa1, fb0, fb1, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const int knext = (kt + 1 < nk ? kt + 1 : kt) * 32;
    if (V == 0) {
      GLOAD(knext); SB;
      FRAG(ga0, ga1, gb0, gb1, cur, 1); SB;
      M16(fa0, fa1, fb0, fb1) SB;
      FRAG(fa0, fa1, fb0, fb1, cur, 2); SB;
      M16(ga0, ga1, gb0, gb1) SB;
      FRAG(ga0, ga1, gb0, gb1, cur, 3); SB;
      SSTORE(cur ^ 1); SB;
      M16(fa0, fa1, fb0, fb1) SB;
      __syncthreads();
      FRAG(fa0, fa1, fb0, fb1, cur ^ 1, 0); SB;
      M16(ga0, ga1, gb0, gb1) SB;
    } else {
      // group 0: f (k-block 0) | 8 global loads of the next tile, then the 4 fragment reads of k-block 1
      MM(acc00, fa0.x, fb0.x); SB; GL(ra0, Ag, lda, 0, knext); SB;
      MM(acc01, fa0.x, fb1.x); SB; GL(ra1, Ag, lda, 1, knext); SB;
      MM(acc10, fa1.x, fb0.x); SB; GL(ra2, Ag, lda, 2, knext); SB;
      MM(acc11, fa1.x, fb1.x); SB; GL(ra3, Ag, lda, 3, knext); SB;
      MM(acc00, fa0.y, fb0.y); SB; GL(rb0, Wg, ldw, 0, knext); SB;
      MM(acc01, fa0.y, fb1.y); SB; GL(rb1, Wg, ldw, 1, knext); SB;
      MM(acc10, fa1.y, fb0.y); SB; GL(rb2, Wg, ldw, 2, knext); SB;
      MM(acc11, fa1.y, fb1.y); SB; GL(rb3, Wg, ldw, 3, knext); SB;
      MM(acc00, fa0.z, fb0.z); SB; FR(ga0, As, a_off, cur, 1, 0); SB;
      MM(acc01, fa0.z, fb1.z); SB; FR(ga1, As, a_off, cur, 1, 1); SB;
      MM(acc10, fa1.z, fb0.z); SB; FR(gb0, Bs, b_off, cur, 1, 0); SB;
      MM(acc11, fa1.z, fb1.z); SB; FR(gb1, Bs, b_off, cur, 1, 1); SB;
      M4(fa0, fa1, fb0, fb1, w) SB;
      // group 1: g (k-block 1) | fragment reads of k-block 2 into f
      MM(acc00, ga0.x, gb0.x); SB; FR(fa0, As, a_off, cur, 2, 0); SB;
      MM(acc01, ga0.x, gb1.x); SB; FR(fa1, As, a_off, cur, 2, 1); SB;
      MM(acc10, ga1.x, gb0.x); SB; FR(fb0, Bs, b_off, cur, 2, 0); SB;
      MM(acc11, ga1.x, gb1.x); SB; FR(fb1, Bs, b_off, cur, 2, 1); SB;
      M4(ga0, ga1, gb0, gb1, y) M4(ga0, ga1, gb0, gb1, z) M4(ga0, ga1, gb0, gb1, w) SB;
      // group 2: f (k-block 2) | fragment reads of k-block 3 into g, then the 8 LDS writes of the next tile
      MM(acc00, fa0.x, fb0.x); SB; FR(ga0, As, a_off, cur, 3, 0); SB;
      MM(acc01, fa0.x, fb1.x); SB; FR(ga1, As, a_off, cur, 3, 1); SB;
      MM(acc10, fa1.x, fb0.x); SB; FR(gb0, Bs, b_off, cur, 3, 0); SB;
      MM(acc11, fa1.x, fb1.x); SB; FR(gb1, Bs, b_off, cur, 3, 1); SB;
      MM(acc00, fa0.y, fb0.y); SB; SW(ra0, As, cur ^ 1, 0); SB;
      MM(acc01, fa0.y, fb1.y); SB; SW(ra1, As, cur ^ 1, 1); SB;
      MM(acc10, fa1.y, fb0.y); SB; SW(ra2, As, cur ^ 1, 2); SB;
      MM(acc11, fa1.y, fb1.y); SB; SW(ra3, As, cur ^ 1, 3); SB;
      MM(acc00, fa0.z, fb0.z); SB; SW(rb0, Bs, cur ^ 1, 0); SB;
      MM(acc01, fa0.z, fb1.z); SB; SW(rb1, Bs, cur ^ 1, 1); SB;
      MM(acc10, fa1.z, fb0.z); SB; SW(rb2, Bs, cur ^ 1, 2); SB;
      MM(acc11, fa1.z, fb1.z); SB; SW(rb3, Bs, cur ^ 1, 3); SB;
      M4(fa0, fa1, fb0, fb1, w) SB;
      __syncthreads();
      // group 3: g (k-block 3) | fragment reads of k-block 0 of the next tile into f
      MM(acc00, ga0.x, gb0.x); SB; FR(fa0, As, a_off, cur ^ 1, 0, 0); SB;
      MM(acc01, ga0.x, gb1.x); SB; FR(fa1, As, a_off, cur ^ 1, 0, 1); SB;
      MM(acc10, ga1.x, gb0.x); SB; FR(fb0, Bs, b_off, cur ^ 1, 0, 0); SB;
      MM(acc11, ga1.x, gb1.x); SB; FR(fb1, Bs, b_off, cur ^ 1, 0, 1); SB;
      M4(ga0, ga1, gb0, gb1, y) M4(ga0, ga1, gb0, gb1, z) M4(ga0, ga1, gb0, gb1, w) SB;
    }
  }
  __syncthreads();
  float* const Cs = lds;
  { float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 64 + r;
    for (int e = 0; e < 16; ++e) { const int ro = ((e & 3) + 8 * (e >> 2)) * kCs; cw[ro] = acc00[e]; cw[ro + 32] = acc01[e]; cw[ro + 32 * kCs] = acc10[e]; cw[ro + 32 * kCs + 32] = acc11[e]; } }
  __syncthreads();
  const int c4 = tid & 31, rsub = tid >> 5;
#pragma unroll 4
  for (int pass = 0; pass < 16; ++pass) { const int row = pass * 8 + rsub;
    *(float4*)(C + (size_t)(m0 + row) * ldc + n0 + 4 * c4) = *(const float4*)(Cs + row * kCs + 4 * c4); }
}

template <int V> float run(const float* A, const float* W, float* C, int M, int N, int K, int iters) {
  const int tiles_n = N / 128, blocks = (M / 128) * tiles_n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_il<V>, dim3(blocks), dim3(256), 0, 0, A, W, C, K, K, N, K / 32, tiles_n, blocks);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_il<V>, dim3(blocks), dim3(256), 0, 0, A, W, C, K, K, N, K / 32, tiles_n, blocks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int shapes[4][3] = {{16000, 2048, 512}, {16000, 512, 2048}, {16000, 1536, 512}, {16000, 512, 512}};
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
    run<0>(A, W, C, M, N, K, 1); hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
    hipMemset(C, 0, c0.size() * 4);
    run<1>(A, W, C, M, N, K, 1); hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
    double md = 0; for (size_t i = 0; i < c0.size(); ++i) md = fmax(md, fabs((double)c0[i] - c1[i]));
    for (int rep = 0; rep < 2; ++rep) {
      const float t0 = run<0>(A, W, C, M, N, K, 20), t1 = run<1>(A, W, C, M, N, K, 20);
      printf("%5dx%4dx%4d  burst %7.1f us %6.1f TF | interleaved %7.1f us %6.1f TF | max diff %g\n", M, N, K, t0 * 1e3, fl / t0 / 1e9,
             t1 * 1e3, fl / t1 / 1e9, md);
    }
    hipFree(A); hipFree(W); hipFree(C);
  }
  return 0;
}
