// Dev probe (GPU): where does the fp32 MFMA GEMM lose its matrix-pipe time?  Builds the main loop up
// stage by stage.  hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kLds = 36, kStage = 128 * kLds;

template <int V>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ A, const float* __restrict__ W, float* C, const float* R,
                                                int lda, int ldw, int ldc, int nk, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kStage];
  float* const As = lds; float* const Bs = lds + 2 * kStage;
  int bid = blockIdx.x;
  if (V >= 5) { const int n_tiles = gridDim.x; const int q = n_tiles >> 3, rr = n_tiles & 7, xcd = bid & 7;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3); }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int lrow = tid >> 3, lc4 = tid & 7;
  const float* Ag = A + (size_t)(m0 + lrow) * lda + 4 * lc4;
  const float* Wg = W + (size_t)(n0 + lrow) * ldw + 4 * lc4;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(1.f, 2.f, 3.f, 4.f);
#define GLOAD(k0) do { ra0 = *(const float4*)(Ag + (k0)); ra1 = *(const float4*)(Ag + (size_t)32 * lda + (k0)); \
  ra2 = *(const float4*)(Ag + (size_t)64 * lda + (k0)); ra3 = *(const float4*)(Ag + (size_t)96 * lda + (k0)); \
  rb0 = *(const float4*)(Wg + (k0)); rb1 = *(const float4*)(Wg + (size_t)32 * ldw + (k0)); \
  rb2 = *(const float4*)(Wg + (size_t)64 * ldw + (k0)); rb3 = *(const float4*)(Wg + (size_t)96 * ldw + (k0)); } while (0)
#define SSTORE(buf) do { float* as_ = As + (buf) * kStage + lrow * kLds + 4 * lc4; float* bs_ = Bs + (buf) * kStage + lrow * kLds + 4 * lc4; \
  *(float4*)(as_) = ra0; *(float4*)(as_ + 32 * kLds) = ra1; *(float4*)(as_ + 64 * kLds) = ra2; *(float4*)(as_ + 96 * kLds) = ra3; \
  *(float4*)(bs_) = rb0; *(float4*)(bs_ + 32 * kLds) = rb1; *(float4*)(bs_ + 64 * kLds) = rb2; *(float4*)(bs_ + 96 * kLds) = rb3; } while (0)
  f32x16 acc00, acc01, acc10, acc11;
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  if (V >= 3) GLOAD(0);
  SSTORE(0);
  __syncthreads();
  const int a_off = (wr * 64 + r) * kLds + 4 * h, b_off = (wc * 64 + r) * kLds + 4 * h;
  float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define FRAG(A0, A1, B0, B1, buf, kb) do { const float* as_ = As + (buf) * kStage + a_off + (kb) * 8; const float* bs_ = Bs + (buf) * kStage + b_off + (kb) * 8; \
  A0 = *(const float4*)(as_); A1 = *(const float4*)(as_ + 32 * kLds); B0 = *(const float4*)(bs_); B1 = *(const float4*)(bs_ + 32 * kLds); } while (0)
#define M4(A0, A1, B0, B1, c) acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B0.c, acc00, 0, 0, 0); acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B1.c, acc01, 0, 0, 0); \
  acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B0.c, acc10, 0, 0, 0); acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B1.c, acc11, 0, 0, 0);
#define M16(A0, A1, B0, B1) M4(A0, A1, B0, B1, x) M4(A0, A1, B0, B1, y) M4(A0, A1, B0, B1, z) M4(A0, A1, B0, B1, w)
#define SB __builtin_amdgcn_sched_barrier(0)
  if ((V == 38 || V == 39 || V == 37) && ((blockIdx.x >> 8) & 1)) { const int reps = V == 37 ? 2 : (V == 38 ? 4 : 8); for (int q = 0; q < reps; ++q) __builtin_amdgcn_s_sleep(127); }
  if (V >= 33 && V < 37 && ((blockIdx.x >> 8) & 1)) { if (V == 33) __builtin_amdgcn_s_sleep(32); if (V == 34) __builtin_amdgcn_s_sleep(64); if (V == 35) { __builtin_amdgcn_s_sleep(64); __builtin_amdgcn_s_sleep(64);} }
  FRAG(fa0, fa1, fb0, fb1, 0, 0);
  ga0 = fa0; ga1 = fa1; gb0 = fb0; gb1 = fb1;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const int knext = (kt + 1 < nk ? kt + 1 : kt) * 32;
    if ((V >= 3 && V != 31)) { GLOAD(knext); SB; }
    if (V >= 1) { FRAG(ga0, ga1, gb0, gb1, cur, 1); SB; }
    M16(fa0, fa1, fb0, fb1) SB;
    if (V >= 1) { FRAG(fa0, fa1, fb0, fb1, cur, 2); SB; }
    M16(ga0, ga1, gb0, gb1) SB;
    if (V >= 1) { FRAG(ga0, ga1, gb0, gb1, cur, 3); SB; }
    if (V >= 32 && V < 40) { SSTORE(cur ^ 1); SB; }
    M16(fa0, fa1, fb0, fb1) SB;
    if (V >= 3 && V != 30 && !(V >= 32 && V < 40)) SSTORE(cur ^ 1);
    if (V == 30) { asm volatile("" :: "v"(ra0.x), "v"(ra1.x), "v"(ra2.x), "v"(ra3.x), "v"(rb0.x), "v"(rb1.x), "v"(rb2.x), "v"(rb3.x)); }
    if (V >= 2) __syncthreads();
    if (V >= 1) { FRAG(fa0, fa1, fb0, fb1, cur ^ 1, 0); SB; }
    M16(ga0, ga1, gb0, gb1) SB;
  }
  // minimal epilogue so nothing is dead: one float per lane
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc00[e] + acc01[e] + acc10[e] + acc11[e];
  if (V < 4 || (V >= 30 && V < 36)) { C[(size_t)(m0 + wr * 64 + r) * ldc + n0 + wc * 64 + h] = s; return; }
  __syncthreads();
  float* const Cs = lds; constexpr int kCs = 132;
  { float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 64 + r;
    for (int e = 0; e < 16; ++e) { const int ro = ((e & 3) + 8 * (e >> 2)) * kCs; cw[ro] = acc00[e]; cw[ro + 32] = acc01[e]; cw[ro + 32 * kCs] = acc10[e]; cw[ro + 32 * kCs + 32] = acc11[e]; } }
  __syncthreads();
  const int c4 = tid & 31, rsub = tid >> 5;
  for (int pass = 0; pass < 16; ++pass) { const int row = pass * 8 + rsub;
    float4 v = *(const float4*)(Cs + row * kCs + 4 * c4);
    if (V >= 6) { const float4 t = *(const float4*)(R + (size_t)(m0 + row) * ldc + n0 + 4 * c4); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    *(float4*)(C + (size_t)(m0 + row) * ldc + n0 + 4 * c4) = v; }
}


// ---- V36: prefetch distance 2 (two staging register sets), early LDS store ----
__global__ __launch_bounds__(256, 2) void probe_pf2(const float* __restrict__ A, const float* __restrict__ W, float* C, const float* R,
                                                    int lda, int ldw, int ldc, int nk, int tiles_n) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kStage];
  float* const As = lds; float* const Bs = lds + 2 * kStage;
  const int bid = blockIdx.x;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int lrow = tid >> 3, lc4 = tid & 7;
  const float* Ag = A + (size_t)(m0 + lrow) * lda + 4 * lc4;
  const float* Wg = W + (size_t)(n0 + lrow) * ldw + 4 * lc4;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
#define GLOADS(k0) do { sa0 = *(const float4*)(Ag + (k0)); sa1 = *(const float4*)(Ag + (size_t)32 * lda + (k0)); \
  sa2 = *(const float4*)(Ag + (size_t)64 * lda + (k0)); sa3 = *(const float4*)(Ag + (size_t)96 * lda + (k0)); \
  sb0 = *(const float4*)(Wg + (k0)); sb1 = *(const float4*)(Wg + (size_t)32 * ldw + (k0)); \
  sb2 = *(const float4*)(Wg + (size_t)64 * ldw + (k0)); sb3 = *(const float4*)(Wg + (size_t)96 * ldw + (k0)); } while (0)
#define SSTORES(buf) do { float* as_ = As + (buf) * kStage + lrow * kLds + 4 * lc4; float* bs_ = Bs + (buf) * kStage + lrow * kLds + 4 * lc4; \
  *(float4*)(as_) = sa0; *(float4*)(as_ + 32 * kLds) = sa1; *(float4*)(as_ + 64 * kLds) = sa2; *(float4*)(as_ + 96 * kLds) = sa3; \
  *(float4*)(bs_) = sb0; *(float4*)(bs_ + 32 * kLds) = sb1; *(float4*)(bs_ + 64 * kLds) = sb2; *(float4*)(bs_ + 96 * kLds) = sb3; } while (0)
  f32x16 acc00, acc01, acc10, acc11;
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  GLOAD(0); SSTORE(0);
  GLOAD(32 < nk * 32 ? 32 : 0);            // tile 1 in set R
  __syncthreads();
  const int a_off = (wr * 64 + r) * kLds + 4 * h, b_off = (wc * 64 + r) * kLds + 4 * h;
  float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
  FRAG(fa0, fa1, fb0, fb1, 0, 0);
  // iteration kt computes tile kt from buffer kt&1; tile kt+1 is in registers (set R if kt even, S if odd);
  // tile kt+2 is loaded into the other set at the top.
  for (int kt = 0; kt < nk; kt += 2) {
    { const int k2 = (kt + 2 < nk ? kt + 2 : nk - 1) * 32;
      GLOADS(k2); SB;
      FRAG(ga0, ga1, gb0, gb1, 0, 1); SB; M16(fa0, fa1, fb0, fb1) SB;
      FRAG(fa0, fa1, fb0, fb1, 0, 2); SB; M16(ga0, ga1, gb0, gb1) SB;
      FRAG(ga0, ga1, gb0, gb1, 0, 3); SB; SSTORE(1); SB; M16(fa0, fa1, fb0, fb1) SB;
      __syncthreads();
      FRAG(fa0, fa1, fb0, fb1, 1, 0); SB; M16(ga0, ga1, gb0, gb1) SB; }
    { const int k3 = (kt + 3 < nk ? kt + 3 : nk - 1) * 32;
      GLOAD(k3); SB;
      FRAG(ga0, ga1, gb0, gb1, 1, 1); SB; M16(fa0, fa1, fb0, fb1) SB;
      FRAG(fa0, fa1, fb0, fb1, 1, 2); SB; M16(ga0, ga1, gb0, gb1) SB;
      FRAG(ga0, ga1, gb0, gb1, 1, 3); SB; SSTORES(0); SB; M16(fa0, fa1, fb0, fb1) SB;
      __syncthreads();
      FRAG(fa0, fa1, fb0, fb1, 0, 0); SB; M16(ga0, ga1, gb0, gb1) SB; }
  }
  __syncthreads();
  float* const Cs = lds; constexpr int kCs = 132;
  { float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 64 + r;
    for (int e = 0; e < 16; ++e) { const int ro = ((e & 3) + 8 * (e >> 2)) * kCs; cw[ro] = acc00[e]; cw[ro + 32] = acc01[e]; cw[ro + 32 * kCs] = acc10[e]; cw[ro + 32 * kCs + 32] = acc11[e]; } }
  __syncthreads();
  const int c4 = tid & 31, rsub = tid >> 5;
  for (int pass = 0; pass < 16; ++pass) { const int row = pass * 8 + rsub;
    *(float4*)(C + (size_t)(m0 + row) * ldc + n0 + 4 * c4) = *(const float4*)(Cs + row * kCs + 4 * c4); }
}
float run_pf2(const float* A, const float* W, float* C, const float* R, int M, int N, int K, int iters) {
  const int tiles_n = N / 128, blocks = (M / 128) * tiles_n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe_pf2, dim3(blocks), dim3(256), 0, 0, A, W, C, R, K, K, N, K / 32, tiles_n);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe_pf2, dim3(blocks), dim3(256), 0, 0, A, W, C, R, K, K, N, K / 32, tiles_n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

// ---- V40: LDS-DMA staging (global_load_lds_dwordx4), unpadded 128-B rows, source-side XOR swizzle ----
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
template <int EARLY>
__global__ __launch_bounds__(256, 2) void probe_dma(const float* __restrict__ A, const float* __restrict__ W, float* C, const float* R,
                                                    int lda, int ldw, int ldc, int nk, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * 4096];   // A0 A1 B0 B1, each 128 x 32 floats
  const int bid = blockIdx.x;
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  // DMA source addresses: instruction j of this wave fills LDS rows (wave*4+j)*8 + (lane>>3), chunk slot lane&7
  const int drow = wave * 32 + (lane >> 3);           // + 8*j
  const int dcp = lane & 7;
  const float* Asrc[4]; const float* Wsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = drow + 8 * j;
    const int c = dcp ^ ((row >> 1) & 7);
    Asrc[j] = A + (size_t)(m0 + row) * lda + 4 * c;
    Wsrc[j] = W + (size_t)(n0 + row) * ldw + 4 * c;
  }
#define DMA(buf, k0) do { _Pragma("unroll") for (int j = 0; j < 4; ++j) { \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(Asrc[j] + (k0)), (lds_ptr_t)(lds + (buf) * 4096 + (wave * 4 + j) * 256), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(Wsrc[j] + (k0)), (lds_ptr_t)(lds + 8192 + (buf) * 4096 + (wave * 4 + j) * 256), 16, 0, 0); } } while (0)
  f32x16 acc00, acc01, acc10, acc11;
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  DMA(0, 0);
  __syncthreads();
  const int ra0_ = wr * 64 + r, ra1_ = ra0_ + 32, rb0_ = wc * 64 + r, rb1_ = rb0_ + 32;
  const int xa0 = (ra0_ >> 1) & 7, xa1 = (ra1_ >> 1) & 7, xb0 = (rb0_ >> 1) & 7, xb1 = (rb1_ >> 1) & 7;
  float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define FRAGD(A0, A1, B0, B1, buf, kb) do { const float* as_ = lds + (buf) * 4096; const float* bs_ = lds + 8192 + (buf) * 4096; const int c_ = 2 * (kb) + h; \
  A0 = *(const float4*)(as_ + ra0_ * 32 + 4 * (c_ ^ xa0)); A1 = *(const float4*)(as_ + ra1_ * 32 + 4 * (c_ ^ xa1)); \
  B0 = *(const float4*)(bs_ + rb0_ * 32 + 4 * (c_ ^ xb0)); B1 = *(const float4*)(bs_ + rb1_ * 32 + 4 * (c_ ^ xb1)); } while (0)
  FRAGD(fa0, fa1, fb0, fb1, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const int knext = (kt + 1 < nk ? kt + 1 : kt) * 32;
    DMA(cur ^ 1, knext); SB;
    FRAGD(ga0, ga1, gb0, gb1, cur, 1); SB;
    M16(fa0, fa1, fb0, fb1) SB;
    FRAGD(fa0, fa1, fb0, fb1, cur, 2); SB;
    M16(ga0, ga1, gb0, gb1) SB;
    FRAGD(ga0, ga1, gb0, gb1, cur, 3); SB;
    M16(fa0, fa1, fb0, fb1) SB;
    __syncthreads();
    FRAGD(fa0, fa1, fb0, fb1, cur ^ 1, 0); SB;
    M16(ga0, ga1, gb0, gb1) SB;
  }
  __syncthreads();
  float* const Cs = lds; constexpr int kCs = 132;   // 128*132 = 16896 floats > 16384: use 2-pass? keep simple: stride 128 (conflicts ok for probe)
  { float* cw = Cs + (wr * 64 + 4 * h) * 128 + wc * 64 + r;
    for (int e = 0; e < 16; ++e) { const int ro = ((e & 3) + 8 * (e >> 2)) * 128; cw[ro] = acc00[e]; cw[ro + 32] = acc01[e]; cw[ro + 32 * 128] = acc10[e]; cw[ro + 32 * 128 + 32] = acc11[e]; } }
  __syncthreads();
  const int c4 = tid & 31, rsub = tid >> 5;
  for (int pass = 0; pass < 16; ++pass) { const int row = pass * 8 + rsub;
    *(float4*)(C + (size_t)(m0 + row) * ldc + n0 + 4 * c4) = *(const float4*)(Cs + row * 128 + 4 * c4); }
}
float run_dma(const float* A, const float* W, float* C, const float* R, int M, int N, int K, int iters) {
  const int tiles_n = N / 128, blocks = (M / 128) * tiles_n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe_dma<0>, dim3(blocks), dim3(256), 0, 0, A, W, C, R, K, K, N, K / 32, tiles_n);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe_dma<0>, dim3(blocks), dim3(256), 0, 0, A, W, C, R, K, K, N, K / 32, tiles_n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

template <int V> float run(const float* A, const float* W, float* C, const float* R, int M, int N, int K, int iters) {
  const int tiles_n = N / 128, blocks = (M / 128) * tiles_n;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, A, W, C, R, K, K, N, K / 32, tiles_n);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, A, W, C, R, K, K, N, K / 32, tiles_n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int M = 16000, N = 2048, K = 512;
  float *A, *W, *C, *R; hipMalloc(&R, (size_t)M * N * 4); hipMemset(R, 0, (size_t)M * N * 4);
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
  const double fl = 2.0 * M * N * K;
  for (int rep = 0; rep < 2; ++rep) {
    float t0 = run<0>(A, W, C, R, M, N, K, 20), t1 = run<1>(A, W, C, R, M, N, K, 20), t2 = run<2>(A, W, C, R, M, N, K, 20),
          t3 = run<3>(A, W, C, R, M, N, K, 20), t4 = run<4>(A, W, C, R, M, N, K, 20), t5 = run<5>(A, W, C, R, M, N, K, 20), t6 = run<6>(A, W, C, R, M, N, K, 20);
    printf("V0 mfma only      %7.1f us %6.1f TF\n", t0 * 1e3, fl / t0 / 1e9);
    printf("V1 + lds frags    %7.1f us %6.1f TF\n", t1 * 1e3, fl / t1 / 1e9);
    printf("V2 + barrier      %7.1f us %6.1f TF\n", t2 * 1e3, fl / t2 / 1e9);
    printf("V3 + global/stage %7.1f us %6.1f TF\n", t3 * 1e3, fl / t3 / 1e9);
    printf("V4 + epilogue     %7.1f us %6.1f TF\n", t4 * 1e3, fl / t4 / 1e9);
    { float a = run<30>(A, W, C, R, M, N, K, 20), b = run<31>(A, W, C, R, M, N, K, 20);
      printf("V30 gload only    %7.1f us %6.1f TF\n", a * 1e3, fl / a / 1e9); printf("V31 dswrite only  %7.1f us %6.1f TF\n", b * 1e3, fl / b / 1e9);
      float c = run<32>(A, W, C, R, M, N, K, 20); printf("V32 early sstore  %7.1f us %6.1f TF\n", c * 1e3, fl / c / 1e9);
      float d3 = run<33>(A, W, C, R, M, N, K, 20), d4 = run<34>(A, W, C, R, M, N, K, 20), d5 = run<35>(A, W, C, R, M, N, K, 20);
      printf("V33 +sleep32      %7.1f us %6.1f TF\nV34 +sleep64      %7.1f us %6.1f TF\nV35 +sleep128     %7.1f us %6.1f TF\n", d3 * 1e3, fl / d3 / 1e9, d4 * 1e3, fl / d4 / 1e9, d5 * 1e3, fl / d5 / 1e9); }
    { float e6 = run<36>(A, W, C, R, M, N, K, 20), e7 = run<37>(A, W, C, R, M, N, K, 20), e8 = run<38>(A, W, C, R, M, N, K, 20), e9 = run<39>(A, W, C, R, M, N, K, 20);
      printf("V36' early+epi     %7.1f us %6.1f TF\nV37 +stagger 7us   %7.1f us %6.1f TF\nV38 +stagger 14us  %7.1f us %6.1f TF\nV39 +stagger 28us  %7.1f us %6.1f TF\n",
             e6 * 1e3, fl / e6 / 1e9, e7 * 1e3, fl / e7 / 1e9, e8 * 1e3, fl / e8 / 1e9, e9 * 1e3, fl / e9 / 1e9); }
    { float d = run_pf2(A, W, C, R, M, N, K, 20); printf("V36 prefetch dist 2 %7.1f us %6.1f TF\n", d * 1e3, fl / d / 1e9);
      std::vector<float> c1((size_t)128 * N), c2((size_t)128 * N);
      run<4>(A, W, C, R, M, N, K, 1); hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
      run_pf2(A, W, C, R, M, N, K, 1); hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost);
      double md = 0; for (size_t i = 0; i < c1.size(); ++i) md = fmax(md, fabs((double)c1[i] - c2[i])); printf("    V36 vs V4 max diff %g\n", md); }
    { float d = run_dma(A, W, C, R, M, N, K, 20); printf("V40 LDS-DMA full   %7.1f us %6.1f TF\n", d * 1e3, fl / d / 1e9);
      std::vector<float> c1((size_t)128 * N), c2((size_t)128 * N);
      run<4>(A, W, C, R, M, N, K, 1); hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
      run_dma(A, W, C, R, M, N, K, 1); hipMemcpy(c2.data(), C, c2.size() * 4, hipMemcpyDeviceToHost);
      double md = 0; for (size_t i = 0; i < c1.size(); ++i) md = fmax(md, fabs((double)c1[i] - c2[i])); printf("    V40 vs V4 max diff %g (c=%g)\n", md, (double)c1[12345]); }
    printf("V5 + xcd swizzle  %7.1f us %6.1f TF\n", t5 * 1e3, fl / t5 / 1e9);
    printf("V6 + residual     %7.1f us %6.1f TF\n", t6 * 1e3, fl / t6 / 1e9);
  }
  return 0;
}
