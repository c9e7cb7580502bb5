// Dev probe (GPU): per-block timeline of the production-shaped fp32 MFMA GEMM.
// Each block records wall-clock stamps (100 MHz s_memrealtime) at entry, after the first tile is staged,
// after the K loop and after its stores have drained, plus its XCC / SE / CU ids, so the host can print how
// the 2 co-resident blocks of a CU overlap and how long each phase really takes.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off gemm_timeline.hip -o gemm_timeline && ./gemm_timeline
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "gemm_loop_gen.h"
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kLds = 36, kStage = 128 * kLds, kCs = 132;

struct Rec { unsigned long long t0, t1, t2, t3; unsigned hw, xcc; };

__global__ __launch_bounds__(256, 2) void gemm_tl(const float* __restrict__ A, const float* __restrict__ W, float* C,
                                                  const float* R, int lda, int ldw, int ldc, int nk, int tiles_n,
                                                  int n_tiles, Rec* rec) {
  __shared__ __attribute__((aligned(16))) float lds[4 * kStage];
  float* const As = lds; float* const Bs = lds + 2 * kStage;
  const unsigned long long t0 = wall_clock64();
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
#define GLOAD(k0) do { ra0 = *(const float4*)(Ag + (k0)); ra1 = *(const float4*)(Ag + (size_t)32 * lda + (k0)); \
  ra2 = *(const float4*)(Ag + (size_t)64 * lda + (k0)); ra3 = *(const float4*)(Ag + (size_t)96 * lda + (k0)); \
  rb0 = *(const float4*)(Wg + (k0)); rb1 = *(const float4*)(Wg + (size_t)32 * ldw + (k0)); \
  rb2 = *(const float4*)(Wg + (size_t)64 * ldw + (k0)); rb3 = *(const float4*)(Wg + (size_t)96 * ldw + (k0)); } while (0)
#define SSTORE(buf) do { float* as_ = As + (buf) * kStage + lrow * kLds + 4 * lc4; float* bs_ = Bs + (buf) * kStage + lrow * kLds + 4 * lc4; \
  *(float4*)(as_) = ra0; *(float4*)(as_ + 32 * kLds) = ra1; *(float4*)(as_ + 64 * kLds) = ra2; *(float4*)(as_ + 96 * kLds) = ra3; \
  *(float4*)(bs_) = rb0; *(float4*)(bs_ + 32 * kLds) = rb1; *(float4*)(bs_ + 64 * kLds) = rb2; *(float4*)(bs_ + 96 * kLds) = rb3; } while (0)
  f32x16 acc00, acc01, acc10, acc11;
  for (int e = 0; e < 16; ++e) { acc00[e] = 0.f; acc01[e] = 0.f; acc10[e] = 0.f; acc11[e] = 0.f; }
  GLOAD(0);
  SSTORE(0);
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  const int a_off = (wr * 64 + r) * kLds + 4 * h, b_off = (wc * 64 + r) * kLds + 4 * h;
  float4 fa0, fa1, fb0, fb1, ga0, ga1, gb0, gb1;
#define FRAG(A0, A1, B0, B1, buf, kb) do { const float* as_ = As + (buf) * kStage + a_off + (kb) * 8; const float* bs_ = Bs + (buf) * kStage + b_off + (kb) * 8; \
  A0 = *(const float4*)(as_); A1 = *(const float4*)(as_ + 32 * kLds); B0 = *(const float4*)(bs_); B1 = *(const float4*)(bs_ + 32 * kLds); } while (0)
#define M4(A0, A1, B0, B1, c) acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B0.c, acc00, 0, 0, 0); acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.c, B1.c, acc01, 0, 0, 0); \
  acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B0.c, acc10, 0, 0, 0); acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.c, B1.c, acc11, 0, 0, 0);
#define M16(A0, A1, B0, B1) M4(A0, A1, B0, B1, x) M4(A0, A1, B0, B1, y) M4(A0, A1, B0, B1, z) M4(A0, A1, B0, B1, w)
#define SB __builtin_amdgcn_sched_barrier(0)
  FRAG(fa0, fa1, fb0, fb1, 0, 0);
#define GL(reg, base, ld, j, k0) reg = *(const float4*)(base + (size_t)(32 * (j)) * ld + (k0))
#define SW(reg, base, buf, j) *(float4*)(base + (buf) * kStage + lrow * kLds + 4 * lc4 + 32 * (j) * kLds) = reg
#define FR(reg, base, off, buf, kb, j) reg = *(const float4*)(base + (buf) * kStage + off + (kb) * 8 + 32 * (j) * kLds)
#define MM(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
  for (int kt = 0; kt < nk; kt += 2) {
    { const int knext = (kt + 1 < nk ? kt + 1 : nk - 1) * 32; LOOP_BODY_A_0 }
    { const int knext = (kt + 2 < nk ? kt + 2 : nk - 1) * 32; LOOP_BODY_A_1 }
  }
  __syncthreads();
  const unsigned long long t2 = wall_clock64();
  float* const Cs = lds;
  { float* cw = Cs + (wr * 64 + 4 * h) * kCs + wc * 64 + r;
    for (int e = 0; e < 16; ++e) { const int ro = ((e & 3) + 8 * (e >> 2)) * kCs; cw[ro] = acc00[e]; cw[ro + 32] = acc01[e]; cw[ro + 32 * kCs] = acc10[e]; cw[ro + 32 * kCs + 32] = acc11[e]; } }
  __syncthreads();
  const int c4 = tid & 31, rsub = tid >> 5;
#pragma unroll 4
  for (int pass = 0; pass < 16; ++pass) { const int row = pass * 8 + rsub;
    float4 v = *(const float4*)(Cs + row * kCs + 4 * c4);
    if (R) { const float4 t = *(const float4*)(R + (size_t)(m0 + row) * ldc + n0 + 4 * c4); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
    *(float4*)(C + (size_t)(m0 + row) * ldc + n0 + 4 * c4) = v; }
  if (rec) {
    __builtin_amdgcn_s_waitcnt(0);          // stores acknowledged
    const unsigned long long t3 = wall_clock64();
    if (tid == 0) { Rec q; q.t0 = t0; q.t1 = t1; q.t2 = t2; q.t3 = t3; q.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      q.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); rec[blockIdx.x] = q; }
  }
}

static void analyse(const std::vector<Rec>& v, const char* name, double flops, float ev_ms) {
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto& q : v) { tmin = std::min(tmin, q.t0); tmax = std::max(tmax, q.t3); }
  double pro = 0, loop = 0, epi = 0;
  for (auto& q : v) { pro += q.t1 - q.t0; loop += q.t2 - q.t1; epi += q.t3 - q.t2; }
  const double n = (double)v.size(), tick = 0.01;   // us per tick (100 MHz)
  printf("%s: %zu blocks, span %.1f us (event %.1f us, %.1f TF); avg per block: prologue %.2f  loop %.2f  epilogue %.2f us\n", name,
         v.size(), (tmax - tmin) * tick, ev_ms * 1e3, flops / ev_ms / 1e9, pro / n * tick, loop / n * tick, epi / n * tick);
  // per-CU occupancy: how long does a CU hold 0 / 1 / 2 blocks, and how long are 0 / 1 / 2 of them inside the K loop
  std::map<unsigned, std::vector<const Rec*>> cu;
  for (auto& q : v) { const unsigned key = ((q.xcc & 15) << 16) | (q.hw & 0xff00); cu[key].push_back(&q); }
  double res[3] = {0, 0, 0}, inl[3] = {0, 0, 0};
  size_t mn = 1 << 30, mx = 0;
  for (auto& kv : cu) {
    mn = std::min(mn, kv.second.size()); mx = std::max(mx, kv.second.size());
    std::vector<std::pair<unsigned long long, int>> ev, el;
    for (auto* q : kv.second) { ev.push_back({q->t0, 1}); ev.push_back({q->t3, -1}); el.push_back({q->t1, 1}); el.push_back({q->t2, -1}); }
    auto sweep = [&](std::vector<std::pair<unsigned long long, int>>& e, double* out) {
      std::sort(e.begin(), e.end());
      unsigned long long prev = tmin; int c = 0;
      for (auto& x : e) { out[std::min(c, 2)] += (double)(x.first - prev); prev = x.first; c += x.second; }
      out[0] += (double)(tmax - prev);
    };
    sweep(ev, res); sweep(el, inl);
  }
  const double tot = (double)(tmax - tmin) * cu.size();
  printf("   %zu CUs seen, %zu..%zu blocks per CU; CU time with 0/1/2 resident blocks: %.1f / %.1f / %.1f %%;  with 0/1/2 blocks inside the K loop: %.1f / %.1f / %.1f %%\n",
         cu.size(), mn, mx, 100 * res[0] / tot, 100 * res[1] / tot, 100 * res[2] / tot, 100 * inl[0] / tot, 100 * inl[1] / tot, 100 * inl[2] / tot);
  // loop duration of a block vs. the share of that time during which another block of the same CU was in its K loop
  { double sum[5] = {0, 0, 0, 0, 0}; int cnt[5] = {0, 0, 0, 0, 0};
    for (auto& kv : cu)
      for (auto* q : kv.second) {
        double ov = 0;
        for (auto* o : kv.second) if (o != q) { const unsigned long long lo = std::max(q->t1, o->t1), hi = std::min(q->t2, o->t2); if (hi > lo) ov += (double)(hi - lo); }
        const double f = ov / (double)(q->t2 - q->t1);
        const int b = std::min(4, (int)(f * 5));
        sum[b] += (double)(q->t2 - q->t1) * tick; cnt[b]++;
      }
    printf("   K-loop time by co-resident overlap share [0-20%%,..,80-100%%]:");
    for (int b = 0; b < 5; ++b) printf("  %.1f us (n=%d)", cnt[b] ? sum[b] / cnt[b] : 0.0, cnt[b]);
    printf("\n"); }
  // start-time histogram in 10 us buckets: does the grid move in lock-step rounds?
  const int nb = (int)((tmax - tmin) * tick / 10) + 1;
  std::vector<int> hs(nb, 0), he(nb, 0);
  for (auto& q : v) { hs[(int)((q.t0 - tmin) * tick / 10)]++; he[(int)((q.t2 - tmin) * tick / 10)]++; }
  printf("   starts per 10us:"); for (int i = 0; i < nb; ++i) printf(" %d", hs[i]);
  printf("\n   loop-ends per 10us:"); for (int i = 0; i < nb; ++i) printf(" %d", he[i]);
  printf("\n");
}

int main(int argc, char** argv) {
  const size_t dyn = getenv("SOLO") ? 60 * 1024 : 0;
  if (dyn) hipFuncSetAttribute((const void*)gemm_tl, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
  struct Shape { int M, N, K; bool res; const char* name; };
  const Shape shapes[] = {{16000, 2048, 512, false, "ffn1 16000x2048x512"}, {16000, 512, 2048, true, "ffn2 16000x512x2048 +res"},
                          {16000, 1536, 512, false, "qkv 16000x1536x512"}, {16000, 512, 512, true, "out 16000x512x512 +res"}};
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    float *A, *W, *C, *R; Rec* rec;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&R, (size_t)M * N * 4);
    hipMemset(R, 0, (size_t)M * N * 4);
    std::vector<float> h((size_t)std::max(M, N) * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    const int tiles_n = N / 128, blocks = (M / 128) * tiles_n;
    hipMalloc(&rec, blocks * sizeof(Rec));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_tl, dim3(blocks), dim3(256), dyn, 0, A, W, C, s.res ? R : nullptr, K, K, N, K / 32, tiles_n, blocks, (Rec*)nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gemm_tl, dim3(blocks), dim3(256), dyn, 0, A, W, C, s.res ? R : nullptr, K, K, N, K / 32, tiles_n, blocks, (Rec*)nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    hipLaunchKernelGGL(gemm_tl, dim3(blocks), dim3(256), dyn, 0, A, W, C, s.res ? R : nullptr, K, K, N, K / 32, tiles_n, blocks, rec);
    hipDeviceSynchronize();
    std::vector<Rec> v(blocks);
    hipMemcpy(v.data(), rec, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
    analyse(v, s.name, 2.0 * M * N * K, ms);
    hipFree(A); hipFree(W); hipFree(C); hipFree(R); hipFree(rec);
  }
  return 0;
}
