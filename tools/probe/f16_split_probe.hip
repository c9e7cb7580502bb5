// Dev probe (GPU): facts the fp16 two-plane GEMM (gemm_x3.hip) rests on.
//   (1) does v_mfma_f32_32x32x16_f16 keep SUBNORMAL fp16 operands, or flush them?
//   (2) what do v_cvt_pkrtz_f16_f32 / a plain (_Float16) cast make of values below 2^-14 and above 65504?
//   (3) cycles: the f16 and bf16 forms of the 32x32x16 MFMA back to back.
// hipcc -O3 --offload-arch=gfx950 f16_split_probe.hip -o f16_split_probe && ./f16_split_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__global__ void mfma_subnormal(float a_val, float b_val, float* out) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void cvt_probe(const float* in, int n, uint32_t* bits_rtz, uint32_t* bits_rn) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const auto p = __builtin_amdgcn_cvt_pkrtz(in[i], 0.f);
  bits_rtz[i] = __builtin_bit_cast(uint32_t, p) & 0xFFFFu;
  bits_rn[i] = (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)in[i]);
}
template <int F16>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
  half8 ah, bh; bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(seed * (threadIdx.x % 13 + i)); bh[i] = (_Float16)(0.37f * (threadIdx.x % 7 + i));
                                ab[i] = (__bf16)(seed * (threadIdx.x % 13 + i)); bb[i] = (__bf16)(0.37f * (threadIdx.x % 7 + i)); }
  f32x16 c0, c1, c2, c3;
  for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; c2[e] = 0.f; c3[e] = 0.f; }
  for (int it = 0; it < iters; ++it) {
    if (F16) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
  float* d; hipMalloc(&d, 1 << 22);
  float h;
  const float cases[][2] = {{ldexpf(1.f, -20), 1.f}, {ldexpf(1.f, -24), 1.f}, {1.f, ldexpf(1.f, -20)}, {ldexpf(1.f, -15), ldexpf(1.f, -15)},
                            {ldexpf(1.f, -14), 1.f}, {ldexpf(3.f, -22), 2.f}};
  for (auto& c : cases) {
    mfma_subnormal<<<1, 64>>>(c[0], c[1], d);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("mfma f16: a=%.3e b=%.3e -> acc[0]=%.6e (exact 16*a*b = %.6e)%s\n", c[0], c[1], h, 16.0 * c[0] * c[1],
           h == 16.f * c[0] * c[1] ? "  KEPT" : "  (flushed or rounded)");
  }
  const float vals[] = {1e-6f, 3e-8f, 6.1e-5f, 6.0e-5f, 65504.f, 65520.f, 70000.f, 1e6f, -1e6f, 1.0009765625f, 1.00146484375f, INFINITY};
  const int n = sizeof(vals) / sizeof(float);
  float* din; uint32_t *b1, *b2; hipMalloc(&din, 4 * n); hipMalloc(&b1, 4 * n); hipMalloc(&b2, 4 * n);
  hipMemcpy(din, vals, 4 * n, hipMemcpyHostToDevice);
  cvt_probe<<<1, 64>>>(din, n, b1, b2);
  uint32_t h1[32], h2[32]; hipMemcpy(h1, b1, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(h2, b2, 4 * n, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt %.9g: pkrtz 0x%04x  cast(rn) 0x%04x\n", vals[i], h1[i], h2[i]);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int f16 = 0; f16 < 2; ++f16) for (int rep = 0; rep < 2; ++rep) {
    const int iters = 20000, blocks = 256 * 4;
    hipEventRecord(e0);
    if (f16) mfma_loop<1><<<blocks, 256>>>(d, iters, 0.01f); else mfma_loop<0><<<blocks, 256>>>(d, iters, 0.01f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 4 * 32768.0;
    printf("%s 32x32x16 loop: %.3f ms -> %.0f TFLOP/s\n", f16 ? "f16 " : "bf16", ms, flops / ms / 1e9);
  }
  return 0;
}
