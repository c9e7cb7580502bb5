// Probe: what a dependent launch costs on this box (same stream, eager), by kernel shape.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(1024) void touch_kernel(const float4* __restrict__ w, float* out, size_t n4) {
  // every thread reads a strided share of n4 float4s (cold HBM stream), one add per element
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = w[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d; hipMalloc(&d, 4);
  const size_t big = (size_t)1 << 30;          // 1 GiB: larger than the Infinity Cache, so every pass is cold
  float4* w; hipMalloc(&w, big); hipMemset(w, 0, big);
  float* o; hipMalloc(&o, 4);
  struct Cfg { int blocks, threads; size_t bytes; const char* what; };
  std::vector<Cfg> cfgs = {{1, 64, 0, "empty 1x64"}, {128, 1024, 0, "empty 128x1024"}, {256, 1024, 0, "empty 256x1024"},
                           {128, 1024, 1u << 20, "read 1 MB"}, {128, 1024, 4u << 20, "read 4 MB"}, {256, 1024, 4u << 20, "read 4 MB / 256 blocks"},
                           {256, 512, 4u << 20, "read 4 MB / 256x512"}, {512, 256, 4u << 20, "read 4 MB / 512x256"}, {256, 1024, 16u << 20, "read 16 MB"}};
  for (const Cfg& c : cfgs) {
    const int n = 400;
    for (int rep = 0; rep < 2; ++rep) {
      hipStreamSynchronize(s);
      const double t0 = now();
      for (int i = 0; i < n; ++i) {
        if (c.bytes == 0) hipLaunchKernelGGL(empty_kernel, dim3(c.blocks), dim3(c.threads), 0, s, d);
        else hipLaunchKernelGGL(touch_kernel, dim3(c.blocks), dim3(c.threads), 0, s, w + ((size_t)i * c.bytes / 16) % (big / 16 - c.bytes / 16), o, c.bytes / 16);
      }
      hipStreamSynchronize(s);
      const double dt = now() - t0;
      if (rep) std::printf("%-28s %7.2f us per launch\n", c.what, dt / n * 1e6);
    }
  }
  return 0;
}
