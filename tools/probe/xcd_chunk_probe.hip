// Feasibility probe for a per-XCD persistent streaming chunk ("structural idea"): 32 workgroups confined to ONE XCD walk
// P phases; a phase streams one weight block (each workgroup its 1/32 slice, cold: the blocks form a ring larger than the
// Infinity Cache), then all 32 meet at an XCD-local barrier (L2 atomic + bounded spin, as blstm.hip's step barrier).  Prints the time
// per phase for several block sizes, with and without the next block's loads issued before the barrier — to hold against the
// launch-per-phase path (7-8 us per 3-4-MB launch, boundary included).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/xcd_chunk_probe.hip -o build/xcd_chunk_probe && build/xcd_chunk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 512, kGroup = 32;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <bool PREFETCH>
__global__ __launch_bounds__(kThreads, 1) void chunk_kernel(const float* __restrict__ w, size_t block_floats, int n_blocks_ring, int phases,
                                                            unsigned* bar, float* sink, int per_thread_f4) {
  if (blockIdx.x % 8 != 0) return;                           // round-robin placement: these 32 share an XCD (verified below)
  const int j = blockIdx.x / 8, tid = threadIdx.x;
  const unsigned my_xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
  if (tid == 0) {
    if (j == 0) bar[2] = my_xcc + 1u;
  }
  float acc = 0.f;
  const size_t slice = block_floats / kGroup;
  auto src = [&](int p) { return w + (size_t)(p % n_blocks_ring) * block_floats + (size_t)j * slice + (size_t)tid * 4; };
  f32x4 cur[16];
  if (PREFETCH) {
#pragma unroll
    for (int u = 0; u < 16; ++u) if (u < per_thread_f4) cur[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src(0) + (size_t)u * kThreads * 4));
  }
  for (int p = 0; p < phases; ++p) {
    f32x4 nxt[16];
    if (PREFETCH) {
      if (p + 1 < phases) {
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < per_thread_f4) nxt[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src(p + 1) + (size_t)u * kThreads * 4));
      }
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) if (u < per_thread_f4) cur[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src(p) + (size_t)u * kThreads * 4));
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) if (u < per_thread_f4) acc += (cur[u][0] + cur[u][1]) + (cur[u][2] + cur[u][3]);
    // XCD-local barrier
    __syncthreads();
    if (tid == 0) {
      atomicAdd(&bar[0], 1u);
      const unsigned want = (unsigned)kGroup * (unsigned)(p + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins > (1u << 22)) { bar[1] = 1u; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (__hip_atomic_load(&bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_xcc + 1u && p == 0) bar[3] = 1u;   // placement check
    }
    __syncthreads();
    if (PREFETCH) {
#pragma unroll
      for (int u = 0; u < 16; ++u) cur[u] = nxt[u];
    }
  }
  if (acc == 1.2345e-30f) sink[0] = acc;
}

int main() {
  const int phases = 250;
  const size_t ring_bytes = (size_t)768 << 20;
  float* w; unsigned* bar; float* sink;
  CK(hipMalloc(&w, ring_bytes)); CK(hipMalloc(&bar, 64)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(w, 0, ring_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t kb : {0ul, 1024ul, 3072ul, 4096ul}) {
    for (int pf = 0; pf < 2; ++pf) {
      const size_t block_floats = kb * 256;                    // kb KB / 4
      const int per_thread = (int)(block_floats / kGroup / (kThreads * 4));      // float4 per thread per phase (<= 16)
      const int ring = block_floats ? (int)(ring_bytes / (block_floats * 4)) : 1;
      float best = 1e9f;
      unsigned flags[4] = {0, 0, 0, 0};
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(bar, 0, 64));
        CK(hipEventRecord(e0));
        if (pf) hipLaunchKernelGGL(chunk_kernel<true>, dim3(256), dim3(kThreads), 0, 0, w, block_floats, ring, phases, bar, sink, per_thread);
        else hipLaunchKernelGGL(chunk_kernel<false>, dim3(256), dim3(kThreads), 0, 0, w, block_floats, ring, phases, bar, sink, per_thread);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
        CK(hipMemcpy(flags, bar, 16, hipMemcpyDeviceToHost));
      }
      printf("block %5zu KB  prefetch %d: %7.2f us per phase (%6.1f GB/s on one XCD)  timeout %u  misplaced %u\n", kb, pf,
             1e3f * best / phases, kb ? kb * 1024.0 / (best / phases * 1e-3) / 1e9 : 0.0, flags[1], flags[3]);
    }
  }
  return 0;
}
