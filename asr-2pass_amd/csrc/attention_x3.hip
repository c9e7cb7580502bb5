// Fused attention with both products on the FP16 matrix cores, THREE products per block: Q, K, V and the probabilities are staged
// as two fp16 planes (hi = fp16_rtz(x), lo = fp16_rn(x - hi): 22-23 significant bits, absolute floor 2^-25 — the scheme of
// gemm_x3.hip, whose header states the precision argument), `v_mfma_f32_32x32x16_f16`, fp32 accumulation, d_k = 128, flash-style
// online softmax.  Sibling of attention_x6.hip (three bf16 planes, six products: PFHIP_ATT_X3=0) and attention.hip (fp32 MFMA:
// PFHIP_ATT_X6=0), same interface; the default for d_k = 128.
//
// One workgroup = 8 waves = 256 query rows of one (utterance, head); a wave keeps its 32 queries' Q planes in registers
// (pre-multiplied by scale*log2 e, then split).  Per 32-key tile:
//   S^T = K Q^T   A operand = K planes from LDS ([key][d] rows, ds_read_b128), B operand = Q planes (registers): 8 k-steps x 3;
//                 a query's 32 scores sit in one lane pair, softmax as in attention.hip;
//   O^T += V^T P^T  B operand = the probabilities, split in registers — lane half h holds keys (e&3)+8(e>>2)+4h, so k-slot i of
//                 step t is key 16t + 8(i>>2) + (i&3) + 4h; the A operand takes the SAME keys from the transposed V planes in LDS
//                 ([d][key] rows: two ds_read_b64), so the permutation cancels: 4 d-tiles x 2 k-steps x 3.
// Range: the probabilities are kept <= 2^10 by the lazy rescale (fp16's largest finite value is 65504); |q| scale log2 e, |k|, |v|
// must stay below 65504 — they are LayerNorm-ed activations times a weight matrix.
// K / V tiles are split while they are staged (global fp32 -> registers -> fp16 planes in LDS; V transposed on the way by
// loading 4 keys x 2 d per thread), double-buffered, one barrier per tile.
// Output: fp32 rows, or — the encoder on large batches — the context as the two fp16 plane images that the output projection
// (gemm_p3.hip) stages by LDS-DMA; then no fp32 context is written at all.
#include "kernels.h"

#include <math.h>

#include <atomic>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using float2v = __attribute__((ext_vector_type(2))) float;

// Timing-only builds (tools/x3_variant.sh <name> "-DPFHIP_ATT_ABLATE=n" attention_x3.hip; results are WRONG for n != 0):
//   1 no FSMN prologue   2 no tile barrier   3 no softmax arithmetic   4 V fragments of one address only   5 no staging of the next tile
//   6 = 2 + 3 + 4 + 5 (MFMAs and K-fragment reads only)
#ifndef PFHIP_ATT_ABLATE
#define PFHIP_ATT_ABLATE 0
#endif
constexpr bool kAbNoFsmn = PFHIP_ATT_ABLATE == 1, kAbNoBar = PFHIP_ATT_ABLATE == 2 || PFHIP_ATT_ABLATE == 6,
               kAbNoSoftmax = PFHIP_ATT_ABLATE == 3 || PFHIP_ATT_ABLATE == 6, kAbOneV = PFHIP_ATT_ABLATE == 4 || PFHIP_ATT_ABLATE == 6,
               kAbNoStage = PFHIP_ATT_ABLATE == 5 || PFHIP_ATT_ABLATE == 6;
constexpr int kHD = 128, kQW = 32, kNW = 8, kQB = kNW * kQW, kKT = 32;   // 8 waves = 256 queries per workgroup, two waves per SIMD
constexpr int kKRow = 272;                       // bytes per key row of a K plane (128 bf16 + 16 pad: conflict-free b128 reads)
constexpr int kKPlane = kKT * kKRow;             // 8,704
constexpr int kVRow = 72;                        // bytes per d row of a V^T plane (32 keys bf16 + 8 pad: conflict-free b64 reads)
constexpr int kVPlane = kHD * kVRow;             // 9,216
constexpr int kBuf = 2 * kKPlane + 2 * kVPlane;  // 35,840
constexpr int kOS = kHD + 4;                     // floats per row of the output transpose tile
constexpr int kLdsBytes = kNW * kQW * kOS * 4;    // 135,168: the output transpose tile (>= 2 * kBuf = 71,680)
static_assert(kLdsBytes >= 2 * kBuf, "K/V buffers must fit");

// x - (float)h for the low / high half of a packed fp16 pair, one instruction each (see gemm_x3.hip)
__device__ __forceinline__ float sub_lo(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float sub_hi(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ unsigned hi_pair(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ unsigned lo_pair(float a, float b) {
  const float2v r = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(r, half2v));
}

// 8 fp32 values -> their two fp16 planes, packed as MFMA operands
__device__ __forceinline__ void split8(const float (&v)[8], half8& p0, half8& p1) {
  uint4 a, b;
  a.x = hi_pair(v[0], v[1]); a.y = hi_pair(v[2], v[3]); a.z = hi_pair(v[4], v[5]); a.w = hi_pair(v[6], v[7]);
  b.x = lo_pair(sub_lo(v[0], a.x), sub_hi(v[1], a.x)); b.y = lo_pair(sub_lo(v[2], a.y), sub_hi(v[3], a.y));
  b.z = lo_pair(sub_lo(v[4], a.z), sub_hi(v[5], a.z)); b.w = lo_pair(sub_lo(v[6], a.w), sub_hi(v[7], a.w));
  p0 = __builtin_bit_cast(half8, a); p1 = __builtin_bit_cast(half8, b);
}

__global__ __launch_bounds__(512, 1) void attention_x3_kernel(
    const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk, const float* __restrict__ V, int ldv,
    float* __restrict__ O, int ldo, const int* __restrict__ q_off, const int* __restrict__ q_len,
    const int* __restrict__ kv_off, const int* __restrict__ kv_len, float scale, const float* __restrict__ fsmn_w,
    float* mem, int ldmem, int mem_accumulate, unsigned char* __restrict__ Ph, unsigned char* __restrict__ Pl, int rows_p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int b = blockIdx.y, head = blockIdx.x;
  const int Lq = q_len[b];
  const int q0 = blockIdx.z * kQB;
  if (q0 >= Lq) return;
  const int Lk = kv_len[b];
  const size_t qbase = (size_t)q_off[b], kbase = (size_t)kv_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- SAN-M memory block of the encoder layer, for this workgroup's 256 rows x this head's 128 channels (self-attention
  // only: q rows = kv rows): mem[t][c] = v[t][c] + sum_j w[c][j] v[t + j - 5][c], zero outside the utterance — the fsmn_kernel of
  // rowops.hip (same operation order: bit-identical), folded in here because the (head, utterance, query block) grid covers
  // every (row, channel) exactly once and the V rows are about to be streamed anyway.  One launch and one 16.9-us kernel per
  // encoder layer less.  With mem_accumulate the memory is added straight into the residual stream (mem = x): the output
  // projection that follows — bandwidth-bound at N = K = 512: 128 MB of operand, two residuals and result per 8.4 GFLOP — then
  // reads one residual instead of two; the 33 MB move into this kernel, which has bandwidth to spare.
  if (fsmn_w && !kAbNoFsmn) {
    constexpr int kTaps = 11, kStrip = 16;
    const int cg = tid & 31, strip = tid >> 5;
    const int c = head * kHD + 4 * cg, t0 = q0 + strip * kStrip;
    if (t0 < Lk) {
      float wk[4][kTaps];
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int j = 0; j < kTaps; ++j) wk[ch][j] = fsmn_w[(size_t)(c + ch) * kTaps + j];
      float4 rows[kStrip + kTaps - 1];
#pragma unroll
      for (int j = 0; j < kStrip + kTaps - 1; ++j) {
        const int t = t0 - 5 + j;
        rows[j] = (t < 0 || t >= Lk) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(V + (kbase + t) * ldv + c);
      }
      // mem_accumulate: mem IS the residual stream, x += memory (each element has exactly one owner).  Its old values are
      // requested eight rows at a time BEFORE the stores of those rows: a load behind a store to the same array cannot be
      // hoisted by the compiler, and one load -> wait -> store per row was sixteen memory latencies in a row.
#pragma unroll
      for (int s0 = 0; s0 < kStrip; s0 += 8) {
        float4 xo[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int t = t0 + s0 + s;
          xo[s] = (mem_accumulate && t < Lk) ? *reinterpret_cast<const float4*>(mem + (kbase + t) * ldmem + c)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int s1 = 0; s1 < 8; ++s1) {
          const int s = s0 + s1, t = t0 + s;
          if (t < Lk) {
            float4 o = rows[s + 5];
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < kTaps; ++j) {
              a.x += wk[0][j] * rows[s + j].x;
              a.y += wk[1][j] * rows[s + j].y;
              a.z += wk[2][j] * rows[s + j].z;
              a.w += wk[3][j] * rows[s + j].w;
            }
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            if (mem_accumulate) { o.x += xo[s1].x; o.y += xo[s1].y; o.z += xo[s1].z; o.w += xo[s1].w; }
            *reinterpret_cast<float4*>(mem + (kbase + t) * ldmem + c) = o;
          }
        }
      }
    }
  }

  // ---- Q planes of this lane: query row q0 + wave*32 + r, k-step s covers d = 16s + 8h + (0..7) ----------------------------
  half8 qf[8][2];
  {
    int qrow = q0 + wave * kQW + r;
    if (qrow >= Lq) qrow = Lq - 1;
    const float* qp = Q + (qbase + qrow) * ldq + head * kHD + 8 * h;
    const float qs = scale * 1.44269504088896340736f;          // scores come out in the base-2 softmax domain
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(qp + 16 * s);
      const float4 c = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
      const float v[8] = {a.x * qs, a.y * qs, a.z * qs, a.w * qs, c.x * qs, c.y * qs, c.z * qs, c.w * qs};
      split8(v, qf[s][0], qf[s][1]);
    }
  }

  // ---- staging maps (512 threads) ---------------------------------------------------------------------------------------
  // K: thread t holds key t/16, d = 64 i + 4 (t%16) + (0..3), i = 0..1 (a load instruction covers 256 contiguous bytes per key)
  const int kkey = tid >> 4, kc = tid & 15;
  const float* Kh = K + kbase * ldk + head * kHD + 4 * kc;
  // V: thread t holds keys 4 (t/64) + (0..3), d = 2 (t%64) + (0..1): its 4 x 2 patch transposes in registers
  const int vd2 = tid & 63, vkg = tid >> 6;
  const float* Vh = V + kbase * ldv + head * kHD + 2 * vd2;
  float4 rk0, rk1;
  float2 rv0, rv1, rv2, rv3;
  auto load_tile = [&](int kt) {
    int key = kt * kKT + kkey;
    key = key < Lk ? key : Lk - 1;
    const float* kp = Kh + (size_t)key * ldk;
    rk0 = *reinterpret_cast<const float4*>(kp);
    rk1 = *reinterpret_cast<const float4*>(kp + 64);
    const int k0 = kt * kKT + 4 * vkg;
    const int last = Lk - 1;
    rv0 = *reinterpret_cast<const float2*>(Vh + (size_t)min(k0, last) * ldv);
    rv1 = *reinterpret_cast<const float2*>(Vh + (size_t)min(k0 + 1, last) * ldv);
    rv2 = *reinterpret_cast<const float2*>(Vh + (size_t)min(k0 + 2, last) * ldv);
    rv3 = *reinterpret_cast<const float2*>(Vh + (size_t)min(k0 + 3, last) * ldv);
  };
  auto store4 = [&](float a, float c, float e, float g, unsigned char* base, int plane_bytes) {   // 4 values -> 2 planes, 8 B each
    const unsigned h0 = hi_pair(a, c), h1 = hi_pair(e, g);
    *reinterpret_cast<uint2*>(base) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(base + plane_bytes) = make_uint2(lo_pair(sub_lo(a, h0), sub_hi(c, h0)), lo_pair(sub_lo(e, h1), sub_hi(g, h1)));
  };
  auto store_tile = [&](int buf) {
    unsigned char* kb = lds + buf * kBuf + kkey * kKRow + 8 * kc;                    // d = 4 kc -> byte 8 kc; + 128 B per i
    store4(rk0.x, rk0.y, rk0.z, rk0.w, kb, kKPlane);
    store4(rk1.x, rk1.y, rk1.z, rk1.w, kb + 128, kKPlane);
    unsigned char* vb = lds + buf * kBuf + 2 * kKPlane + (2 * vd2) * kVRow + 8 * vkg;   // row d, keys 4 vkg .. + 3
    store4(rv0.x, rv1.x, rv2.x, rv3.x, vb, kVPlane);
    store4(rv0.y, rv1.y, rv2.y, rv3.y, vb + kVRow, kVPlane);
  };

  f32x16 oacc0, oacc1, oacc2, oacc3;
#pragma unroll
  for (int e = 0; e < 16; ++e) { oacc0[e] = 0.f; oacc1[e] = 0.f; oacc2[e] = 0.f; oacc3[e] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const int nkt = (Lk + kKT - 1) / kKT;
  load_tile(0);
  store_tile(0);
  load_tile(nkt > 1 ? 1 : 0);                          // raw registers run one tile ahead of the LDS buffers
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const unsigned char* kb = lds + cur * kBuf + r * kKRow + 16 * h;
    const unsigned char* vb = lds + cur * kBuf + 2 * kKPlane + r * kVRow + 8 * h;

    // S^T[key][q]: 8 k-steps x 3 plane products (k_lo q_hi, k_hi q_lo, k_hi q_hi).  The split of the NEXT tile (8 stages of 4-6
    // VALU ops + one LDS write into the other buffer) and the K-fragment reads of the next k-step are placed by hand between the
    // MFMAs and pinned with scheduling barriers, as in attention_x6.hip.
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    unsigned char* const kst = lds + (cur ^ 1) * kBuf + kkey * kKRow + 8 * kc;
    unsigned char* const vst = lds + (cur ^ 1) * kBuf + 2 * kKPlane + (2 * vd2) * kVRow + 8 * vkg;
    unsigned th0, th1;
    float t0, t1, t2, t3;
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
#define PFHIP_KF(dst, p, s_) dst = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(kb + (p) * kKPlane + 32 * (s_)))
#define PFHIP_MM(a_, b_) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, sacc, 0, 0, 0)
    // one stage of the next tile's split: HI writes the high plane of four values and keeps their residuals, LO writes the low plane
#define PFHIP_HI(a_, c_, e_, g_, dst)                                                                     \
  if (!kAbNoStage) { th0 = hi_pair(a_, c_); th1 = hi_pair(e_, g_); *reinterpret_cast<uint2*>(dst) = make_uint2(th0, th1);  \
    t0 = sub_lo(a_, th0); t1 = sub_hi(c_, th0); t2 = sub_lo(e_, th1); t3 = sub_hi(g_, th1); }
#define PFHIP_LO(dst) if (!kAbNoStage) { *reinterpret_cast<uint2*>(dst) = make_uint2(lo_pair(t0, t1), lo_pair(t2, t3)); }
    half8 k0, k1, n0, n1;
    PFHIP_KF(k0, 0, 0); PFHIP_KF(k1, 1, 0);
    PFHIP_SB;
    PFHIP_MM(k1, qf[0][0]); PFHIP_SB;
    PFHIP_KF(n0, 0, 1); PFHIP_KF(n1, 1, 1); PFHIP_SB;
    PFHIP_MM(k0, qf[0][1]); PFHIP_SB;
    PFHIP_HI(rk0.x, rk0.y, rk0.z, rk0.w, kst) PFHIP_SB;
    PFHIP_MM(k0, qf[0][0]); PFHIP_SB;
    PFHIP_MM(n1, qf[1][0]); PFHIP_SB;
    PFHIP_KF(k0, 0, 2); PFHIP_KF(k1, 1, 2); PFHIP_SB;
    PFHIP_MM(n0, qf[1][1]); PFHIP_SB;
    PFHIP_LO(kst + kKPlane) PFHIP_SB;
    PFHIP_MM(n0, qf[1][0]); PFHIP_SB;
    PFHIP_MM(k1, qf[2][0]); PFHIP_SB;
    PFHIP_KF(n0, 0, 3); PFHIP_KF(n1, 1, 3); PFHIP_SB;
    PFHIP_MM(k0, qf[2][1]); PFHIP_SB;
    PFHIP_HI(rk1.x, rk1.y, rk1.z, rk1.w, kst + 128) PFHIP_SB;
    PFHIP_MM(k0, qf[2][0]); PFHIP_SB;
    PFHIP_MM(n1, qf[3][0]); PFHIP_SB;
    PFHIP_KF(k0, 0, 4); PFHIP_KF(k1, 1, 4); PFHIP_SB;
    PFHIP_MM(n0, qf[3][1]); PFHIP_SB;
    PFHIP_LO(kst + 128 + kKPlane) PFHIP_SB;
    PFHIP_MM(n0, qf[3][0]); PFHIP_SB;
    PFHIP_MM(k1, qf[4][0]); PFHIP_SB;
    PFHIP_KF(n0, 0, 5); PFHIP_KF(n1, 1, 5); PFHIP_SB;
    PFHIP_MM(k0, qf[4][1]); PFHIP_SB;
    PFHIP_HI(rv0.x, rv1.x, rv2.x, rv3.x, vst) PFHIP_SB;
    PFHIP_MM(k0, qf[4][0]); PFHIP_SB;
    PFHIP_MM(n1, qf[5][0]); PFHIP_SB;
    PFHIP_KF(k0, 0, 6); PFHIP_KF(k1, 1, 6); PFHIP_SB;
    PFHIP_MM(n0, qf[5][1]); PFHIP_SB;
    PFHIP_LO(vst + kVPlane) PFHIP_SB;
    PFHIP_MM(n0, qf[5][0]); PFHIP_SB;
    PFHIP_MM(k1, qf[6][0]); PFHIP_SB;
    PFHIP_KF(n0, 0, 7); PFHIP_KF(n1, 1, 7); PFHIP_SB;
    PFHIP_MM(k0, qf[6][1]); PFHIP_SB;
    PFHIP_HI(rv0.y, rv1.y, rv2.y, rv3.y, vst + kVRow) PFHIP_SB;
    PFHIP_MM(k0, qf[6][0]); PFHIP_SB;
    PFHIP_MM(n1, qf[7][0]); PFHIP_SB;
    PFHIP_MM(n0, qf[7][1]); PFHIP_SB;
    PFHIP_LO(vst + kVRow + kVPlane) PFHIP_SB;
    PFHIP_MM(n0, qf[7][0]); PFHIP_SB;
#undef PFHIP_LO
#undef PFHIP_HI
#undef PFHIP_MM
#undef PFHIP_KF
#undef PFHIP_SB
    __builtin_amdgcn_sched_barrier(0);
    if (!kAbNoStage) load_tile(kt + 2 < nkt ? kt + 2 : nkt - 1);      // past the end: re-fetch the last tile (never used)
    __builtin_amdgcn_sched_barrier(0);

    // online softmax (base 2) for query column r; this lane holds keys (e&3) + 8*(e>>2) + 4*h of the tile
    float tmax = -INFINITY;
    if (kAbNoSoftmax) {
      tmax = m_run;
    } else if ((kt + 1) * kKT <= Lk) {
#pragma unroll
      for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, sacc[e]);
    } else {
      const int key0 = kt * kKT + 4 * h;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = key0 + (e & 3) + 8 * (e >> 2);
        const float sv = (key < Lk) ? sacc[e] : -INFINITY;
        sacc[e] = sv;
        tmax = fmaxf(tmax, sv);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    // Lazy rescale: the running reference m_run only moves when some query's maximum has outgrown it by more than 2^10 —
    // the probabilities are then at most 2^10 (fp16 planes: largest finite value 65504) and the 64 accumulator multiplies per
    // tile disappear from all but the first tiles; p / l is the same quotient either way.
    if (__any(tmax > m_run + 10.0f)) {
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int e = 0; e < 16; ++e) { oacc0[e] *= alpha; oacc1[e] *= alpha; oacc2[e] *= alpha; oacc3[e] *= alpha; }
    }
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = kAbNoSoftmax ? sacc[e] : __builtin_amdgcn_exp2f(sacc[e] - m_run);
      sacc[e] = pv;
      psum += pv;
    }
    psum += __shfl_xor(psum, 32);
    l_run += psum;

    // O^T[d][q] += V^T P^T: two k-steps of 16 keys; k-slot i of step t is register e = 8t + i of the score tile
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float pv[8] = {sacc[8 * t + 0], sacc[8 * t + 1], sacc[8 * t + 2], sacc[8 * t + 3],
                           sacc[8 * t + 4], sacc[8 * t + 5], sacc[8 * t + 6], sacc[8 * t + 7]};
      half8 p0, p1;
      split8(pv, p0, p1);
#define PFHIP_PV(OACC, dt)                                                                                        \
      {                                                                                                           \
        const unsigned char* vp = kAbOneV ? vb : vb + (dt) * 32 * kVRow + 32 * t;                                 \
        half8 v0, v1;                                                                                             \
        {                                                                                                         \
          const uint2 lo = *reinterpret_cast<const uint2*>(vp), hi = *reinterpret_cast<const uint2*>(vp + 16);    \
          v0 = __builtin_bit_cast(half8, make_uint4(lo.x, lo.y, hi.x, hi.y));                                     \
        }                                                                                                         \
        {                                                                                                         \
          const uint2 lo = *reinterpret_cast<const uint2*>(vp + kVPlane), hi = *reinterpret_cast<const uint2*>(vp + kVPlane + 16); \
          v1 = __builtin_bit_cast(half8, make_uint4(lo.x, lo.y, hi.x, hi.y));                                     \
        }                                                                                                         \
        OACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, p0, OACC, 0, 0, 0);                                     \
        OACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, p1, OACC, 0, 0, 0);                                     \
        OACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, p0, OACC, 0, 0, 0);                                     \
      }
      PFHIP_PV(oacc0, 0) PFHIP_PV(oacc1, 1) PFHIP_PV(oacc2, 2) PFHIP_PV(oacc3, 3)
#undef PFHIP_PV
    }
    if (!kAbNoBar) __syncthreads();
  }

  // ---- normalise, transpose through LDS, store full rows (as attention.hip) ---------------------------------------------
  const float inv_l = 1.0f / l_run;
  float* os = reinterpret_cast<float*>(lds) + wave * (kQW * kOS);
#define PFHIP_O_STORE(OACC, dt)                                                          \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                        \
    float4 o4;                                                                           \
    o4.x = OACC[4 * g + 0] * inv_l; o4.y = OACC[4 * g + 1] * inv_l;                      \
    o4.z = OACC[4 * g + 2] * inv_l; o4.w = OACC[4 * g + 3] * inv_l;                      \
    *reinterpret_cast<float4*>(os + r * kOS + (dt) * 32 + 8 * g + 4 * h) = o4;           \
  }
  PFHIP_O_STORE(oacc0, 0) PFHIP_O_STORE(oacc1, 1) PFHIP_O_STORE(oacc2, 2) PFHIP_O_STORE(oacc3, 3)
#undef PFHIP_O_STORE
  __syncthreads();
  if (Ph) {
    // The context as the two fp16 plane images the output projection stages by LDS-DMA (gemm_p3.hip: [K / 16][rows][16] per plane,
    // the 16-byte halves of a row swapped where row bit 3 is set): lane = (query row, 8-column piece), so one store instruction of a
    // wave covers 32 rows x 32 bytes = 1 KB of an image, contiguous.
    const int pc = h, qrow = q0 + wave * kQW + r;
    const size_t grow = qbase + (size_t)qrow;
    if (qrow < Lq) {
#pragma unroll
      for (int ks = 0; ks < kHD / 16; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(os + r * kOS + 16 * ks + 8 * pc);
        const float4 c = *reinterpret_cast<const float4*>(os + r * kOS + 16 * ks + 8 * pc + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        half8 p0, p1;
        split8(v, p0, p1);
        const size_t off = ((size_t)(head * (kHD / 16) + ks) * rows_p + grow) * 32 + (size_t)((pc ^ (int)((grow >> 3) & 1)) << 4);
        *reinterpret_cast<uint4*>(Ph + off) = __builtin_bit_cast(uint4, p0);
        *reinterpret_cast<uint4*>(Pl + off) = __builtin_bit_cast(uint4, p1);
      }
    }
  } else {
    constexpr int C4 = kHD / 4, RW = 64 / C4;
#pragma unroll
    for (int pass = 0; pass < kQW / RW; ++pass) {
      const int row = pass * RW + lane / C4, cc = lane % C4;
      const int qrow = q0 + wave * kQW + row;
      if (qrow < Lq) {
        const float4 o4 = *reinterpret_cast<const float4*>(os + row * kOS + 4 * cc);
        *reinterpret_cast<float4*>(O + (qbase + qrow) * ldo + head * kHD + 4 * cc) = o4;
      }
    }
  }
}

}  // namespace

void launch_attention_x3(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                         const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H, int max_q_len,
                         float scale, hipStream_t s, const float* fsmn_w, float* mem, int ldmem, bool mem_accumulate, void* planes_hi,
                         void* planes_lo, int plane_rows) {
  if (B <= 0 || max_q_len <= 0) return;
  static std::atomic<unsigned long long> attr_done{0};      // > 64 KB of dynamic LDS needs the opt-in once per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_done.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsBytes);
    attr_done.fetch_or(1ull << (dev & 63));
  }
  const dim3 grid(H, B, (max_q_len + kQB - 1) / kQB), block(512);
  hipLaunchKernelGGL(attention_x3_kernel, grid, block, kLdsBytes, s, Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off,
                     kv_len, scale, fsmn_w, mem, ldmem, mem_accumulate ? 1 : 0, static_cast<unsigned char*>(planes_hi),
                     static_cast<unsigned char*>(planes_lo), plane_rows);
}

}  // namespace pfhip
