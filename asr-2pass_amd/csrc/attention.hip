// Fused multi-head attention (SAN-M self-attention and decoder cross-attention), d_k = 128, fp32 on
// v_mfma_f32_32x32x2_f32, flash-style online softmax.  Stands in for the MatMul/Softmax/MatMul nodes
// of every attention block inside the reference's `m_session_->Run` (onnxruntime/src/paraformer.cpp:541).
//
// One workgroup = 4 waves = 128 query rows of one (utterance, head); each wave owns 32 query rows and
// keeps its Q slice in registers.  Key/value tiles of 32 keys are staged global -> registers -> LDS,
// double buffered.  Per tile a wave computes S^T = K * Q^T (key on the MFMA row, query on the lane:
// "swapped QK^T"), so that
//   * a query's 32 scores live in one lane pair (l, l^32): row max / row sum = 16 register ops + one
//     cross-half shuffle, no LDS;
//   * P^T is already laid out as the B operand of O^T += V^T * P^T — the probabilities never leave
//     their registers; V^T fragments are read from the row-major V tile with conflict-free
//     ds_read_b32 (32 consecutive d per lane half);
//   * O^T keeps the query on the lane too, so the online-softmax rescale is a per-lane scalar multiply.
// The output tile is transposed through LDS once at the end and stored as full 512-B rows.
// Keys >= kv_len are masked to -inf; tile rows past the segment end are clamped to the last valid
// row, so no out-of-segment memory is read.
#include "kernels.h"

#include <math.h>
#include <stdlib.h>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kQW = 32;            // query rows per wave
constexpr int kQB = 128;           // query rows per block
constexpr int kKT = 32;            // keys per tile

// HD = head dimension: 128 (Paraformer SAN-M / cross-attention) or 32 (CT-Transformer, 256 / 8 heads)
template <int HD>
__global__ __launch_bounds__(256, 2) void attention_kernel(
    const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
    const float* __restrict__ V, int ldv, float* __restrict__ O, int ldo,
    const int* __restrict__ q_off, const int* __restrict__ q_len, const int* __restrict__ kv_off,
    const int* __restrict__ kv_len, const int* __restrict__ q_kv_limit, float scale) {
  constexpr int kHeadDim = HD;
  constexpr int kKS = HD + 4;                        // padded K-tile row stride: conflict-free ds_read_b128
  constexpr int kKVBuf = kKT * kKS + kKT * HD;       // floats per (K,V) buffer
  constexpr int kLdsFloats = (2 * kKVBuf > 4 * kQW * kKS) ? 2 * kKVBuf : 4 * kQW * kKS;
  constexpr int NKB = HD / 8;                        // MFMA k-blocks of the QK^T product
  constexpr int ND = HD / 32;                        // 32-wide d tiles of the output
  constexpr int NP = HD / 32;                        // staging passes (256 threads move 1024 floats per pass)
  constexpr int C4 = HD / 4;                         // float4 chunks per row
  constexpr int RPP = 256 / C4;                      // rows per staging pass
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];

  // grid = (head, utterance, query block): consecutive workgroups go to consecutive XCDs, so with the query block as the
  // SLOWEST index all query blocks of one (utterance, head) land on the same XCD and share its K/V tiles in that L2
  const int b = blockIdx.y, head = blockIdx.x;
  const int Lq = q_len[b];
  const int q0 = blockIdx.z * kQB;
  if (q0 >= Lq) return;
  const int Lk = kv_len[b];
  const size_t qbase = (size_t)q_off[b], kbase = (size_t)kv_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- Q slice of this lane: row q0 + wave*32 + r, d = 8*kb + 4*h + kk -------------------------
  float4 qreg[NKB];
  int klim = Lk;              // keys this lane's query may see: all, or a per-query prefix (CT-Transformer VadMask)
  {
    int qrow = q0 + wave * kQW + r;
    if (qrow >= Lq) qrow = Lq - 1;
    if (q_kv_limit) { const int l = q_kv_limit[qbase + qrow]; klim = l < Lk ? l : Lk; }
    const float* qp = Q + (qbase + qrow) * ldq + head * kHeadDim + 4 * h;
    // Q is pre-multiplied by scale * log2(e): the scores come out of the MFMAs in the base-2 softmax domain, which
    // saves a multiply per score per tile (the softmax below is exp2(s - m); its ratios are those of exp)
    const float qs = scale * 1.44269504088896340736f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      float4 qv = *reinterpret_cast<const float4*>(qp + kb * 8);
      qv.x *= qs; qv.y *= qs; qv.z *= qs; qv.w *= qs;
      qreg[kb] = qv;
    }
  }
  const bool limited = q_kv_limit != nullptr;      // per-query key limits: every tile takes the masked path

  // ---- K/V tile staging (named registers + sched_barriers: hipcc otherwise spills the staging
  //      arrays to scratch and waits for the loads right where they are issued) -------------------------
  const int lrow = tid / C4, lc4 = tid % C4;   // RPP rows x C4 float4 per pass, NP passes
  float4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
  rk1 = rk2 = rk3 = rv1 = rv2 = rv3 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* Kh = K + kbase * ldk + head * kHeadDim + 4 * lc4;
  const float* Vh = V + kbase * ldv + head * kHeadDim + 4 * lc4;
  // one row group (RPP keys) of the next K or V tile -> staging registers / staging registers -> LDS
#define PFHIP_K_LOAD(RK, i, kt)                                                         \
  do {                                                                                  \
    int key_ = (kt) * kKT + lrow + RPP * (i);                                           \
    key_ = key_ < Lk ? key_ : Lk - 1;                                                   \
    RK = *reinterpret_cast<const float4*>(Kh + (size_t)key_ * ldk);                     \
  } while (0)
#define PFHIP_V_LOAD(RV, i, kt)                                                         \
  do {                                                                                  \
    int key_ = (kt) * kKT + lrow + RPP * (i);                                           \
    key_ = key_ < Lk ? key_ : Lk - 1;                                                   \
    RV = *reinterpret_cast<const float4*>(Vh + (size_t)key_ * ldv);                     \
  } while (0)
#define PFHIP_K_STORE(RK, i, buf) \
  *reinterpret_cast<float4*>(lds + (buf) * kKVBuf + (lrow + RPP * (i)) * kKS + 4 * lc4) = RK
#define PFHIP_V_STORE(RV, i, buf) \
  *reinterpret_cast<float4*>(lds + (buf) * kKVBuf + kKT * kKS + (lrow + RPP * (i)) * kHeadDim + 4 * lc4) = RV

  f32x16 oacc0, oacc1, oacc2, oacc3;
#pragma unroll
  for (int e = 0; e < 16; ++e) { oacc0[e] = 0.f; oacc1[e] = 0.f; oacc2[e] = 0.f; oacc3[e] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const int nkt = (Lk + kKT - 1) / kKT;
  PFHIP_K_LOAD(rk0, 0, 0); PFHIP_V_LOAD(rv0, 0, 0);
  if (NP > 1) {
    PFHIP_K_LOAD(rk1, 1, 0); PFHIP_V_LOAD(rv1, 1, 0); PFHIP_K_LOAD(rk2, 2, 0); PFHIP_V_LOAD(rv2, 2, 0);
    PFHIP_K_LOAD(rk3, 3, 0); PFHIP_V_LOAD(rv3, 3, 0);
  }
  PFHIP_K_STORE(rk0, 0, 0); PFHIP_V_STORE(rv0, 0, 0);
  if (NP > 1) {
    PFHIP_K_STORE(rk1, 1, 0); PFHIP_V_STORE(rv1, 1, 0); PFHIP_K_STORE(rk2, 2, 0); PFHIP_V_STORE(rv2, 2, 0);
    PFHIP_K_STORE(rk3, 3, 0); PFHIP_V_STORE(rv3, 3, 0);
  }
  __syncthreads();

  // Memory instructions are issued singly between MFMA groups (the next tile's 2*NP global loads under the QK^T
  // MFMAs, its 2*NP LDS writes under the PV MFMAs): in bursts they starve the matrix pipe (see gemm.hip).
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    // unconditional prefetch of the next tile (the last iteration re-fetches its own): straight-line body
    const int ktn = kt + 1 < nkt ? kt + 1 : kt;
    const float* ks = lds + cur * kKVBuf;
    const float* vs = ks + kKT * kKS;

    // S^T[key][q] = sum_d K[key][d] * Q[q][d]; the K fragment of k-block kb+1 is read under the 4 MFMAs of kb.
    // Two accumulators, alternated, so consecutive MFMAs never wait for each other's result.
    f32x16 sacc, sacb;
#pragma unroll
    for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; sacb[e] = 0.f; }
    const float* kp = ks + r * kKS + 4 * h;
    float4 ka = *reinterpret_cast<const float4*>(kp);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const float4 kn = *reinterpret_cast<const float4*>(kp + (kb < NKB - 1 ? kb + 1 : kb) * 8);
      if (kb == 0) PFHIP_K_LOAD(rk0, 0, ktn);
      if (kb == 1) PFHIP_V_LOAD(rv0, 0, ktn);
      if (NP > 1) {
        if (kb == 2) PFHIP_K_LOAD(rk1, 1, ktn);
        if (kb == 3) PFHIP_V_LOAD(rv1, 1, ktn);
        if (kb == 4) PFHIP_K_LOAD(rk2, 2, ktn);
        if (kb == 5) PFHIP_V_LOAD(rv2, 2, ktn);
        if (kb == 6) PFHIP_K_LOAD(rk3, 3, ktn);
        if (kb == 7) PFHIP_V_LOAD(rv3, 3, ktn);
      }
      __builtin_amdgcn_sched_barrier(0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.x, qreg[kb].x, sacc, 0, 0, 0);
      sacb = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.y, qreg[kb].y, sacb, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.z, qreg[kb].z, sacc, 0, 0, 0);
      sacb = __builtin_amdgcn_mfma_f32_32x32x2f32(ka.w, qreg[kb].w, sacb, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ka = kn;
    }

    // online softmax (base 2) for query column r; this lane holds keys (e&3) + 8*(e>>2) + 4*h of the tile.
    // Tiles that lie wholly inside the key range skip the masking (wave-uniform branch).
    float tmax = -INFINITY;
    if (!limited && (kt + 1) * kKT <= Lk) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float sv = sacc[e] + sacb[e];
        sacc[e] = sv;
        tmax = fmaxf(tmax, sv);
      }
    } else {
      const int key0 = kt * kKT + 4 * h;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = key0 + (e & 3) + 8 * (e >> 2);
        const float sv = (key < klim) ? sacc[e] + sacb[e] : -INFINITY;
        sacc[e] = sv;
        tmax = fmaxf(tmax, sv);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pv = __builtin_amdgcn_exp2f(sacc[e] - m_new);
      sacc[e] = pv;
      psum += pv;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        oacc0[e] *= alpha;
        if (ND > 1) { oacc1[e] *= alpha; oacc2[e] *= alpha; oacc3[e] *= alpha; }
      }
    }

    // O^T[d][q] += sum_key V[key][d] * P^T[key][q]; V values of step e+1 are read under the 4 MFMAs of e
    const float* vp = vs + (4 * h) * kHeadDim + r;
    float va0 = vp[0], va1 = 0.f, va2 = 0.f, va3 = 0.f;
    if (ND > 1) { va1 = vp[32]; va2 = vp[64]; va3 = vp[96]; }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int en = e < 15 ? e + 1 : e;
      const float* vrow = vp + ((en & 3) + 8 * (en >> 2)) * kHeadDim;
      const float vn0 = vrow[0];
      float vn1 = 0.f, vn2 = 0.f, vn3 = 0.f;
      if (ND > 1) { vn1 = vrow[32]; vn2 = vrow[64]; vn3 = vrow[96]; }
      if (e == 0) PFHIP_K_STORE(rk0, 0, cur ^ 1);
      if (e == 1) PFHIP_V_STORE(rv0, 0, cur ^ 1);
      if (NP > 1) {
        if (e == 2) PFHIP_K_STORE(rk1, 1, cur ^ 1);
        if (e == 3) PFHIP_V_STORE(rv1, 1, cur ^ 1);
        if (e == 4) PFHIP_K_STORE(rk2, 2, cur ^ 1);
        if (e == 5) PFHIP_V_STORE(rv2, 2, cur ^ 1);
        if (e == 6) PFHIP_K_STORE(rk3, 3, cur ^ 1);
        if (e == 7) PFHIP_V_STORE(rv3, 3, cur ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float pb = sacc[e];
      oacc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, pb, oacc0, 0, 0, 0);
      if (ND > 1) {
        oacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, pb, oacc1, 0, 0, 0);
        oacc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(va2, pb, oacc2, 0, 0, 0);
        oacc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(va3, pb, oacc3, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      va0 = vn0; va1 = vn1; va2 = vn2; va3 = vn3;
    }
    __syncthreads();
  }
#undef PFHIP_K_LOAD
#undef PFHIP_V_LOAD
#undef PFHIP_K_STORE
#undef PFHIP_V_STORE

  // ---- normalise, transpose through LDS, store full rows ------------------------------------------
  const float inv_l = 1.0f / l_run;
  float* os = lds + wave * (kQW * kKS);
#define PFHIP_O_STORE(OACC, dt)                                                          \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                        \
    float4 o4;                                                                           \
    o4.x = OACC[4 * g + 0] * inv_l; o4.y = OACC[4 * g + 1] * inv_l;                      \
    o4.z = OACC[4 * g + 2] * inv_l; o4.w = OACC[4 * g + 3] * inv_l;                      \
    /* registers 4g..4g+3 are d = dt*32 + 8g + 4h + (0..3) of query column r */          \
    *reinterpret_cast<float4*>(os + r * kKS + (dt) * 32 + 8 * g + 4 * h) = o4;           \
  }
  PFHIP_O_STORE(oacc0, 0)
  if (ND > 1) { PFHIP_O_STORE(oacc1, 1) PFHIP_O_STORE(oacc2, 2) PFHIP_O_STORE(oacc3, 3) }
#undef PFHIP_O_STORE
  __syncthreads();
  // each wave stores its own 32 x HD tile as full rows: C4 lanes x float4 per row, 64/C4 rows per pass
  {
    constexpr int RW = 64 / C4;
#pragma unroll
    for (int pass = 0; pass < kQW / RW; ++pass) {
      const int row = pass * RW + lane / C4, cc = lane % C4;
      const int qrow = q0 + wave * kQW + row;
      if (qrow < Lq) {
        const float4 o4 = *reinterpret_cast<const float4*>(os + row * kKS + 4 * cc);
        *reinterpret_cast<float4*>(O + (qbase + qrow) * ldo + head * kHeadDim + 4 * cc) = o4;
      }
    }
  }
}

}  // namespace

void launch_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                      float* O, int ldo, const int* q_off, const int* q_len, const int* kv_off,
                      const int* kv_len, int B, int H, int max_q_len, float scale, hipStream_t s) {
  launch_attention_hd(Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off, kv_len, B, H, max_q_len, scale, kHeadDim, s);
}

void launch_attention_masked(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                             const int* q_off, const int* q_len, const int* kv_off, const int* kv_len,
                             const int* q_kv_limit, int B, int H, int max_q_len, float scale, int head_dim, hipStream_t s) {
  if (B <= 0 || max_q_len <= 0) return;
  const dim3 grid(H, B, (max_q_len + kQB - 1) / kQB), block(256);
  if (head_dim == 32)
    hipLaunchKernelGGL(attention_kernel<32>, grid, block, 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off,
                       kv_len, q_kv_limit, scale);
  else
    hipLaunchKernelGGL(attention_kernel<128>, grid, block, 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off,
                       kv_len, q_kv_limit, scale);
}

// the split-operand attention: two fp16 planes / three products (attention_x3.hip, default) or three bf16 planes / six products
// (attention_x6.hip, PFHIP_ATT_X3=0)
static bool att_x3_on() {
  static const bool x3 = [] { const char* e = getenv("PFHIP_ATT_X3"); return !(e && e[0] == '0'); }();
  return x3 && !launch_ctx().exact;
}
static void launch_attention_split(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                                   const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H, int max_q_len,
                                   float scale, hipStream_t s, const float* fsmn_w = nullptr, float* mem = nullptr, int ldmem = 0,
                                   bool mem_accumulate = false, void* planes_hi = nullptr, void* planes_lo = nullptr, int plane_rows = 0) {
  if (att_x3_on()) launch_attention_x3(Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off, kv_len, B, H, max_q_len, scale, s, fsmn_w, mem, ldmem, mem_accumulate, planes_hi, planes_lo, plane_rows);
  else launch_attention_x6(Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off, kv_len, B, H, max_q_len, scale, s, fsmn_w, mem, ldmem, mem_accumulate);
}

static bool att_x6_on() {
  static const bool x6 = [] { const char* e = getenv("PFHIP_ATT_X6"); return !(e && e[0] == '0'); }();
  return x6;
}

bool attention_fsmn_is_fused(int max_len) {
  static const bool fuse = [] { const char* e = getenv("PFHIP_ATT_FSMN"); return !(e && e[0] == '0'); }();
  return fuse && att_x6_on() && max_len > 64;
}

bool attention_planes_ok(int max_len) { return attention_fsmn_is_fused(max_len) && att_x3_on(); }

void launch_attention_fsmn(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                           const int* off, const int* len, int B, int H, int max_len, float scale, const float* fsmn_w, float* mem,
                           int ldmem, hipStream_t s, bool mem_accumulate, void* planes_hi, void* planes_lo, int plane_rows) {
  if (B <= 0 || max_len <= 0) return;
  if (attention_fsmn_is_fused(max_len)) {
    launch_attention_split(Q, ldq, K, ldk, V, ldv, O, ldo, off, len, off, len, B, H, max_len, scale, s, fsmn_w, mem, ldmem, mem_accumulate,
                           planes_hi, planes_lo, plane_rows);
    return;
  }
  launch_fsmn(V, ldv, fsmn_w, nullptr, 0, mem, ldmem, off, len, B, max_len, H * kHeadDim, s);
  launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, off, len, off, len, B, H, max_len, scale, s);
}

void launch_attention_hd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                         const int* q_off, const int* q_len, const int* kv_off, const int* kv_len, int B, int H,
                         int max_q_len, float scale, int head_dim, hipStream_t s) {
  if (B <= 0 || max_q_len <= 0) return;
  // d_k = 128 without per-query limits (encoder self-attention, decoder cross-attention): both products on the BF16 matrix
  // cores with the exact three-way split (attention_x6.hip) — 1.5 x this file's fp32-MFMA kernel.  PFHIP_ATT_X6=0 keeps fp32.
  static const bool x6 = [] { const char* e = getenv("PFHIP_ATT_X6"); return !(e && e[0] == '0'); }();
  if (x6 && head_dim == 128 && max_q_len > 64) {      // streaming windows (20 queries) would leave 7 of its 8 waves idle
    launch_attention_split(Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off, kv_len, B, H, max_q_len, scale, s);
    return;
  }
  const dim3 grid(H, B, (max_q_len + kQB - 1) / kQB), block(256);
  if (head_dim == 32)
    hipLaunchKernelGGL(attention_kernel<32>, grid, block, 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off,
                       kv_len, static_cast<const int*>(nullptr), scale);
  else
    hipLaunchKernelGGL(attention_kernel<128>, grid, block, 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, q_off, q_len, kv_off,
                       kv_len, static_cast<const int*>(nullptr), scale);
}

}  // namespace pfhip
