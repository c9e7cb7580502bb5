// The fused attention of attention_x3.hip on K / V that ARRIVE as fp16 planes (round 4).
//
// attention_x3.hip stages its K / V tiles global fp32 -> registers -> split into two fp16 planes -> LDS; timing-only builds of that
// kernel say the staging is the one part of its loop that does not hide behind the MFMAs (65 us as built, 50 without it, encoder
// shape 32 x 4 heads x 500 x 500).  Here the QKV projection's epilogue (gemm_p3.hip, OUT = 5) has already written K and V as ROW-MAJOR
// planes — hi = fp16_rtz(x), lo = fp16_rn(x - hi), [rows][2 d] fp16 per plane, K in columns 0 .. d - 1 and V in d .. 2 d - 1: the
// 2 + 2 bytes of the fp32 value they replace, the very planes attention_x3.hip would have computed — and a tile is four LDS-DMAs per
// wave (`global_load_lds_dwordx4`: no VGPR, no vector instruction, no ds_write).
//
// LDS image of a 32-key tile, per plane: plain 256-byte rows [key][128 d] with the sixteen 16-byte chunks of a row permuted,
// chunk ch of row r at slot ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) — the permutation is applied to the per-lane SOURCE address of the DMA
// (its LDS side is lane-linear), and serves both kinds of read without a conflict (cdna_hip_programming.md T10 image (b)):
//   S^T = K Q^T     A operand = K rows by `ds_read_b128` (lane = key, 8 consecutive d);
//   O^T += V^T P^T  A operand = V COLUMNS by `ds_read_b64_tr_b16` (gfx950's transposing read: a 16-lane group reads 4 keys x 16 d
//                   and each lane receives one d of the 4 keys) — no transposed copy of V exists anywhere; the key order of the
//                   k-slots is the one the probabilities have in the accumulator registers (attention_x3.hip), so two reads (keys
//                   16 t + 4 h + 0..3 and 16 t + 8 + 4 h + 0..3) make one operand.
// Everything else — Q planes in registers, base-2 online softmax with the lazy rescale, MFMA order per accumulator, epilogue, the
// SAN-M memory block folded in front (reading V as hi + lo: the 22-23-bit value the products see) — is attention_x3.hip's, so the
// context equals that kernel's bit for bit on the same K / V planes.
#include "kernels.h"

#include <math.h>

#include <atomic>

namespace pfhip {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using float2v = __attribute__((ext_vector_type(2))) float;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef PFHIP_ATTP_ABLATE
#define PFHIP_ATTP_ABLATE 0      // timing-only builds (results WRONG): 1 no FSMN prologue, 2 no DMA of the next tile, 10 no K-fragment reads, 11 no V reads either, 12 no softmax either
#endif
constexpr int kHD = 128, kQW = 32, kNW = 8, kQB = kNW * kQW, kKT = 32;
constexpr int kRowB = 2 * kHD;                   // 256 bytes per key row of a plane
constexpr int kPl = kKT * kRowB;                 // 8,192: one plane of one tile
constexpr int kBuf = 4 * kPl;                    // 32,768: K hi | K lo | V hi | V lo
constexpr int kOS = kHD + 4;                     // floats per row of the output transpose tile
constexpr int kLdsBytes = kNW * kQW * kOS * 4;   // 135,168: the output transpose tile (>= 2 * kBuf)
constexpr int kRing = 4;                         // tile buffers: the DMA runs three tiles ahead of the MFMAs
static_assert(kLdsBytes >= kRing * kBuf, "K/V buffers must fit");

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
__device__ __forceinline__ void split8(const float (&v)[8], half8& p0, half8& p1) {
  uint4 a, b;
  a.x = hi_pair(v[0], v[1]); a.y = hi_pair(v[2], v[3]); a.z = hi_pair(v[4], v[5]); a.w = hi_pair(v[6], v[7]);
  b.x = lo_pair(sub_lo(v[0], a.x), sub_hi(v[1], a.x)); b.y = lo_pair(sub_lo(v[2], a.y), sub_hi(v[3], a.y));
  b.z = lo_pair(sub_lo(v[4], a.z), sub_hi(v[5], a.z)); b.w = lo_pair(sub_lo(v[6], a.w), sub_hi(v[7], a.w));
  p0 = __builtin_bit_cast(half8, a); p1 = __builtin_bit_cast(half8, b);
}
// byte offset of 16-byte chunk `ch` of key row `row` inside a plane of a tile
__device__ __forceinline__ int tile_off(int row, int ch) { return kRowB * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
// hi + lo of two packed fp16 pairs -> the fp32 values (exact: the planes do not overlap)
__device__ __forceinline__ float2 join2(unsigned h, unsigned l) {
  const half2v a = __builtin_bit_cast(half2v, h), b = __builtin_bit_cast(half2v, l);
  return make_float2((float)a[0] + (float)b[0], (float)a[1] + (float)b[1]);
}

// X[rows, cols] fp32 -> row-major planes [rows][ldp] (tests, tools): one thread per 8 columns
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ X, int ld, int rows, int cols, unsigned char* __restrict__ hi,
                                                         unsigned char* __restrict__ lo, int ldp) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int pieces = cols / 8;
  if (idx >= (size_t)rows * pieces) return;
  const int row = (int)(idx / pieces), pc = (int)(idx % pieces);
  const float4 a = *reinterpret_cast<const float4*>(X + (size_t)row * ld + 8 * pc);
  const float4 b = *reinterpret_cast<const float4*>(X + (size_t)row * ld + 8 * pc + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  half8 p0, p1;
  split8(v, p0, p1);
  const size_t off = ((size_t)row * ldp + 8 * pc) * 2;
  *reinterpret_cast<uint4*>(hi + off) = __builtin_bit_cast(uint4, p0);
  *reinterpret_cast<uint4*>(lo + off) = __builtin_bit_cast(uint4, p1);
}

__global__ __launch_bounds__(512, 1) void attention_p3_kernel(
    const float* __restrict__ Q, int ldq, const unsigned char* __restrict__ KVh, const unsigned char* __restrict__ KVl, int ldkv, int v_col,
    float* __restrict__ O, int ldo, const int* __restrict__ q_off, const int* __restrict__ q_len, const int* __restrict__ kv_off,
    const int* __restrict__ kv_len, float scale, const float* __restrict__ fsmn_w, float* mem, int ldmem, int mem_accumulate,
    unsigned char* __restrict__ Ph, unsigned char* __restrict__ Pl, int rows_p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int b = blockIdx.y, head = blockIdx.x;
  const int Lq = q_len[b];
  const int q0 = blockIdx.z * kQB;
  if (q0 >= Lq) return;
  const int Lk = kv_len[b];
  const size_t qbase = (size_t)q_off[b], kbase = (size_t)kv_off[b];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const size_t ldb = (size_t)ldkv * 2;                       // bytes per row of a plane
#ifdef PFHIP_ATTP_STAMPS      // dev build (tools/att_stamps.py): s_memtime at the phase boundaries of every wave of workgroup (0, 0, 0), into O
  unsigned long long* const dbg = reinterpret_cast<unsigned long long*>(O) + wave * 128;
  const bool stamping = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0;
  int n_stamp = 0;
  unsigned long long* const dbg2 = dbg + 100;      // outside the loop: entry, after the memory block, after Q, after barrier #-1, after the final barrier
  int n_stamp2 = 0;
#define PFHIP_STAMP2 { asm volatile("s_nop 0" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (stamping) dbg2[n_stamp2] = t_; ++n_stamp2; }
#define PFHIP_STAMP { asm volatile("s_nop 0" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (stamping) dbg[n_stamp] = t_; ++n_stamp; }
  if (stamping) dbg[110] = __builtin_amdgcn_s_memrealtime();
#else
#define PFHIP_STAMP
#define PFHIP_STAMP2
#endif
  PFHIP_STAMP2

  // ---- tile transport: thread -> (key row tid / 16, 16-byte chunk tid % 16) of each of the four planes -----------------------------
  // Default: global -> registers -> LDS (four `global_load_dwordx4` a tile ahead, four `ds_write_b128` into the swizzled image: 16 lanes
  // write one row's sixteen chunks, permuted inside the row — conflict-free).  -DPFHIP_ATTP_DMA=1 builds the LDS-DMA form instead
  // (`global_load_lds_dwordx4`, the permutation on the source address): measured 64 us per launch against attention_x3.hip's 65 whatever
  // the lead (one or three tiles) and wherever in the iteration the four pieces were issued, 55 with the pieces taken out — the issue
  // cost of an LDS-DMA piece inside an MFMA loop (MI355X_MICROARCH.md: 100-185 cycles per 1-KB piece; cdna_hip_programming.md's
  // staging table says the same of attention forwards).
#ifndef PFHIP_ATTP_DMA
#define PFHIP_ATTP_DMA 0
#endif
  const int nkt = (Lk + kKT - 1) / kKT;
  auto tclamp = [&](int t) { return t < nkt ? t : nkt - 1; };
  const int drow = tid >> 4, dslot = tid & 15;
  const int dsw = ((drow & 3) << 2) | ((drow >> 2) & 3);
  const int gch = PFHIP_ATTP_DMA ? (dslot ^ dsw) : dslot;                       // the chunk this thread fetches
  const unsigned char* const gk_h = KVh + kbase * ldb + (size_t)head * kRowB + 16 * gch;
  const unsigned char* const gk_l = KVl + kbase * ldb + (size_t)head * kRowB + 16 * gch;
  const int st_off = kRowB * drow + 16 * (dslot ^ dsw);                          // where it goes inside a plane of the tile image
  uint4 rkh, rkl, rvh, rvl;                                                      // one tile in flight through registers
#define PFHIP_DMA1(src, dst)                                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                   \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
  // fetch tile t (past the end: the last tile again — never read): into registers, or (DMA form) straight into ring slot t % 4
  auto fetch_tile = [&](int t) __attribute__((always_inline)) {
    if (PFHIP_ATTP_ABLATE == 2) return;
    const int tt = tclamp(t);
    int key = tt * kKT + drow;
    key = key < Lk ? key : Lk - 1;                           // past the last key: that key again (those scores are masked)
    const size_t ro = (size_t)key * ldb;
    if (PFHIP_ATTP_DMA) {
      unsigned char* dst = lds + (t & (kRing - 1)) * kBuf + wave * 1024;
      PFHIP_DMA1(gk_h + ro, dst)
      PFHIP_DMA1(gk_l + ro, dst + kPl)
      PFHIP_DMA1(gk_h + ro + 2 * (size_t)v_col, dst + 2 * kPl)
      PFHIP_DMA1(gk_l + ro + 2 * (size_t)v_col, dst + 3 * kPl)
    } else {
      rkh = *reinterpret_cast<const uint4*>(gk_h + ro);
      rkl = *reinterpret_cast<const uint4*>(gk_l + ro);
      rvh = *reinterpret_cast<const uint4*>(gk_h + ro + 2 * (size_t)v_col);
      rvl = *reinterpret_cast<const uint4*>(gk_l + ro + 2 * (size_t)v_col);
    }
  };
  // registers -> ring slot t % 4 (register form only)
  auto store_tile = [&](int t) __attribute__((always_inline)) {
    if (PFHIP_ATTP_DMA || PFHIP_ATTP_ABLATE == 2) return;
    unsigned char* dst = lds + (t & (kRing - 1)) * kBuf + st_off;
    *reinterpret_cast<uint4*>(dst) = rkh;
    *reinterpret_cast<uint4*>(dst + kPl) = rkl;
    *reinterpret_cast<uint4*>(dst + 2 * kPl) = rvh;
    *reinterpret_cast<uint4*>(dst + 3 * kPl) = rvl;
  };
  // the same, one plane at a time — placed between the MFMAs of the loop (register form): a tile's four ds_writes inside the S phase of
  // the tile before it, its four global loads inside the PV phase two tiles before it.  All eight waves doing the whole transport at
  // once behind the barrier cost 700-1000 of a tile's 4800 cycles (in-kernel stamps: 32 KB through the CU's 64 B/clk vector-memory
  // path and its ~80 B/clk LDS write path with nothing else running).
  auto store_piece = [&](int t, int i) __attribute__((always_inline)) {
    if (PFHIP_ATTP_DMA || PFHIP_ATTP_ABLATE == 2) return;
    unsigned char* dst = lds + (t & (kRing - 1)) * kBuf + st_off + i * kPl;
    *reinterpret_cast<uint4*>(dst) = i == 0 ? rkh : i == 1 ? rkl : i == 2 ? rvh : rvl;
  };
  size_t fetch_ro = 0;
  auto fetch_piece = [&](int t, int i) __attribute__((always_inline)) {
    if (PFHIP_ATTP_DMA || PFHIP_ATTP_ABLATE == 2) return;
    if (i == 0) {
      int key = tclamp(t) * kKT + drow;
      key = key < Lk ? key : Lk - 1;
      fetch_ro = (size_t)key * ldb;
    }
    if (i == 0) rkh = *reinterpret_cast<const uint4*>(gk_h + fetch_ro);
    if (i == 1) rkl = *reinterpret_cast<const uint4*>(gk_l + fetch_ro);
    if (i == 2) rvh = *reinterpret_cast<const uint4*>(gk_h + fetch_ro + 2 * (size_t)v_col);
    if (i == 3) rvl = *reinterpret_cast<const uint4*>(gk_l + fetch_ro + 2 * (size_t)v_col);
  };
  if (!PFHIP_ATTP_DMA) fetch_tile(0);      // in flight under the prologue below

  PFHIP_STAMP2
  // ---- SAN-M memory block (attention_x3.hip / rowops.hip fsmn_kernel, same operation order) on v = hi + lo ------------------------
  if (fsmn_w && PFHIP_ATTP_ABLATE != 1) {
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
      const unsigned char* vh = KVh + kbase * ldb + 2 * (size_t)(v_col + c);
      const unsigned char* vl = KVl + kbase * ldb + 2 * (size_t)(v_col + c);
#pragma unroll
      for (int j = 0; j < kStrip + kTaps - 1; ++j) {      // out-of-range rows: a clamped (valid) address, the value replaced by zero
        const int t = t0 - 5 + j;
        const bool in = t >= 0 && t < Lk;
        const size_t ro = (size_t)(in ? t : t0) * ldb;
        const uint2 a = *reinterpret_cast<const uint2*>(vh + ro), bb = *reinterpret_cast<const uint2*>(vl + ro);
        const float2 lo2 = join2(a.x, bb.x), hi2 = join2(a.y, bb.y);
        rows[j] = in ? make_float4(lo2.x, lo2.y, hi2.x, hi2.y) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
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

  // ---- Q planes of this lane: query row q0 + wave*32 + r, k-step s covers d = 16s + 8h + (0..7) ----------------------------------
  // Fetched as whole rows (32 lanes x 16 B = the head's 512 bytes of a row, two rows per instruction) into the wave's own corner of LDS
  // (its output-transpose area) and picked up from there in fragment order: a lane reading ITS row's 32-byte pieces straight from
  // global memory makes every load instruction touch 32 cache lines for 1 KB (in-kernel stamps: 2-10 k cycles of a 20-k prologue).
  half8 qf[8][2];
  {
    float* const qst = reinterpret_cast<float*>(lds) + wave * (kQW * kOS);
    const int c4 = lane & 31, r2 = lane >> 5;
#pragma unroll
    for (int half = 0; half < 2; ++half) {      // two batches of eight loads: sixteen rows in flight, 32 registers
      float4 qv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int qrow = q0 + wave * kQW + 16 * half + 2 * i + r2;
        if (qrow >= Lq) qrow = Lq - 1;
        qv[i] = *reinterpret_cast<const float4*>(Q + (qbase + qrow) * ldq + head * kHD + 4 * c4);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(qst + (16 * half + 2 * i + r2) * kOS + 4 * c4) = qv[i];
    }
    const float qs = scale * 1.44269504088896340736f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(qst + r * kOS + 16 * s + 8 * h);
      const float4 c = *reinterpret_cast<const float4*>(qst + r * kOS + 16 * s + 8 * h + 4);
      const float v[8] = {a.x * qs, a.y * qs, a.z * qs, a.w * qs, c.x * qs, c.y * qs, c.z * qs, c.w * qs};
      split8(v, qf[s][0], qf[s][1]);
    }
  }
  PFHIP_STAMP2
  if (PFHIP_ATTP_DMA) {      // the DMA lands in LDS: only once every wave is done with its Q staging corner
    __syncthreads();
    fetch_tile(0);
    fetch_tile(1);
  }

  // ---- fragment addresses inside a tile (loop-invariant) ----------------------------------------------------------------------------
  // K: key row r, k-step s = chunk 2 s + h.  V: 16-lane group g = lane >> 4 (h = g >> 1), lane 4 q + p of the group addresses key row
  // (block's first key + q), d = 32 dt + 16 (g & 1) + 4 p .. + 3 and receives d = 32 dt + (lane & 31) of the block's four keys.
  // The swizzle is an XOR on address bits that the other terms leave alone, so every fragment address is ONE per-lane offset XOR a
  // constant: K, k-step s: kf0 ^ 32 s;  V, d-tile dt, second block: vf0 ^ 64 dt ^ (2048 + 32).  They are formed inside the loop from
  // (slot base + offset) — sixteen loop-invariant address registers do not fit beside the accumulators, the Q planes and the staging set.
  const int kf0 = tile_off(r, h);
  const int vq = (lane & 15) >> 2, vp = lane & 3, vc = 2 * ((lane >> 4) & 1) + (vp >> 1);
  const int vf0 = tile_off(4 * h + vq, vc) + 8 * (vp & 1);      // relative to the V hi plane; + 16 t key rows = + 4096 bytes
  const unsigned lds_base = (unsigned)(size_t)lds;      // the LDS offset is the low half of the flat address

  f32x16 oacc0, oacc1, oacc2, oacc3;
#pragma unroll
  for (int e = 0; e < 16; ++e) { oacc0[e] = 0.f; oacc1[e] = 0.f; oacc2[e] = 0.f; oacc3[e] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  // ---- the loop: three phases per tile — S (24 MFMAs on one accumulator), softmax (vector only), PV (24 MFMAs) — and one barrier ------
  // Barrier #n (n = -1 before the loop): tile n + 1 is in LDS for every wave, and every wave is past its reads of tile n.
  f32x16 sacc;
  u32x2 va[4][4];      // V operands of a PV step: [d-tile][hi first half, hi second half, lo first, lo second]
#define PFHIP_SB __builtin_amdgcn_sched_barrier(0)
  // The transposing reads are written as asm: behind the builtin the compiler puts `s_waitcnt vmcnt(0)` in front of the first of them
  // (it cannot tell the read from an LDS-DMA in flight into ANOTHER buffer), which serialises every tile's DMA with its compute.
  // So the waits are explicit too: PFHIP_VWAIT before the first MFMA that consumes a set (LDS returns in order).
#if PFHIP_ATTP_ABLATE >= 11
#define PFHIP_TR(dst, addr, imm) if (first_tile) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#else
#define PFHIP_TR(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")
#endif
#define PFHIP_VF(R, dt, t)                                                                                        \
    {                                                                                                             \
      PFHIP_TR(R[dt][0], vad[dt][0], 2 * kPl + 4096 * (t)); PFHIP_TR(R[dt][1], vad[dt][1], 2 * kPl + 4096 * (t));  \
      PFHIP_TR(R[dt][2], vad[dt][0], 3 * kPl + 4096 * (t)); PFHIP_TR(R[dt][3], vad[dt][1], 3 * kPl + 4096 * (t));  \
    }
#define PFHIP_VWAIT(R)                                                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                      \
                 : "+v"(R[0][0]), "+v"(R[0][1]), "+v"(R[0][2]), "+v"(R[0][3]), "+v"(R[1][0]), "+v"(R[1][1]), "+v"(R[1][2]), "+v"(R[1][3]),    \
                   "+v"(R[2][0]), "+v"(R[2][1]), "+v"(R[2][2]), "+v"(R[2][3]), "+v"(R[3][0]), "+v"(R[3][1]), "+v"(R[3][2]), "+v"(R[3][3])     \
                 :: "memory")
#define PFHIP_VHI(R, dt) __builtin_bit_cast(half8, make_uint4(R[dt][0].x, R[dt][0].y, R[dt][1].x, R[dt][1].y))
#define PFHIP_VLO(R, dt) __builtin_bit_cast(half8, make_uint4(R[dt][2].x, R[dt][2].y, R[dt][3].x, R[dt][3].y))

  // S^T[key][q] of tile kt: 8 k-steps x 3 plane products (k_lo q_hi, k_hi q_lo, k_hi q_hi — the order of attention_x3.hip).  Fragment
  // reads run one k-step ahead of their MFMAs and are pinned there (the compiler's own order is read -> wait -> MFMA on one register
  // set); the V operands of the first PV step are requested behind the last score MFMAs.
  auto phase_s = [&](int kt, int lead) __attribute__((always_inline)) {
    const int cur = kt & (kRing - 1);
    const bool first_tile = kt < 0;      // (timing-only builds: never true, unknown to the compiler)
    (void)first_tile;
    const int tk = cur * kBuf + kf0, tv = cur * kBuf + vf0;
    unsigned vad[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { vad[dt][0] = lds_base + (tv ^ (64 * dt)); vad[dt][1] = lds_base + (tv ^ (64 * dt) ^ 2080); }
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#define PFHIP_KF(dst, p, s_) if (PFHIP_ATTP_ABLATE < 10 || first_tile) dst = __builtin_bit_cast(half8, *reinterpret_cast<const uint4*>(lds + (p) * kPl + (tk ^ (32 * (s_)))))
#define PFHIP_MM(a_, b_) sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, sacc, 0, 0, 0)
    half8 k0 = qf[0][0], k1 = qf[0][1], n0 = qf[1][0], n1 = qf[1][1];
    PFHIP_KF(k0, 0, 0); PFHIP_KF(k1, 1, 0); PFHIP_SB;
#define PFHIP_S2(sa)                                                                                              \
    PFHIP_MM(k1, qf[sa][0]); PFHIP_SB; PFHIP_KF(n0, 0, sa + 1); PFHIP_KF(n1, 1, sa + 1); PFHIP_SB;                \
    PFHIP_MM(k0, qf[sa][1]); PFHIP_SB; PFHIP_MM(k0, qf[sa][0]); PFHIP_SB;                                         \
    PFHIP_MM(n1, qf[sa + 1][0]); PFHIP_SB;
    PFHIP_S2(0) PFHIP_KF(k0, 0, 2); PFHIP_KF(k1, 1, 2); PFHIP_SB; PFHIP_MM(n0, qf[1][1]); PFHIP_SB; store_piece(kt + lead, 0); PFHIP_SB; PFHIP_MM(n0, qf[1][0]); PFHIP_SB;
    PFHIP_S2(2) PFHIP_KF(k0, 0, 4); PFHIP_KF(k1, 1, 4); PFHIP_SB; PFHIP_MM(n0, qf[3][1]); PFHIP_SB; store_piece(kt + lead, 1); PFHIP_SB; PFHIP_MM(n0, qf[3][0]); PFHIP_SB;
    PFHIP_S2(4) PFHIP_KF(k0, 0, 6); PFHIP_KF(k1, 1, 6); PFHIP_SB; PFHIP_MM(n0, qf[5][1]); PFHIP_SB; store_piece(kt + lead, 2); PFHIP_SB; PFHIP_MM(n0, qf[5][0]); PFHIP_SB;
    PFHIP_MM(k1, qf[6][0]); PFHIP_SB; PFHIP_KF(n0, 0, 7); PFHIP_KF(n1, 1, 7); PFHIP_SB;
    PFHIP_MM(k0, qf[6][1]); PFHIP_SB; store_piece(kt + lead, 3); PFHIP_SB; PFHIP_MM(k0, qf[6][0]); PFHIP_SB;
    // (the compiler's wait for the last K fragments is `lgkmcnt(0)`: the V reads go behind it, and have the softmax to land)
    PFHIP_MM(n1, qf[7][0]); PFHIP_SB; PFHIP_VF(va, 0, 0) PFHIP_VF(va, 1, 0) PFHIP_SB;
    PFHIP_MM(n0, qf[7][1]); PFHIP_SB; PFHIP_VF(va, 2, 0) PFHIP_VF(va, 3, 0) PFHIP_SB; PFHIP_MM(n0, qf[7][0]); PFHIP_SB;
#undef PFHIP_S2
#undef PFHIP_MM
#undef PFHIP_KF
  };

  // online softmax (base 2) of tile kt for query column r; this lane holds keys (e&3) + 8*(e>>2) + 4*h of the tile
  auto phase_sm = [&](int kt) __attribute__((always_inline)) {
    if (PFHIP_ATTP_ABLATE >= 12) return;
    float tmax = -INFINITY;
    if ((kt + 1) * kKT <= Lk) {
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
    if (__any(tmax > m_run + 10.0f)) {      // lazy rescale: probabilities stay <= 2^10 (attention_x3.hip)
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
      const float pv = __builtin_amdgcn_exp2f(sacc[e] - m_run);
      sacc[e] = pv;
      psum += pv;
    }
    psum += __shfl_xor(psum, 32);
    l_run += psum;
  };

  // O^T[d][q] += V^T P^T of tile kt: two k-steps of 16 keys; k-slot i of step t is register e = 8t + i of the score tile = key
  // 16 t + 8 (i >> 2) + 4 h + (i & 3): the two transposing reads of an operand take keys 16 t + 4 h .. and 16 t + 8 + 4 h ..
  // Per d-tile (an accumulator of its own) v_lo p_hi, v_hi p_lo, v_hi p_hi, as attention_x3.hip; ONE set of V operand registers: a
  // d-tile's operands of step 1 are requested as soon as its three MFMAs of step 0 are issued.
  auto phase_pv = [&](int kt, int lead) __attribute__((always_inline)) {
    const int cur = kt & (kRing - 1);
    const bool first_tile = kt < 0;
    (void)first_tile;
    const int tv = cur * kBuf + vf0;
    unsigned vad[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { vad[dt][0] = lds_base + (tv ^ (64 * dt)); vad[dt][1] = lds_base + (tv ^ (64 * dt) ^ 2080); }
#define PFHIP_PM(OACC, A_, B_) OACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_, B_, OACC, 0, 0, 0); PFHIP_SB;
#define PFHIP_PV3(OACC, dt) PFHIP_PM(OACC, PFHIP_VLO(va, dt), p0) PFHIP_PM(OACC, PFHIP_VHI(va, dt), p1) PFHIP_PM(OACC, PFHIP_VHI(va, dt), p0)
    {
      const float pv[8] = {sacc[0], sacc[1], sacc[2], sacc[3], sacc[4], sacc[5], sacc[6], sacc[7]};
      half8 p0, p1;
      split8(pv, p0, p1);
      PFHIP_SB;
      PFHIP_VWAIT(va);
      PFHIP_SB;
      PFHIP_PV3(oacc0, 0) PFHIP_VF(va, 0, 1) PFHIP_SB;
      PFHIP_PV3(oacc1, 1) PFHIP_VF(va, 1, 1) PFHIP_SB;
      PFHIP_PV3(oacc2, 2) PFHIP_VF(va, 2, 1) PFHIP_SB;
      PFHIP_PV3(oacc3, 3) PFHIP_VF(va, 3, 1) PFHIP_SB;
    }
    {
      const float pv[8] = {sacc[8], sacc[9], sacc[10], sacc[11], sacc[12], sacc[13], sacc[14], sacc[15]};
      half8 p0, p1;
      split8(pv, p0, p1);
      PFHIP_SB;
      PFHIP_VWAIT(va);
      PFHIP_SB;
      PFHIP_PV3(oacc0, 0) fetch_piece(kt + lead + 1, 0); fetch_piece(kt + lead + 1, 1); PFHIP_SB;
      PFHIP_PV3(oacc1, 1) fetch_piece(kt + lead + 1, 2); fetch_piece(kt + lead + 1, 3); PFHIP_SB;
      PFHIP_PV3(oacc2, 2) PFHIP_PV3(oacc3, 3)
    }
#undef PFHIP_PV3
#undef PFHIP_PM
  };
  // Between barrier #n and barrier #n + 1 a wave moves tile n + 2 one stage on: registers -> LDS (fetched an interval ago), then the fetch of
  // tile n + 3 (DMA form: the DMA of tile n + 3, two intervals ahead of its first read).  Ring slot (n + 2) % 4 was last read before
  // barrier #n - 1.
  auto advance = [&](int n) __attribute__((always_inline)) {
    if (PFHIP_ATTP_DMA) fetch_tile(n + 3);      // (register form: the pieces travel inside the phases)
    PFHIP_SB;
  };
  // every wave's part of the next tile is in LDS (register form: its ds_writes are done; DMA form: all but its newest pieces landed)
#if PFHIP_ATTP_DMA
#define PFHIP_TILE_BARRIER asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define PFHIP_TILE_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

#ifndef PFHIP_ATTP_SHIFT
#define PFHIP_ATTP_SHIFT 0
#endif
  // barrier #-1: tile 0 is in LDS for every wave.  (DMA form behind the memory block: its STORES are in the count too, and stores do not
  // retire in order with loads: wait for everything.)
  if (PFHIP_ATTP_DMA) {
    if (fsmn_w) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else PFHIP_TILE_BARRIER;
    fetch_tile(2);
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has its Q planes: the staging corners are free
    store_tile(0);
    fetch_tile(1);      // stored inside the S phase of tile 0
    PFHIP_TILE_BARRIER;
  }
  PFHIP_STAMP2
  if (!PFHIP_ATTP_SHIFT || PFHIP_ATTP_DMA) {
    for (int kt = 0; kt < nkt; ++kt) {
      PFHIP_STAMP
      phase_s(kt, 1);
      PFHIP_STAMP
      phase_sm(kt);
      PFHIP_STAMP
      phase_pv(kt, 1);
      PFHIP_STAMP
      PFHIP_TILE_BARRIER;      // #kt
      PFHIP_STAMP
      advance(kt);
    }
    PFHIP_STAMP
  } else {
    // -DPFHIP_ATTP_SHIFT=1: the SIMD partners run the phases of an interval in different orders — waves 0-3: S(t) softmax(t) PV(t);
    // waves 4-7: softmax(t) PV(t) S(t + 1) — so that one wave's vector phase sits under the other's MFMAs.  Tile t + 2 is then complete
    // by barrier #t (waves 0-3 store it inside S(t), one tile further ahead; waves 4-7 inside S(t + 1)): three tiles live + one being
    // written = the ring of four.
    if (wave < kNW / 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      store_tile(1);
      fetch_tile(2);
      PFHIP_TILE_BARRIER;      // #-1: tile 1 complete
      for (int kt = 0; kt < nkt; ++kt) {
        PFHIP_STAMP
        phase_s(kt, 2);
        PFHIP_STAMP
        phase_sm(kt);
        PFHIP_STAMP
        phase_pv(kt, 2);
        PFHIP_STAMP
        PFHIP_TILE_BARRIER;      // #kt
        PFHIP_STAMP
      }
      PFHIP_STAMP
    } else {
      phase_s(0, 1);
      PFHIP_TILE_BARRIER;      // #-1
      for (int kt = 0; kt < nkt; ++kt) {
        PFHIP_STAMP
        phase_sm(kt);
        PFHIP_STAMP
        phase_pv(kt, 1);
        PFHIP_STAMP
        if (kt + 1 < nkt) phase_s(kt + 1, 1);
        PFHIP_STAMP
        PFHIP_TILE_BARRIER;      // #kt
        PFHIP_STAMP
      }
      PFHIP_STAMP
    }
  }
#undef PFHIP_TILE_BARRIER
#undef PFHIP_VLO
#undef PFHIP_VHI
#undef PFHIP_VWAIT
#undef PFHIP_VF
#undef PFHIP_TR
#undef PFHIP_SB
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // the last (redundant) pieces must not land in the output tile
  PFHIP_STAMP2
#undef PFHIP_DMA1

  // ---- normalise, transpose through LDS, store (attention_x3.hip) -------------------------------------------------------------------
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
#ifdef PFHIP_ATTP_STAMPS
  PFHIP_STAMP2
  if (blockIdx.y == 0 && blockIdx.z == 0) { if (stamping) dbg[111] = __builtin_amdgcn_s_memrealtime(); return; }      // its O rows hold the stamps
#endif
  if (Ph) {      // the context as the plane images of gemm_p3.hip
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

void launch_split_rows(const float* X, int ld, int rows, int cols, void* hi, void* lo, int ldp, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  const size_t n = (size_t)rows * (cols / 8);
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, ld, rows, cols, static_cast<unsigned char*>(hi),
                     static_cast<unsigned char*>(lo), ldp);
}

void launch_attention_p3(const float* Q, int ldq, const void* kv_hi, const void* kv_lo, int ldkv, int v_col, float* O, int ldo, const int* q_off,
                         const int* q_len, const int* kv_off, const int* kv_len, int B, int H, int max_q_len, float scale, hipStream_t s,
                         const float* fsmn_w, float* mem, int ldmem, bool mem_accumulate, void* planes_hi, void* planes_lo, int plane_rows) {
  if (B <= 0 || max_q_len <= 0) return;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!(attr_done.load(std::memory_order_relaxed) >> (dev & 63) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_p3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    attr_done.fetch_or(1ull << (dev & 63));
  }
  const dim3 grid(H, B, (max_q_len + kQB - 1) / kQB), block(512);
  hipLaunchKernelGGL(attention_p3_kernel, grid, block, kLdsBytes, s, Q, ldq, static_cast<const unsigned char*>(kv_hi),
                     static_cast<const unsigned char*>(kv_lo), ldkv, v_col, O, ldo, q_off, q_len, kv_off, kv_len, scale, fsmn_w, mem, ldmem,
                     mem_accumulate ? 1 : 0, static_cast<unsigned char*>(planes_hi), static_cast<unsigned char*>(planes_lo), plane_rows);
}

}  // namespace pfhip
