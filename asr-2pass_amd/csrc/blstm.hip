// CifPredictorV3 timestamp head (the producer of the us_alphas / us_cif_peak tensors that
// Paraformer::Forward hands to TimestampOnnx, onnxruntime/src/paraformer.cpp:545-562): the bidirectional LSTM over the
// x3-upsampled encoder frames, the alpha head and the cif_wo_hidden scan.  The two GEMM-shaped parts (ConvTranspose1d as
// a [T,512]x[512,1536] product, the LSTM input projection for all frames and both directions) run on gemm.hip.
//
// The recurrence is the only serial part of the whole model: 3T steps (1500 for 30 s) of h[B,512] x W_hh^T[512,2048] per
// direction, with all of h exchanged between the steps.  One PERSISTENT kernel runs all steps.  Where the blocks run is
// the design decision: the 8 XCDs' L2 caches are not coherent with each other, so a step barrier across XCDs needs an
// agent-scope release + acquire (L2 write-back + invalidate) — measured 37 us per step with cooperative_groups'
// grid.sync() and the same with hand-rolled fences, 56 ms for a 30-s batch.  Instead EACH DIRECTION LIVES ON ONE XCD
// (workgroup i is dispatched to XCD i % 8: direction d = blocks with blockIdx % 8 == d): its 32 blocks exchange h
// through that XCD's own L2, which is coherent for them — plain stores (L1 is write-through) + `s_waitcnt vmcnt(0)`,
// a counter in the same L2, and `sc1` loads that bypass the reader's L1.  No cache-wide operation in the loop.
// A block owns 16 hidden units x 4 gates = 64 rows of W_hh, held in VGPRs as MFMA B fragments (fp16 planes) for the whole sequence;
// per step it reads the previous h[B,512] (64 KB, L2), runs 2 x 4 x 6 v_mfma_f32_16x16x32_f16 per wave (two fp16 planes per
// operand, three products; the 8 waves split K), reduces through LDS, updates its 16 x B cells (c stays in registers) and publishes 16 x B new h values.
// All utterances of a launch (<= 32) advance together; each walks its own frames (packed layout, no padding frames: a
// padded batch would feed pad frames into the backward direction).  Placement is verified on the device (XCC_ID).
#include "kernels.h"

#include <math.h>

namespace pfhip {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kH = 512;              // LSTM hidden size (= d_model)
constexpr int kUnits = 16;           // hidden units per block
constexpr int kBlocksPerDir = kH / kUnits;     // 32 = the CUs of one XCD
constexpr int kMaxB = 32;            // utterances per launch (2 MFMA row tiles)
constexpr int kXcds = 8;
constexpr int kBlstmThreads = 512;   // 8 waves split K = 512 eight ways: 16 consecutive k per lane
constexpr int kWaves = kBlstmThreads / 64;
constexpr int kKL = kH / kWaves / 4;  // k per lane (4 lane groups per wave)
// state words after the exchange ring: [2] error flag, [4],[5] XCC id + 1 of each direction ([0],[1] unused)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- the recurrent product on the FP16 matrix cores ----------------------------------------------------------------------------
// h[B,512] x W_hh^T on `v_mfma_f32_16x16x4_f32` is 2.1 MFLOP per block and step at the fp32 matrix rate (= the vector rate):
// 3.4 us of a 7.7-us step.  Both operands are staged as two fp16 planes instead (hi = rtz(x), lo = rn(x - hi): 22-23 significant
// bits, the scheme of gemm_x3.hip, whose header states the precision argument) and multiplied with three
// `v_mfma_f32_16x16x32_f16` per 32 k (hi lo + lo hi + hi hi, fp32 accumulation): 48 MFMAs of 16 cycles per wave and step instead
// of 128 of 32.  |h| < 1 always; W_hh is staged times 2^10 (|w| < 32) so that its low plane stays out of fp16's subnormal range,
// and the accumulators are scaled back before the reduction.
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using float2v = __attribute__((ext_vector_type(2))) float;
constexpr float kWScale = 1024.0f, kWScaleInv = 1.0f / 1024.0f;

__device__ __forceinline__ float sub_lo(float x, unsigned h) {      // x - (float)low half of h, one instruction, exact
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float sub_hi(float x, unsigned h) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
// eight consecutive-k values of one lane -> the hi / lo operand fragments of a 16x16x32 MFMA
__device__ __forceinline__ void split8(const float (&v)[8], half8& hi, half8& lo) {
  uint4 a, b;
  unsigned* ap = &a.x;
  unsigned* bp = &b.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ap[i] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2 * i], v[2 * i + 1]));
    const float2v r = {sub_lo(v[2 * i], ap[i]), sub_hi(v[2 * i + 1], ap[i])};
    bp[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, half2v));
  }
  hi = __builtin_bit_cast(half8, a);
  lo = __builtin_bit_cast(half8, b);
}
// W_hh rows of this lane: [gate][16 consecutive k] fp32 -> fragments [gate][k half]
struct WFrags { half8 hi[4][2], lo[4][2]; };
__device__ __forceinline__ void load_w_frags(const float* __restrict__ whh, int dir, int unit0, int n, int kbase, WFrags& w) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* wrow = whh + ((size_t)dir * 4 * 512 + (size_t)g * 512 + unit0 + n) * 512 + kbase;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float4 v0 = *reinterpret_cast<const float4*>(wrow + 8 * m), v1 = *reinterpret_cast<const float4*>(wrow + 8 * m + 4);
      const float v[8] = {v0.x * kWScale, v0.y * kWScale, v0.z * kWScale, v0.w * kWScale,
                          v1.x * kWScale, v1.y * kWScale, v1.z * kWScale, v1.w * kWScale};
      split8(v, w.hi[g][m], w.lo[g][m]);
    }
  }
}
// one utterance tile: 16 consecutive k of h per lane (four f32x4) x the W fragments -> four gate accumulators (scaled by kWScale)
using f32x4_ = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void recur_mfma(const f32x4_ (&h)[4], const WFrags& w, f32x4_ (&acc)[4]) {
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float v[8] = {h[2 * m][0], h[2 * m][1], h[2 * m][2], h[2 * m][3], h[2 * m + 1][0], h[2 * m + 1][1], h[2 * m + 1][2], h[2 * m + 1][3]};
    half8 ahi, alo;
    split8(v, ahi, alo);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, w.lo[g][m], acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, w.hi[g][m], acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, w.hi[g][m], acc[g], 0, 0, 0);
    }
  }
}

// both utterance tiles of a step at once: ONE L2 round trip instead of two on the step's critical path
__device__ __forceinline__ void load32_l2(const float* p, const float* q, f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f,
                                          f32x4& g, f32x4& h) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n"
      "global_load_dwordx4 %1, %8, off offset:16 sc1\n"
      "global_load_dwordx4 %2, %8, off offset:32 sc1\n"
      "global_load_dwordx4 %3, %8, off offset:48 sc1\n"
      "global_load_dwordx4 %4, %9, off sc1\n"
      "global_load_dwordx4 %5, %9, off offset:16 sc1\n"
      "global_load_dwordx4 %6, %9, off offset:32 sc1\n"
      "global_load_dwordx4 %7, %9, off offset:48 sc1\n"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h)
      : "v"(p), "v"(q)
      : "memory");
}
// 64 contiguous bytes per lane, loaded past this CU's L1 (`sc1`: the data was written by other CUs of the same XCD since
// the last step).  The loads AND their wait are one asm statement: the compiler does not count asm loads, so outside
// of it the destination registers could be copied before the data has landed.
__device__ __forceinline__ void load16_l2(const float* p, f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n"
      "global_load_dwordx4 %1, %4, off offset:16 sc1\n"
      "global_load_dwordx4 %2, %4, off offset:32 sc1\n"
      "global_load_dwordx4 %3, %4, off offset:48 sc1\n"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(p)
      : "memory");
}
static_assert(kH / (kBlstmThreads / 64) / 4 == 16, "load16_l2 moves exactly one lane's K slice");
static_assert(kBlstmScratchFloats == 2 * 4 * kMaxB * kH + 8 && kBlstmFlagWord == 2 * 4 * kMaxB * kH + 2, "kernels.h states the scratch layout");

// gx   [rows, 2*4H]  input projections + biases, forward gates then backward gates (i,f,g,o blocks of H each)
// whh  [2][4H][H]
// y    [rows, 2H]    forward h | backward h
// hx   [2 dir][kRing][kMaxB][H]  exchange ring; the host fills buffer 0 with zeros (h_0) and the others with the sentinel
//
// THE EXCHANGE CARRIES ITS OWN SYNCHRONISATION.  Round 2's step ended in a barrier (stores acknowledged -> atomic counter -> spin
// -> load h: four L2 round trips in a row on a chain of 1500 steps).  Here a block publishes its 16 x 32 values of h(t) with plain
// stores into ring buffer (t + 1) % 4 and the readers POLL the data: every cell of a buffer holds the sentinel (+inf; |h| < 1)
// until its value arrives, each lane re-reads its own 64 or 128 bytes until none of them is the sentinel.  One store -> L2 -> load
// hand-off per step.  The ring has FOUR buffers so that the reset needs no wait of its own:
//   step t: poll buffer t % 4 (h(t-1)) -> MFMAs -> block barrier (now every wave's poll has ended: the block as a whole has seen
//   every h(t-1)) -> reset my cells of buffer (t + 3) % 4 to the sentinel (it held h(t-2), which every block finished reading
//   before it published h(t-1)) -> cell update -> publish h(t) into buffer (t + 1) % 4.
//   A reader polls buffer (t + 3) % 4 at step t + 3, after it saw my h(t+1); I issued h(t+1) after my poll of step t + 1, whose
//   `s_waitcnt vmcnt(0)` also waited for the acknowledgement of the reset of step t: the reset is in L2 before any reader can look,
//   so nobody ever takes h(t-2) for h(t+2).
constexpr int kRing = 4;
constexpr unsigned kSentinelBits = 0x7f800000u;          // +inf
constexpr unsigned kSpinLimit = 1u << 16;                // ~0.1 s of polling, then the error flag (the host redoes the call stepwise)

__device__ __forceinline__ float max4(const f32x4& v) { return fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])); }

__global__ __launch_bounds__(kBlstmThreads, 1) void blstm_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                       float* __restrict__ y, float* hx, unsigned* bar,
                                                       const int* __restrict__ off, const int* __restrict__ len, int B,
                                                       int Lmax) {
  // exactly 64 KB; the 16-column block index is XOR-swizzled with the row's lane group so the four lane groups of a wave
  // (rows 4 kq + r) write different banks
  __shared__ float red[kWaves][kMaxB][4 * kUnits];
  const int dir = blockIdx.x % kXcds, blk = blockIdx.x / kXcds;
  if (dir > 1 || blk >= kBlocksPerDir) return;             // the other XCDs' blocks have nothing to do
  const int unit0 = blk * kUnits;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int kbase = 4 * kKL * wave + kKL * kq;            // this lane's kKL consecutive k (same mapping for A and B)
  const unsigned my_xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
  if (blk == 0 && tid == 0) bar[4 + dir] = my_xcc + 1u;        // 0 = never written
  // B fragments (two fp16 planes, resident for the whole sequence): n-tile g = gate g, column n = unit n of this block
  WFrags wf;
  load_w_frags(whh, dir, unit0, n, kbase, wf);
  // cell owner: thread tid -> utterance tid >> 4, unit tid & 15 (32 x 16 = 512 = one cell per thread; rows >= B publish zeros)
  const int cb = tid >> 4, cu = tid & 15;
  const bool own = cb < B;
  const int my_off = own ? off[cb] : 0, my_len = own ? len[cb] : 0;
  float c_state = 0.f, h_state = 0.f;
  float* const hdir = hx + (size_t)dir * kRing * kMaxB * kH;
  const size_t cell = (size_t)cb * kH + unit0 + cu;
  const bool two_tiles = B > 16;
  bool dead = false;                                       // the exchange timed out (here or in another block): stop waiting

  // gate pre-activations of the input projection: those of step t + 1 are requested right after the poll of step t, a whole
  // step before the next `s_waitcnt vmcnt(0)` (HBM latency off the chain)
  auto gx_ptr = [&](int t, bool& live_t, int& frame_t) {
    live_t = t < my_len;
    frame_t = live_t ? my_off + (dir == 0 ? t : my_len - 1 - t) : 0;
    return gx + (size_t)frame_t * (8 * kH) + (size_t)dir * 4 * kH + unit0 + cu;
  };
  float gpre[4], gnext[4] = {0.f, 0.f, 0.f, 0.f};
  bool live, live_n = false;
  int frame, frame_n = 0;
  {
    const float* gp = gx_ptr(0, live, frame);
#pragma unroll
    for (int g = 0; g < 4; ++g) gpre[g] = live ? gp[g * kH] : 0.f;
  }

  for (int t = 0; t < Lmax; ++t) {
    const float* hprev = hdir + (size_t)(t & 3) * kMaxB * kH;
    float* hnext = hdir + (size_t)((t + 1) & 3) * kMaxB * kH;
    float* hreset = hdir + (size_t)((t + 3) & 3) * kMaxB * kH;
    // previous h of utterance rows n and n + 16: polled until every value has arrived
    f32x4 ha[kKL / 4], hb[kKL / 4];
    {
      const float* a0 = hprev + (size_t)n * kH + kbase;
      unsigned spins = 0;
      for (;;) {
        float m;
        if (two_tiles) {
          load32_l2(a0, a0 + 16 * kH, ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]);
          m = fmaxf(fmaxf(fmaxf(max4(ha[0]), max4(ha[1])), fmaxf(max4(ha[2]), max4(ha[3]))),
                    fmaxf(fmaxf(max4(hb[0]), max4(hb[1])), fmaxf(max4(hb[2]), max4(hb[3]))));
        } else {
          load16_l2(a0, ha[0], ha[1], ha[2], ha[3]);
          m = fmaxf(fmaxf(max4(ha[0]), max4(ha[1])), fmaxf(max4(ha[2]), max4(ha[3])));
        }
        if (m <= 1.5f || dead) break;
        if ((++spins & 31u) == 0u) {                       // a lost block must not hang the GPU: flag it, tell the others, go on
          if (__hip_atomic_load(&bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { dead = true; break; }
          if (spins > kSpinLimit) { bar[2] = 1u; dead = true; break; }
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!two_tiles) {
#pragma unroll
        for (int s4 = 0; s4 < kKL / 4; ++s4) hb[s4] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (t + 1 < Lmax) {
      const float* gp = gx_ptr(t + 1, live_n, frame_n);
#pragma unroll
      for (int g = 0; g < 4; ++g) gnext[g] = live_n ? gp[g * kH] : 0.f;
    }
    // every block of a direction must sit on the same XCD, or the exchange above is not coherent (block 0's note was acknowledged
    // before it published h(0), which this block has just seen)
    if (t == 1 && tid == 0 && __hip_atomic_load(&bar[4 + dir], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != my_xcc + 1u) bar[2] = 2u;
    f32x4 acc[2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc[0][g] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    recur_mfma(ha, wf, acc[0]);
    if (two_tiles) recur_mfma(hb, wf, acc[1]);
    // D[i][j]: j = lane & 15 (unit), i = 4 * (lane >> 4) + r (utterance within the tile); column g*16 + unit
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wave][4 * kq + r][((g ^ kq) * kUnits) + n] = acc[0][g][r] * kWScaleInv;
        red[wave][16 + 4 * kq + r][((g ^ kq) * kUnits) + n] = acc[1][g][r] * kWScaleInv;
      }
    __syncthreads();
    // EVERY wave of this block has finished its poll of step t by now, i.e. all 32 blocks have published h(t-1) — and a block
    // publishes h(t-1) only after ITS poll of step t-1, its last read of h(t-2): nobody reads buffer (t + 3) % 4 any more.
    // (Right after this wave's own poll that would not hold: a wave sees the h(t-1) of only the 4 blocks whose units fall in its
    // k slice, while the cells it resets are read by all 32.)
    hreset[cell] = __builtin_bit_cast(float, kSentinelBits);
    if (live) {
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = (g ^ ((cb >> 2) & 3)) * kUnits + cu;
        float a = gpre[g];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) a += red[w][cb][col];
        pre[g] = a;
      }
      c_state = sigmoidf_(pre[1]) * c_state + sigmoidf_(pre[0]) * tanhf(pre[2]);
      h_state = sigmoidf_(pre[3]) * tanhf(c_state);
      y[(size_t)frame * (2 * kH) + (size_t)dir * kH + unit0 + cu] = h_state;
    }
    hnext[cell] = h_state;        // every cell, every step: finished utterances republish their last h, rows >= B a zero (unused)
    __syncthreads();              // `red` is rewritten by the next step
#pragma unroll
    for (int g = 0; g < 4; ++g) gpre[g] = gnext[g];
    live = live_n; frame = frame_n;
  }
}

// ---- fallback: one launch per time step ---------------------------------------------------------------------------------------
// The persistent kernel needs the 32 blocks of a direction co-resident on one XCD; with another model's kernels on the device
// (the 2-pass server runs VAD, punctuation and the streaming model beside this one) that is not guaranteed, and its bounded
// spin then ends in an error flag.  This form has no inter-block exchange inside a launch: step t is one launch of 2 x 32
// blocks, h travels through HBM/L2 between launches (kernel boundaries order it), the cell state lives in `cst`.  Same thread
// mapping and MFMA sequence as above (bit-identical results); W_hh is re-read every step (8 MB from L2 / Infinity Cache).
// ~3 us of launch boundary + ~6 us of work per step: about as fast as the persistent form's 9 us, but 1500 launches per 30-s batch.
__global__ __launch_bounds__(kBlstmThreads, 1) void blstm_step_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                            float* __restrict__ y, float* hx, float* cst,
                                                            const int* __restrict__ off, const int* __restrict__ len, int B, int t) {
  __shared__ float red[kWaves][kMaxB][4 * kUnits];
  const int dir = blockIdx.x & 1, blk = blockIdx.x >> 1;
  const int unit0 = blk * kUnits;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const int kbase = 4 * kKL * wave + kKL * kq;
  WFrags wf;
  load_w_frags(whh, dir, unit0, n, kbase, wf);
  const int cb = tid >> 4, cu = tid & 15;
  const bool own = cb < B;
  const int my_off = own ? off[cb] : 0, my_len = own ? len[cb] : 0;
  const bool live = t < my_len;
  const int frame = live ? my_off + (dir == 0 ? t : my_len - 1 - t) : 0;
  float gpre[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const float* gp = gx + (size_t)frame * (8 * kH) + (size_t)dir * 4 * kH + unit0 + cu;
#pragma unroll
    for (int g = 0; g < 4; ++g) gpre[g] = live ? gp[g * kH] : 0.f;
  }
  float* const hdir = hx + (size_t)dir * 2 * kMaxB * kH;
  const float* hprev = hdir + (size_t)(t & 1) * kMaxB * kH;
  float* hnext = hdir + (size_t)((t & 1) ^ 1) * kMaxB * kH;
  const bool two_tiles = B > 16;
  f32x4 ha[kKL / 4], hb[kKL / 4];
  {
    const float* a0 = hprev + (size_t)n * kH + kbase;
#pragma unroll
    for (int s4 = 0; s4 < kKL / 4; ++s4) {
      ha[s4] = *reinterpret_cast<const f32x4*>(a0 + 4 * s4);
      hb[s4] = two_tiles ? *reinterpret_cast<const f32x4*>(a0 + 16 * kH + 4 * s4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  f32x4 acc[2][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) { acc[0][g] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  recur_mfma(ha, wf, acc[0]);
  if (two_tiles) recur_mfma(hb, wf, acc[1]);
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[wave][4 * kq + r][((g ^ kq) * kUnits) + n] = acc[0][g][r] * kWScaleInv;
      red[wave][16 + 4 * kq + r][((g ^ kq) * kUnits) + n] = acc[1][g][r] * kWScaleInv;
    }
  __syncthreads();
  if (own) {
    float* cp = cst + ((size_t)dir * kMaxB + cb) * kH + unit0 + cu;
    float c_state = t == 0 ? 0.f : *cp;
    float h_state = t == 0 ? 0.f : hprev[(size_t)cb * kH + unit0 + cu];
    if (live) {
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = (g ^ ((cb >> 2) & 3)) * kUnits + cu;
        float a = gpre[g];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) a += red[w][cb][col];
        pre[g] = a;
      }
      c_state = sigmoidf_(pre[1]) * c_state + sigmoidf_(pre[0]) * tanhf(pre[2]);
      h_state = sigmoidf_(pre[3]) * tanhf(c_state);
      y[(size_t)frame * (2 * kH) + (size_t)dir * kH + unit0 + cu] = h_state;
    }
    *cp = c_state;
    hnext[(size_t)cb * kH + unit0 + cu] = h_state;
  }
}

// a2[row] = relu(sigmoid(y[row] . w + b) * smooth - noise); one wave per row
__global__ __launch_bounds__(256) void alpha2_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ w, float b,
                                                     float smooth, float noise, float* __restrict__ a2, int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* yr = y + (size_t)row * ldy;
  float acc = 0.f;
  for (int c = 4 * lane; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(yr + c);
    const float4 ww = *reinterpret_cast<const float4*>(w + c);
    acc += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) a2[row] = fmaxf(sigmoidf_(acc + b) * smooth - noise, 0.f);
}

// per utterance: rescale a2 so that it sums to token_num, then cif_wo_hidden(threshold): peaks[t] = integrate after
// adding alpha[t]; a fire subtracts the threshold.  One block per utterance; the scan is serial from LDS.
__global__ __launch_bounds__(256) void us_cif_kernel(const float* __restrict__ a2, const int* __restrict__ off,
                                                     const int* __restrict__ len, const int* __restrict__ token_num,
                                                     float threshold, float* __restrict__ us_alphas, float* __restrict__ us_peaks) {
  extern __shared__ float sh[];
  __shared__ float part[256];
  const int b = blockIdx.x, L = len[b], o = off[b], tid = threadIdx.x;
  // left-to-right fp32 sum in 256 interleaved lanes then a serial combine: order differs from torch.sum only in rounding
  float s = 0.f;
  for (int t = tid; t < L; t += 256) { const float v = a2[o + t]; sh[t] = v; s += v; }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int i = 0; i < 256; ++i) tot += part[i];
    part[0] = (float)token_num[b] / tot;
  }
  __syncthreads();
  const float scale = part[0];
  for (int t = tid; t < L; t += 256) { const float v = sh[t] * scale; sh[t] = v; us_alphas[o + t] = v; }
  __syncthreads();
  if (tid == 0) {
    float integrate = 0.f;
    for (int t = 0; t < L; ++t) {
      integrate += sh[t];
      us_peaks[o + t] = integrate;
      if (integrate >= threshold) integrate -= threshold;
    }
  }
}

}  // namespace

hipError_t launch_blstm(const float* gx, const float* whh, float* y, float* hx, const int* off, const int* len, int B, int Lmax,
                        hipStream_t s) {
  if (B <= 0 || Lmax <= 0) return hipSuccess;
  if (B > kMaxB) return hipErrorInvalidValue;
  // hx: the exchange ring of both directions (buffer 0 = h_0 = zeros, the others the sentinel), then 8 words of state; the error
  // flag (word 2) survives across launches
  const size_t buf = (size_t)kMaxB * kH;
  hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(hx), (int)kSentinelBits, 2 * kRing * buf, s);
  if (e != hipSuccess) return e;
  for (int d = 0; d < 2; ++d) {
    e = hipMemsetAsync(hx + (size_t)d * kRing * buf, 0, buf * sizeof(float), s);
    if (e != hipSuccess) return e;
  }
  unsigned* bar = reinterpret_cast<unsigned*>(hx + 2 * kRing * buf);
  // one block per (XCD, slot): blocks 8 i + d land on XCD d; only d = 0, 1 work, the rest return at once
  hipLaunchKernelGGL(blstm_kernel, dim3(kXcds * kBlocksPerDir), dim3(kBlstmThreads), 0, s, gx, whh, y, hx, bar, off, len, B, Lmax);
  return hipGetLastError();
}

hipError_t launch_blstm_stepwise(const float* gx, const float* whh, float* y, float* hx, float* cst, const int* off, const int* len, int B,
                                 int Lmax, hipStream_t s) {
  if (B <= 0 || Lmax <= 0) return hipSuccess;
  if (B > kMaxB) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(hx, 0, (size_t)2 * 2 * kMaxB * kH * sizeof(float), s);
  if (e != hipSuccess) return e;
  for (int t = 0; t < Lmax; ++t)
    hipLaunchKernelGGL(blstm_step_kernel, dim3(2 * kBlocksPerDir), dim3(kBlstmThreads), 0, s, gx, whh, y, hx, cst, off, len, B, t);
  return hipGetLastError();
}

void launch_alpha2(const float* y, int ldy, const float* w, float b, float smooth, float noise, float* a2, int rows, int D,
                   hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(alpha2_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, y, ldy, w, b, smooth, noise, a2, rows, D);
}

void launch_us_cif(const float* a2, const int* off, const int* len, const int* token_num, int B, int max_len, float threshold,
                   float* us_alphas, float* us_peaks, hipStream_t s) {
  if (B <= 0 || max_len <= 0) return;
  hipLaunchKernelGGL(us_cif_kernel, dim3(B), dim3(256), (size_t)max_len * sizeof(float), s, a2, off, len, token_num, threshold,
                     us_alphas, us_peaks);
}

}  // namespace pfhip
