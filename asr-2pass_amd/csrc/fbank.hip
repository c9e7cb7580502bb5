// fbank + LFR + CMVN fused, sixteen lanes per 25-ms frame, four frames per wavefront (gfx950).
//
// What it computes, per frame, in the reference's own order of operations:
//   x32768 (onnxruntime/src/paraformer.cpp:312-314), DC removal (knf feature-window.cc:179-190),
//   pre-emphasis 0.97 (:200-211), Hamming (:33-42,58-64), zero-pad to 512, real FFT in fp64
//   (rfft.cc:41-52 does it in double too), cast to fp32 + power spectrum (feature-functions.cc:28-47),
//   80 mel triangles accumulated left-to-right in fp32 (mel-computations.cc:224-247),
//   log(max(e, FLT_EPSILON)) (feature-fbank.cc:102-107); then the LFR gather (7 frames, stride 6,
//   first/last frame replicated) and (x+mean)*istd of Paraformer::LfrCmvn (paraformer.cpp:421-461)
//   written straight into the [T,560] feature matrix — the [F,80] matrix never exists in HBM.
//
// The 512-point real transform is a 256-point complex one (z[n] = y[2n] + i y[2n+1]) done as 16 x 16: every lane runs a 16-point
// transform on registers, one twiddle multiply, ONE exchange through LDS (real parts, then imaginary parts, through the same 2 KB),
// a second 16-point transform, and a second exchange for the conjugate partners of the real-input split — two LDS round trips per
// frame where the radix-2 form of rounds 1-2 (one
// wave per frame, eight stages) made eight, and the window never touches LDS at all: lane l loads the samples 32 n1 + 2 l, + 1
// it will transform, the neighbour of the pre-emphasis comes over DPP (row_shr / row_ror inside the 16-lane row).
//
// Built with -ffp-contract=off so that a*b+c sequences round exactly like the reference's scalar code.
// HBM-bound: 4*S bytes in, 4*560*T out per utterance; the 400-sample windows overlap
// 240/400, the re-reads are served by L2.
#include "kernels.h"

#include <float.h>

namespace pfhip {

namespace {

struct FbankParams {
  const float* pcm;
  const int64_t* sample_off;
  const int* frame_off;
  const int* nframes;
  const int* row_off;
  int B;
  int total_frames;
  FbankTables tb;
  float* feats;     // LFR+CMVN output [M,560] (offline), or null
  float* fb_out;    // raw log-mel frames [total_frames,80] (streaming: ParaformerOnline::FbankKaldi), or null
};

constexpr int kFrameLen = 400, kFrameShift = 160, kMels = 80, kLfrM = 7, kLfrN = 6;
constexpr int kFramesPerBlock = 16;      // 4 waves x 4 frames
constexpr int kMelTaps = 20;             // unrolled taps per triangle (80 bins at 512 points need <= 19; longer rows take the loop)

// Each 16-lane row owns its slice of the LDS array: exchanges between its lanes need no workgroup barrier.  A wave's LDS
// instructions execute in program order, so a ds_write followed by another lane's ds_read is ordered by the hardware; the fence
// only stops the COMPILER from moving LDS accesses across it.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// fused on purpose (the file is built with contraction off for the fp32 chain that mirrors the reference's scalar code; the fp64
// transform has ~30 bits to spare before the cast to fp32)
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(__builtin_fma(a.x, b.x, -(a.y * b.y)), __builtin_fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// forward 4-point transform in place (W4 = -i)
__device__ __forceinline__ void fft4(double2& a0, double2& a1, double2& a2, double2& a3) {
  const double2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
  const double2 t3 = make_double2(d.y, -d.x);
  a0 = cadd(t0, t2); a1 = cadd(t1, t3); a2 = csub(t0, t2); a3 = csub(t1, t3);
}
// forward 16-point transform, X[k] = sum_n x[n] exp(-2 pi i n k / 16), as 4 x 4
__device__ __forceinline__ void fft16(const double2 (&x)[16], double2 (&X)[16]) {
  constexpr double c1 = 0.92387953251128674, s1 = 0.38268343236508977, r = 0.70710678118654752;
  double2 y[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    y[b][0] = x[b]; y[b][1] = x[4 + b]; y[b][2] = x[8 + b]; y[b][3] = x[12 + b];
    fft4(y[b][0], y[b][1], y[b][2], y[b][3]);
  }
  // W16^(b c): (cos, -sin) of pi b c / 8
  y[1][1] = cmul(y[1][1], make_double2(c1, -s1));
  y[1][2] = cmul(y[1][2], make_double2(r, -r));
  y[1][3] = cmul(y[1][3], make_double2(s1, -c1));
  y[2][1] = cmul(y[2][1], make_double2(r, -r));
  y[2][2] = make_double2(y[2][2].y, -y[2][2].x);              // W16^4 = -i
  y[2][3] = cmul(y[2][3], make_double2(-r, -r));
  y[3][1] = cmul(y[3][1], make_double2(s1, -c1));
  y[3][2] = cmul(y[3][2], make_double2(-r, -r));
  y[3][3] = cmul(y[3][3], make_double2(-c1, s1));
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    fft4(y[0][c], y[1][c], y[2][c], y[3][c]);
    X[c] = y[0][c]; X[c + 4] = y[1][c]; X[c + 8] = y[2][c]; X[c + 12] = y[3][c];
  }
}

// lane l of a 16-lane row <- lane l - 1 of `v`; lane 0 <- `first`
__device__ __forceinline__ float row_prev(float v, float first) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false));
}
// lane l <- lane (l - 1) mod 16
__device__ __forceinline__ float row_ror1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));
}

__global__ __launch_bounds__(256) void fbank_lfr_cmvn_kernel(FbankParams p) {
  __shared__ double ex[4][4][256];        // [wave][frame of the wave][256 doubles]: 32 KB, four workgroups per CU.  Exchanges move
                                          // the real parts, then the imaginary parts, through the same 2 KB of a frame.

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane >> 4, l = lane & 15;
  const int g = (blockIdx.x * 4 + wave) * 4 + fr;
  const bool active = g < p.total_frames;
  const int gg = active ? g : p.total_frames - 1;

  // utterance of this frame: last b with frame_off[b] <= gg.  The table is read 64 entries at a time, one per lane, and each row of
  // 16 lanes counts the entries at or below its frame (one load latency instead of a binary search's five dependent ones).
  int b = -1;
  for (int base = 0; base < p.B; base += 64) {
    const int fo = base + lane < p.B ? p.frame_off[base + lane] : 0x7fffffff;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int g4 = __builtin_amdgcn_readlane(gg, 16 * r4);
      const int cnt = __popcll(__ballot(fo <= g4));
      if (fr == r4) b += cnt;
    }
  }
  const int f = gg - p.frame_off[b];
  const int F = p.nframes[b];
  const float* x = p.pcm + p.sample_off[b] + (int64_t)f * kFrameShift;

  // ---- window extraction (samples 32 n1 + 2 l and + 1), x32768, DC removal ------------------------------------------------------
  float ze[13], zo[13];
  float s = 0.f;
#pragma unroll
  for (int n1 = 0; n1 < 13; ++n1) {
    const int i = 32 * n1 + 2 * l;
    const bool ok = i < kFrameLen;                 // i is even and the frame length is even: i + 1 is valid with i
    ze[n1] = ok ? x[i] * 32768.f : 0.f;
    zo[n1] = ok ? x[i + 1] * 32768.f : 0.f;
    s += ze[n1];
    s += zo[n1];
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)kFrameLen;
  // ---- pre-emphasis + window; the 256 complex inputs of the transform ------------------------------------------------------------
  double2 u[16];
  {
    float prev_odd = 0.f;                          // d[32 (n1 - 1) + 2 l + 1]
#pragma unroll
    for (int n1 = 0; n1 < 13; ++n1) {
      const int i = 32 * n1 + 2 * l;
      const bool ok = i < kFrameLen;
      const float de = ze[n1] - mean, dd = zo[n1] - mean;
      // d[i - 1]: the odd sample of lane l - 1; for lane 0 the odd sample of lane 15 one n1 earlier; d[0] for i = 0
      const float first = n1 == 0 ? de : row_ror1(prev_odd);
      const float pe = row_prev(dd, first);
      float ye = de - 0.97f * pe;
      ye = ye * p.tb.window[ok ? i : 0];
      float yo = dd - 0.97f * de;
      yo = yo * p.tb.window[ok ? i + 1 : 0];
      u[n1] = ok ? make_double2((double)ye, (double)yo) : make_double2(0.0, 0.0);
      prev_odd = dd;
    }
    u[13] = u[14] = u[15] = make_double2(0.0, 0.0);
  }

  // ---- 256-point complex transform as 16 x 16: n = 16 n1 + n2 (n2 = l), k = k1 + 16 k2 -------------------------------------------
  const double2* tw = reinterpret_cast<const double2*>(p.tb.tw512);        // tw[k] = exp(-2 pi i k / 512), k < 256
  double* T = ex[wave][fr];
  double2 A[16];
  fft16(u, A);                                     // over n1 -> k1
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) {                // x W256^(n2 k1) = tw[2 j], or -tw[2 j - 256] past the half turn
    const int j = l * k1;
    double2 w = tw[(2 * j) & 255];
    if (j >= 128) w = make_double2(-w.x, -w.y);
    A[k1] = cmul(A[k1], w);
  }
  // row k1, column n2, xor-swizzled: the column reads are conflict-free
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) T[k1 * 16 + (l ^ k1)] = A[k1].x;
  wave_sync();
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) u[n2].x = T[l * 16 + (n2 ^ l)];      // now l = k1
  wave_sync();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) T[k1 * 16 + (l ^ k1)] = A[k1].y;
  wave_sync();
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) u[n2].y = T[l * 16 + (n2 ^ l)];
  fft16(u, A);                                     // over n2 -> k2: A[k2] = Z[l + 16 k2]
  wave_sync();
  // ---- split into the real-input spectrum (conjugate partner Z[256 - k] through LDS), cast to fp32, power ----------------------
  double2 zn[16];
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) T[l + 16 * k2] = A[k2].x;
  wave_sync();
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) zn[k2].x = T[(256 - (l + 16 * k2)) & 255];
  wave_sync();
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) T[l + 16 * k2] = A[k2].y;
  wave_sync();
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) zn[k2].y = T[(256 - (l + 16 * k2)) & 255];
  wave_sync();
  float* ps = reinterpret_cast<float*>(T);
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) {
    const int k = l + 16 * k2;
    const double2 zk = A[k2];
    const double2 e = make_double2(0.5 * (zk.x + zn[k2].x), 0.5 * (zk.y - zn[k2].y));
    const double2 d = make_double2(zk.x - zn[k2].x, zk.y + zn[k2].y);
    const double2 o = make_double2(0.5 * d.y, -0.5 * d.x);
    const double2 xo = cmul(tw[k], o);
    const float re = (float)(e.x + xo.x);
    const float im = (float)(e.y + xo.y);
    ps[k] = re * re + im * im;
  }
  wave_sync();

  // ---- mel + log: lane l takes the bins l, l + 16, ... ----------------------------------------------------------------------------
  // Taps in the reference's left-to-right order (mel-computations.cc:236-241); the weight rows are fetched as vectors up front and
  // the taps unrolled with a predicate, so that no tap waits for a load (a `for (k < size)` loop paid one L1 round trip per tap).
  float melv[5];
#pragma unroll
  for (int u5 = 0; u5 < 5; ++u5) {
    const int bin = l + 16 * u5;
    const int off = p.tb.mel_off[bin];
    const int sz = p.tb.mel_size[bin];
    const float4* w4 = reinterpret_cast<const float4*>(p.tb.mel_w + bin * kMelW);
    // the longest triangle among the 16 bins of this pass (the same in the wave's four rows): taps run in groups of four up to it
    int smax = sz;
#pragma unroll
    for (int o2 = 8; o2 >= 1; o2 >>= 1) smax = max(smax, __shfl_xor(smax, o2));
    smax = __builtin_amdgcn_readfirstlane(smax);
    float e = 0.f;
#pragma unroll
    for (int q = 0; q < kMelTaps / 4; ++q) {
      if (4 * q < smax) {
        const float4 t4 = w4[q];
        const float wv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float t = e + wv[k] * ps[min(off + 4 * q + k, 255)];
          e = 4 * q + k < sz ? t : e;
        }
      }
    }
    for (int k = kMelTaps; k < sz; ++k) e += p.tb.mel_w[bin * kMelW + k] * ps[off + k];      // not reached at 80 bins / 512 points
    melv[u5] = logf(fmaxf(e, FLT_EPSILON));
  }
  if (!active) return;
  if (p.fb_out) {
#pragma unroll
    for (int u5 = 0; u5 < 5; ++u5) p.fb_out[(size_t)g * kMels + l + 16 * u5] = melv[u5];
    return;
  }

  // ---- LFR gather + CMVN ----------------------------------------------------------------------------------------------------------
  const int T_lfr = (F + kLfrN - 1) / kLfrN;
  float* out = p.feats + (size_t)p.row_off[b] * (kLfrM * kMels);
  auto emit = [&](int t, int j) {
#pragma unroll
    for (int u5 = 0; u5 < 5; ++u5) {
      const int c = j * kMels + l + 16 * u5;
      out[(size_t)t * (kLfrM * kMels) + c] = (melv[u5] + p.tb.cmvn_mean[c]) * p.tb.cmvn_istd[c];
    }
  };
  const int pi = f + (kLfrM - 1) / 2;     // index in the left-padded sequence
  const int t1 = pi / kLfrN, j1 = pi % kLfrN;
  if (t1 < T_lfr) emit(t1, j1);
  if (j1 == 0 && t1 >= 1) emit(t1 - 1, kLfrN);
  if (f == 0) {
    for (int j = 0; j < (kLfrM - 1) / 2; ++j) emit(0, j);
  }
  if (f == F - 1) {
    for (int j = 0; j < kLfrM; ++j)
      if (kLfrN * (T_lfr - 1) + j > F + 2) emit(T_lfr - 1, j);
  }
}

}  // namespace

void launch_fbank_lfr_cmvn(const float* pcm, const int64_t* sample_off, const int* frame_off,
                           const int* nframes, const int* row_off, int B, int total_frames,
                           FbankTables tb, float* feats, hipStream_t s) {
  if (total_frames <= 0) return;
  FbankParams p{pcm, sample_off, frame_off, nframes, row_off, B, total_frames, tb, feats, nullptr};
  const int blocks = (total_frames + kFramesPerBlock - 1) / kFramesPerBlock;
  hipLaunchKernelGGL(fbank_lfr_cmvn_kernel, dim3(blocks), dim3(256), 0, s, p);
}

void launch_fbank_frames(const float* pcm, const int64_t* sample_off, const int* frame_off, const int* nframes,
                         int total_frames, FbankTables tb, float* fb_out, hipStream_t s) {
  if (total_frames <= 0) return;
  FbankParams p{pcm, sample_off, frame_off, nframes, nullptr, 1, total_frames, tb, nullptr, fb_out};
  hipLaunchKernelGGL(fbank_lfr_cmvn_kernel, dim3((total_frames + kFramesPerBlock - 1) / kFramesPerBlock), dim3(256), 0, s, p);
}

void launch_fbank_frames_batch(const float* pcm, const int64_t* sample_off, const int* frame_off, const int* nframes, int B,
                               int total_frames, FbankTables tb, float* fb_out, hipStream_t s) {
  if (total_frames <= 0) return;
  FbankParams p{pcm, sample_off, frame_off, nframes, nullptr, B, total_frames, tb, nullptr, fb_out};
  hipLaunchKernelGGL(fbank_lfr_cmvn_kernel, dim3((total_frames + kFramesPerBlock - 1) / kFramesPerBlock), dim3(256), 0, s, p);
}

// ---- embed: x*sqrt(d_model) + sinusoidal PE (paraformer-online.cpp:549-555, 240-268) ----------
namespace {
__global__ __launch_bounds__(192) void embed_kernel(const float* feats, int D, float* x0, int ldx,
                                                    const int* row_pos, int M,
                                                    const float* inv_ts, float scale) {
  const int row = blockIdx.x;
  if (row >= M) return;
  const float pos = (float)(row_pos[row] + 1);
  const int half = D >> 1;
  for (int c = threadIdx.x; c < ldx; c += blockDim.x) {
    float o = 0.f;
    if (c < D) {
      const int i = c < half ? c : c - half;
      const float coe = inv_ts[i] * pos;
      const float pe = c < half ? sinf(coe) : cosf(coe);
      o = feats[(size_t)row * D + c] * scale + pe;
    }
    x0[(size_t)row * ldx + c] = o;
  }
}
}  // namespace

void launch_embed(const float* feats, int D, float* x0, int ldx, const int* row_pos, int M,
                  const float* inv_timescale, float scale, hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL(embed_kernel, dim3(M), dim3(192), 0, s, feats, D, x0, ldx, row_pos, M,
                     inv_timescale, scale);
}

}  // namespace pfhip
