// fbank + LFR + CMVN fused, one wavefront per 25-ms frame (gfx950).
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

constexpr int kFrameLen = 400, kFrameShift = 160, kNfft = 512, kMels = 80, kLfrM = 7, kLfrN = 6;

// Each wave owns its slices of the LDS arrays (xs / za / zb / ps [wave]): exchanges between its lanes need no workgroup
// barrier.  A wave's LDS instructions execute in program order, so a ds_write followed by another lane's ds_read is ordered by
// the hardware; the fence only stops the COMPILER from moving LDS accesses across it.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ __launch_bounds__(256) void fbank_lfr_cmvn_kernel(FbankParams p) {
  __shared__ float xs[4][kNfft];
  __shared__ double2 za[4][256];
  __shared__ double2 zb[4][256];
  __shared__ float ps[4][264];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + wave;
  const bool active = g < p.total_frames;
  const int gg = active ? g : p.total_frames - 1;

  // utterance of this frame: last b with frame_off[b] <= gg (wave-uniform search)
  int lo = 0, hi = p.B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (p.frame_off[mid] <= gg) lo = mid; else hi = mid;
  }
  const int b = lo;
  const int f = gg - p.frame_off[b];
  const int F = p.nframes[b];
  const float* x = p.pcm + p.sample_off[b] + (int64_t)f * kFrameShift;

  // ---- window extraction, x32768, DC removal ---------------------------------------------------
  float v[7];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int i = lane + 64 * j;
    v[j] = (i < kFrameLen) ? x[i] * 32768.f : 0.f;
    s += v[j];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)kFrameLen;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int i = lane + 64 * j;
    if (i < kFrameLen) xs[wave][i] = v[j] - mean;
  }
  wave_sync();
  // ---- pre-emphasis + window ---------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int i = lane + 64 * j;
    float y = 0.f;
    if (i < kFrameLen) {
      const float cur = xs[wave][i];
      const float prev = xs[wave][i > 0 ? i - 1 : 0];
      y = cur - 0.97f * prev;
      y = y * p.tb.window[i];
    }
    v[j] = y;
  }
  wave_sync();
#pragma unroll
  for (int j = 0; j < 7; ++j) xs[wave][lane + 64 * j] = v[j];
  xs[wave][448 + lane] = 0.f;
  wave_sync();

  // ---- 512-pt real FFT as a 256-pt complex Stockham FFT in fp64 -------------------------------------
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = lane + 64 * q;
    za[wave][n] = make_double2((double)xs[wave][2 * n], (double)xs[wave][2 * n + 1]);
  }
  wave_sync();
  double2* X = za[wave];
  double2* Y = zb[wave];
  const double2* tw = reinterpret_cast<const double2*>(p.tb.tw512);
#pragma unroll
  for (int stage = 0; stage < 8; ++stage) {
    const int sstr = 1 << stage;          // stride
    const int m = 128 >> stage;           // half of the current sub-transform length
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = lane + 64 * u;
      const int pidx = t >> stage;
      const int q = t & (sstr - 1);
      const double2 a = X[q + sstr * pidx];
      const double2 c = X[q + sstr * (pidx + m)];
      const double2 w = tw[pidx << (stage + 1)];
      Y[q + sstr * (2 * pidx)] = make_double2(a.x + c.x, a.y + c.y);
      Y[q + sstr * (2 * pidx + 1)] = cmul(make_double2(a.x - c.x, a.y - c.y), w);
    }
    wave_sync();
    double2* tmp = X; X = Y; Y = tmp;
  }
  // ---- split into the real-input spectrum, cast to fp32, power ----------------------------------
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = lane + 64 * q;
    const double2 zk = X[k];
    const double2 zn = X[(256 - k) & 255];
    const double2 e = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
    const double2 d = make_double2(zk.x - zn.x, zk.y + zn.y);
    const double2 o = make_double2(0.5 * d.y, -0.5 * d.x);
    const double2 xo = cmul(tw[k], o);
    const float re = (float)(e.x + xo.x);
    const float im = (float)(e.y + xo.y);
    ps[wave][k] = re * re + im * im;
  }
  wave_sync();

  // ---- mel + log -----------------------------------------------------------------------------
  float melv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int bin = lane + 64 * u;
    float val = 0.f;
    if (bin < kMels) {
      const int off = p.tb.mel_off[bin];
      const int sz = p.tb.mel_size[bin];
      const float* w = p.tb.mel_w + bin * kMelW;
      float e = 0.f;
      for (int k = 0; k < sz; ++k) e += w[k] * ps[wave][off + k];
      val = logf(fmaxf(e, FLT_EPSILON));
    }
    melv[u] = val;
  }
  if (!active) return;
  if (p.fb_out) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int bin = lane + 64 * u;
      if (bin < kMels) p.fb_out[(size_t)g * kMels + bin] = melv[u];
    }
    return;
  }

  // ---- LFR gather + CMVN -----------------------------------------------------------------------
  const int T = (F + kLfrN - 1) / kLfrN;
  float* out = p.feats + (size_t)p.row_off[b] * (kLfrM * kMels);
  auto emit = [&](int t, int j) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int bin = lane + 64 * u;
      if (bin < kMels) {
        const int c = j * kMels + bin;
        out[(size_t)t * (kLfrM * kMels) + c] = (melv[u] + p.tb.cmvn_mean[c]) * p.tb.cmvn_istd[c];
      }
    }
  };
  const int pi = f + (kLfrM - 1) / 2;     // index in the left-padded sequence
  const int t1 = pi / kLfrN, j1 = pi % kLfrN;
  if (t1 < T) emit(t1, j1);
  if (j1 == 0 && t1 >= 1) emit(t1 - 1, kLfrN);
  if (f == 0) {
    for (int j = 0; j < (kLfrM - 1) / 2; ++j) emit(0, j);
  }
  if (f == F - 1) {
    for (int j = 0; j < kLfrM; ++j)
      if (kLfrN * (T - 1) + j > F + 2) emit(T - 1, j);
  }
}

}  // namespace

void launch_fbank_lfr_cmvn(const float* pcm, const int64_t* sample_off, const int* frame_off,
                           const int* nframes, const int* row_off, int B, int total_frames,
                           FbankTables tb, float* feats, hipStream_t s) {
  if (total_frames <= 0) return;
  FbankParams p{pcm, sample_off, frame_off, nframes, row_off, B, total_frames, tb, feats, nullptr};
  const int blocks = (total_frames + 3) / 4;
  hipLaunchKernelGGL(fbank_lfr_cmvn_kernel, dim3(blocks), dim3(256), 0, s, p);
}

void launch_fbank_frames(const float* pcm, const int64_t* sample_off, const int* frame_off, const int* nframes,
                         int total_frames, FbankTables tb, float* fb_out, hipStream_t s) {
  if (total_frames <= 0) return;
  FbankParams p{pcm, sample_off, frame_off, nframes, nullptr, 1, total_frames, tb, nullptr, fb_out};
  hipLaunchKernelGGL(fbank_lfr_cmvn_kernel, dim3((total_frames + 3) / 4), dim3(256), 0, s, p);
}

void launch_fbank_frames_batch(const float* pcm, const int64_t* sample_off, const int* frame_off, const int* nframes, int B,
                               int total_frames, FbankTables tb, float* fb_out, hipStream_t s) {
  if (total_frames <= 0) return;
  FbankParams p{pcm, sample_off, frame_off, nframes, nullptr, B, total_frames, tb, nullptr, fb_out};
  hipLaunchKernelGGL(fbank_lfr_cmvn_kernel, dim3((total_frames + 3) / 4), dim3(256), 0, s, p);
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
