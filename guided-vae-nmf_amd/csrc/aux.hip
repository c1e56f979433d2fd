// Small kernels around the hot loop: W multiplicative update, cost reduction,
// dense layers (encoder / classifier), |X|^2, STFT / iSTFT, SI-SDR Gram sums.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned u32x4p;

namespace {

// ----------------------------------------------------------------------------
// W update (mcem.py:107-110) + L1 column normalisation (mcem.py:129-133), two stages:
//   w_partial_kernel: num[f,k] = sum_n P[n,f] H[k,n], den[f,k] = sum_n A1[n,f] H[k,n] over one of
//                     NCH frame chunks of an utterance (P = X2 * sum_r Vx^-2, A1 = sum_r Vx^-1 come from
//                     decode_kernel<MODE_WSTATS>); grid (NCH, U), one thread per bin
//   w_update_kernel:  fixed-order sum of the NCH partials, W <- W sqrt(num/den), column norms
// ----------------------------------------------------------------------------
constexpr int W_NCH = 8;

// Ranks are independent in the update (the column norms are per rank): w_update_kernel splits the padded ranks above 8
// over blockIdx.y in groups of KG = 8 -- one workgroup per utterance before, 64 workgroups on 256 CUs; rank 32: 56 -> 21 us.
constexpr int W_KG = 8;

template <int KP>
__global__ __launch_bounds__(640) void w_partial_kernel(const float* __restrict__ A1, const float* __restrict__ P,
                                                        const float* __restrict__ Ht, float* __restrict__ part,
                                                        const int32_t* __restrict__ frame_off, int Fs) {
  // (not split over rank groups: every group would read the A1 / P rows again -- measured 64 -> 71 us at rank 32)
  const int u = blockIdx.y, ch = blockIdx.x, f = threadIdx.x;
  const int nb = frame_off[u], ne = frame_off[u + 1];
  const int per = (ne - nb + W_NCH - 1) / W_NCH;
  const int n0 = nb + ch * per, n1 = (n0 + per < ne) ? n0 + per : ne;
  float num[KP], den[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) num[k] = den[k] = 0.f;
  if (f < Fs) {
#pragma unroll 8
    for (int n = n0; n < n1; ++n) {
      const float pv = P[(size_t)n * Fs + f], av = A1[(size_t)n * Fs + f];
#pragma unroll
      for (int k = 0; k < KP; k += 4) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(Ht + (size_t)n * KP + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) { num[k + t] += pv * h[t]; den[k + t] += av * h[t]; }
      }
    }
    float* dst = part + (((size_t)u * W_NCH + ch) * Fs + f) * 2 * KP;
#pragma unroll
    for (int k = 0; k < KP; k += 4) {
      *reinterpret_cast<f32x4*>(dst + k) = f32x4{num[k], num[k + 1], num[k + 2], num[k + 3]};
      *reinterpret_cast<f32x4*>(dst + KP + k) = f32x4{den[k], den[k + 1], den[k + 2], den[k + 3]};
    }
  }
}

template <int KP>
__global__ __launch_bounds__(640) void w_update_kernel(const float* __restrict__ part, float* __restrict__ W,
                                                       float* __restrict__ normW, int F, int Fs, int K) {
  __shared__ float colred[10][W_KG];
  const int u = blockIdx.x, f = threadIdx.x, k0 = blockIdx.y * W_KG;
  float wn[W_KG];
#pragma unroll
  for (int k = 0; k < W_KG; ++k) wn[k] = 0.f;
  if (f < F) {
    float num[W_KG], den[W_KG];
#pragma unroll
    for (int k = 0; k < W_KG; ++k) num[k] = den[k] = 0.f;
    for (int ch = 0; ch < W_NCH; ++ch) {                 // fixed order: deterministic
      const float* src = part + (((size_t)u * W_NCH + ch) * Fs + f) * 2 * KP;
#pragma unroll
      for (int k = 0; k < W_KG; k += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + k0 + k), b = *reinterpret_cast<const f32x4*>(src + KP + k0 + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) { num[k + t] += a[t]; den[k + t] += b[t]; }
      }
    }
#pragma unroll
    for (int k = 0; k < W_KG; ++k)
      if (k0 + k < K) wn[k] = W[((size_t)u * Fs + f) * KP + k0 + k] * sqrtf(num[k] / den[k]);          // mcem.py:110
  }
  // column L1 norms (mcem.py:129): DPP/permlane wave sum, then across waves in fixed order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < W_KG; ++k) {
    const float v = sum_rows4(sum_row16(fabsf(wn[k])));
    if (lane == 0) colred[wv][k] = v;
  }
  __syncthreads();
  float nrm[W_KG];
#pragma unroll
  for (int k = 0; k < W_KG; ++k) {
    float s = 0.f;
    for (int ww = 0; ww < nwv; ++ww) s += colred[ww][k];
    nrm[k] = s;
  }
  if (f < Fs) {
#pragma unroll
    for (int k = 0; k < W_KG; ++k)
      W[((size_t)u * Fs + f) * KP + k0 + k] = (k0 + k < K && f < F) ? wn[k] / nrm[k] : 0.f;              // mcem.py:131
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < W_KG; ++k) normW[(size_t)u * KP + k0 + k] = (k0 + k < K) ? nrm[k] : 0.f;         // applied to H (mcem.py:133)
  }
}

// Small batches: wstats_group_kernel (stream.hip) leaves one block of sums per 16-frame GROUP (the groups of utterance u
// follow those of the utterances before it, ceil(N / 16) each).  A 64-frame tile's partial is rebuilt here from its (up to)
// four groups in the order wstats_fused_kernel adds its four wavefronts -- ((g0 + g1) + g2) + g3 -- into the tile layout
// w_update_tiles_kernel reads: the same bits as the tile kernel's.  Block (tile, slot), one thread per bin.
__global__ void w_combine_groups_kernel(const float* __restrict__ part16, const int32_t* __restrict__ tile_g0, const int32_t* __restrict__ tile_cnt,
                                        int F, int Fs, int slots, float* __restrict__ part64) {
  const int T = blockIdx.x, slot = blockIdx.y, f = threadIdx.x;
  if (f >= F) return;
  const int g0 = tile_g0[T], nj = (tile_cnt[T] + 15) / 16;      // the tile's first group and its 1..4 groups (tables of vaenmf_bind_batch)
  const float* src = part16 + ((size_t)g0 * slots + slot) * Fs + f;
  float v = src[0];
  for (int j = 1; j < nj; ++j) v += src[(size_t)j * slots * Fs];
  part64[((size_t)T * slots + slot) * Fs + f] = v;
}

// The same update from the per-tile partial sums of wstats_fused_kernel (stream.hip): part [tile][2 KP][Fs], the tiles of
// utterance u are tile_first[u] .. tile_first[u+1]-1, added in that order; a tile's block is slot-major: [2 k + stat][Fs].
template <int KP>
__global__ __launch_bounds__(640) void w_update_tiles_kernel(const float* __restrict__ part, const int32_t* __restrict__ tile_first,
                                                             float* __restrict__ W, float* __restrict__ normW, int F, int Fs, int K) {
  __shared__ float colred[10][KP];
  const int u = blockIdx.x, f = threadIdx.x;
  float wn[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) wn[k] = 0.f;
  if (f < F) {
    float num[KP], den[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) num[k] = den[k] = 0.f;
    const int t0 = tile_first[u], t1 = tile_first[u + 1];
    for (int t = t0; t < t1; ++t) {                       // fixed order: deterministic
      const float* src = part + (size_t)t * 2 * KP * Fs + f;        // [slot = 2 k + stat][Fs]: coalesced over the bins
#pragma unroll
      for (int k = 0; k < KP; ++k) { num[k] += src[(size_t)(2 * k) * Fs]; den[k] += src[(size_t)(2 * k + 1) * Fs]; }
    }
#pragma unroll
    for (int k = 0; k < KP; ++k)
      if (k < K) wn[k] = W[((size_t)u * Fs + f) * KP + k] * sqrtf(num[k] / den[k]);          // mcem.py:110
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const float v = sum_rows4(sum_row16(fabsf(wn[k])));
    if (lane == 0) colred[wv][k] = v;
  }
  __syncthreads();
  float nrm[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    float s = 0.f;
    for (int ww = 0; ww < nwv; ++ww) s += colred[ww][k];
    nrm[k] = s;
  }
  if (f < Fs) {
#pragma unroll
    for (int k = 0; k < KP; ++k)
      W[((size_t)u * Fs + f) * KP + k] = (k < K && f < F) ? wn[k] / nrm[k] : 0.f;              // mcem.py:131
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < KP; ++k) normW[(size_t)u * KP + k] = (k < K) ? nrm[k] : 0.f;         // applied to H (mcem.py:133)
  }
}

// cost[u][it] = mean_{r,f,n}(log Vx + X2/Vx)  (mcem.py:70) from per-frame sums; block (u, j): iteration it0 + j, whose
// per-frame sums are row j of cost_frames [n_it][stride]
__global__ void cost_reduce_kernel(const double* __restrict__ cost_frames, size_t stride, const int32_t* __restrict__ frame_off,
                                   int R, int F, double* __restrict__ cost, int niter, int it0) {
  __shared__ double red[256];
  const int u = blockIdx.x, j = blockIdx.y;
  const int nb = frame_off[u], ne = frame_off[u + 1];
  const double* cf = cost_frames + (size_t)j * stride;
  double s = 0.0;
  for (int n = nb + threadIdx.x; n < ne; n += blockDim.x) s += cf[n];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) cost[(size_t)u * niter + it0 + j] = red[0] / ((double)R * F * (ne - nb));
}

// ----------------------------------------------------------------------------
// Dense layer Y = act(X Wt^T + b), fp32 FMA, 64x64 output tile, K-step 16.
// (encoder / classifier: once per utterance, models.py:101-104, 57-62)
// ----------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case VAENMF_ACT_TANH: return tanhf(v);
    case VAENMF_ACT_RELU: return fmaxf(v, 0.f);
    case VAENMF_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case VAENMF_ACT_STEP: return v > 0.f ? 1.f : 0.f;
    default: return v;
  }
}
__global__ __launch_bounds__(256) void dense_kernel(const float* __restrict__ X, int M, int in, int ldx,
                                                    const float* __restrict__ Wt, const float* __restrict__ b, int out,
                                                    int act, float* __restrict__ Y, int ldy) {
  __shared__ float xs[16][65], ws[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, o0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < in; k0 += 16) {
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      const int r = e >> 4, kk = e & 15;
      xs[kk][r] = (m0 + r < M && k0 + kk < in) ? X[(size_t)(m0 + r) * ldx + k0 + kk] : 0.f;
      ws[kk][r] = (o0 + r < out && k0 + kk < in) ? Wt[(size_t)(o0 + r) * in + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float xv[4], wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { xv[i] = xs[kk][ty * 4 + i]; wv[i] = ws[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += xv[i] * wv[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, o = o0 + tx * 4 + j;
      if (m < M && o < out) Y[(size_t)m * ldy + o] = apply_act(acc[i][j] + (b ? b[o] : 0.f), act);
    }
}

__global__ void power_spec_kernel(const float2* __restrict__ X, float* __restrict__ X2, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float2 v = X[i]; X2[i] = v.x * v.x + v.y * v.y; }
}

// EM.init_parameters on the device (mcem.py:42-44, :51): W = max(U(0,1), eps) over (F, K), H = max(U(0,1), eps) over
// (K, N), g = 1, padding rows / ranks zero.  Counter-based: the draw of element (utterance, f, k) / (frame inside the
// utterance, k) is splitmix64(utterance seed ^ salt ^ tag ^ index), so an utterance's initialisation does not depend on the
// batch it sits in.  One thread per element of W [U][Fs][Kp] followed by Ht [NT][Kp] and g [NT].
__global__ void nmf_init_kernel(float* __restrict__ W, float* __restrict__ Ht, float* __restrict__ g,
                                const uint64_t* __restrict__ utt_seed, const int32_t* __restrict__ frame_utt,
                                const int32_t* __restrict__ frame_loc, int n_utt, int NT, int F, int Fs, int K, int Kp,
                                uint64_t salt, float eps) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nW = (size_t)n_utt * Fs * Kp, nH = (size_t)NT * Kp;
  auto u01 = [](uint64_t key) {
    uint64_t x = key;
    const uint64_t r = splitmix64(x);
    return (float)(r >> 40) * 5.9604644775390625e-8f;      // 24 bits, [0, 1)
  };
  if (i < nW) {
    const int k = (int)(i % Kp), f = (int)((i / Kp) % Fs), u = (int)(i / ((size_t)Kp * Fs));
    float v = 0.f;
    if (f < F && k < K) v = fmaxf(u01(utt_seed[u] ^ salt ^ (0x57ull << 56) ^ (uint64_t)(f * K + k)), eps);
    W[i] = v;
  } else if (i < nW + nH) {
    const size_t j = i - nW;
    const int k = (int)(j % Kp), n = (int)(j / Kp);
    float v = 0.f;
    if (k < K) v = fmaxf(u01(utt_seed[frame_utt[n]] ^ salt ^ (0x48ull << 56) ^ ((uint64_t)frame_loc[n] * (uint64_t)K + (uint64_t)k)), eps);
    Ht[j] = v;
  } else if (i < nW + nH + (size_t)NT) {
    g[i - nW - nH] = 1.f;
  }
}

// ----------------------------------------------------------------------------
// STFT / iSTFT (python/processing/stft.py -> librosa): radix-2 FFT in LDS, fp64.
// ----------------------------------------------------------------------------
__device__ __forceinline__ int bitrev(int x, int bits) { return (int)(__brev((unsigned)x) >> (32 - bits)); }

// in-place complex FFT of length n (power of two) on LDS arrays; sign = -1 forward, +1 inverse
__device__ void fft_lds(double* re, double* im, const double* twr, const double* twi, int n, int bits, int sign) {
  for (int len = 2, st = n >> 1, lh = 0; len <= n; len <<= 1, st >>= 1, ++lh) {
    const int half = len >> 1;                     // = 1 << lh
    for (int b = threadIdx.x; b < (n >> 1); b += blockDim.x) {
      const int grp = b >> lh, pos = b & (half - 1);
      const int i0 = grp * len + pos, i1 = i0 + half;
      const double wr = twr[pos * st], wi = sign * twi[pos * st];
      const double xr = re[i1] * wr - im[i1] * wi, xi = re[i1] * wi + im[i1] * wr;
      re[i1] = re[i0] - xr; im[i1] = im[i0] - xi;
      re[i0] += xr;         im[i0] += xi;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ wav, const int64_t* __restrict__ samp_off,
                                                   const int32_t* __restrict__ frame_off, const int32_t* __restrict__ frame_utt,
                                                   const int32_t* __restrict__ pad_len, int nfft, int bits, int hop, int Fs,
                                                   float2* __restrict__ X) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* re = reinterpret_cast<double*>(smem);
  double* im = re + nfft;
  double* twr = im + nfft;
  double* twi = twr + nfft / 2;
  const int n = blockIdx.x, u = frame_utt[n], i = n - frame_off[u];
  const int64_t off = samp_off[u];
  const int64_t T = samp_off[u + 1] - off;
  const int64_t Tp = pad_len[u];                 // length after the end-pad rule (stft.py:48-53)
  for (int t = threadIdx.x; t < nfft / 2; t += blockDim.x) {
    double s, c;
    sincospi(-2.0 * t / nfft, &s, &c);           // exp(-2 pi i t / nfft)
    twr[t] = c; twi[t] = s;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nfft; t += blockDim.x) {
    int64_t p = (int64_t)i * hop + t - nfft / 2;  // centre=True, reflect padding
    if (p < 0) p = -p;
    if (p >= Tp) p = 2 * (Tp - 1) - p;
    const double v = (p >= 0 && p < T) ? (double)wav[off + p] : 0.0;
    // cos(2 pi t / nfft) from the twiddle table (two thirds of this kernel's time went into a second fp64 sincospi per sample)
    const double cw = t < nfft / 2 ? twr[t] : -twr[t - nfft / 2];
    const int r = bitrev(t, bits);
    re[r] = v * (0.5 - 0.5 * cw);                 // periodic Hann
    im[r] = 0.0;
  }
  __syncthreads();
  fft_lds(re, im, twr, twi, nfft, bits, 1);       // table holds exp(-i..): sign +1 keeps it
  const int F = nfft / 2 + 1;
  for (int f = threadIdx.x; f < Fs; f += blockDim.x)
    X[(size_t)n * Fs + f] = f < F ? make_float2((float)re[f], (float)im[f]) : make_float2(0.f, 0.f);
}

__global__ __launch_bounds__(256) void istft_frames_kernel(const float2* __restrict__ S, int nfft, int bits, int Fs,
                                                           float* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* re = reinterpret_cast<double*>(smem);
  double* im = re + nfft;
  double* twr = im + nfft;
  double* twi = twr + nfft / 2;
  const int n = blockIdx.x;
  for (int t = threadIdx.x; t < nfft / 2; t += blockDim.x) {
    double s, c;
    sincospi(2.0 * t / nfft, &s, &c);             // exp(+2 pi i t / nfft)
    twr[t] = c; twi[t] = s;
  }
  const int half = nfft / 2;
  for (int k = threadIdx.x; k < nfft; k += blockDim.x) {
    const int kk = k <= half ? k : nfft - k;
    const float2 v = S[(size_t)n * Fs + kk];
    double vr = v.x, vi = (k <= half) ? v.y : -v.y;
    if (k == 0 || k == half) vi = 0.0;            // c2r ignores the imaginary part of DC / Nyquist
    const int r = bitrev(k, bits);
    re[r] = vr; im[r] = vi;
  }
  __syncthreads();
  fft_lds(re, im, twr, twi, nfft, bits, 1);
  for (int t = threadIdx.x; t < nfft; t += blockDim.x) {
    const double cw = t < nfft / 2 ? twr[t] : -twr[t - nfft / 2];     // cos(2 pi t / nfft), see stft_kernel
    work[(size_t)n * nfft + t] = (float)(re[t] / nfft * (0.5 - 0.5 * cw));
  }
}

__global__ void istft_ola_kernel(const float* __restrict__ work, const int64_t* __restrict__ samp_off,
                                 const int32_t* __restrict__ frame_off, int n_utt, int nfft, int hop,
                                 float* __restrict__ out) {
  const int u = blockIdx.y;
  const int64_t off = samp_off[u], T = samp_off[u + 1] - off;
  const int nb = frame_off[u], nfr = frame_off[u + 1] - nb;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = t + nfft / 2;
    float y = 0.f;
    double wss = 0.0;
    if (p < (int64_t)nfft + (int64_t)hop * (nfr - 1)) {
      int64_t i_lo = (p - nfft + hop) / hop;        // ceil((p - nfft + 1)/hop)
      if (p - nfft + 1 <= 0) i_lo = 0;
      int64_t i_hi = p / hop;
      if (i_hi > nfr - 1) i_hi = nfr - 1;
      for (int64_t i = i_lo; i <= i_hi; ++i) {
        const int tt = (int)(p - i * hop);
        double sw, cw;
        sincospi(2.0 * tt / nfft, &sw, &cw);
        const double wv = 0.5 - 0.5 * cw;
        y += work[(size_t)(nb + i) * nfft + tt];
        wss += wv * wv;
      }
      if (wss > 1.1754943508222875e-38) y = (float)(y / wss);
    }
    out[off + t] = y;
  }
}

// Gram matrix of (s_hat, s, n) per utterance in float64 (python/metrics.py:12-60)
__global__ __launch_bounds__(256) void gram3_kernel(const float* __restrict__ sh, const float* __restrict__ s,
                                                    const float* __restrict__ nz, const int64_t* __restrict__ samp_off,
                                                    double* __restrict__ out) {
  __shared__ double red[6][256];
  const int u = blockIdx.x;
  const int64_t b = samp_off[u], e = samp_off[u + 1];
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    const double x = sh[i], y = s[i], z = nz[i];
    a[0] += x * x; a[1] += x * y; a[2] += x * z; a[3] += y * y; a[4] += y * z; a[5] += z * z;
  }
  for (int j = 0; j < 6; ++j) red[j][threadIdx.x] = a[j];
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m)
      for (int j = 0; j < 6; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x < 6) out[(size_t)u * 6 + threadIdx.x] = red[threadIdx.x][0];
}


// HBM read probe (bench.py's measured ceiling for the streaming M-step kernels): every lane keeps UNR 16-byte loads in
// flight, a workgroup sweeps contiguous 4 KB pieces in a grid-stride loop, the grid is the resident set (8 workgroups of
// 256 threads per CU = 8 wavefronts per SIMD).  The loaded words are folded into one value per lane that is stored only
// if it equals a sentinel the data never produces, so nothing is written and nothing is optimised away.
template <int UNR>
__global__ __launch_bounds__(256) void hbm_read_probe_kernel(const u32x4p* __restrict__ buf, size_t n16, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * 256 * UNR;
  unsigned acc = 0;
  size_t i = (size_t)blockIdx.x * 256 * UNR + threadIdx.x;
  for (; i + (size_t)(UNR - 1) * 256 < n16; i += stride) {
    u32x4p v[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) v[k] = __builtin_nontemporal_load(buf + i + (size_t)k * 256);
#pragma unroll
    for (int k = 0; k < UNR; ++k) acc ^= v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
  }
  for (; i < n16; i += 256) { const u32x4p v = buf[i]; acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (acc == 0x9E3779B9u) sink[0] = acc;
}

int ilog2(int n) { int b = 0; while ((1 << b) < n) ++b; return b; }

}  // namespace

int vn_launch_w_update(const vaenmf_plan* p, float* W, const float* Ht, hipStream_t st) {
  const int threads = ((p->Fs + 63) / 64) * 64;        // one thread per bin (Fs <= 640)
#define VN_WU(KP)                                                                                               \
  do {                                                                                                          \
    hipLaunchKernelGGL((w_partial_kernel<KP>), dim3(W_NCH, p->n_utt), dim3(threads), 0, st, p->A1, p->P, Ht, \
                       p->wpart, p->d_frame_off, p->Fs);                                                        \
    hipLaunchKernelGGL((w_update_kernel<KP>), dim3(p->n_utt, KP / W_KG), dim3(threads), 0, st, p->wpart, W, p->normW, \
                       p->cfg.F, p->Fs, p->cfg.K);                                                               \
  } while (0)
  switch (p->Kp) { case 8: VN_WU(8); break; case 16: VN_WU(16); break; default: VN_WU(32); break; }
#undef VN_WU
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

int vn_launch_w_update_tiles(const vaenmf_plan* p, float* W, hipStream_t st, bool groups) {
  const int threads = ((p->Fs + 63) / 64) * 64;
  if (groups) {                                           // group sums -> tile partials (small batches)
    VN_REQUIRE(p->wpart16, "group partials: rank <= 8 only");
    hipLaunchKernelGGL(w_combine_groups_kernel, dim3(p->n_t64, 2 * p->Kp), dim3(threads), 0, st, p->wpart16, p->d_t64_g0, p->d_t64_cnt,
                       p->cfg.F, p->Fs, 2 * p->Kp, p->wpart64);
  }
  switch (p->Kp) {
    case 8: hipLaunchKernelGGL((w_update_tiles_kernel<8>), dim3(p->n_utt), dim3(threads), 0, st, p->wpart64, p->d_t64_first, W, p->normW, p->cfg.F, p->Fs, p->cfg.K); break;
    case 16: hipLaunchKernelGGL((w_update_tiles_kernel<16>), dim3(p->n_utt), dim3(threads), 0, st, p->wpart64, p->d_t64_first, W, p->normW, p->cfg.F, p->Fs, p->cfg.K); break;
    default: hipLaunchKernelGGL((w_update_tiles_kernel<32>), dim3(p->n_utt), dim3(threads), 0, st, p->wpart64, p->d_t64_first, W, p->normW, p->cfg.F, p->Fs, p->cfg.K); break;
  }
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

// n_it consecutive iterations it0 .. it0 + n_it - 1 in one launch (cost_frames: n_it rows of `stride` doubles)
int vn_launch_cost_reduce(const vaenmf_plan* p, const double* cost_frames, size_t stride, int n_it, int R, double* cost, int niter, int it0, hipStream_t st) {
  hipLaunchKernelGGL(cost_reduce_kernel, dim3(p->n_utt, n_it), dim3(256), 0, st, cost_frames, stride, p->d_frame_off, R, p->cfg.F, cost, niter, it0);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vaenmf_dense(const float* X, int32_t M, int32_t in, int32_t ldx, const float* Wt, const float* b,
                            int32_t out, int32_t act, float* Y, int32_t ldy, void* stream) {
  VN_REQUIRE(X && Wt && Y && M > 0 && in > 0 && out > 0, "vaenmf_dense: bad arguments");
  hipLaunchKernelGGL(dense_kernel, dim3((out + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, X, M, in, ldx,
                     Wt, b, out, act, Y, ldy);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vaenmf_power_spec(const float* X, float* X2, int64_t n, void* stream) {
  VN_REQUIRE(X && X2 && n > 0, "vaenmf_power_spec: bad arguments");
  hipLaunchKernelGGL(power_spec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(X), X2, n);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vaenmf_init_nmf(vaenmf_plan* p, float* W, float* Ht, float* g, uint64_t salt, float eps, void* stream) {
  VN_REQUIRE(p && W && Ht && g && p->NT > 0, "vaenmf_init_nmf: null argument or no batch bound");
  const size_t n = (size_t)p->n_utt * p->Fs * p->Kp + (size_t)p->NT * p->Kp + (size_t)p->NT;
  hipLaunchKernelGGL(nmf_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, Ht, g, p->d_utt_seed,
                     p->d_frame_utt, p->d_frame_loc, p->n_utt, p->NT, p->cfg.F, p->Fs, p->cfg.K, p->Kp, salt, eps);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vaenmf_stft_num_frames(int64_t n_samples, double fs, double wlen_sec, double hop_percent, int32_t* nfft,
                                      int32_t* hop, int32_t* n_frames, int32_t* n_padded) {
  VN_REQUIRE(wlen_sec * fs == (double)(int64_t)(wlen_sec * fs), "wlen_sample of STFT is not an integer.");  // stft.py:37-38
  const int nf = (int)(wlen_sec * fs);
  const int hp = (int)(hop_percent * nf);
  VN_REQUIRE(nf >= 16 && nf <= 2048 && (nf & (nf - 1)) == 0, "n_fft=%d: this build needs a power of two in [16,2048]", nf);
  VN_REQUIRE(hp > 0, "hop must be positive");
  const double utt_len = (double)n_samples / fs;                                                            // stft.py:49
  const double ratio = utt_len / wlen_sec / hop_percent;
  int64_t Tp = n_samples;
  if (ceil(ratio) != (double)(int64_t)ratio) Tp += hp;                                                      // stft.py:50-51
  *nfft = nf; *hop = hp; *n_padded = (int32_t)Tp; *n_frames = (int32_t)(1 + Tp / hp);
  return 0;
}

extern "C" int vaenmf_stft_batch(const float* wav, int32_t n_frames_total, const int64_t* sample_offsets,
                                 const int32_t* frame_offsets, const int32_t* frame_utt, const int32_t* padded_len,
                                 int32_t nfft, int32_t hop, int32_t Fs, float* X, void* stream) {
  VN_REQUIRE(wav && X && n_frames_total > 0, "vaenmf_stft_batch: bad arguments");
  VN_REQUIRE((nfft & (nfft - 1)) == 0 && nfft <= 2048 && Fs >= nfft / 2 + 1, "vaenmf_stft_batch: bad nfft/Fs");
  const size_t lds = (size_t)nfft * 3 * sizeof(double);
  hipLaunchKernelGGL(stft_kernel, dim3(n_frames_total), dim3(256), lds, (hipStream_t)stream, wav, sample_offsets,
                     frame_offsets, frame_utt, padded_len, nfft, ilog2(nfft), hop, Fs, reinterpret_cast<float2*>(X));
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vaenmf_istft_batch(const float* S, int32_t n_utt, int32_t n_frames_total, const int64_t* sample_offsets,
                                  const int32_t* frame_offsets, int32_t nfft, int32_t hop, int32_t Fs, float* work,
                                  float* wav_out, void* stream) {
  VN_REQUIRE(S && work && wav_out && n_utt > 0, "vaenmf_istft_batch: bad arguments");
  VN_REQUIRE((nfft & (nfft - 1)) == 0 && nfft <= 2048 && Fs >= nfft / 2 + 1, "vaenmf_istft_batch: bad nfft/Fs");
  const size_t lds = (size_t)nfft * 3 * sizeof(double);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(istft_frames_kernel, dim3(n_frames_total), dim3(256), lds, st, reinterpret_cast<const float2*>(S), nfft,
                     ilog2(nfft), Fs, work);
  hipLaunchKernelGGL(istft_ola_kernel, dim3(64, n_utt), dim3(256), 0, st, work, sample_offsets, frame_offsets, n_utt, nfft, hop, wav_out);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vaenmf_gram3_batch(const float* s_hat, const float* s, const float* n, int32_t n_utt,
                                  const int64_t* sample_offsets, double* out, void* stream) {
  VN_REQUIRE(s_hat && s && n && out && n_utt > 0, "vaenmf_gram3_batch: bad arguments");
  hipLaunchKernelGGL(gram3_kernel, dim3(n_utt), dim3(256), 0, (hipStream_t)stream, s_hat, s, n, sample_offsets, out);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

// Measurement aid of bench.py (not on the hot path): the rate at which this GPU delivers a plain streaming read of `bytes`
// bytes of `buf` (DEV, 16-byte aligned) to a hand-written kernel -- 16 bytes per lane, 8 wavefronts per SIMD, 4 loads in
// flight per lane.  `reps` timed sweeps after one untimed sweep, HIP events on `stream`; synchronises.  sink DEV: 4 bytes.
extern "C" int vaenmf_hbm_read_probe(const void* buf, int64_t bytes, int32_t reps, void* sink, double* gbps_out, void* stream) {
  VN_REQUIRE(buf && sink && gbps_out && bytes >= (1 << 20) && reps > 0, "vaenmf_hbm_read_probe: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, cus = 256;
  VN_CHECK_HIP(hipGetDevice(&dev));
  VN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  hipEvent_t e0, e1;
  VN_CHECK_HIP(hipEventCreate(&e0));
  VN_CHECK_HIP(hipEventCreate(&e1));
  const size_t n16 = (size_t)bytes / 16;
  auto sweep = [&]() { hipLaunchKernelGGL((hbm_read_probe_kernel<4>), dim3(cus * 8), dim3(256), 0, st, reinterpret_cast<const u32x4p*>(buf), n16, reinterpret_cast<unsigned*>(sink)); };
  sweep();
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) sweep();
  (void)hipEventRecord(e1, st);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  VN_CHECK_HIP(e);
  VN_CHECK_HIP(hipGetLastError());
  *gbps_out = (double)(n16 * 16) * reps / ((double)ms * 1e-3) / 1e9;
  return 0;
}
