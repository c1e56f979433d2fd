// Label / guide front-ends of the M2 path: Lorenz-quantile IBM and VAD labels
// (python/processing/target.py:7-50), the ideal Wiener mask (:104-116) and the
// mask-times-spectrogram step of the supervised baseline (scripts/evaluate_wiener_filter.py:99).
//
// The labels are 0/1 decisions `power > threshold` where the threshold is the last entry of the
// descending-sorted powers whose Lorenz value cumsum/sum is below the quantile fraction.  A decision
// flips if a single float32 rounding differs, so the float32 arithmetic of the reference's numpy calls
// is reproduced operation for operation (the test suite's CPU checker holds the same statement and is
// pinned against the reference's own outputs, tests/golden/labels_f257.npz):
//   power     fma(re, re, round(im*im))                      (complex64 product, target.py:16/:37)
//   total     numpy's pairwise float32 sum                   (np.sum, :19/:40)
//   cumsum    running float32 sum                            (np.cumsum)
//   per-frame power (VAD): pairwise sum over the F bins of the frame (power.sum(axis=0) on the
//             Fortran-ordered STFT the reference's stft returns, :38)
// The sort itself is order-free: rocPRIM segmented radix sort (one segment per utterance).
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace {

// numpy pairwise_sum over a[0..n) (float32), restated; called by single threads
__device__ float pw_block(const float* a, int n) {        // n <= 128
  if (n < 8) {
    float r = 0.f;
    for (int i = 0; i < n; ++i) r = __fadd_rn(r, a[i]);
    return r;
  }
  float r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = a[k];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = __fadd_rn(r[k], a[i + k]);
  }
  float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                        __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
  for (; i < n; ++i) res = __fadd_rn(res, a[i]);
  return res;
}

// the recursion of pairwise_sum unrolled with an explicit stack (depth <= 32: n < 2^31)
__device__ float pw_sum(const float* a, int64_t n) {
  struct Item { int64_t off, n; int state; float left; };
  Item st[40];
  int sp = 0;
  st[0] = {0, n, 0, 0.f};
  float ret = 0.f;
  while (sp >= 0) {
    Item& it = st[sp];
    if (it.n <= 128) { ret = pw_block(a + it.off, (int)it.n); --sp; continue; }
    int64_t n2 = it.n / 2;
    n2 -= n2 % 8;
    if (it.state == 0) { it.state = 1; st[++sp] = {it.off, n2, 0, 0.f}; }
    else if (it.state == 1) { it.left = ret; it.state = 2; st[++sp] = {it.off + n2, it.n - n2, 0, 0.f}; }
    else { ret = __fadd_rn(it.left, ret); --sp; }
  }
  return ret;
}

// compact powers [NT][F] (no padding bins)
__global__ void power_compact_kernel(const float2* __restrict__ X, int NT, int F, int Fs, float* __restrict__ P) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)NT * F) return;
  const int n = (int)(i / F), f = (int)(i - (int64_t)n * F);
  const float2 v = X[(size_t)n * Fs + f];
  P[i] = __fmaf_rn(v.x, v.x, __fmul_rn(v.y, v.y));
}

__global__ void frame_power_kernel(const float* __restrict__ P, int NT, int F, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < NT) out[n] = pw_sum(P + (size_t)n * F, F);
}

// one workgroup per utterance, thread 0 walks the sorted segment
__global__ void lorenz_threshold_kernel(const float* __restrict__ sorted, const int* __restrict__ seg_off, float q,
                                        float* __restrict__ thr, int* __restrict__ err) {
  if (threadIdx.x != 0) return;
  const int u = blockIdx.x;
  const float* s = sorted + seg_off[u];
  const int64_t n = seg_off[u + 1] - seg_off[u];
  const float total = pw_sum(s, n);
  float run = 0.f, t = 0.f;
  bool any = false;
  for (int64_t i = 0; i < n; ++i) {
    run = __fadd_rn(run, s[i]);
    if (__fdiv_rn(run, total) < q) { t = s[i]; any = true; }
    else break;                                   // non-decreasing curve: nothing further qualifies
  }
  thr[u] = t;
  if (!any) atomicExch(err, 1 + u);               // target.py:22 raises IndexError here
}

__global__ void ibm_labels_kernel(const float* __restrict__ P, const int* __restrict__ frame_utt, const float* __restrict__ thr,
                                  int NT, int F, float lo, float hi, float* __restrict__ out, int ld) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)NT * F) return;
  const int n = (int)(i / F), f = (int)(i - (int64_t)n * F);
  out[(size_t)n * ld + f] = P[i] > thr[frame_utt[n]] ? hi : lo;
}

__global__ void vad_labels_kernel(const float* __restrict__ pf, const int* __restrict__ frame_utt, const float* __restrict__ thr,
                                  int NT, float lo, float hi, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < NT) out[n] = pf[n] > thr[frame_utt[n]] ? hi : lo;
}

// ideal Wiener mask |S|^2 / (|S|^2 + |N|^2 + eps) in float32 like np.power(abs(c64), 2); abs = the correctly
// rounded hypot (numpy's vectorised abs is within 1 ulp of it: the mask agrees to ~1e-7)
__global__ void wiener_mask_kernel(const float2* __restrict__ S, const float2* __restrict__ Nz, int64_t n, float eps, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 s = S[i], z = Nz[i];
  const float as = (float)sqrt((double)s.x * s.x + (double)s.y * s.y), an = (float)sqrt((double)z.x * z.x + (double)z.y * z.y);
  const float sp = __fmul_rn(as, as), np = __fmul_rn(an, an);
  out[i] = __fdiv_rn(sp, __fadd_rn(__fadd_rn(sp, np), eps));
}

__global__ void apply_mask_kernel(const float2* __restrict__ X, const float* __restrict__ mask, int ldm, int NT, int F, int Fs,
                                  float2* __restrict__ S) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)NT * Fs) return;
  const int n = (int)(i / Fs), f = (int)(i - (int64_t)n * Fs);
  float2 v = {0.f, 0.f};
  if (f < F) { const float m = mask[(size_t)n * ldm + f]; v = X[i]; v.x *= m; v.y *= m; }
  S[i] = v;
}

struct Carve {
  char* base; size_t off = 0;
  template <typename T> T* take(size_t count) {
    off = (off + 255) / 256 * 256;
    T* p = reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    return p;
  }
};

size_t sort_temp_bytes(int64_t items, int segs) {
  size_t bytes = 0;
  hipcub::DeviceSegmentedRadixSort::SortKeysDescending(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (int)items, segs,
                                                       (const int*)nullptr, (const int*)nullptr, 0, 32, (hipStream_t)0);
  return bytes;
}

}  // namespace

extern "C" int64_t vaenmf_lorenz_work_bytes(int32_t n_frames_total, int32_t F, int32_t n_utt, int32_t mode) {
  const int64_t M = mode == VAENMF_LABEL_IBM ? (int64_t)n_frames_total * F : n_frames_total;
  size_t b = 4096;
  b += ((size_t)n_frames_total * F * 4 + 256);          // compact powers
  b += 2 * ((size_t)M * 4 + 256);                       // values (VAD: frame powers) + sorted
  b += ((size_t)n_frames_total * 4 + 256) + ((size_t)(n_utt + 1) * 4 + 256) * 2 + 512;
  b += sort_temp_bytes(M, n_utt) + 256;
  return (int64_t)b;
}

extern "C" int vaenmf_lorenz_labels(const float* X, int32_t n_utt, const int32_t* frame_offsets, int32_t F, int32_t Fs,
                                    int32_t mode, float quantile_fraction, float lo, float hi, float* labels, int32_t ld,
                                    float* thr_out, void* work, int64_t work_bytes, void* stream) {
  VN_REQUIRE(X && labels && frame_offsets && work && n_utt > 0 && F > 0 && Fs >= F, "vaenmf_lorenz_labels: bad arguments");
  VN_REQUIRE(mode == VAENMF_LABEL_IBM || mode == VAENMF_LABEL_VAD, "vaenmf_lorenz_labels: mode %d", mode);
  const int NT = frame_offsets[n_utt];
  VN_REQUIRE(NT > 0 && (int64_t)NT * F < (1ll << 31), "vaenmf_lorenz_labels: %d frames x %d bins does not fit one sort", NT, F);
  VN_REQUIRE(work_bytes >= vaenmf_lorenz_work_bytes(NT, F, n_utt, mode), "vaenmf_lorenz_labels: work buffer too small");
  VN_REQUIRE(mode == VAENMF_LABEL_VAD || ld >= F, "vaenmf_lorenz_labels: ld < F");
  hipStream_t st = (hipStream_t)stream;
  const int64_t M = mode == VAENMF_LABEL_IBM ? (int64_t)NT * F : NT;
  Carve cv{reinterpret_cast<char*>(work)};
  float* P = cv.take<float>((size_t)NT * F);
  float* vals = mode == VAENMF_LABEL_IBM ? P : cv.take<float>(NT);
  float* sorted = cv.take<float>(M);
  int* frame_utt = cv.take<int>(NT);
  int* seg = cv.take<int>(n_utt + 1);
  float* thr = cv.take<float>(n_utt);
  int* err = cv.take<int>(1);
  size_t tb = sort_temp_bytes(M, n_utt);
  void* temp = cv.take<char>(tb);
  // segment tables (host -> device)
  std::vector<int> h_seg(n_utt + 1), h_fu(NT);
  for (int u = 0; u <= n_utt; ++u) h_seg[u] = mode == VAENMF_LABEL_IBM ? frame_offsets[u] * F : frame_offsets[u];
  for (int u = 0; u < n_utt; ++u) {
    VN_REQUIRE(frame_offsets[u + 1] > frame_offsets[u], "vaenmf_lorenz_labels: empty utterance %d", u);
    for (int n = frame_offsets[u]; n < frame_offsets[u + 1]; ++n) h_fu[n] = u;
  }
  VN_CHECK_HIP(hipMemcpyAsync(seg, h_seg.data(), h_seg.size() * sizeof(int), hipMemcpyHostToDevice, st));
  VN_CHECK_HIP(hipMemcpyAsync(frame_utt, h_fu.data(), h_fu.size() * sizeof(int), hipMemcpyHostToDevice, st));
  VN_CHECK_HIP(hipMemsetAsync(err, 0, sizeof(int), st));
  VN_CHECK_HIP(hipStreamSynchronize(st));               // the host tables above go out of scope
  const int64_t np = (int64_t)NT * F;
  hipLaunchKernelGGL(power_compact_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float2*>(X), NT, F, Fs, P);
  if (mode == VAENMF_LABEL_VAD)
    hipLaunchKernelGGL(frame_power_kernel, dim3((NT + 63) / 64), dim3(64), 0, st, P, NT, F, vals);
  VN_CHECK_HIP(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(temp, tb, vals, sorted, (int)M, n_utt, seg, seg + 1, 0, 32, st));
  hipLaunchKernelGGL(lorenz_threshold_kernel, dim3(n_utt), dim3(64), 0, st, sorted, seg, quantile_fraction, thr, err);
  if (mode == VAENMF_LABEL_IBM)
    hipLaunchKernelGGL(ibm_labels_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, P, frame_utt, thr, NT, F, lo, hi, labels, ld);
  else
    hipLaunchKernelGGL(vad_labels_kernel, dim3((NT + 255) / 256), dim3(256), 0, st, vals, frame_utt, thr, NT, lo, hi, labels);
  if (thr_out) VN_CHECK_HIP(hipMemcpyAsync(thr_out, thr, n_utt * sizeof(float), hipMemcpyDeviceToDevice, st));
  int h_err = 0;
  VN_CHECK_HIP(hipMemcpyAsync(&h_err, err, sizeof(int), hipMemcpyDeviceToHost, st));
  VN_CHECK_HIP(hipStreamSynchronize(st));
  VN_REQUIRE(h_err == 0, "index -1 is out of bounds for axis 0 with size 0 (utterance %d: no Lorenz value below the fraction)", h_err - 1);
  return 0;
}

extern "C" int vaenmf_wiener_mask(const float* S, const float* N, int64_t n, float eps, float* mask, void* stream) {
  VN_REQUIRE(S && N && mask && n > 0, "vaenmf_wiener_mask: bad arguments");
  hipLaunchKernelGGL(wiener_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(S), reinterpret_cast<const float2*>(N), n, eps, mask);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vaenmf_apply_mask(const float* X, const float* mask, int32_t ldm, int32_t NT, int32_t F, int32_t Fs, float* S_hat,
                                 void* stream) {
  VN_REQUIRE(X && mask && S_hat && NT > 0 && F > 0 && Fs >= F && ldm >= F, "vaenmf_apply_mask: bad arguments");
  const int64_t n = (int64_t)NT * Fs;
  hipLaunchKernelGGL(apply_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(X), mask, ldm, NT, F, Fs, reinterpret_cast<float2*>(S_hat));
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

// ============================================================================
// SPP-based noise PSD / speech-presence estimator (python/models/spp_estimation.py:17-160, Gerkmann &
// Hendriks 2011): a per-bin recursion over the frames of an utterance, float64 state like the reference
// (numpy float64 arrays), one thread per (utterance, bin).
// ============================================================================
namespace {

struct SppParams { double fixed_smooth, prob_smooth, inv_glr_factor, inv_glr_exp_factor; int num_frames_init; };

// per: [NT][ld] float32 periodogram |Y|^2; spp_out / psd_out float32 [NT][ldo] (either may be null)
__global__ void spp_kernel(const float* __restrict__ per, int ld, const int* __restrict__ frame_off, int F, SppParams p,
                           float* __restrict__ spp_out, float* __restrict__ psd_out, int ldo) {
  const int u = blockIdx.y, f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int n0 = frame_off[u], n1 = frame_off[u + 1];
  double old_psd = 0.0, smooth = 0.0;
  int done = 0;
  for (int n = n0; n < n1; ++n) {
    const double y = (double)per[(size_t)n * ld + f];
    double psd, spp;
    if (done < p.num_frames_init) {                                     // spp_estimation.py:105-117
      old_psd = old_psd + y / (double)p.num_frames_init;
      ++done;
      psd = y;                                                          // (the call returns the periodogram itself)
      spp = 0.0;
    } else {
      const double inv_glr = p.inv_glr_factor * exp(-y / (old_psd + 1e-8) * p.inv_glr_exp_factor);   // :120-121
      spp = 1.0 / (1.0 + inv_glr);                                                                  // :124
      smooth = (1.0 - p.prob_smooth) * spp + p.prob_smooth * smooth;                                // :128-129
      if (smooth > 0.99) spp = fmin(spp, 0.99);                                                     // :130-131
      const double nper = (1.0 - spp) * y + spp * old_psd;                                          // :135-136
      psd = (1.0 - p.fixed_smooth) * nper + p.fixed_smooth * old_psd;                               // :138-139
      old_psd = psd;                                                                                // :142
    }
    if (spp_out) spp_out[(size_t)n * ldo + f] = (float)spp;
    if (psd_out) psd_out[(size_t)n * ldo + f] = (float)psd;
  }
}

// the `v_spp_in` branch (:145-153): the old PSD is never updated there, so it stays at its initial zeros
__global__ void spp_given_kernel(const float* __restrict__ per, const float* __restrict__ spp_in, int64_t n, double fixed_smooth,
                                 float* __restrict__ psd_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double nper = (double)__fmul_rn(__fsub_rn(1.f, spp_in[i]), per[i]);      // float32 product (float32 operands), then float64
  psd_out[i] = (float)((1.0 - fixed_smooth) * nper);
}

}  // namespace

extern "C" int vaenmf_spp_estimate(const float* per, int32_t ld, int32_t n_utt, const int32_t* frame_offsets, int32_t F,
                                   double fixed_smooth, double prob_smooth, double prior, double snr_opt_db,
                                   int32_t num_frames_init, float* spp_out, float* psd_out, int32_t ldo, void* stream) {
  VN_REQUIRE(per && frame_offsets && n_utt > 0 && F > 0 && ld >= F && ldo >= F && (spp_out || psd_out), "vaenmf_spp_estimate: bad arguments");
  VN_REQUIRE(prior > 0.0 && prior < 1.0 && num_frames_init >= 0, "vaenmf_spp_estimate: bad prior / num_frames_init");
  const double snr_lin = pow(10.0, snr_opt_db / 10.0);                                              // :76
  SppParams p{fixed_smooth, prob_smooth, (1.0 - prior) / prior * (1.0 + snr_lin), snr_lin / (1.0 + snr_lin), num_frames_init};   // :85-86
  hipLaunchKernelGGL(spp_kernel, dim3((F + 63) / 64, n_utt), dim3(64), 0, (hipStream_t)stream, per, ld, frame_offsets, F, p, spp_out,
                     psd_out, ldo);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int vaenmf_spp_noise_given(const float* per, const float* spp_in, int64_t n, double fixed_smooth, float* psd_out, void* stream) {
  VN_REQUIRE(per && spp_in && psd_out && n > 0, "vaenmf_spp_noise_given: bad arguments");
  hipLaunchKernelGGL(spp_given_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, per, spp_in, n, fixed_smooth, psd_out);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}
