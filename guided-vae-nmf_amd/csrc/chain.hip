// Wave-private Metropolis-Hastings chains (MCEM_M1.sample_posterior mcem.py:371-441, MCEM_M2 :218-294).
//
// One wavefront owns 16 frames for a whole chain and evaluates the complete decoder
// (python/models/models.py:118-121) for them on v_mfma_f32_16x16x32_bf16: the frames are the MFMA column
// dimension, every feature tile of every layer belongs to the same wavefront.  An accumulator tile
// (lane (q,c): features 4q..4q+3 of frame c) is, after tanh and the bf16 rounding, exactly half of the
// B fragment of the next layer IN THE SAME LANE (k-permutation phi(s,q,j) = 32 s + 16 (j>>2) + 4 q + (j&3),
// applied to the weights on the host, plan.hip: pack_weights), so activations never leave the registers:
// no LDS exchange, no barrier anywhere in the chain.  The sum over bins of the acceptance ratio is in-lane
// plus two permlane swaps, the noise of a frame's 32 latents is drawn by the four lanes that hold them, and
// the wavefronts of a workgroup share nothing but the LDS-resident weights: two wavefronts on a SIMD drift
// apart and fill each other's MFMA / transcendental latencies.
//
// Bin order of the last layer (bf16 mode): tiles are paired so that a lane holds 8 CONSECUTIVE bins per
// pair (bin = 32 (t>>1) + 8 q + 4 (t&1) + j) and writes its part of a sample-variance row with 16-byte
// stores that cover 64 contiguous bytes per frame and instruction; the bf16x3 mode stores float rows and
// keeps the natural order (bin = 16 t + 4 q + j).  The permutation is applied to W3 / b3 on the host.
#include "common.h"

// Diagnostic builds only (-DVN_STAMP): per-phase tick sums (s_memtime) of workgroup 0 / wave 0 into a debug buffer that
// nothing else reads.  The shipped library never executes a stamp.
#ifdef VN_STAMP
__device__ long long g_wc_stamp[32];
// (sums are kept in registers and written once per wave tile: a read-modify-write of the debug buffer inside the loop
// would wait, through vmcnt, for every variance store in flight and time those instead)
#define WC_STAMP_DECL long long _t_prev = (long long)__builtin_amdgcn_s_memtime(); long long _st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int _st_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WC_STAMP(slot) do { long long _t = (long long)__builtin_amdgcn_s_memtime(); _st_acc[slot] += _t - _t_prev; _st_cnt[slot] += 1; _t_prev = _t; } while (0)
#define WC_STAMP_FLUSH do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int _i = 0; _i < 8; ++_i) { g_wc_stamp[_i] += _st_acc[_i]; g_wc_stamp[16 + _i] += _st_cnt[_i]; } } } while (0)
#else
#define WC_STAMP_DECL
#define WC_STAMP(slot)
#define WC_STAMP_FLUSH
#endif
#ifndef VN_WC_SPREAD
#define VN_WC_SPREAD 1      // wave tiles spread over the CUs first (one 4 s utterance through the drop-in classes: 80.5 -> 69.0 ms; the 64-utterance batch: unchanged)
#endif
#ifndef VN_PF
#define VN_PF 1
#endif
#ifndef VN_PACKED
#define VN_PACKED 1
#endif
#ifndef VN_ST_AUX
#define VN_ST_AUX 0      // cache policy of the sample-variance row stores (2 = nt)
#endif
#ifndef VN_PFL
#define VN_PFL 2      // bf16x3 mode, F > 80: tiles of W3-lo fragments (streamed from L2, ~1 us away) in flight ahead of their MFMAs -- with one
                      // tile the single wavefront of a SIMD sat out the L2 latency once per bin tile (1.00 -> 0.89 ms per launch)
#endif
#ifndef VN_PFL_HOIST
#define VN_PFL_HOIST 2     // of those, tiles requested at the top of the evaluation, two layers before their use (registers live through the
                           // hidden layers: 2 + 2 fits the 512 registers, 3 + 1 spills; 132.5 -> 131.8 ms per bf16x3 step)
#endif
#ifndef VN_TPM
#define VN_TPM 2      // transcendentals scheduled right behind each MFMA (a packed instruction waits for a matrix instruction in flight); 0: mixed with the other VALU work
#endif
#ifndef VN_TPR
#define VN_TPR 1
#endif
#ifndef VN_VPER
#define VN_VPER 8
#endif
#ifndef VN_SB
#define VN_SB __builtin_amdgcn_sched_barrier(0)
#endif

namespace {

constexpr int NK = HID / 32;      // k-steps over a hidden layer
constexpr int NTH = HID / 16;     // feature tiles of a hidden layer
constexpr int WT_FRAMES = 16;     // frames per wavefront
constexpr unsigned WC_OOB = 0xF0000000u;   // byte offset behind every buffer of the chain (the host keeps them below 3.5 GB)
#ifndef VN_WC_WAVES_BF16
#define VN_WC_WAVES_BF16 8
#endif

struct WcArgs {
  const __bf16 *w1f, *w2f, *w3f;   // fragment order [tile][kstep][part hi/lo][lane][8]; w3f in the chain's bin order
  const float *b1, *b2, *b3;       // b3 in the chain's bin order, padded to 16 NT3 with -200 (2^-200 = 0: padding bins store exact zeros)
  int NT3;                         // bin tiles (the last one may be partial)
  int Tm;                          // tiles in paired order (bf16 mode); the rest is natural
  int F, Fs;
  const float *X2, *W, *Ht, *g, *B1, *Vb;
  float *Z, *Zs, *acc_out;
  void* VsS;                       // sample-variance store [NT+1][Rs][Fs] (float: bf16x3 mode, bf16: bf16 mode) or null
  unsigned VsS_bytes;
  int32_t* src;                    // [Rs][NT]
  int Rs;
  const int32_t *wt_utt, *wt_n0, *wt_cnt;   // wave tiles: <= 16 frames of one utterance
  int n_wtiles;
  const int32_t* frame_off;
  const uint64_t* utt_seed;
  const float *eps, *u;            // replay draws or null
  int Kp, NT, n_utts, Rcap, nsamples, burnin, rng_mode, update_Z;
  int n_hi_lds;                    // W3 tiles whose hi fragments are in LDS (the rest streams from L2)
  int b1_lds;                      // M2 at 8 wavefronts: byte offset of the per-wave stash of the layer-1 bias rows
  uint32_t call;
  float sd;
  float sd_hi;                     // step of latents 16..31 (0: they are padding of a 16-dimensional latent space)
  int one_hidden;                  // decoder with one hidden layer: layer 2 is skipped
};

template <bool SPLIT>
struct WcLds {
  static constexpr int PARTS = SPLIT ? 2 : 1;
  static constexpr int W1 = 0;                                  // [8][PARTS][1 KB]
  static constexpr int W2 = W1 + NTH * PARTS * 1024;            // [8][4][PARTS][1 KB]
  static constexpr int B1 = W2 + NTH * NK * PARTS * 1024;       // float[128]
  static constexpr int B2 = B1 + HID * 4;
  static constexpr int B3 = B2 + HID * 4;                       // float[16 * 40]
  static constexpr int W3 = B3 + 640 * 4;                       // hi blocks [tile][kstep][1 KB] of the first n_hi_lds tiles,
                                                                // then (LOL) the lo blocks of all tiles
  static constexpr int fixed_bytes = W3;
};

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned pk2(float a, float b) { return __builtin_bit_cast(unsigned, bf16x2{(__bf16)a, (__bf16)b}); }   // v_cvt_pk_bf16_f32
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }
__device__ __forceinline__ bf16x8 cat8(const bf16x4 a, const bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

template <bool SPLIT>
__device__ __forceinline__ void pack4(const f32x4 h, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const __bf16 x = (__bf16)h[t];
    hi[t] = x;
    lo[t] = SPLIT ? (__bf16)(h[t] - (float)x) : (__bf16)0.f;
  }
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Two elements per non-transcendental instruction (v_pk_add/mul/fma_f32): next to transcendentals a packed instruction
// costs the issue slot of a plain one (tools/ubench/overlap.hip: 4.4 against 4.2 ticks, for twice the work).
__device__ __forceinline__ f32x4 tanh4(const f32x4 a) {
#if VN_PACKED
  const f32x2 one = {1.f, 1.f}, m2 = {-2.f, -2.f};
  const f32x2 d0 = f32x2{fast_exp(a[0]), fast_exp(a[1])} + one, d1 = f32x2{fast_exp(a[2]), fast_exp(a[3])} + one;
  const f32x2 r0 = {fast_rcp(d0[0]), fast_rcp(d0[1])}, r1 = {fast_rcp(d1[0]), fast_rcp(d1[1])};
  const f32x2 h0 = m2 * r0 + one, h1 = m2 * r1 + one;
  return f32x4{h0[0], h0[1], h1[0], h1[1]};
#else
  f32x4 h;
#pragma unroll
  for (int t = 0; t < 4; ++t) h[t] = fast_tanh(a[t]);
  return h;
#endif
}
template <bool SPLIT>
__device__ __forceinline__ f32x4 mma(const bf16x8 whi, const bf16x8 wlo, const bf16x8 ahi, const bf16x8 alo, f32x4 acc) {
  if (SPLIT) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, ahi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, alo, acc, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, ahi, acc, 0, 0, 0);
}

// MAXT : compile-time bound of the bin tiles; EXACT: NT3 == MAXT (no per-tile checks)
// LOL  : the lo fragments of W3 are in LDS too (bf16x3 mode, small F); otherwise they stream from L2
// HIALL: every hi fragment of W3 is in LDS
// GT   : the hi fragments of the FIRST GT bin tiles stay in global memory (F = 513: W3 does not fit the LDS whole); a
//        wavefront requests them at the top of every evaluation, two layers before their use, and before any store of
//        that evaluation (vmcnt counts in order); LDS holds tiles GT..NT3-1
// M2   : per-frame layer-1 bias B1 = b1 + W1[:, L:] y_n, else b1 from LDS.  At 4 wavefronts (512 registers each) the
//        lane's 32 values stay in registers; at 8 (bf16 mode, 256 registers) each lane parks them as bf16 in a private
//        8 x 8 bytes of LDS (4 KB per wavefront) and reads one 8-byte word back per layer-1 tile
template <int MAXT, bool EXACT, bool SPLIT, bool STORE, int NWAVES, bool LOL, bool HIALL, bool M2, int GT>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void wchain_kernel(const WcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = WcLds<SPLIT>;
  constexpr int PARTS = L::PARTS;
  using store_t = typename std::conditional<SPLIT, float, __bf16>::type;
  const int NT3 = EXACT ? MAXT : a.NT3;
  const int n_hi = HIALL ? NT3 - GT : a.n_hi_lds;

  // ---- workgroup prologue: weights and biases into LDS (the only barrier of the kernel)
  {
    const int nthr = NWAVES * 64;
    auto stage = [&](char* dst, const __bf16* srcp, int nblk) {    // blocks [blk][2 parts] -> [blk][PARTS]
      for (int e = threadIdx.x; e < nblk * PARTS * 64; e += nthr) {
        const int chunk = e & 63, pb = e >> 6, b = pb / PARTS, part = pb - b * PARTS;
        *reinterpret_cast<f32x4*>(dst + (size_t)e * 16) =
            *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(srcp) + ((size_t)(b * 2 + part) * 64 + chunk) * 16);
      }
    };
    stage(smem + L::W1, a.w1f, NTH);
    stage(smem + L::W2, a.w2f, NTH * NK);
    for (int e = threadIdx.x; e < n_hi * NK * 64; e += nthr) {      // hi blocks of W3 (tiles GT..)
      const int chunk = e & 63, b = (e >> 6) + GT * NK;
      *reinterpret_cast<f32x4*>(smem + L::W3 + (size_t)e * 16) =
          *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.w3f) + ((size_t)(b * 2) * 64 + chunk) * 16);
    }
    if (SPLIT && LOL)
      for (int e = threadIdx.x; e < NT3 * NK * 64; e += nthr) {     // lo blocks of W3, behind the hi blocks
        const int chunk = e & 63, b = e >> 6;
        *reinterpret_cast<f32x4*>(smem + L::W3 + (size_t)n_hi * NK * 1024 + (size_t)e * 16) =
            *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.w3f) + ((size_t)(b * 2 + 1) * 64 + chunk) * 16);
      }
    float* b1s = reinterpret_cast<float*>(smem + L::B1);
    float* b2s = reinterpret_cast<float*>(smem + L::B2);
    float* b3s = reinterpret_cast<float*>(smem + L::B3);
    for (int i = threadIdx.x; i < HID; i += nthr) { b1s[i] = a.b1[i]; b2s[i] = a.b2[i]; }
    for (int i = threadIdx.x; i < 16 * NT3; i += nthr) b3s[i] = a.b3[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
  const unsigned l16 = (unsigned)lane * 16u;
  const float* b1l = reinterpret_cast<const float*>(smem + L::B1);
  const float* b2l = reinterpret_cast<const float*>(smem + L::B2);
  const float* b3l = reinterpret_cast<const float*>(smem + L::B3);
  const char* w3lo_lds = smem + L::W3 + (size_t)n_hi * NK * 1024;
  const char* w3g = reinterpret_cast<const char*>(a.w3f);
  const int S = a.nsamples + a.burnin;
  // first of this lane's 4 consecutive bins in tile t
  const int Tm = EXACT ? ((MAXT - 1) & ~1) : a.Tm;
  auto bin0 = [&](int t) { return (!SPLIT && t < Tm) ? 32 * (t >> 1) + 8 * q + 4 * (t & 1) : 16 * t + 4 * q; };
  auto tile_on = [&](int t) { return EXACT || t < NT3; };
  auto tile_on3 = [&](int n, int t) { return n != MAXT || EXACT || t < NT3; };      // (hidden layers: every tile)

  __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(a.VsS, 0, STORE ? (int)a.VsS_bytes : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t zs_rs = __builtin_amdgcn_make_buffer_rsrc(a.Zs, 0, a.Zs ? (int)((unsigned)a.NT * (unsigned)a.Rcap * LAT * 4u) : 0, 0x00020000);   // (no Zs: the sample stores fall outside the resource and are dropped)
  __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(a.src, 0, STORE ? (int)((unsigned)a.NT * (unsigned)a.Rs * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t acc_rs = __builtin_amdgcn_make_buffer_rsrc(a.acc_out, 0, a.acc_out ? (int)((unsigned)a.NT * (unsigned)S * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t eps_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.eps), 0, a.eps ? (int)((unsigned)a.NT * (unsigned)S * LAT * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t x2in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X2), 0, (int)((unsigned)a.NT * (unsigned)a.Fs * 4u), 0x00020000);
  __amdgpu_buffer_rsrc_t vbin_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Vb), 0, a.Vb ? (int)((unsigned)a.NT * (unsigned)a.Fs * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W), 0, a.W ? (int)((unsigned)a.n_utts * (unsigned)a.Fs * (unsigned)a.Kp * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Ht), 0, a.Ht ? (int)((unsigned)a.NT * (unsigned)a.Kp * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc(a.Z, 0, (int)((unsigned)a.NT * LAT * 4u), 0x00020000);
  __amdgpu_buffer_rsrc_t b1in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B1), 0, M2 ? (int)((unsigned)a.NT * HID * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u ? (int)((unsigned)a.NT * (unsigned)S * 4u) : 0, 0x00020000);

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // wave tile of (workgroup b, wavefront w, round i): (i NWAVES + w) gridDim + b -- a batch with fewer tiles than wavefront
  // slots (one utterance through the drop-in classes: 32 tiles) spreads one wavefront per CU instead of filling four
  // workgroups, and a lone wavefront has its SIMD's issue slots to itself
#if VN_WC_SPREAD
  for (int wt = wave * (int)gridDim.x + (int)blockIdx.x; wt < a.n_wtiles; wt += (int)gridDim.x * NWAVES) {
#else
  for (int wt = blockIdx.x * NWAVES + wave; wt < a.n_wtiles; wt += gridDim.x * NWAVES) {
#endif
    const int utt = a.wt_utt[wt], n0 = a.wt_n0[wt], cnt = a.wt_cnt[wt];
    const bool fvalid = c < cnt;
    const int nrow = n0 + (fvalid ? c : cnt - 1);          // idle lanes shadow the last frame (no stores)
    const float gn = a.g[nrow];
    // ---- per-(bin, frame) constants in accumulator layout: X2 and Vb = W H (mcem.py:81-82) or the given noise PSD.
    // Padding bins: X2 = 0, Vb = 1 and (b3 = -200, W3 = 0) Vs = 0, so their term is exactly 0.
    f32x4 x2[MAXT], vb[MAXT];
    {
      // Through buffer resources: a tile's address is the resource (SGPRs) + one per-lane byte offset + a scalar or
      // immediate tile offset.  With 64-bit pointers the compiler kept an address pair per (tile, bin) alive and spilled
      // them -- the only scratch of the kernel, and a kernel with scratch pays a scratch-memory set-up per dispatch.
      const unsigned rowF = (unsigned)nrow * (unsigned)a.Fs;                     // first element of the frame's row
      const unsigned uF = (unsigned)utt * (unsigned)a.Fs;
      int qo = q;                                            // (opaque: the per-tile bin numbers are recomputed here, once per
      asm volatile("" : "+v"(qo));                           //  wave tile, instead of being hoisted out of the loop and spilled)
      auto bin0o = [&](int t) { return (!SPLIT && t < Tm) ? 32 * (t >> 1) + 8 * qo + 4 * (t & 1) : 16 * t + 4 * qo; };
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        x2[t] = f32x4{0, 0, 0, 0};
        vb[t] = f32x4{1, 1, 1, 1};
        if (tile_on(t)) {
          const int f0 = bin0o(t);
          f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x2in_rs, (rowF + (unsigned)f0) * 4u, 0, 0));
          f32x4 v = {0, 0, 0, 0};
          if (a.Vb) {
            v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vbin_rs, (rowF + (unsigned)f0) * 4u, 0, 0));
          } else {
            for (int k = 0; k < a.Kp; k += 4) {              // (once per launch: a plain loop, operands from L1/L2)
              const f32x4 h4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h_rs, ((unsigned)nrow * (unsigned)a.Kp + (unsigned)k) * 4u, 0, 0));
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const f32x4 w4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, ((uF + (unsigned)(f0 + j)) * (unsigned)a.Kp + (unsigned)k) * 4u, 0, 0));
                v[j] += w4[0] * h4[0] + w4[1] * h4[1] + w4[2] * h4[2] + w4[3] * h4[3];
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (f0 + j >= a.F) { xv[j] = 0.f; v[j] = 1.f; }
          x2[t] = xv;
          vb[t] = v;
        }
      }
    }
    constexpr bool B1L = M2 && NWAVES == 8;                // layer-1 bias rows parked in LDS (see above)
    f32x4 b1r[(M2 && !B1L) ? NTH : 1];                     // M2: layer-1 accumulator init of this lane's frame
    char* b1stash = smem + a.b1_lds + wave * (NTH * 512) + lane * 8;     // [tile][lane][4 bf16], this lane's words only
    if (M2) {
#pragma unroll
      for (int t = 0; t < NTH; ++t) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b1in_rs, ((unsigned)nrow * HID + 4u * (unsigned)q) * 4u + 64u * (unsigned)t, 0, 0));
        if (B1L) *reinterpret_cast<u32x2*>(b1stash + t * 512) = u32x2{pk2(v[0], v[1]), pk2(v[2], v[3])};
        else b1r[(M2 && !B1L) ? t : 0] = v;
      }
    }
    // ---- current latent state, fragment order: latents 4q..4q+3 and 16+4q..16+4q+3 of frame c
    float z[8];
    {
      // (buffer-addressed like everything else per frame: a 64-bit address pair here was loop-invariant in its lane part,
      //  hoisted out of the wave-tile loop and spilled -- the kernel's only scratch)
      const unsigned zo = ((unsigned)nrow * LAT + 4u * (unsigned)q) * 4u;
      const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rs, zo, 0, 0));
      const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rs, zo + 64u, 0, 0));
#pragma unroll
      for (int t = 0; t < 4; ++t) { z[t] = lo[t]; z[4 + t] = hi[t]; }
    }
    // ---- sample-variance store: slot r of the frame holds the variances of the proposal of post-burn-in step r,
    // slot R the state the chain is in when the burn-in ends; src[r][frame] names the slot of the state after
    // step r (mcem.py:429-437).
    // Every per-frame address of the loop is a buffer resource (SGPRs) + one 32-bit byte offset per lane: no 64-bit
    // address pairs live across the chain (spilled, they were reloaded from scratch before each store, and a scratch
    // reload waits for every store in flight: vmcnt counts in order).  Lanes without a frame get an offset behind
    // the buffer's size: the hardware drops their stores and returns 0 for their loads, no predicate needed.
    const unsigned fbase = STORE ? (fvalid ? (unsigned)nrow * (unsigned)(a.Rs * a.Fs) * (unsigned)sizeof(store_t) : WC_OOB) : 0u;
    const unsigned zs_off = fvalid ? ((unsigned)nrow * (unsigned)a.Rcap * LAT + 4u * q) * 4u : WC_OOB;      // Zs[nrow][r][4q..]
    const unsigned fr_off = (fvalid && q == 0) ? (unsigned)nrow * 4u : WC_OOB;                                  // [step][nrow] tables
    const unsigned rp_off = ((unsigned)nrow * LAT + 4u * q) * 4u;                                               // eps[step][nrow][4q..]
    int cur_src = a.nsamples;
    // ---- noise streams: this lane draws latents 4q..4q+3 (stream q) and 16+4q.. (stream 4+q) of its frame;
    // the streams are keyed by (utterance seed, frame inside the utterance, latent quad, chain call)
    Xs128 st0, st1;
    if (a.rng_mode == VAENMF_RNG_DEVICE) {
      const uint32_t floc = (uint32_t)(nrow - a.frame_off[utt]);
      int qs = q;                                          // (opaque: the key words are built here, once per wave tile, not hoisted and spilled)
      asm volatile("" : "+v"(qs));
      st0 = xs_seed(a.utt_seed[utt], floc, (uint32_t)qs, a.call);
      st1 = xs_seed(a.utt_seed[utt], floc, (uint32_t)(4 + qs), a.call);
    }

    // E(z) = sum_f [log Vx + X2 / Vx] of this lane's frame (all lanes of the frame get the sum).  fp64 across
    // the tiles: the reference sums per-bin DIFFERENCES of two states (mcem.py:415-416); summing each state
    // separately needs the extra bits.
    WC_STAMP_DECL
    auto energy = [&](const float (&zz)[8], int slot, auto dost) -> double {
      constexpr bool DOST = STORE && decltype(dost)::value;
      // byte offset of this lane's part of the row: frame block + slot + lane part; the tile is the instruction's
      // immediate offset.  The slot is folded in here and NOT passed as the scalar offset of the buffer store: hipcc
      // 7.2 leaves out the wait states a store of more than 8 bytes needs before its data registers are rewritten when
      // the scalar offset is a register (GCNHazardRecognizer assumes no hazard then; gfx950 has it: sporadic garbage
      // in the first dword of the 16-byte stores).
      const unsigned voff = DOST ? fbase + (unsigned)slot * (unsigned)a.Fs * (unsigned)sizeof(store_t) : 0u;
      // B fragments of a layer's input: k-step s <-> feature tiles 2s (elements 0..3) and 2s+1 (elements 4..7)
      u32x4 bh[NK], bl[NK], ch[NK], cl[NK];
      auto put = [&](u32x4 (&dh)[NK], u32x4 (&dl)[NK], int t, const f32x4 h) {      // tile t of the next layer's input
        dh[t >> 1][2 * (t & 1)] = pk2(h[0], h[1]);
        dh[t >> 1][2 * (t & 1) + 1] = pk2(h[2], h[3]);
        if (SPLIT) {
          dl[t >> 1][2 * (t & 1)] = pk2(h[0] - bf_lo(dh[t >> 1][2 * (t & 1)]), h[1] - bf_hi(dh[t >> 1][2 * (t & 1)]));
          dl[t >> 1][2 * (t & 1) + 1] = pk2(h[2] - bf_lo(dh[t >> 1][2 * (t & 1) + 1]), h[3] - bf_hi(dh[t >> 1][2 * (t & 1) + 1]));
        }
      };
      // One layer, software-pipelined over its output tiles: the weight fragments of tile t+1 are requested, the
      // MFMAs of tile t issued and the epilogue of tile t-1 computed in the same scheduling region, so a
      // wavefront covers its own LDS and MFMA latencies with epilogue work.
      //   NKS k-steps; frag(t, s, hi, lo) loads; bias(t); bop(s, hi, lo) the input fragments; epi(t, acc)
      //   frag_lo(t, s, lo): the lo fragments (bf16x3 mode), PFL tiles ahead
      auto run_layer = [&](auto nks_c, auto ntiles_c, auto pfl_c, auto frag, auto frag_lo, auto frag_lo_pro, auto bias, auto bop, auto epi) {
        constexpr int NKS = decltype(nks_c)::value, N = decltype(ntiles_c)::value;
        // VN_PF tiles of weight fragments (and bias) in flight ahead of the MFMAs that use them
        constexpr int PF = VN_PF, NB = PF + 1;
        constexpr int PFL = SPLIT ? decltype(pfl_c)::value : 0, NBL = PFL + 1;
        bf16x8 wh[NB][NKS], wl[NBL][NKS];
        f32x4 acc[2], bq[NB];
#pragma unroll
        for (int p = 0; p < PFL; ++p)
          if (p < N) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) frag_lo_pro(p, s, wl[p % NBL][s]);
          }
#pragma unroll
        for (int p = 0; p < PF; ++p)
          if (p < N) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) frag(p, s, wh[p % NB][s]);
            bq[p % NB] = bias(p);
          }
#pragma unroll
        for (int t = 0; t <= N; ++t) {
          if (SPLIT && t + PFL < N && tile_on3(N, t + PFL)) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) frag_lo(t + PFL, s, wl[(t + PFL) % NBL][s]);
          }
          if (t + PF < N && tile_on3(N, t + PF)) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) frag(t + PF, s, wh[(t + PF) % NB][s]);
            bq[(t + PF) % NB] = bias(t + PF);
          }
          if (t < N && tile_on3(N, t)) {
            f32x4 ac = bq[t % NB];
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
              bf16x8 ah, al;
              bop(s, ah, al);
              ac = mma<SPLIT>(wh[t % NB][s], SPLIT ? wl[t % NBL][s] : wh[t % NB][s], ah, al, ac);
            }
            acc[t & 1] = ac;
          }
          if (t > 0 && tile_on3(N, t - 1)) epi(t - 1, acc[(t - 1) & 1]);
          // order inside the region: every MFMA of the tile followed by a share of the previous tile's epilogue.  A wave
          // whose next instruction is an MFMA waiting for the matrix pipe (or for the accumulator of the MFMA before it)
          // holds the SIMD's vector issue port -- measured: a wave of back-to-back MFMAs starves its SIMD partner's VALU
          // stream (tools/ubench/overlap.hip) -- so the MFMAs are spaced by VALU work of the same wave
          if (NKS > 1 && t > 0 && t < N) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
#if VN_TPM > 0
              __builtin_amdgcn_sched_group_barrier(0x400, VN_TPM, 0);     // transcendentals first (see VN_TPM)
              __builtin_amdgcn_sched_group_barrier(0x002, VN_VPER - VN_TPM, 0);
#else
              __builtin_amdgcn_sched_group_barrier(0x402, VN_VPER, 0);    // VALU / transcendental
#endif
            }
          }
          if (VN_TPR <= 1 || (t % VN_TPR) == VN_TPR - 1 || t == N) VN_SB;      // VN_TPR tiles per scheduling region
        }
      };
      // bf16x3 mode, W3-lo streamed from L2: the first NHOIST tiles' lo fragments are requested here, two layers before their
      // use (the single wavefront of a SIMD otherwise sits out one L2 round trip per evaluation at the head of the output layer)
      constexpr int NHOIST = (SPLIT && !LOL && !M2) ? VN_PFL_HOIST : 0;      // (M2 keeps its 32 bias registers instead)
      bf16x8 lo_pre[NHOIST > 0 ? NHOIST : 1][NK];
      if (NHOIST > 0) {
#pragma unroll
        for (int t = 0; t < NHOIST; ++t)
#pragma unroll
          for (int s = 0; s < NK; ++s) lo_pre[t][s] = *reinterpret_cast<const bf16x8*>(w3g + (size_t)((t * NK + s) * 2 + 1) * 1024 + l16);
      }
      bf16x8 gfr[GT > 0 ? GT : 1][NK];                 // hi fragments of the bin tiles that live in global memory (L2)
      if (GT > 0) {
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
          for (int s = 0; s < NK; ++s) gfr[t][s] = *reinterpret_cast<const bf16x8*>(w3g + (size_t)((t * NK + s) * 2) * 1024 + l16);
      }
      // ---- layer 1: input = the latents of this lane's frame (one k-step)
      {
        bh[0][0] = pk2(zz[0], zz[1]); bh[0][1] = pk2(zz[2], zz[3]); bh[0][2] = pk2(zz[4], zz[5]); bh[0][3] = pk2(zz[6], zz[7]);
        if (SPLIT) {
          bl[0][0] = pk2(zz[0] - bf_lo(bh[0][0]), zz[1] - bf_hi(bh[0][0]));
          bl[0][1] = pk2(zz[2] - bf_lo(bh[0][1]), zz[3] - bf_hi(bh[0][1]));
          bl[0][2] = pk2(zz[4] - bf_lo(bh[0][2]), zz[5] - bf_hi(bh[0][2]));
          bl[0][3] = pk2(zz[6] - bf_lo(bh[0][3]), zz[7] - bf_hi(bh[0][3]));
        }
        run_layer(std::integral_constant<int, 1>{}, std::integral_constant<int, NTH>{}, std::integral_constant<int, VN_PF>{},
                  [&](int t, int, bf16x8& hi) { hi = *reinterpret_cast<const bf16x8*>(smem + L::W1 + t * PARTS * 1024 + l16); },
                  [&](int t, int, bf16x8& lo) { lo = *reinterpret_cast<const bf16x8*>(smem + L::W1 + t * PARTS * 1024 + (SPLIT ? 1024 : 0) + l16); },
                  [&](int t, int, bf16x8& lo) { lo = *reinterpret_cast<const bf16x8*>(smem + L::W1 + t * PARTS * 1024 + (SPLIT ? 1024 : 0) + l16); },
                  [&](int t) {
                    if (B1L) {
                      const u32x2 w = *reinterpret_cast<const u32x2*>(b1stash + t * 512);
                      return f32x4{bf_lo(w[0]), bf_hi(w[0]), bf_lo(w[1]), bf_hi(w[1])};
                    }
                    return M2 ? b1r[(M2 && !B1L) ? t : 0] : *reinterpret_cast<const f32x4*>(b1l + 16 * t + 4 * q);
                  },
                  [&](int, bf16x8& hi, bf16x8& lo) { hi = __builtin_bit_cast(bf16x8, bh[0]); lo = SPLIT ? __builtin_bit_cast(bf16x8, bl[0]) : hi; },
                  [&](int t, const f32x4 acc) { put(ch, cl, t, tanh4(acc)); });
      }
      WC_STAMP(1);
      // ---- layer 2 (a decoder with ONE hidden layer, models.py:107-121 with h_dim = [128], hands layer 1's output on)
      if (a.one_hidden) {
#pragma unroll
        for (int s = 0; s < NK; ++s) { bh[s] = ch[s]; if (SPLIT) bl[s] = cl[s]; }
      } else
      run_layer(std::integral_constant<int, NK>{}, std::integral_constant<int, NTH>{}, std::integral_constant<int, VN_PF>{},
                [&](int t, int s, bf16x8& hi) { hi = *reinterpret_cast<const bf16x8*>(smem + L::W2 + (t * NK + s) * PARTS * 1024 + l16); },
                [&](int t, int s, bf16x8& lo) { lo = *reinterpret_cast<const bf16x8*>(smem + L::W2 + (t * NK + s) * PARTS * 1024 + (SPLIT ? 1024 : 0) + l16); },
                [&](int t, int s, bf16x8& lo) { lo = *reinterpret_cast<const bf16x8*>(smem + L::W2 + (t * NK + s) * PARTS * 1024 + (SPLIT ? 1024 : 0) + l16); },
                [&](int t) { return *reinterpret_cast<const f32x4*>(b2l + 16 * t + 4 * q); },
                [&](int s, bf16x8& hi, bf16x8& lo) { hi = __builtin_bit_cast(bf16x8, ch[s]); lo = SPLIT ? __builtin_bit_cast(bf16x8, cl[s]) : hi; },
                [&](int t, const f32x4 acc) { put(bh, bl, t, tanh4(acc)); });
      WC_STAMP(2);
      // ---- output layer: each finished tile straight into the energy epilogue
      double e = 0.0;
      float ef = 0.f;
      f32x2 pl2 = {0.f, 0.f}, px2 = {0.f, 0.f};
      (void)ef;
      unsigned pk_even0 = 0, pk_even1 = 0;
      run_layer(std::integral_constant<int, NK>{}, std::integral_constant<int, MAXT>{}, std::integral_constant<int, (SPLIT && !LOL) ? VN_PFL : VN_PF>{},
                [&](int t, int s, bf16x8& hi) {
                  if (GT > 0 && t < GT) hi = gfr[t < GT ? t : 0][s];
                  else if (HIALL || t - GT < n_hi) hi = *reinterpret_cast<const bf16x8*>(smem + L::W3 + ((t - GT) * NK + s) * 1024 + l16);
                  else hi = *reinterpret_cast<const bf16x8*>(w3g + (size_t)((t * NK + s) * 2) * 1024 + l16);
                },
                [&](int t, int s, bf16x8& lo) {
                  if (LOL) lo = *reinterpret_cast<const bf16x8*>(w3lo_lds + (t * NK + s) * 1024 + l16);
                  else lo = *reinterpret_cast<const bf16x8*>(w3g + (size_t)((t * NK + s) * 2 + 1) * 1024 + l16);
                },
                [&](int t, int s, bf16x8& lo) {           // the first tiles' lo fragments: requested at the top of the evaluation (VN_PFL_HOIST)
                  if (LOL) lo = *reinterpret_cast<const bf16x8*>(w3lo_lds + (t * NK + s) * 1024 + l16);
                  else if (t < NHOIST) lo = lo_pre[t < NHOIST ? t : 0][s];
                  else lo = *reinterpret_cast<const bf16x8*>(w3g + (size_t)((t * NK + s) * 2 + 1) * 1024 + l16);
                },
                [&](int t) { return *reinterpret_cast<const f32x4*>(b3l + 16 * t + 4 * q); },
                [&](int s, bf16x8& hi, bf16x8& lo) { hi = __builtin_bit_cast(bf16x8, bh[s]); lo = SPLIT ? __builtin_bit_cast(bf16x8, bl[s]) : hi; },
                [&](int t, const f32x4 acc) {
                  // two bins at a time: log Vx0 + log Vx1 = log(Vx0 Vx1), X0/Vx0 + X1/Vx1 = (X0 Vx1 + X1 Vx0)/(Vx0 Vx1)
                  f32x4 ev;
#pragma unroll
                  for (int j = 0; j < 4; ++j) ev[j] = fast_exp(acc[j]);
#if VN_PACKED
                  {
                    // pairs (bin 0, bin 2) and (bin 1, bin 3): the two pairs are the two lanes of the packed instructions
                    const f32x2 g2 = {gn, gn};
                    const f32x2 v0 = g2 * ev.lo + vb[t].lo, v1 = g2 * ev.hi + vb[t].hi;
                    const f32x2 pp = v0 * v1;
                    pl2 += f32x2{fast_log2(pp[0]), fast_log2(pp[1])};
                    const f32x2 rc = {fast_rcp(pp[0]), fast_rcp(pp[1])};
                    px2 = (x2[t].lo * v1 + x2[t].hi * v0) * rc + px2;
                  }
#else
                  float pl = 0.f, px = 0.f;
#pragma unroll
                  for (int j = 0; j < 4; j += 2) {
                    const float v0 = gn * ev[j] + vb[t][j];
                    const float v1 = gn * ev[j + 1] + vb[t][j + 1];
                    const float pp = v0 * v1;
                    pl += fast_log2(pp);
                    px += (x2[t][j] * v1 + x2[t][j + 1] * v0) * fast_rcp(pp);
                  }
#endif
                  if (DOST) {
                    if (SPLIT) {
                      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ev), vrs, voff + 16u * q + 64u * t, 0, VN_ST_AUX);
                    } else {
                      const unsigned p0 = pk2(ev[0], ev[1]), p1 = pk2(ev[2], ev[3]);
                      if (t < Tm) {
                        if ((t & 1) == 0) { pk_even0 = p0; pk_even1 = p1; }
                        else __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk_even0, pk_even1, p0, p1}, vrs, voff + 16u * q + 64u * (t >> 1), 0, VN_ST_AUX);
                      } else {
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{p0, p1}, vrs, voff + 8u * q + 32u * t, 0, VN_ST_AUX);
                      }
                    }
                  }
#if VN_PACKED
                  if ((t & 1) == 1) {          // fp32 over two tiles, fp64 across them
                    e += (double)((pl2[0] + pl2[1]) * LN2_F + (px2[0] + px2[1]));
                    pl2 = px2 = f32x2{0.f, 0.f};
                  }
#else
                  ef += pl * LN2_F + px;
                  if ((t & 1) == 1) { e += (double)ef; ef = 0.f; }
#endif
                });
#if VN_PACKED
      e += (double)((pl2[0] + pl2[1]) * LN2_F + (px2[0] + px2[1]));
#else
      e += (double)ef;
#endif
      WC_STAMP(3);
#ifdef VN_EXP_F32SUM
      return (double)sum_rows4((float)e);
#elif defined(VN_EXP_PRIO)
      __builtin_amdgcn_s_setprio(3);
      const double r_ = sum_rows4_d(e);
      return r_;
#else
      return sum_rows4_d(e);
#endif
    };

    double Ecur = 0.0;
    // it = -1 evaluates the initial state (mcem.py:392-400); it >= 0 are the MH steps.  With the store on and a
    // burn-in, one more pass after the burn-in re-evaluates the state the chain is in (nothing drawn, nothing
    // decided) so that its variances are on record in slot R.
    const bool reeval = STORE && a.burnin > 0;
    // retire the prologue's loads here: a counted vmcnt wait for them placed inside the loop would, on every later
    // step, wait for the previous step's stores instead (vmcnt counts loads and stores in order)
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0), gfx9 encoding
    for (int it = -1; it < S + (reeval ? 1 : 0); ++it) {
      asm volatile("" ::: "memory");            // the LDS-resident weights are re-read every step (no hoisting into registers)
      const bool re = reeval && it == a.burnin;
      const int m = (reeval && it > a.burnin) ? it - 1 : it;
      const bool step = m >= 0 && !re;
      // ---- noise of this step and the proposal Z' = Z + sqrt(var) randn (mcem.py:407)
      float zp[8];
      float lu = 0.f;                                   // log U(0,1) of the frame (mcem.py:420), lanes q = 0
      if (step) {
        // (the proposal is finished inside each generator branch and pinned there: were the two branches to join
        // with the noise still in flight, the compiler would place the replay loads' vmcnt wait on the common path,
        // where it drains the variance stores of the device-generator run: vmcnt counts loads and stores in order)
        if (a.rng_mode == VAENMF_RNG_DEVICE) {
          const f32x4 e0 = normal4(st0);
          const float uu = q == 0 ? uniform01(st0) : 0.5f;
          const f32x4 e1 = normal4(st1);
#pragma unroll
          for (int t = 0; t < 4; ++t) { zp[t] = z[t] + a.sd * e0[t]; zp[4 + t] = z[4 + t] + a.sd_hi * e1[t]; }
          lu = q == 0 ? fast_log(uu) : 0.f;
        } else {
          const unsigned so = (unsigned)m * (unsigned)a.NT;      // step offset in rows
          const f32x4 e0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(eps_rs, rp_off + so * (LAT * 4u), 0, 0));
          const f32x4 e1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(eps_rs, rp_off + so * (LAT * 4u) + 64u, 0, 0));
          const float uu = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(u_rs, (unsigned)nrow * 4u + so * 4u, 0, 0));
#pragma unroll
          for (int t = 0; t < 4; ++t) { zp[t] = z[t] + a.sd * e0[t]; zp[4 + t] = z[4 + t] + a.sd_hi * e1[t]; }
          lu = q == 0 ? fast_log(uu) : 0.f;
#pragma unroll
          for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(zp[t]));
          asm volatile("" : "+v"(lu));
        }
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) zp[t] = z[t];
      }
#ifdef VN_EXP_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      WC_STAMP(0);
      const int slot = !STORE ? -1 : (re ? a.nsamples : (m >= a.burnin ? m - a.burnin : ((m < 0 && a.burnin == 0) ? a.nsamples : -1)));
      double Ep;
      if (STORE && slot >= 0) Ep = energy(zp, slot, std::true_type{});
      else Ep = energy(zp, slot, std::false_type{});
      WC_STAMP(5);
      if (re) continue;
      float pr = 0.f;                                   // .5 * sum(Z^2 - Z'^2)   (mcem.py:417)
#pragma unroll
      for (int j = 0; j < 8; ++j) pr += z[j] * z[j] - zp[j] * zp[j];
      pr = sum_rows4(pr);
      lu = sum_rows4(lu);                               // every lane of the frame gets log u
      WC_STAMP(6);
      const float accp = (float)(Ecur - Ep) + 0.5f * pr;
      const bool ok = m < 0 || lu < accp;               // mcem.py:420
      if (a.acc_out && m >= 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, accp), acc_rs, fr_off + (unsigned)m * (unsigned)a.NT * 4u, 0, 0);
      // mcem.py:429-433, as selects (no divergent branch in the loop)
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = ok ? zp[j] : z[j];
      Ecur = ok ? Ep : Ecur;
      WC_STAMP(7);
      if (m >= a.burnin) {                              // (uniform) mcem.py:435-437
        const unsigned r = (unsigned)(m - a.burnin);
        if (STORE) {
          cur_src = ok ? (int)r : cur_src;
          __builtin_amdgcn_raw_buffer_store_b32((unsigned)cur_src, src_rs, fr_off + r * (unsigned)a.NT * 4u, 0, 0);
        }
        const unsigned zo = zs_off + r * (LAT * 4u);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{z[0], z[1], z[2], z[3]}), zs_rs, zo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{z[4], z[5], z[6], z[7]}), zs_rs, zo + 64u, 0, 0);
      }
      WC_STAMP(4);
    }
    WC_STAMP_FLUSH;
    if (a.update_Z) {                                   // self.Z = last draw (mcem.py:466); idle lanes: offset behind the buffer
      const unsigned zo = fvalid ? rp_off : WC_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{z[0], z[1], z[2], z[3]}), z_rs, zo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{z[4], z[5], z[6], z[7]}), z_rs, zo + 64u, 0, 0);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Small batches (no more wave tiles than CUs: one utterance through the drop-in classes is 32 tiles): FOUR wavefronts,
// one per SIMD of a CU, share one tile of 16 frames.  bf16 mode, 17 bin tiles (F = 257..272).  Every wavefront draws the
// noise and takes the accept decision redundantly (same keys, same instructions: same bits).  Of each hidden layer it
// computes TWO of the eight feature tiles -- in the accumulator = operand layout of this file exactly ONE k-step of the
// next layer's input fragments, 16 bytes per lane -- and the four k-steps are exchanged through LDS; of the output layer it
// owns TWO of the eight bin-tile pairs (wavefront 0 also the odd 17th tile): X2 / Vb of 4-5 tiles instead of 17, a quarter
// of the MFMAs, transcendentals and row stores.  The per-pair fp32 energy sums go through LDS as well and every wavefront
// adds all nine in the order the one-wavefront kernel adds them, so the acceptance decisions -- and with them every
// result -- are bit-identical to wchain_kernel's (tested).  Three workgroup barriers per evaluation.  A lone wavefront per
// SIMD needed 7 450 cycles per evaluation for the whole decoder.
// ---------------------------------------------------------------------------------------------------------------
template <int MAXT, bool STORE, bool M2>
__global__ __launch_bounds__(256, 1) void wchain4_kernel(const WcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = (MAXT - 1) / 2;                     // bin-tile pairs (8 / 16); tile MAXT-1 is the odd last one
  constexpr int PPW = NP / 4, NOWN = 2 * PPW + 1;        // pairs and tiles per wavefront (wavefront 0 owns the last tile too)
  constexpr bool B1BF = MAXT == 17;                      // M2 bias rows as wchain_kernel holds them for this shape (bf16 at 8 wavefronts, fp32 at 4)
  static_assert(NP % 4 == 0, "pairs must split over four wavefronts");
  // LDS: exchange areas only -- hidden activations u32x4 [4][64] per hidden layer, pair energies float [NP + 1][64]
  u32x4* xh1 = reinterpret_cast<u32x4*>(smem);
  u32x4* xh2 = xh1 + 4 * 64;
  float* exl = reinterpret_cast<float*>(xh2 + 4 * 64);

  const int lane = threadIdx.x & 63, q = lane >> 4, c = lane & 15;
  const unsigned l16 = (unsigned)lane * 16u;
  const int S = a.nsamples + a.burnin;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // owned bin tiles: the pairs wave, wave + 4, ... (tiles 2p, 2p+1); wavefront 0 also tile MAXT-1
  auto tile_of = [&](int i) { return i < 2 * PPW ? 2 * (wave + 4 * (i >> 1)) + (i & 1) : MAXT - 1; };
  auto own = [&](int i) { return i < 2 * PPW || wave == 0; };

  // ---- this wavefront's weight fragments and biases, in registers for the whole launch (fragment order
  // [tile][kstep][hi/lo][lane][8], plan.hip): 2 hidden tiles per hidden layer, NOWN bin tiles -- no LDS-resident weights,
  // no weight loads inside the chain
  bf16x8 w1r[2], w2r[2][NK], w3r[NOWN][NK];
  f32x4 bias1[2], bias2[2], bias3[NOWN];
  {
    const char* w1g = reinterpret_cast<const char*>(a.w1f);
    const char* w2g = reinterpret_cast<const char*>(a.w2f);
    const char* w3g = reinterpret_cast<const char*>(a.w3f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = 2 * wave + i;
      w1r[i] = *reinterpret_cast<const bf16x8*>(w1g + (size_t)(t * 2) * 1024 + l16);
#pragma unroll
      for (int s = 0; s < NK; ++s) w2r[i][s] = *reinterpret_cast<const bf16x8*>(w2g + (size_t)((t * NK + s) * 2) * 1024 + l16);
      bias1[i] = *reinterpret_cast<const f32x4*>(a.b1 + 16 * t + 4 * q);
      bias2[i] = *reinterpret_cast<const f32x4*>(a.b2 + 16 * t + 4 * q);
    }
#pragma unroll
    for (int i = 0; i < NOWN; ++i) {
      const int t = tile_of(i);
#pragma unroll
      for (int s = 0; s < NK; ++s) w3r[i][s] = *reinterpret_cast<const bf16x8*>(w3g + (size_t)((t * NK + s) * 2) * 1024 + l16);
      bias3[i] = *reinterpret_cast<const f32x4*>(a.b3 + 16 * t + 4 * q);
    }
  }

  __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(a.VsS, 0, STORE ? (int)a.VsS_bytes : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t zs_rs = __builtin_amdgcn_make_buffer_rsrc(a.Zs, 0, a.Zs ? (int)((unsigned)a.NT * (unsigned)a.Rcap * LAT * 4u) : 0, 0x00020000);   // (no Zs: the sample stores fall outside the resource and are dropped)
  __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(a.src, 0, STORE ? (int)((unsigned)a.NT * (unsigned)a.Rs * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t acc_rs = __builtin_amdgcn_make_buffer_rsrc(a.acc_out, 0, a.acc_out ? (int)((unsigned)a.NT * (unsigned)S * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t eps_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.eps), 0, a.eps ? (int)((unsigned)a.NT * (unsigned)S * LAT * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t x2in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X2), 0, (int)((unsigned)a.NT * (unsigned)a.Fs * 4u), 0x00020000);
  __amdgpu_buffer_rsrc_t vbin_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Vb), 0, a.Vb ? (int)((unsigned)a.NT * (unsigned)a.Fs * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W), 0, a.W ? (int)((unsigned)a.n_utts * (unsigned)a.Fs * (unsigned)a.Kp * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Ht), 0, a.Ht ? (int)((unsigned)a.NT * (unsigned)a.Kp * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc(a.Z, 0, (int)((unsigned)a.NT * LAT * 4u), 0x00020000);
  __amdgpu_buffer_rsrc_t b1in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B1), 0, M2 ? (int)((unsigned)a.NT * HID * 4u) : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.u), 0, a.u ? (int)((unsigned)a.NT * (unsigned)S * 4u) : 0, 0x00020000);

  for (int wt = blockIdx.x; wt < a.n_wtiles; wt += (int)gridDim.x) {
    const int utt = a.wt_utt[wt], n0 = a.wt_n0[wt], cnt = a.wt_cnt[wt];
    const bool fvalid = c < cnt;
    const int nrow = n0 + (fvalid ? c : cnt - 1);
    const float gn = a.g[nrow];
    f32x4 x2[NOWN], vb[NOWN];
    {
      const unsigned rowF = (unsigned)nrow * (unsigned)a.Fs, uF = (unsigned)utt * (unsigned)a.Fs;
#pragma unroll
      for (int i = 0; i < NOWN; ++i) {
        x2[i] = f32x4{0, 0, 0, 0};
        vb[i] = f32x4{1, 1, 1, 1};
        if (own(i)) {
          const int t = tile_of(i);
          const int f0 = t < MAXT - 1 ? 32 * (t >> 1) + 8 * q + 4 * (t & 1) : 16 * t + 4 * q;      // the chain's bin order (see the file header)
          f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x2in_rs, (rowF + (unsigned)f0) * 4u, 0, 0));
          f32x4 v = {0, 0, 0, 0};
          if (a.Vb) {
            v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vbin_rs, (rowF + (unsigned)f0) * 4u, 0, 0));
          } else {
            for (int k = 0; k < a.Kp; k += 4) {          // the same order of operations as wchain_kernel's prologue
              const f32x4 h4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(h_rs, ((unsigned)nrow * (unsigned)a.Kp + (unsigned)k) * 4u, 0, 0));
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const f32x4 w4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, ((uF + (unsigned)(f0 + j)) * (unsigned)a.Kp + (unsigned)k) * 4u, 0, 0));
                v[j] += w4[0] * h4[0] + w4[1] * h4[1] + w4[2] * h4[2] + w4[3] * h4[3];
              }
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (f0 + j >= a.F) { xv[j] = 0.f; v[j] = 1.f; }
          x2[i] = xv;
          vb[i] = v;
        }
      }
    }
    f32x4 b1r[2];                                          // layer-1 accumulator init of this wavefront's two hidden tiles
    b1r[0] = bias1[0]; b1r[1] = bias1[1];
    if (M2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b1in_rs, ((unsigned)nrow * HID + 4u * (unsigned)q) * 4u + 64u * (unsigned)(2 * wave + i), 0, 0));
        if (B1BF) {      // (rounded to bf16 like the rows wchain_kernel parks in LDS at 8 wavefronts: the same bits in both kernels)
          const unsigned w0 = pk2(v[0], v[1]), w1 = pk2(v[2], v[3]);
          b1r[i] = f32x4{bf_lo(w0), bf_hi(w0), bf_lo(w1), bf_hi(w1)};
        } else b1r[i] = v;
      }
    }
    float z[8];
    {
      const unsigned zo = ((unsigned)nrow * LAT + 4u * (unsigned)q) * 4u;
      const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rs, zo, 0, 0));
      const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(z_rs, zo + 64u, 0, 0));
#pragma unroll
      for (int t = 0; t < 4; ++t) { z[t] = lo[t]; z[4 + t] = hi[t]; }
    }
    // rows: every wavefront writes its own tiles' part; samples, slot map, acceptances and Z: wavefront 0 alone
    const bool w0 = wave == 0;
    const unsigned fbase = STORE ? (fvalid ? (unsigned)nrow * (unsigned)(a.Rs * a.Fs) * 2u : WC_OOB) : 0u;
    const unsigned zs_off = (fvalid && w0) ? ((unsigned)nrow * (unsigned)a.Rcap * LAT + 4u * q) * 4u : WC_OOB;
    const unsigned fr_off = (fvalid && w0 && q == 0) ? (unsigned)nrow * 4u : WC_OOB;
    const unsigned rp_off = ((unsigned)nrow * LAT + 4u * q) * 4u;
    int cur_src = a.nsamples;
    Xs128 st0, st1;
    if (a.rng_mode == VAENMF_RNG_DEVICE) {
      const uint32_t floc = (uint32_t)(nrow - a.frame_off[utt]);
      st0 = xs_seed(a.utt_seed[utt], floc, (uint32_t)q, a.call);
      st1 = xs_seed(a.utt_seed[utt], floc, (uint32_t)(4 + q), a.call);
    }

    auto energy = [&](const float (&zz)[8], int slot, auto dost) -> double {
      constexpr bool DOST = STORE && decltype(dost)::value;
      const unsigned voff = DOST ? fbase + (unsigned)slot * (unsigned)a.Fs * 2u : 0u;
      u32x4 bh[NK], ch[NK];
      // hidden layers: this wavefront computes feature tiles 2w, 2w+1 -- after tanh and the bf16 rounding exactly k-step w of
      // the next layer's input fragments (file header) -- and the four k-steps are exchanged through LDS
      u32x4 mine;
      auto put2 = [&](int i, const f32x4 h) { mine[2 * i] = pk2(h[0], h[1]); mine[2 * i + 1] = pk2(h[2], h[3]); };
      auto exchange = [&](u32x4* xh, u32x4 (&dst)[NK]) {
        xh[wave * 64 + lane] = mine;
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < NK; ++s2) dst[s2] = xh[s2 * 64 + lane];
      };
      {
        const bf16x8 zin = __builtin_bit_cast(bf16x8, u32x4{pk2(zz[0], zz[1]), pk2(zz[2], zz[3]), pk2(zz[4], zz[5]), pk2(zz[6], zz[7])});
        f32x4 ac[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) ac[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[i], zin, b1r[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) put2(i, tanh4(ac[i]));
      }
      exchange(xh1, ch);
      if (a.one_hidden) {
#pragma unroll
        for (int s = 0; s < NK; ++s) bh[s] = ch[s];
      } else {
        f32x4 ac[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ac[i] = bias2[i];
#pragma unroll
          for (int s = 0; s < NK; ++s) ac[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2r[i][s], __builtin_bit_cast(bf16x8, ch[s]), ac[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) put2(i, tanh4(ac[i]));
        exchange(xh2, bh);
      }
      // ---- this wavefront's share of the output layer: tile i+1's MFMAs are issued before tile i's epilogue
      f32x2 pl2 = {0.f, 0.f}, px2 = {0.f, 0.f};
      float pv[PPW + 1];
#pragma unroll
      for (int j = 0; j <= PPW; ++j) pv[j] = 0.f;
      unsigned pk_even0 = 0, pk_even1 = 0;
      auto mm3 = [&](int i) {
        f32x4 ac = bias3[i];
#pragma unroll
        for (int s = 0; s < NK; ++s) ac = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3r[i][s], __builtin_bit_cast(bf16x8, bh[s]), ac, 0, 0, 0);
        return ac;
      };
      auto epi = [&](int i, const f32x4 acc) {
        f32x4 ev;
#pragma unroll
        for (int j = 0; j < 4; ++j) ev[j] = fast_exp(acc[j]);
        {
          // two bins at a time, exactly as wchain_kernel's epilogue
          const f32x2 g2 = {gn, gn};
          const f32x2 v0 = g2 * ev.lo + vb[i].lo, v1 = g2 * ev.hi + vb[i].hi;
          const f32x2 pp = v0 * v1;
          pl2 += f32x2{fast_log2(pp[0]), fast_log2(pp[1])};
          const f32x2 rc = {fast_rcp(pp[0]), fast_rcp(pp[1])};
          px2 = (x2[i].lo * v1 + x2[i].hi * v0) * rc + px2;
        }
        if (DOST) {
          const unsigned p0 = pk2(ev[0], ev[1]), p1 = pk2(ev[2], ev[3]);
          const unsigned t = (unsigned)tile_of(i);
          if (i < 2 * PPW) {
            if ((i & 1) == 0) { pk_even0 = p0; pk_even1 = p1; }
            else __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk_even0, pk_even1, p0, p1}, vrs, voff + 16u * q + 64u * (t >> 1), 0, VN_ST_AUX);
          } else {
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{p0, p1}, vrs, voff + 8u * q + 32u * t, 0, VN_ST_AUX);
          }
        }
        if ((i & 1) == 1 || i == 2 * PPW) {     // fp32 over a pair's two tiles (the odd last tile alone), as wchain_kernel
          pv[i >> 1] = (pl2[0] + pl2[1]) * LN2_F + (px2[0] + px2[1]);
          pl2 = px2 = f32x2{0.f, 0.f};
        }
      };
      {
        f32x4 acn = mm3(0);
#pragma unroll
        for (int i = 0; i < NOWN; ++i) {
          const f32x4 acc = acn;
          if (i + 1 < NOWN && own(i + 1)) acn = mm3(i + 1);
          if (own(i)) epi(i, acc);
        }
      }
      // ---- exchange: NP + 1 fp32 values per lane, added in fp64 in the one-wavefront kernel's order (pairs 0.., last tile)
      // (single buffers: between a read of an exchange area and its next write lie the two other barriers of the evaluation)
      float* ex = exl + lane;
#pragma unroll
      for (int j = 0; j < PPW; ++j) ex[(wave + 4 * j) * 64] = pv[j];
      if (wave == 0) ex[NP * 64] = pv[PPW];
      __syncthreads();
      double e = 0.0;
#pragma unroll
      for (int p = 0; p <= NP; ++p) e += (double)ex[p * 64];
      return sum_rows4_d(e);
    };

    double Ecur = 0.0;
    const bool reeval = STORE && a.burnin > 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): retire the prologue's loads (see wchain_kernel)
    for (int it = -1; it < S + (reeval ? 1 : 0); ++it) {
      const bool re = reeval && it == a.burnin;
      const int m = (reeval && it > a.burnin) ? it - 1 : it;
      const bool step = m >= 0 && !re;
      float zp[8];
      float lu = 0.f;
      if (step) {
        if (a.rng_mode == VAENMF_RNG_DEVICE) {
          const f32x4 e0 = normal4(st0);
          const float uu = q == 0 ? uniform01(st0) : 0.5f;
          const f32x4 e1 = normal4(st1);
#pragma unroll
          for (int t = 0; t < 4; ++t) { zp[t] = z[t] + a.sd * e0[t]; zp[4 + t] = z[4 + t] + a.sd_hi * e1[t]; }
          lu = q == 0 ? fast_log(uu) : 0.f;
        } else {
          const unsigned so = (unsigned)m * (unsigned)a.NT;
          const f32x4 e0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(eps_rs, rp_off + so * (LAT * 4u), 0, 0));
          const f32x4 e1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(eps_rs, rp_off + so * (LAT * 4u) + 64u, 0, 0));
          const float uu = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(u_rs, (unsigned)nrow * 4u + so * 4u, 0, 0));
#pragma unroll
          for (int t = 0; t < 4; ++t) { zp[t] = z[t] + a.sd * e0[t]; zp[4 + t] = z[4 + t] + a.sd_hi * e1[t]; }
          lu = q == 0 ? fast_log(uu) : 0.f;
#pragma unroll
          for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(zp[t]));
          asm volatile("" : "+v"(lu));
        }
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) zp[t] = z[t];
      }
      const int slot = !STORE ? -1 : (re ? a.nsamples : (m >= a.burnin ? m - a.burnin : ((m < 0 && a.burnin == 0) ? a.nsamples : -1)));
      double Ep;
      if (STORE && slot >= 0) Ep = energy(zp, slot, std::true_type{});
      else Ep = energy(zp, slot, std::false_type{});
      if (re) continue;
      float pr = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) pr += z[j] * z[j] - zp[j] * zp[j];
      pr = sum_rows4(pr);
      lu = sum_rows4(lu);
      const float accp = (float)(Ecur - Ep) + 0.5f * pr;
      const bool ok = m < 0 || lu < accp;
      if (a.acc_out && m >= 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, accp), acc_rs, fr_off + (unsigned)m * (unsigned)a.NT * 4u, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = ok ? zp[j] : z[j];
      Ecur = ok ? Ep : Ecur;
      if (m >= a.burnin) {
        const unsigned r = (unsigned)(m - a.burnin);
        if (STORE) {
          cur_src = ok ? (int)r : cur_src;
          __builtin_amdgcn_raw_buffer_store_b32((unsigned)cur_src, src_rs, fr_off + r * (unsigned)a.NT * 4u, 0, 0);
        }
        const unsigned zo = zs_off + r * (LAT * 4u);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{z[0], z[1], z[2], z[3]}), zs_rs, zo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{z[4], z[5], z[6], z[7]}), zs_rs, zo + 64u, 0, 0);
      }
    }
    if (a.update_Z) {
      const unsigned zo = (fvalid && w0) ? rp_off : WC_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{z[0], z[1], z[2], z[3]}), z_rs, zo, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{z[4], z[5], z[6], z[7]}), z_rs, zo + 64u, 0, 0);
    }
  }
}

}  // namespace

// ============================================================================
// Host side
// ============================================================================
int vn_ensure_dyn_lds(const void* fn, int bytes);     // plan.hip: per-device hipFuncSetAttribute, checked

namespace {

constexpr int WC_LDS_LIMIT = 160 * 1024;

template <int MAXT, bool EXACT, bool SPLIT, bool STORE, int NWAVES, bool LOL, bool HIALL, bool M2, int GT = 0>
int wc_launch(const WcArgs& a, int grid, size_t lds, hipStream_t st) {
  auto* fn = wchain_kernel<MAXT, EXACT, SPLIT, STORE, NWAVES, LOL, HIALL, M2, GT>;
  if (int e = vn_ensure_dyn_lds((const void*)fn, WC_LDS_LIMIT)) return e;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(NWAVES * 64), lds, st, a);
  return 0;
}
// M1 runs NW1 wavefronts per workgroup; M2 the same (at 8 the per-frame layer-1 bias rows live in LDS: b1_lds)
template <int MAXT, bool EXACT, bool SPLIT, int NW1, bool LOL, int GT = 0>
int wc_launch_s(const WcArgs& a, int nwt, int n_sms, size_t lds, hipStream_t st) {
  const int nw = NW1;
#if VN_WC_SPREAD
  int grid = nwt;                                       // (wave tiles spread over the CUs first, then over a workgroup's wavefronts)
#else
  int grid = (nwt + nw - 1) / nw;
#endif
  if (grid > n_sms) grid = n_sms;                       // one workgroup per CU (LDS), wave tiles in a grid-stride loop
  if (a.B1) {
#if !defined(VN_DEV_FAST) || defined(VN_DEV_M2)
    return a.VsS ? wc_launch<MAXT, EXACT, SPLIT, true, NW1, LOL, true, true, GT>(a, grid, lds, st)
                 : wc_launch<MAXT, EXACT, SPLIT, false, NW1, LOL, true, true, GT>(a, grid, lds, st);
#else
    return -1;
#endif
  }
  return a.VsS ? wc_launch<MAXT, EXACT, SPLIT, true, NW1, LOL, true, false, GT>(a, grid, lds, st)
               : wc_launch<MAXT, EXACT, SPLIT, false, NW1, LOL, true, false, GT>(a, grid, lds, st);
}

}  // namespace

// dev / test switch: VAENMF_WCHAIN4=0 keeps small batches on wchain_kernel (read per call)
static bool wc4_enabled() {
  const char* e = getenv("VAENMF_WCHAIN4");
  return !(e && e[0] == '0');
}

// Shapes the wave-private chain covers (the rest runs engine.hip's team kernel): every W3 hi fragment in LDS
bool vn_wchain_supported(const vaenmf_plan* p) {
  // up to 17 bin tiles (F <= 272) in both precision modes; F = 513..528 (33 tiles, the reference scripts' 1024-pt STFT)
  // in bf16 mode with four tiles' fragments streamed from L2
  if (p->NT3c > 17 && !(p->NT3c == 33 && p->cfg.precision == VAENMF_PREC_BF16)) return false;
  const char* e = getenv("VAENMF_TEAM_CHAIN");          // dev / test override: force the team kernel of engine.hip
  return !(e && e[0] == '1');
}

// The wave chain addresses every per-frame buffer through a buffer resource with a 32-bit byte offset per lane and marks
// idle lanes with WC_OOB: each buffer must end below it (with a margin for the per-step / per-tile immediate offsets).
// Larger batches run the team kernel of engine.hip (64-bit addresses).
extern "C" int vaenmf_wchain_addressable(int64_t NT, int32_t Rcap, int32_t steps, int32_t Fs, int32_t Kp, int32_t n_utt, int32_t replay) {
  const uint64_t lim = 0xE0000000ull;
  const uint64_t nt = (uint64_t)(NT > 0 ? NT : 0), S = (uint64_t)(steps > 0 ? steps : 0);
  if (nt * (uint64_t)Rcap * LAT * 4 >= lim) return 0;                  // Zs
  if (nt * (uint64_t)Fs * 4 >= lim) return 0;                          // X2, Vb
  if (nt * (uint64_t)HID * 4 >= lim) return 0;                         // B1 (M2), Z
  if ((uint64_t)n_utt * (uint64_t)Fs * (uint64_t)Kp * 4 >= lim) return 0;   // W
  if (nt * (uint64_t)Kp * 4 >= lim) return 0;                          // Ht
  if (nt * S * 4 >= lim) return 0;                                     // acc_out, u
  if (replay && nt * S * LAT * 4 >= lim) return 0;                     // eps
  return 1;
}
bool vn_wchain_fits(const vaenmf_plan* p, const VnChainCall& cc) {
  return vaenmf_wchain_addressable(p->NT, cc.Rcap, cc.nsamples + cc.burnin, p->Fs, p->Kp, p->n_utt, cc.eps != nullptr) != 0;
}

int vn_launch_wchain(vaenmf_plan* p, const VnChainCall& cc, hipStream_t st) {
  VN_REQUIRE(vn_wchain_fits(p, cc), "wave chain: a buffer of this batch (%d frames) passes the kernel's 32-bit byte offsets", p->NT);
  const bool split = p->cfg.precision == VAENMF_PREC_BF16X3;
  WcArgs a = {};
  a.w1f = p->w1f; a.w2f = p->w2f; a.w3f = p->w3c; a.b1 = p->b1; a.b2 = p->b2; a.b3 = p->b3c;
  a.NT3 = p->NT3c; a.Tm = split ? 0 : ((p->NT3c - 1) & ~1); a.F = p->cfg.F; a.Fs = p->Fs;
  a.X2 = cc.X2; a.W = cc.W; a.Ht = cc.Ht; a.g = cc.g; a.B1 = cc.B1; a.Vb = p->Vb_ext;
  a.Z = cc.Z; a.Zs = cc.Zs; a.acc_out = cc.acc_out;
  a.VsS = cc.VsS; a.VsS_bytes = (unsigned)cc.VsS_bytes; a.src = cc.src; a.Rs = cc.Rs;
  a.wt_utt = p->d_wt_utt; a.wt_n0 = p->d_wt_n0; a.wt_cnt = p->d_wt_cnt; a.n_wtiles = p->n_wtiles;
  a.frame_off = p->d_frame_off; a.utt_seed = p->d_utt_seed; a.eps = cc.eps; a.u = cc.u;
  a.Kp = p->Kp; a.NT = p->NT; a.n_utts = p->n_utt; a.Rcap = cc.Rcap; a.nsamples = cc.nsamples; a.burnin = cc.burnin;
  a.rng_mode = cc.rng_mode; a.update_Z = cc.update_Z; a.call = cc.call; a.sd = cc.sd; a.sd_hi = cc.sd_hi; a.one_hidden = cc.one_hidden;
  a.n_hi_lds = p->NT3c;
  const size_t fixed = split ? WcLds<true>::fixed_bytes : WcLds<false>::fixed_bytes;
  constexpr int GT33 = 4;                                // F = 513: bin tiles whose fragments stay in global memory
  const size_t w3hi = (size_t)(p->NT3c == 33 ? p->NT3c - GT33 : p->NT3c) * NK * 1024;
  const bool lol = split && p->NT3c <= 5;              // bf16x3: the lo fragments of W3 fit in LDS up to 5 tiles (F <= 80), else they stream from L2
  constexpr int NW_BF16 = VN_WC_WAVES_BF16, NW_X3 = 4;
  const int nwaves = (split || p->NT3c == 33) ? 4 : NW_BF16;
  a.b1_lds = (int)(fixed + w3hi * (lol ? 2 : 1));
  const size_t lds = (size_t)a.b1_lds + ((cc.B1 && nwaves == 8) ? (size_t)nwaves * NTH * 512 : 0);
  VN_REQUIRE(lds <= (size_t)WC_LDS_LIMIT, "wave chain: %zu bytes of LDS needed", lds);
  // small batches (at most one wave tile per CU) in bf16 mode at 17 / 33 bin tiles: four wavefronts per wave tile (wchain4_kernel)
  if (!split && (p->NT3c == 17 || p->NT3c == 33) && p->n_wtiles <= p->n_sms && wc4_enabled()) {
    const size_t lds4 = 2 * 4 * 64 * 16 + 20 * 64 * 4;          // exchange areas only
    const int grid = p->n_wtiles;
    auto go = [&](auto* fn) -> int {
      hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds4, st, a);
      return 0;
    };
    a.w3f = p->w3c; a.b3 = p->b3c;
    int rc4;
    if (p->NT3c == 17)
      rc4 = cc.B1 ? (a.VsS ? go(wchain4_kernel<17, true, true>) : go(wchain4_kernel<17, false, true>))
                  : (a.VsS ? go(wchain4_kernel<17, true, false>) : go(wchain4_kernel<17, false, false>));
    else
      rc4 = cc.B1 ? (a.VsS ? go(wchain4_kernel<33, true, true>) : go(wchain4_kernel<33, false, true>))
                  : (a.VsS ? go(wchain4_kernel<33, true, false>) : go(wchain4_kernel<33, false, false>));
    if (rc4) return rc4;
    VN_CHECK_HIP(hipGetLastError());
    p->last_chain_kernel = 2;
    return 0;
  }
  // wavefronts per workgroup: 8 (two per SIMD, 256 registers each) in bf16 mode, 4 (512 registers) in bf16x3 mode
  int rc = -1;
  if (p->NT3c == 33)     rc = wc_launch_s<33, true, false, 4, false, GT33>(a, p->n_wtiles, p->n_sms, lds, st);
  else if (p->NT3c == 17) rc = split ? wc_launch_s<17, true, true, NW_X3, false>(a, p->n_wtiles, p->n_sms, lds, st) : wc_launch_s<17, true, false, NW_BF16, false>(a, p->n_wtiles, p->n_sms, lds, st);
  else if (p->NT3c == 5) rc = split ? wc_launch_s<5, true, true, NW_X3, true>(a, p->n_wtiles, p->n_sms, lds, st)   : wc_launch_s<5, true, false, NW_BF16, false>(a, p->n_wtiles, p->n_sms, lds, st);
#ifndef VN_DEV_FAST
  else if (p->NT3c < 5)  rc = split ? wc_launch_s<5, false, true, NW_X3, true>(a, p->n_wtiles, p->n_sms, lds, st)  : wc_launch_s<5, false, false, NW_BF16, false>(a, p->n_wtiles, p->n_sms, lds, st);
  else                   rc = split ? wc_launch_s<17, false, true, NW_X3, false>(a, p->n_wtiles, p->n_sms, lds, st) : wc_launch_s<17, false, false, NW_BF16, false>(a, p->n_wtiles, p->n_sms, lds, st);
#endif
  VN_REQUIRE(rc != -1, "wave chain: shape not compiled in (dev build)");
  if (rc) return rc;
  VN_CHECK_HIP(hipGetLastError());
  p->last_chain_kernel = 1;
  return 0;
}

#ifdef VN_STAMP
extern "C" int vaenmf_debug_stamps_wc(long long* out32, int reset) {
  long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wc_stamp), sizeof(h)) != hipSuccess) return -2;
  for (int i = 0; i < 32; ++i) out32[i] = h[i];
  if (reset) { long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wc_stamp), z, sizeof(z)); }
  return 0;
}
#endif
