// Shared declarations for libvaenmf.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <type_traits>
#include <vector>
#include "../../include/vaenmf.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- host side -------------------------------------------------------------
void vaenmf_set_error(const char* fmt, ...);
#define VN_CHECK_HIP(expr)                                                         \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      vaenmf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -2;                                                                   \
    }                                                                              \
  } while (0)
#define VN_REQUIRE(cond, ...)                \
  do {                                       \
    if (!(cond)) {                           \
      vaenmf_set_error(__VA_ARGS__);         \
      return -1;                             \
    }                                        \
  } while (0)

constexpr int VN_COST_CHUNK = 25;     // EM iterations whose per-frame cost sums the fused driver keeps before one reduction launch
constexpr int MAX_TILE_FRAMES = 64;   // MH chain: frames per workgroup = 32 per team (2 MFMA column groups of 16)
constexpr int LAT = 32;               // latent dimension handled by the MFMA path
constexpr int HID = 128;              // hidden width of both decoder layers

struct vaenmf_plan {
  vaenmf_config cfg;
  int Fs, Kp, NT3;           // padded bins, padded rank, feature tiles of 16 on the MFMA path of the last layer
  int Fm;                    // bins on the MFMA path: F-1 when F = 16k+1 (the odd last bin is computed in fp32 FMA), else F
  float* w3n;                // [HID] fp32 row F-1 of the last layer (odd last bin)
  int geom;                  // workgroup geometry (engine.hip: launch_decode)
  int nwaves;                // waves of a team that split the features
  int tile_frames;           // MH-chain frames per workgroup: 64 (2 teams of 4 waves) or 32 (1 team of 8)
  // decoder weights on the device, MFMA fragment order (see weights in plan.hip)
  __bf16 *w1f, *w2f, *w3f;   // [tile][kstep][part hi/lo][lane][8]
  float *b1, *b2, *b3;       // biases (b3 padded to 16*NT3)
  float* w1y;                // [H1][Dy] label columns of W1 (M2)
  int Dy;
  int Lz;                    // latent dimension of the model (16 or 32): latents Lz..31 of the MFMA path are zero padding
  bool one_hidden;           // decoder with ONE hidden layer (h_dim = [128]): layer 2 is skipped
  bool have_weights;
  const float* Vb_ext;       // caller-owned noise variance [NT][Fs] (noNMF variants) or null
  // bound batch
  int n_utt, NT, n_tiles;
  int32_t *d_frame_off;      // [n_utt+1]
  int32_t *d_tile_utt, *d_tile_n0, *d_tile_cnt;   // MH-chain tiles (<=32 frames, one utterance each)
  int32_t *d_frame_utt;      // [NT]
  int32_t *d_frame_loc;      // [NT] frame index inside its utterance
  // wave-private chain (chain.hip): all F bins on the MFMA path, W3 / b3 in the chain's bin order; wave tiles
  __bf16* w3c;               // [NT3c][kstep][part][lane][8]
  float* b3c;                // [16 NT3c], padding -200
  int NT3c;                  // ceil(F / 16)
  int32_t *d_wt_utt, *d_wt_n0, *d_wt_cnt;         // wave tiles (<= 16 frames of one utterance each)
  int n_wtiles;
  // tiles of <= 64 consecutive frames of one utterance (wstats_fused_kernel: one workgroup per tile) and the partial sums of
  // the W update, one [Fs][2 Kp] block per tile
  int32_t *d_t64_n0 = nullptr, *d_t64_cnt = nullptr, *d_t64_first = nullptr, *d_t64_g0 = nullptr;      // [n_t64], [n_t64], [n_utt+1]
  int n_t64 = 0;
  float* wpart64 = nullptr;
  size_t wpart16_groups = 0; // capacity of wpart16 in 16-frame groups
  float* wpart16 = nullptr;  // [n_sms][2 Kp][Fs]: per-16-frame-group partials of wstats_group_kernel (small batches, rank <= 8)
  int last_chain_kernel = 0; // VAENMF_Q_CHAIN_KERNEL
  int last_w_fused = 0;      // VAENMF_Q_W_FUSED: 1 when the last stored M-step ran the fused W-statistics kernel
  int Rcap_store;            // samples per frame the store was sized for at vaenmf_bind_batch / vaenmf_sample_store
  int last_m_step_path;      // VAENMF_Q_MSTEP_PATH: 0 none yet, 1 stored (streaming), 2 decoding
  uint64_t* d_utt_seed;      // [n_utt]
  std::vector<int32_t> h_frame_off;
  // workspace
  float *A1, *P;             // [NT][Fs] W-update statistics
  float* normW;              // [n_utt][Kp]
  float* wpart;              // [n_utt][8 chunks][Fs][2 Kp] partial W-update sums
  double* cost_frames;       // [VN_COST_CHUNK][max_frames] per-frame cost sums of the fused driver (a row per iteration of a chunk)
  // sample-variance store (vaenmf_sample_store): the MH chain keeps the decoded variances of its samples here
  bool store_on;
  void* VsS;                 // [NT][store_Rs][Fs], float (bf16x3 mode) or bf16 (bf16 mode)
  int32_t* src;              // [NT][store_Rs]
  size_t VsS_cap, src_cap;   // allocated bytes / elements
  int store_R, store_Rs;     // samples / slots per frame of the last chain that filled the store (0: empty)
  int n_sms;
  // optional per-kernel timing with HIP events on the launch stream (vaenmf_profile_*)
  bool prof_on;
  std::vector<hipEvent_t> prof_ev;      // pairs (start, stop)
  std::vector<int> prof_kind;
  size_t prof_used;
  // vaenmf_em_run as a HIP graph: the ~600 launches of a call captured once per call signature and replayed, so that
  // the launch path needs the host once per call instead of once per kernel (a loaded host showed as up to 20 % of
  // idle GPU time between the kernels)
  hipStream_t cap_stream = nullptr;     // capture needs a stream of its own (the caller's may be the null stream)
  struct EmGraph { std::vector<uint64_t> key; hipGraphExec_t exec; uint64_t used; };
  std::vector<EmGraph> g_cache;         // captured calls, a few signatures (a job alternates batch shapes: 63 / 62 utterances)
  std::vector<std::vector<uint64_t>> g_seen;   // signatures run eagerly once (a signature is captured at its second appearance)
  uint64_t g_tick = 0;
  bool g_off = false;                   // capture failed once on this plan: stay eager
  // vaenmf_bind_batch_async: the per-batch seeds go up from a ring of pinned slots (a slot is reused when its copy is done)
  static constexpr int SEED_RING = 8;
  uint64_t* h_seed_ring = nullptr;      // pinned [SEED_RING][max_utts]
  hipEvent_t seed_ev[SEED_RING] = {};
  bool seed_ev_used[SEED_RING] = {};
  int seed_pos = 0;
  int last_em_graph = 0;                // VAENMF_Q_EM_GRAPH: 1 when the last vaenmf_em_run was a graph launch
};

// one MH-chain call, as vaenmf_mh_chain hands it to the kernel launchers (engine.hip team kernel, chain.hip wave kernel)
struct VnChainCall {
  const float *X2, *W, *Ht, *g, *B1;
  float *Z, *Zs, *acc_out;
  const float *eps, *u;
  void* VsS; size_t VsS_bytes; int32_t* src; int Rs;
  int Rcap, nsamples, burnin, rng_mode, update_Z;
  uint32_t call;
  float sd;
  float sd_hi;               // random-walk step of latents 16..31: sd, or 0 when they are padding (latent dimension 16)
  int one_hidden;
};

enum { VN_K_CHAIN = 0, VN_K_WSTATS = 1, VN_K_WUPDATE = 2, VN_K_HG = 3, VN_K_WF = 4, VN_K_NKINDS = 5 };
struct ProfScope {   // records a (start, stop) event pair around a launch when profiling is on
  vaenmf_plan* p; hipStream_t st; size_t idx; bool on;
  ProfScope(vaenmf_plan* p_, int kind, hipStream_t st_) : p(p_), st(st_), idx(0), on(false) {
    if (p && p->prof_on && p->prof_used + 2 <= p->prof_ev.size()) {
      on = true; idx = p->prof_used; p->prof_used += 2; p->prof_kind[idx / 2] = kind;
      (void)hipEventRecord(p->prof_ev[idx], st);
    }
  }
  ~ProfScope() { if (on) (void)hipEventRecord(p->prof_ev[idx + 1], st); }
};

// ---- device helpers --------------------------------------------------------
#if defined(__HIPCC__)
// The decoder weights are pre-scaled on the host (plan.hip): hidden layers by 2 log2(e), the output layer by
// log2(e), so the MFMA accumulators are already the arguments of v_exp_f32 (2^x):
//   tanh(x) = 1 - 2/(exp(2x)+1) = 1 - 2/(2^(xs)+1),  xs = 2 log2(e) x      (abs error ~2e-7)
//   exp(a)  = 2^(as),                                 as = log2(e) a
__device__ __forceinline__ float fast_tanh(float xs) {
  float e = __builtin_amdgcn_exp2f(xs);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float fast_exp(float as) { return __builtin_amdgcn_exp2f(as); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
constexpr float LN2_F = 0.6931471805599453f;
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// split 4 floats into bf16 hi and lo parts: v ~= hi + lo (error ~2^-17 |v|)
__device__ __forceinline__ void split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    __bf16 h = (__bf16)v[t];
    hi[t] = h;
    lo[t] = (__bf16)(v[t] - (float)h);
  }
}

// ---- cross-lane sums without LDS traffic (DPP / permlane-swap VALU ops) ----------------
// lane = 16 q + c: sum over c (the 16 lanes of a DPP row), over q (the 4 rows), or over the wave
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum_row16(float v) {       // every lane gets the sum over its row
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return v;
}
// v_permlane{16,32}_swap exchange halves between TWO registers; feeding one value twice lets
// the compiler assign both operands the same register (a no-op swap), so the copy is made opaque.
__device__ __forceinline__ unsigned opaque_copy(unsigned a) {
  unsigned b = a;
  asm volatile("" : "+v"(b));
  return b;
}
__device__ __forceinline__ float swap16_add(float v) {      // v[l] + v[l ^ 16]
  const unsigned a = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(a, opaque_copy(a), false, false);
  unsigned x = r[0], y = r[1];
  asm volatile("" : "+v"(x), "+v"(y));      // keep both halves of the result (hipcc 7.2 folds r[1] into r[0] otherwise)
  return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float swap32_add(float v) {      // v[l] + v[l ^ 32]
  const unsigned a = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(a, opaque_copy(a), false, false);
  unsigned x = r[0], y = r[1];
  asm volatile("" : "+v"(x), "+v"(y));
  return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float sum_rows4(float v) { return swap32_add(swap16_add(v)); }   // sum over q
__device__ __forceinline__ double swap16_add_d(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  auto rl = __builtin_amdgcn_permlane16_swap(lo, opaque_copy(lo), false, false);
  auto rh = __builtin_amdgcn_permlane16_swap(hi, opaque_copy(hi), false, false);
  const double a0 = __builtin_bit_cast(double, ((unsigned long long)rh[0] << 32) | rl[0]);
  const double a1 = __builtin_bit_cast(double, ((unsigned long long)rh[1] << 32) | rl[1]);
  return a0 + a1;
}
__device__ __forceinline__ double swap32_add_d(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
  auto rl = __builtin_amdgcn_permlane32_swap(lo, opaque_copy(lo), false, false);
  auto rh = __builtin_amdgcn_permlane32_swap(hi, opaque_copy(hi), false, false);
  const double a0 = __builtin_bit_cast(double, ((unsigned long long)rh[0] << 32) | rl[0]);
  const double a1 = __builtin_bit_cast(double, ((unsigned long long)rh[1] << 32) | rl[1]);
  return a0 + a1;
}
__device__ __forceinline__ double sum_rows4_d(double v) { return swap32_add_d(swap16_add_d(v)); }

// xoshiro128+ : per-lane stream, 32 random bits per call, no multiplies
struct Xs128 {
  uint32_t s0, s1, s2, s3;
  __device__ __forceinline__ uint32_t next() {
    uint32_t r = s0 + s3;
    uint32_t t = s1 << 9;
    s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t;
    s3 = (s3 << 11) | (s3 >> 21);
    return r;
  }
};
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t& x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// stream key: (utterance seed, frame index inside the utterance, sub-stream id, chain call)
__device__ __forceinline__ Xs128 xs_seed(uint64_t utt_seed, uint32_t frame, uint32_t sub, uint32_t call) {
  uint64_t x = utt_seed ^ (((uint64_t)frame << 32) | ((uint64_t)sub << 24) | (uint64_t)(call & 0xFFFFFFu));
  x ^= (uint64_t)(call >> 24) << 56;
  uint64_t a = splitmix64(x), b = splitmix64(x);
  Xs128 s;
  s.s0 = (uint32_t)a; s.s1 = (uint32_t)(a >> 32); s.s2 = (uint32_t)b; s.s3 = (uint32_t)(b >> 32);
  if ((s.s0 | s.s1 | s.s2 | s.s3) == 0) s.s0 = 1;
  return s;
}
// 4 standard normals (two Box-Muller pairs); v_sin/v_cos take revolutions.
// VN_RNG16 = 1: ONE random word per pair -- radius from its high 16 bits, angle from its low 16 bits -- instead of two
// words at 24 bits each: half the generator steps (the proposals are N(0, var_RW) steps of a random walk: a radius on a
// 2^-16 grid, |eps| <= 4.71 sigma, changes nothing the sampler's distribution can show; any symmetric proposal leaves
// the Metropolis-Hastings target unchanged).
#ifndef VN_RNG16
#define VN_RNG16 1
#endif
__device__ __forceinline__ f32x4 normal4(Xs128& st) {
  f32x4 o;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#if VN_RNG16
    const uint32_t a = st.next();
    const float u1 = ((float)(a >> 16) + 1.0f) * 1.52587890625e-5f;  // (0,1], 2^-16 grid
    const float u2 = (float)(a & 0xFFFFu) * 1.52587890625e-5f;       // [0,1)
#else
    uint32_t a = st.next(), b = st.next();
    float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-8f;   // (0,1]
    float u2 = (float)(b >> 8) * 5.9604644775390625e-8f;            // [0,1)
#endif
    float r = __builtin_amdgcn_sqrtf(-2.0f * fast_log(u1));
    o[2 * p] = r * __builtin_amdgcn_cosf(u2);
    o[2 * p + 1] = r * __builtin_amdgcn_sinf(u2);
  }
  return o;
}
__device__ __forceinline__ float uniform01(Xs128& st) { return (float)(st.next() >> 8) * 5.9604644775390625e-8f; }
#endif
