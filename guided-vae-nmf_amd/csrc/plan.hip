// Plan management: shapes, device-resident decoder weights in MFMA fragment order,
// batch tiling tables, workspace.  All allocation happens here.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <set>
#include <utility>
#include "common.h"
#include <algorithm>

static thread_local char g_err[512] = "";

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember (device, kernel) pairs, check the result
int vn_ensure_dyn_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  VN_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, fn})) return 0;
  VN_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, fn});
  return 0;
}

void vaenmf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* vaenmf_last_error(void) { return g_err; }

long long g_vn_dev_allocs = 0;      // device allocations made by the library in this process (VAENMF_Q_DEV_ALLOCS)

namespace {

uint16_t bf16_rne(float v) {                 // round-to-nearest-even, NaN kept quiet
  uint32_t u;
  memcpy(&u, &v, 4);
  if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float bf16_to_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float v;
  memcpy(&v, &u, 4);
  return v;
}

// W [out][ldw] (first `in` columns used) -> fragments [tile][kstep][part][lane][8]:
// lane (q = lane>>4, i = lane&15), element j  <->  W[16 tile + i][32 s + 16 (j>>2) + 4 q + (j&3)]
std::vector<uint16_t> pack_weights(const float* W, int out, int in, int ldw, int ntiles, int nk) {
  std::vector<uint16_t> f((size_t)ntiles * nk * 2 * 64 * 8, 0);
  for (int t = 0; t < ntiles; ++t)
    for (int s = 0; s < nk; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int q = lane >> 4, i = lane & 15;
        for (int j = 0; j < 8; ++j) {
          const int row = 16 * t + i, col = 32 * s + 16 * (j >> 2) + 4 * q + (j & 3);
          const float v = (row < out && col < in) ? W[(size_t)row * ldw + col] : 0.f;
          const uint16_t hi = bf16_rne(v);
          const uint16_t lo = bf16_rne(v - bf16_to_f(hi));
          const size_t base = ((((size_t)t * nk + s) * 2) * 64 + lane) * 8 + j;
          f[base] = hi;
          f[base + 64 * 8] = lo;
        }
      }
  return f;
}

template <typename T>
int dev_alloc(T** p, size_t n) {
  VN_CHECK_HIP(hipMalloc((void**)p, (n ? n : 1) * sizeof(T)));
  ++g_vn_dev_allocs;
  return 0;
}
template <typename T>
int upload(T* dst, const T* src, size_t n) {
  VN_CHECK_HIP(hipMemcpy(dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

}  // namespace

extern "C" int vaenmf_plan_create(const vaenmf_config* cfg, vaenmf_plan** out) {
  VN_REQUIRE(cfg && out, "null argument");
  // decoder shapes: z(32 or 16) -> 128 [-> 128] -> F.  Latent dimension 16 runs the 32-wide first layer with zero padding
  // (no random-walk noise on the padding); H2 = 0 is a decoder with ONE hidden layer (the reference's h_dim = [128]).
  VN_REQUIRE(cfg->L == LAT || cfg->L == 16, "this build supports latent dims %d and 16 (got %d)", LAT, cfg->L);
  VN_REQUIRE(cfg->H1 == HID && (cfg->H2 == HID || cfg->H2 == 0), "this build supports hidden sizes %d[,%d] (got %d,%d)", HID, HID, cfg->H1, cfg->H2);
  VN_REQUIRE(cfg->F >= 1 && cfg->F <= 640, "F=%d out of range (1..640)", cfg->F);
  VN_REQUIRE(cfg->K >= 1 && cfg->K <= 32, "NMF rank K=%d out of range (1..32)", cfg->K);
  VN_REQUIRE(cfg->max_frames >= 1 && cfg->max_utts >= 1, "bad capacities");
  VN_REQUIRE(cfg->precision == VAENMF_PREC_BF16X3 || cfg->precision == VAENMF_PREC_BF16, "bad precision");
  vaenmf_plan* p = new vaenmf_plan();
  memset((void*)&p->cfg, 0, sizeof(p->cfg));
  p->cfg = *cfg;
  p->Fs = (cfg->F + 15) / 16 * 16;
  p->Fm = (cfg->F % 16 == 1 && cfg->F > 16) ? cfg->F - 1 : cfg->F;    // n_fft/2+1 bins: the last one leaves the tiles
  p->NT3 = (p->Fm + 15) / 16;
  p->Kp = cfg->K <= 8 ? 8 : (cfg->K <= 16 ? 16 : 32);
  {
    const char* g = getenv("VAENMF_GEOM");          // dev override (A/B runs): force one team of 8 waves
    // geometries 3 / 4 (4 bin tiles per wave, no tile checks) need exactly 16 / 32 tiles: F = 257 / 513
    int geom = p->NT3 == 16 ? 3 : (p->NT3 <= 20 ? 0 : (p->NT3 == 32 ? 4 : 2));
    if (g && g[0] == '2') geom = p->NT3 == 32 ? 4 : 2;
    p->geom = geom;
    p->nwaves = (geom == 0 || geom == 3) ? 4 : 8;
    p->tile_frames = (geom == 0 || geom == 3) ? 64 : 32;
  }
  p->NT3c = (cfg->F + 15) / 16;
  p->w1f = p->w2f = p->w3f = p->w3c = nullptr;
  p->b1 = p->b2 = p->b3 = p->w1y = p->w3n = p->b3c = nullptr;
  p->d_wt_utt = p->d_wt_n0 = p->d_wt_cnt = nullptr;
  p->n_wtiles = 0; p->Rcap_store = 0; p->last_m_step_path = 0;
  p->Dy = 0;
  p->Lz = cfg->L;
  p->one_hidden = cfg->H2 == 0;
  p->have_weights = false;
  p->Vb_ext = nullptr;
  p->store_on = false; p->VsS = nullptr; p->src = nullptr; p->VsS_cap = p->src_cap = 0; p->store_R = p->store_Rs = 0;
  p->n_utt = p->NT = p->n_tiles = 0;
  p->prof_on = false;
  p->prof_used = 0;
  int dev = 0;
  hipDeviceProp_t prop;
  VN_CHECK_HIP(hipGetDevice(&dev));
  VN_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  p->n_sms = prop.multiProcessorCount;
  const size_t NTc = cfg->max_frames, Uc = cfg->max_utts;
  const size_t max_tiles = NTc / 32 + Uc + 1;
  int e = 0;
  e |= dev_alloc(&p->w1f, (size_t)(HID / 16) * 1 * 2 * 64 * 8);
  e |= dev_alloc(&p->w2f, (size_t)(HID / 16) * (HID / 32) * 2 * 64 * 8);
  e |= dev_alloc(&p->w3f, (size_t)p->NT3 * (HID / 32) * 2 * 64 * 8);
  e |= dev_alloc(&p->b1, HID);
  e |= dev_alloc(&p->b2, HID);
  e |= dev_alloc(&p->b3, p->Fs);
  e |= dev_alloc(&p->w3n, HID);
  e |= dev_alloc(&p->w3c, (size_t)p->NT3c * (HID / 32) * 2 * 64 * 8);
  e |= dev_alloc(&p->b3c, (size_t)p->NT3c * 16);
  const size_t max_wtiles = NTc / 16 + Uc + 1;
  e |= dev_alloc(&p->d_wt_utt, max_wtiles);
  e |= dev_alloc(&p->d_wt_n0, max_wtiles);
  e |= dev_alloc(&p->d_wt_cnt, max_wtiles);
  e |= dev_alloc(&p->d_frame_off, Uc + 1);
  e |= dev_alloc(&p->d_tile_utt, max_tiles);
  e |= dev_alloc(&p->d_tile_n0, max_tiles);
  e |= dev_alloc(&p->d_tile_cnt, max_tiles);
  e |= dev_alloc(&p->d_frame_utt, NTc);
  e |= dev_alloc(&p->d_frame_loc, NTc);
  e |= dev_alloc(&p->d_utt_seed, Uc);
  e |= dev_alloc(&p->A1, NTc * p->Fs);
  e |= dev_alloc(&p->P, NTc * p->Fs);
  e |= dev_alloc(&p->normW, Uc * p->Kp);
  e |= dev_alloc(&p->wpart, Uc * 8 * p->Fs * 2 * p->Kp);
  e |= dev_alloc(&p->cost_frames, NTc * VN_COST_CHUNK);
  const size_t max_t64 = NTc / 64 + Uc + 1;
  e |= dev_alloc(&p->d_t64_n0, max_t64);
  e |= dev_alloc(&p->d_t64_cnt, max_t64);
  e |= dev_alloc(&p->d_t64_g0, max_t64);
  e |= dev_alloc(&p->d_t64_first, Uc + 1);
  e |= dev_alloc(&p->wpart64, max_t64 * p->Fs * 2 * p->Kp);
  if (p->Kp == 8) {       // group partials of small batches (at most one 16-frame group per CU)
    p->wpart16_groups = (size_t)p->n_sms;
    e |= dev_alloc(&p->wpart16, p->wpart16_groups * p->Fs * 2 * p->Kp);
  }
  if (e) { vaenmf_plan_destroy(p); return -2; }
  *out = p;
  return 0;
}

extern "C" void vaenmf_plan_destroy(vaenmf_plan* p) {
  if (!p) return;
  void* ptrs[] = {p->w1f, p->w2f, p->w3f, p->b1, p->b2, p->b3, p->w3n, p->w1y, p->d_frame_off, p->d_tile_utt, p->d_tile_n0,
                  p->d_tile_cnt, p->d_frame_utt, p->d_frame_loc, p->d_utt_seed, p->A1, p->P, p->normW, p->wpart, p->cost_frames,
                  p->VsS, p->src, p->w3c, p->b3c, p->d_wt_utt, p->d_wt_n0, p->d_wt_cnt, p->d_t64_n0, p->d_t64_cnt, p->d_t64_first, p->d_t64_g0,
                  p->wpart64, p->wpart16};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  for (hipEvent_t e : p->prof_ev) (void)hipEventDestroy(e);
  for (auto& gph : p->g_cache) if (gph.exec) (void)hipGraphExecDestroy(gph.exec);
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  if (p->h_seed_ring) (void)hipHostFree(p->h_seed_ring);
  for (int i = 0; i < vaenmf_plan::SEED_RING; ++i) if (p->seed_ev[i]) (void)hipEventDestroy(p->seed_ev[i]);
  delete p;
}

extern "C" int vaenmf_plan_query(const vaenmf_plan* p, int32_t what) {
  if (!p) return -1;
  switch (what) {
    case VAENMF_Q_FS: return p->Fs;
    case VAENMF_Q_KP: return p->Kp;
    case VAENMF_Q_TILES: return p->n_tiles;
    case VAENMF_Q_NT: return p->NT;
    case VAENMF_Q_NUTT: return p->n_utt;
    case VAENMF_Q_MSTEP_PATH: return p->last_m_step_path;
    case VAENMF_Q_WTILES: return p->n_wtiles;
    case VAENMF_Q_EM_GRAPH: return p->last_em_graph;
    case VAENMF_Q_DEV_ALLOCS: return (int)g_vn_dev_allocs;
    case VAENMF_Q_W_FUSED: return p->last_w_fused;
    case VAENMF_Q_CHAIN_KERNEL: return p->last_chain_kernel;
    default: return -1;
  }
}

extern "C" int vaenmf_set_decoder_weights(vaenmf_plan* p, const float* W1, int32_t in1, const float* b1, const float* W2,
                                          const float* b2, const float* W3, const float* b3) {
  VN_REQUIRE(p && W1 && b1 && W3 && b3, "null argument");
  VN_REQUIRE(p->one_hidden || (W2 && b2), "null second-layer weights (plan with H2 = %d)", p->cfg.H2);
  const int Lz = p->Lz;
  VN_REQUIRE(in1 >= Lz, "decoder input width %d < latent dim %d", in1, Lz);
  const int F = p->cfg.F;
  // first layer on the 32-wide MFMA k-step: [H][32 latent columns (zero beyond Lz) | Dy label columns]
  std::vector<float> W1pad;
  const int in1m = in1 - Lz + LAT;
  if (Lz != LAT) {
    W1pad.assign((size_t)HID * in1m, 0.f);
    for (int h = 0; h < HID; ++h) {
      memcpy(&W1pad[(size_t)h * in1m], W1 + (size_t)h * in1, sizeof(float) * Lz);
      memcpy(&W1pad[(size_t)h * in1m + LAT], W1 + (size_t)h * in1 + Lz, sizeof(float) * (in1 - Lz));
    }
    W1 = W1pad.data();
    in1 = in1m;
  }
  const std::vector<float> W2zero(p->one_hidden ? (size_t)HID * HID : 0, 0.f), b2zero(p->one_hidden ? HID : 0, 0.f);
  if (p->one_hidden) { W2 = W2zero.data(); b2 = b2zero.data(); }   // (never read by the kernels: layer 2 is skipped)
  // pre-scale so that the accumulators are v_exp_f32 arguments (common.h: fast_tanh / fast_exp)
  const double C2 = 2.0 * 1.4426950408889634, C1 = 1.4426950408889634;
  auto scaled = [](const float* src, size_t n, double c) {
    std::vector<float> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (float)((double)src[i] * c);
    return v;
  };
  const std::vector<float> W1s = scaled(W1, (size_t)HID * in1, C2), b1s = scaled(b1, HID, C2);
  const std::vector<float> W2s = scaled(W2, (size_t)HID * HID, C2), b2s = scaled(b2, HID, C2);
  const std::vector<float> W3s = scaled(W3, (size_t)F * HID, C1), b3s = scaled(b3, F, C1);
  std::vector<uint16_t> f1 = pack_weights(W1s.data(), HID, LAT, in1, HID / 16, 1);
  std::vector<uint16_t> f2 = pack_weights(W2s.data(), HID, HID, HID, HID / 16, HID / 32);
  std::vector<uint16_t> f3 = pack_weights(W3s.data(), p->Fm, HID, HID, p->NT3, HID / 32);
  std::vector<float> b3p(p->Fs, -100.f);      // padding bins: W3 rows are 0, so Vs = 2^-100 ~ 0
  memcpy(b3p.data(), b3s.data(), sizeof(float) * F);
  int e = 0;
  e |= upload((uint16_t*)p->w1f, f1.data(), f1.size());
  e |= upload((uint16_t*)p->w2f, f2.data(), f2.size());
  e |= upload((uint16_t*)p->w3f, f3.data(), f3.size());
  e |= upload(p->b1, b1s.data(), HID);
  e |= upload(p->b2, b2s.data(), HID);
  e |= upload(p->b3, b3p.data(), b3p.size());
  e |= upload(p->w3n, W3s.data() + (size_t)(F - 1) * HID, HID);
  {
    // chain.hip: every bin on the MFMA path; bf16 mode pairs the tiles (bin = 32 (t>>1) + 8 q + 4 (t&1) + j for
    // row 4 q + j of tile t < Tm), bf16x3 mode keeps the natural order; padding rows: W3 = 0, b3 = -200 (Vs = 0)
    const int NT3c = p->NT3c, Tm = p->cfg.precision == VAENMF_PREC_BF16X3 ? 0 : ((NT3c - 1) & ~1);
    std::vector<float> Wp((size_t)16 * NT3c * HID, 0.f), bp((size_t)16 * NT3c, -200.f);
    for (int t = 0; t < NT3c; ++t)
      for (int i = 0; i < 16; ++i) {
        const int q = i >> 2, j = i & 3;
        const int bin = t < Tm ? 32 * (t >> 1) + 8 * q + 4 * (t & 1) + j : 16 * t + i;
        if (bin < F) {
          memcpy(&Wp[(size_t)(16 * t + i) * HID], W3s.data() + (size_t)bin * HID, sizeof(float) * HID);
          bp[16 * t + i] = b3s[bin];
        }
      }
    std::vector<uint16_t> f3c = pack_weights(Wp.data(), 16 * NT3c, HID, HID, NT3c, HID / 32);
    e |= upload((uint16_t*)p->w3c, f3c.data(), f3c.size());
    e |= upload(p->b3c, bp.data(), bp.size());
  }
  if (p->w1y) { (void)hipFree(p->w1y); p->w1y = nullptr; }
  p->Dy = in1 - LAT;
  if (p->Dy > 0) {                             // label columns of W1 (scaled like the rest of layer 1), [H1][Dy]
    std::vector<float> wy((size_t)HID * p->Dy);
    for (int h = 0; h < HID; ++h) memcpy(&wy[(size_t)h * p->Dy], W1s.data() + (size_t)h * in1 + LAT, sizeof(float) * p->Dy);
    e |= dev_alloc(&p->w1y, wy.size());
    if (!e) e |= upload(p->w1y, wy.data(), wy.size());
  }
  if (e) return -2;
  p->have_weights = true;
  return 0;
}

// Binds a batch.  The frame tables are rebuilt and uploaded (blocking copies) only when the batch's frame structure
// differs from the bound one; the utterance seeds -- the only per-batch table of a job that repeats its batch shape -- go
// up asynchronously on `stream` from a ring of pinned slots, so the call does not wait for the GPU and a caller that
// never synchronises can prepare the next batch while this one runs.
extern "C" int vaenmf_bind_batch_async(vaenmf_plan* p, int32_t n_utt, const int32_t* frame_offsets, const uint64_t* utt_seeds, void* stream) {
  VN_REQUIRE(p && frame_offsets, "null argument");
  VN_REQUIRE(n_utt >= 1 && n_utt <= p->cfg.max_utts, "n_utt=%d exceeds capacity %d", n_utt, p->cfg.max_utts);
  VN_REQUIRE(frame_offsets[0] == 0, "frame_offsets[0] must be 0");
  const int NT = frame_offsets[n_utt];
  VN_REQUIRE(NT >= 1 && NT <= p->cfg.max_frames, "total frames %d exceeds capacity %d", NT, p->cfg.max_frames);
  hipStream_t st = (hipStream_t)stream;
  const bool same = p->NT == NT && p->n_utt == n_utt && (int)p->h_frame_off.size() == n_utt + 1 &&
                    std::equal(p->h_frame_off.begin(), p->h_frame_off.end(), frame_offsets);
  if (!same) {
    std::vector<int32_t> t_utt, t_n0, t_cnt, f_utt(NT), f_loc(NT);
    for (int u = 0; u < n_utt; ++u) {
      const int b = frame_offsets[u], e = frame_offsets[u + 1];
      VN_REQUIRE(e > b, "utterance %d is empty", u);
      for (int n = b; n < e; n += p->tile_frames) {
        t_utt.push_back(u);
        t_n0.push_back(n);
        t_cnt.push_back(e - n < p->tile_frames ? e - n : p->tile_frames);
      }
      for (int n = b; n < e; ++n) { f_utt[n] = u; f_loc[n] = n - b; }
    }
    std::vector<int32_t> w_utt, w_n0, w_cnt;              // wave tiles of the wave-private chain (chain.hip)
    for (int u = 0; u < n_utt; ++u)
      for (int n = frame_offsets[u]; n < frame_offsets[u + 1]; n += 16) {
        w_utt.push_back(u);
        w_n0.push_back(n);
        w_cnt.push_back(frame_offsets[u + 1] - n < 16 ? frame_offsets[u + 1] - n : 16);
      }
    std::vector<int32_t> t64_n0, t64_cnt, t64_g0, t64_first(n_utt + 1);   // <= 64-frame tiles of the fused W-statistics kernel
    int32_t gcount = 0;                                   // 16-frame groups (= wave tiles) before the utterance
    for (int u = 0; u < n_utt; ++u) {
      t64_first[u] = (int32_t)t64_n0.size();
      for (int n = frame_offsets[u]; n < frame_offsets[u + 1]; n += 64) {
        t64_n0.push_back(n);
        t64_cnt.push_back(frame_offsets[u + 1] - n < 64 ? frame_offsets[u + 1] - n : 64);
        t64_g0.push_back(gcount + (n - frame_offsets[u]) / 16);       // the tile's first group (w_combine_groups_kernel)
      }
      gcount += (frame_offsets[u + 1] - frame_offsets[u] + 15) / 16;
    }
    t64_first[n_utt] = (int32_t)t64_n0.size();
    VN_CHECK_HIP(hipStreamSynchronize(st));               // kernels of the previous batch may still read the tables
    int e = 0;
    e |= upload(p->d_frame_off, frame_offsets, (size_t)n_utt + 1);
    e |= upload(p->d_tile_utt, t_utt.data(), t_utt.size());
    e |= upload(p->d_tile_n0, t_n0.data(), t_n0.size());
    e |= upload(p->d_tile_cnt, t_cnt.data(), t_cnt.size());
    e |= upload(p->d_frame_utt, f_utt.data(), f_utt.size());
    e |= upload(p->d_frame_loc, f_loc.data(), f_loc.size());
    e |= upload(p->d_wt_utt, w_utt.data(), w_utt.size());
    e |= upload(p->d_wt_n0, w_n0.data(), w_n0.size());
    e |= upload(p->d_wt_cnt, w_cnt.data(), w_cnt.size());
    e |= upload(p->d_t64_n0, t64_n0.data(), t64_n0.size());
    e |= upload(p->d_t64_cnt, t64_cnt.data(), t64_cnt.size());
    e |= upload(p->d_t64_g0, t64_g0.data(), t64_g0.size());
    e |= upload(p->d_t64_first, t64_first.data(), t64_first.size());
    p->n_t64 = (int)t64_n0.size();
    if (e) return -2;
    p->n_wtiles = (int)w_utt.size();
    p->n_tiles = (int)t_utt.size();
    p->n_utt = n_utt;
    p->NT = NT;
    p->h_frame_off.assign(frame_offsets, frame_offsets + n_utt + 1);
  }
  // ---- seeds: pinned slot -> device, stream-ordered
  if (!p->h_seed_ring) VN_CHECK_HIP(hipHostMalloc((void**)&p->h_seed_ring, (size_t)vaenmf_plan::SEED_RING * p->cfg.max_utts * sizeof(uint64_t), hipHostMallocDefault));
  const int slot = p->seed_pos;
  p->seed_pos = (p->seed_pos + 1) % vaenmf_plan::SEED_RING;
  if (!p->seed_ev[slot]) VN_CHECK_HIP(hipEventCreateWithFlags(&p->seed_ev[slot], hipEventDisableTiming));
  if (p->seed_ev_used[slot]) VN_CHECK_HIP(hipEventSynchronize(p->seed_ev[slot]));      // (eight binds ago: long done)
  uint64_t* hs = p->h_seed_ring + (size_t)slot * p->cfg.max_utts;
  for (int u = 0; u < n_utt; ++u) {
    uint64_t x = 0x5EEDull + (uint64_t)u;
    hs[u] = utt_seeds ? utt_seeds[u] : splitmix64(x);
  }
  VN_CHECK_HIP(hipMemcpyAsync(p->d_utt_seed, hs, (size_t)n_utt * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  VN_CHECK_HIP(hipEventRecord(p->seed_ev[slot], st));
  p->seed_ev_used[slot] = true;
  p->store_R = p->store_Rs = 0;                         // the store's contents belong to the previous batch
  return 0;
}

// The blocking form: every table is on the device when the call returns.
extern "C" int vaenmf_bind_batch(vaenmf_plan* p, int32_t n_utt, const int32_t* frame_offsets, const uint64_t* utt_seeds) {
  if (int e = vaenmf_bind_batch_async(p, n_utt, frame_offsets, utt_seeds, nullptr)) return e;
  VN_CHECK_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

extern "C" int vaenmf_layer1_bias(vaenmf_plan* p, const float* y, int32_t Dy, float* B1, void* stream) {
  VN_REQUIRE(p && y && B1, "null argument");
  VN_REQUIRE(p->have_weights && p->NT > 0, "plan needs weights and a bound batch");
  VN_REQUIRE(Dy == p->Dy && Dy > 0, "label width %d does not match decoder input (L+%d)", Dy, p->Dy);
  return vaenmf_dense(y, p->NT, Dy, Dy, p->w1y, p->b1, HID, VAENMF_ACT_NONE, B1, HID, stream);
}

// Per-kernel timing: HIP events recorded on the launch stream around every hot-path
// launch (kinds: 0 mh_chain, 1 decode+W-stats, 2 W update, 3 decode+H/g/cost, 4 decode+Wiener).
extern "C" int vaenmf_profile_enable(vaenmf_plan* p, int32_t max_launches) {
  VN_REQUIRE(p, "null plan");
  for (hipEvent_t e : p->prof_ev) (void)hipEventDestroy(e);
  p->prof_ev.clear();
  p->prof_kind.clear();
  p->prof_used = 0;
  p->prof_on = max_launches > 0;
  for (int i = 0; i < 2 * max_launches; ++i) {
    hipEvent_t e;
    VN_CHECK_HIP(hipEventCreate(&e));
    p->prof_ev.push_back(e);
  }
  p->prof_kind.assign(max_launches > 0 ? max_launches : 0, 0);
  return 0;
}
// Synchronises, then ms[k] = summed device time of kind k, counts[k] = launches; resets.
extern "C" int vaenmf_profile_read(vaenmf_plan* p, double* ms, int64_t* counts) {
  VN_REQUIRE(p && ms && counts, "null argument");
  for (int k = 0; k < VN_K_NKINDS; ++k) { ms[k] = 0.0; counts[k] = 0; }
  for (size_t i = 0; i + 1 < p->prof_used; i += 2) {
    VN_CHECK_HIP(hipEventSynchronize(p->prof_ev[i + 1]));
    float t = 0.f;
    VN_CHECK_HIP(hipEventElapsedTime(&t, p->prof_ev[i], p->prof_ev[i + 1]));
    ms[p->prof_kind[i / 2]] += t;
    counts[p->prof_kind[i / 2]] += 1;
  }
  p->prof_used = 0;
  return 0;
}
