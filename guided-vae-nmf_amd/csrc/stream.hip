// M-step and Wiener filter over the sample-variance store (EM.M_step mcem.py:90-152, cost :68-70,
// compute_WF :486-488): the variances Vs[n][r][f] of the chain's samples were written to HBM by
// mh_chain_kernel (engine.hip, STORE), so these kernels stream them instead of running the decoder
// again.  They are plain bandwidth kernels: one wavefront owns one frame at a time, a lane owns four
// consecutive bins per 256-bin chunk (one 1 KiB row segment per load instruction), the samples are a
// sequential loop, and everything a frame needs beyond its own row -- sums over bins for H, g and the
// cost -- is a wavefront reduction (DPP / permlane, no LDS, no barrier).
//
//   wstats_stream   A1 = sum_r 1/Vx, P = X2 sum_r 1/Vx^2 with the pre-update W, H, g        (:107-109)
//   hg_stream       H <- H sqrt(num/den) with the updated W (:118-121), refreshed Vb (:124-125),
//                   g <- g sqrt(num/den) (:138-142), cost with the refreshed variances (:70, :151-152);
//                   three passes over the frame's rows, which stay in registers when they fit one batch
//   wf_stream       Wiener masks mean_r(g Vs/Vx), mean_r(Vb/Vx) and S_hat, N_hat            (:486-488, :175-176)
//
// Bins of the odd last bin and the padding (f >= Fm: F-1 when F = 16k+1) are handled as one extra
// element that every lane computes redundantly (same address: a broadcast load).
#include "common.h"
#ifndef VN_HG_WAVES
#define VN_HG_WAVES 4      // minimum waves per SIMD asked of the compiler (register cap 512 / waves)
#endif
#ifndef VN_WS_WAVES
#define VN_WS_WAVES 4
#endif
#ifndef VN_ROWCHUNK
#define VN_ROWCHUNK 1      // rows per uniform branch in RowBatch::for_rows (2, 3, 5 rows per branch spill more at the 128-register cap: 87 / 106 / 275 scratch operations against 50)
#endif
#ifndef VN_ROT
#define VN_ROT 1          // rotating-register W-statistics / H,g kernels for the bench shapes (dev builds: 0 = the batch forms)
#endif
#ifndef VN_WS_EXACT
#define VN_WS_EXACT 0     // exact-sample-count instantiations of wstats_stream2 (R = 30 / 10)
#endif
#ifndef VN_HG_FULL2
#define VN_HG_FULL2 1
#endif
#ifndef VN_HG_FULLF
#define VN_HG_FULLF 1
#endif
#ifndef VN_HG_FULL2_KMAX
#define VN_HG_FULL2_KMAX 16     // (rank 32: the whole-frame batch still spills, 288 bytes, and gains nothing)
#endif
#ifndef VN_HG_EXACT
#define VN_HG_EXACT 1     // exact-sample-count instantiations of hg_stream (R = 30 / 10, one chunk, rank <= 8): no per-row branches, rows consumed
                          // as they arrive (precise vmcnt counts): 0.198 -> 0.172 ms.  (Slower while the extra-bin addresses still spilled.)
#endif
#ifndef VN_HG_STAGGER
#define VN_HG_STAGGER 0     // s_sleep(127) units (8128 cycles each) per hardware wave slot at the start of hg_stream
#endif
#ifndef VN_HG_PROBE
#define VN_HG_PROBE 0     // dev probes of hg_stream: 1 = rows loaded, arithmetic reduced to one add per element; 2 = every row read from the same cached address
#endif
#ifndef VN_HG_REPACK
#define VN_HG_REPACK 1
#endif
#ifndef VN_ROWGRP
#define VN_ROWGRP 6     // rows whose arithmetic the scheduler may interleave in the exact-count stream kernels
#endif
#ifndef VN_HG_EXACT2
#define VN_HG_EXACT2 1     // exact-count H/g kernel for two-chunk rows at rank <= 16 (F = 513, K = 10: 0.2625 -> 0.250 ms)
#endif
// Non-temporal loads for the sample-variance rows (each row is read once per kernel and the store, 0.54 GB, passes every
// cache): measured on one box, alternating builds -- W statistics 0.131-0.137 -> 0.121-0.124 ms, step 71.9 -> 70.5 ms (the H/g
// kernel, VALU-bound, does not move).  What the rows compete with is the chain's own write-back: with non-temporal STORES in
// the chain the W statistics drop to 0.099 ms (4.9 TB/s), but the chain pays 0.39 -> 0.53 ms for them (sc1 stores: 0.46), so
// the stores stay cached.
#ifndef VN_ROW_AUX
#define VN_ROW_AUX 2     // cache policy of the row loads of wstats_rot / wstats_fused (buffer loads: 2 = nt)
#endif
#ifndef VN_ROW_NT
#define VN_ROW_NT 1      // non-temporal row loads in the RowBatch kernels (wstats_stream, wstats_group, hg_stream, wf_stream)
#endif
#ifndef VN_STREAM2
#define VN_STREAM2 1      // frame-pipelined W-statistics kernel (dev builds: 0 = the batch-at-a-time form)
#endif

int vn_ensure_dyn_lds(const void* fn, int bytes);     // plan.hip

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct StreamArgs {
  const void* VsS;             // [NT][Rs][Fs], float or bf16 (template ST)
  const int32_t* src;          // [Rs][NT] (sample-major)
  const float *X2, *W, *normW, *Vb, *X;
  float *Ht, *g, *A1, *P, *S_hat, *N_hat, *WFs, *WFn;
  double* cost_frames;
  const int32_t* frame_utt;
  int NT, R, Rs, F, Fm, Fs, K;
  int gains_only;              // the *_noNMF M-step (mcem.py:543-578): Vb given, only g moves
  int store_f32;               // rows are float (bf16x3 mode) rather than bf16
  int w_blk_lds;               // rank > 8: W of the workgroup's first utterance is staged in LDS
  int n_sms;
  // W statistics fused with the W update's sums (wstats_fused_kernel): tiles of <= 64 frames of one utterance
  const int32_t *t64_n0, *t64_cnt;
  int n_t64;
  float* wpart64;              // [n_t64][2 KP slots: 2 k + stat][Fs]
};

__device__ __forceinline__ float wave_sum(float v) { return sum_rows4(sum_row16(v)); }
// a value every lane holds alike, kept in a scalar register from here on
__device__ __forceinline__ float vn_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

template <int NCH, int KP, typename ST>
struct FrameCtx {
  const StreamArgs& a;
  int lane;
  bool cv[NCH];                // chunk c holds real bins for this lane
  int f0[NCH];
  unsigned fo[NCH];            // cv ? f0 : 0 as an unsigned element offset: (uniform row pointer)[fo] is addressed as SGPR base + 32-bit
                               // VGPR offset, no 64-bit address pair per load in vector registers
  bool has_x;                  // an extra bin F-1 beside the 4-bin chunks
  // W of the utterance in LDS.  Read from global memory per frame the rows dominate the L1 traffic: a lane's 4
  // rows are 128 B (rank 8) from the next lane's, so every load instruction touches 64 cache lines -- ~3500 line
  // accesses per frame against ~120 for the variances.
  //   rank <= 8: a wave-private piece, refilled by the wavefront when it enters another utterance (no barrier);
  //   rank > 8 : one copy per workgroup, of the utterance of the workgroup's first frame (staged by the kernel
  //              prologue when it fits); a wavefront in another utterance reads global memory.
  static constexpr bool WPRIV = KP <= 8;
  float* wl;                   // [Fs][KP]
  // W[utt][F-1][k] of the extra bin in lane k (refreshed when the wavefront enters another utterance): its eight LDS
  // addresses, loop-invariant, were spilled and every reload (scratch: a memory round trip, waited for with vmcnt(0))
  // stalled the wavefront -- ~25 serial round trips per frame, more than the frame's arithmetic
  float wx;
  float nw;                    // normW[utt][k] in lane k (hg_stream), same refresh
  int wutt, blk_utt, xutt;
  bool in_lds;                 // wave-uniform: this frame's W rows are the ones in LDS
  __device__ FrameCtx(const StreamArgs& a_, float* wl_) : a(a_), wl(wl_) {
    lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < NCH; ++c) { f0[c] = 256 * c + 4 * lane; cv[c] = f0[c] < a.Fm; fo[c] = cv[c] ? (unsigned)f0[c] : 0u; }
    has_x = a.F != a.Fm;
    wutt = -1;
    blk_utt = -1;
    xutt = -1;
    wx = 0.f;
    nw = 1.f;
    in_lds = false;
  }
  __device__ __forceinline__ float nwk(int k) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nw), k));
  }
  __device__ __forceinline__ float wxk(int k) const {       // W[utt][F-1][k], wave-uniform (k: compile time)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wx), k));
  }
  // workgroup prologue (rank > 8): stage W of the utterance of the workgroup's first frame
  __device__ __forceinline__ void stage_block_w() {
    if (WPRIV || !a.w_blk_lds) return;
    const int wpb = blockDim.x >> 6, nw = gridDim.x * wpb, per = (a.NT + nw - 1) / nw;
    const int n_first = blockIdx.x * wpb * per;
    if (n_first < a.NT) {
      blk_utt = a.frame_utt[n_first];
      const f32x4* src = reinterpret_cast<const f32x4*>(a.W + (size_t)blk_utt * a.Fs * KP);
      for (int e = threadIdx.x; e < a.Fs * KP / 4; e += blockDim.x) put_t(e, src[e]);
    }
    __syncthreads();
  }
  __device__ __forceinline__ void set_utt(int utt, const float* normW = nullptr) {        // wave-uniform
    if (utt != xutt) {
      xutt = utt;
      // (uniform base + 32-bit lane offset, the lane number opaque: nothing of this rare branch is precomputed outside the
      // frame loop and kept -- spilled -- across it)
      unsigned lo = (unsigned)lane;
      asm volatile("" : "+v"(lo));
      const unsigned lk = lo < (unsigned)KP ? lo : 0u;
      wx = (has_x && lo < (unsigned)KP) ? (a.W + ((size_t)utt * a.Fs + a.F - 1) * KP)[lk] : 0.f;
      if (normW) nw = lo < (unsigned)KP ? (normW + (size_t)utt * KP)[lk] : 1.f;
      // retired inside the branch: merged with the other path as "maybe pending", their first use -- behind the row
      // loads -- would be waited for with vmcnt(0), i.e. together with every row
      __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
    }
    if (!WPRIV) { in_lds = utt == blk_utt; return; }
    in_lds = true;
    if (utt == wutt) return;
    wutt = utt;
    __builtin_amdgcn_wave_barrier();                        // earlier reads of the previous utterance's rows
    const f32x4* src = reinterpret_cast<const f32x4*>(a.W + (size_t)utt * a.Fs * KP);
    unsigned e0 = (unsigned)lane;
    asm volatile("" : "+v"(e0));
    for (unsigned e = e0; e < (unsigned)(a.Fs * KP / 4); e += 64u) put_t((int)e, src[e]);
    __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): the rows are in LDS (gfx9 encoding)
    __builtin_amdgcn_wave_barrier();
  }
  // LDS copy of W[utt]: TRANSPOSED, wl[k][f].  A lane's bins are f0 + 4 lane .. +3: with the global layout [f][k]
  // (row = 8 ranks = 32 B.. 128 B) the 64 lanes of a read hit one or two LDS banks (32-way conflicts: the LDS, not
  // HBM or the VALU, bounded H/g: 45 conflicted reads per frame); rank-major, a lane reads its 4 bins of one rank
  // with one ds_read_b128 and consecutive lanes are 16 B apart: conflict-free.
  __device__ __forceinline__ void put_t(int e, const f32x4 v) {          // e-th float4 of W[utt] in global order [f][k]
    const int f = e / (KP / 4), k = (e - f * (KP / 4)) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) wl[(k + j) * a.Fs + f] = v[j];
  }
  template <bool L>
  __device__ __forceinline__ const float* w_row(int utt, int f) const {   // global layout only
    return a.W + ((size_t)utt * a.Fs + f) * KP;
  }
  // Vb = sum_k W[f,k] h[k] for this lane's bins (+ the extra bin); set_utt(utt) first
  template <bool L>
  __device__ __forceinline__ void noise_var_(int utt, const float (&h)[KP], f32x4 (&vb)[NCH], float& vbx) const {
    if (L) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (cv[c]) {
#pragma unroll
          for (int k = 0; k < KP; ++k) v += *reinterpret_cast<const f32x4*>(wl + k * a.Fs + f0[c]) * h[k];
        }
        vb[c] = cv[c] ? v : f32x4{1.f, 1.f, 1.f, 1.f};
      }
      vbx = 1.f;
      if (has_x) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) v += wxk(k) * h[k];
        vbx = v;
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      vb[c] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (cv[c]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float v = 0.f;
          const float* wrow = w_row<L>(utt, f0[c] + t);
#pragma unroll
          for (int k = 0; k < KP; k += 4) {
            // (one fused multiply-add per rank, in rank order: the order and rounding of the LDS form above, so that a frame's
            // result does not depend on whether its utterance's W happens to be the copy staged in LDS -- i.e. on the batch)
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) v = __builtin_fmaf(w4[j], h[k + j], v);
          }
          vb[c][t] = v;
        }
      }
    }
    vbx = 1.f;
    if (has_x) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < KP; ++k) v += wxk(k) * h[k];
      vbx = v;
    }
  }
  __device__ __forceinline__ void noise_var(int utt, const float (&h)[KP], f32x4 (&vb)[NCH], float& vbx) const {
    if (WPRIV || in_lds) noise_var_<true>(utt, h, vb, vbx);
    else noise_var_<false>(utt, h, vb, vbx);
  }
  // num_k = sum_f W[f,k] P[f], den_k = sum_f W[f,k] A[f] over this lane's bins (+ the extra bin in lane 0's terms)
  template <bool L>
  __device__ __forceinline__ void w_dot_(int utt, int k, const f32x4 (&P)[NCH], const f32x4 (&A)[NCH], float px, float ax,
                                         float& nu, float& de) const {
    nu = de = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (cv[c]) {
        if (L) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(wl + k * a.Fs + f0[c]);
#pragma unroll
          for (int t = 0; t < 4; ++t) { nu += w4[t] * P[c][t]; de += w4[t] * A[c][t]; }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float w = w_row<L>(utt, f0[c] + t)[k];
            nu += w * P[c][t];
            de += w * A[c][t];
          }
        }
      }
    if (has_x) {
      const float w = wxk(k);
      nu += w * px;
      de += w * ax;
    }
  }
  __device__ __forceinline__ void w_dot(int utt, int k, const f32x4 (&P)[NCH], const f32x4 (&A)[NCH], float px, float ax,
                                        float& nu, float& de) const {
    if (WPRIV || in_lds) w_dot_<true>(utt, k, P, A, px, ax, nu, de);
    else w_dot_<false>(utt, k, P, A, px, ax, nu, de);
  }
  __device__ __forceinline__ void ext_var(int n, f32x4 (&vb)[NCH], float& vbx) const {   // caller-given Vb (noNMF)
    const float* row = a.Vb + (size_t)n * a.Fs;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      vb[c] = cv[c] ? *reinterpret_cast<const f32x4*>(row + fo[c]) : f32x4{1.f, 1.f, 1.f, 1.f};
    vbx = has_x ? row[a.F - 1] : 1.f;
  }
  __device__ __forceinline__ void load_x2(int n, f32x4 (&x2)[NCH], float& x2x) const {
    const float* row = a.X2 + (size_t)n * a.Fs;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      x2[c] = cv[c] ? *reinterpret_cast<const f32x4*>(row + fo[c]) : f32x4{0.f, 0.f, 0.f, 0.f};
    x2x = has_x ? row[a.F - 1] : 0.f;
  }
  // write a per-bin result row (bins >= F zeroed)
  __device__ __forceinline__ void store_row(float* dst, const f32x4 (&v)[NCH], float vx) const {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (cv[c]) *reinterpret_cast<f32x4*>(dst + fo[c]) = v[c];
    if (a.Fm + lane < a.Fs) dst[a.Fm + lane] = (lane == 0 && has_x) ? vx : 0.f;
  }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 rcp2(const f32x2 x) { return f32x2{fast_rcp(x[0]), fast_rcp(x[1])}; }
__device__ __forceinline__ f32x2 log2_2(const f32x2 x) { return f32x2{fast_log2(x[0]), fast_log2(x[1])}; }

// A batch of up to RB rows of one frame in registers (this lane's bins, still packed as stored): every load of
// the batch is issued before the first use, so a wavefront keeps RB x 0.5-1 KiB in flight -- the kernels are
// latency-bound otherwise (86 % of the wave time in s_waitcnt with 4 loads in flight) -- and the H / g / cost
// passes of hg_stream reuse the registers instead of reading the rows again when the frame fits one batch.
// RT > 0: the frame has exactly RT rows (compile time: the row loops carry no checks, so the compiler interleaves the
// rows' dependent chains); RT = 0: any row count up to RB (one uniform branch per row).
template <int NCH, typename ST, int DIV = 1, int RT = 0, int RBX = 0>
struct RowBatch {
  // rows per batch; RBX > 0 overrides (hg_stream with two 256-bin chunks of bf16 rows: 32 rows x 4 registers hold a whole
  // 30-sample frame, so its three passes read the rows once instead of re-reading two batches of 16 in every pass)
  static constexpr int RB = RBX > 0 ? RBX : (sizeof(ST) == 2 ? 32 : 16) / NCH / DIV;
  static_assert(RT <= RB, "RT");
  __device__ __forceinline__ bool on(int r) const { return RT > 0 ? r < RT : r < nr; }
  using raw_t = typename std::conditional<sizeof(ST) == 4, f32x4, bf16x4>::type;
  raw_t raw[RB][NCH];
  unsigned xbits;              // lane j: extra bin (F-1) of row j of the batch, as loaded (bf16 bits in the low half / float bits);
                               // x() converts at the use, so no load site waits for it before issuing the row loads
  int nr;
  // rows r0 .. r0+nr-1 of the frame whose store block starts at `base`; scol = the frame's column of the
  // sample-major slot map (stride NT)
  // lane j: slot of row r0 + j of the frame (the frame's column of the sample-major slot map, stride NT)
  template <typename FC>
  static __device__ __forceinline__ int load_slots(const FC& fc, const int32_t* scol, int r0, int R) {
    const int nr_ = R - r0 < RB ? R - r0 : RB;
    // (a 32-bit BYTE offset from a uniform base: SGPR pair + one VGPR, no 64-bit address pair kept -- and spilled -- per lane)
    return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(scol) + (unsigned)(r0 + (fc.lane < nr_ ? fc.lane : 0)) * (unsigned)fc.a.NT * 4u);
  }
  // rows r0 .. r0+nr-1 of the frame whose store block starts at `base`, slots from load_slots
  template <typename FC>
  __device__ __forceinline__ void load_rows(const FC& fc, const ST* base, int sl, int r0, int R) {
    nr = RT > 0 ? RT : (R - r0 < RB ? R - r0 : RB);
    // (row 0 exists in every batch and is loaded without a check: its readlane of `sl` puts the wait for the slot
    // load on the straight path -- behind a branch, the compiler waits with vmcnt(0) in EVERY row's block, i.e. for
    // the row load issued just before it, and the batch becomes thirty serial round trips)
#pragma unroll
    for (int r = 0; r < RB; ++r)
      if (r == 0 || on(r)) {
#if VN_HG_PROBE == 2
        const ST* row = reinterpret_cast<const ST*>(fc.a.VsS) + (size_t)(__builtin_amdgcn_readlane(sl, r) & 1) * fc.a.Fs;
#else
        const ST* row = base + (size_t)__builtin_amdgcn_readlane(sl, r) * fc.a.Fs;
#endif
#pragma unroll
        for (int c = 0; c < NCH; ++c) raw[r][c] = VN_ROW_NT ? __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(row + fc.fo[c])) : *reinterpret_cast<const raw_t*>(row + fc.fo[c]);
      }
    // extra bin, raw bits (x() converts at the use: a conversion here would wait for every load above)
    if (fc.has_x) {
      if constexpr (sizeof(ST) == 2) xbits = reinterpret_cast<const unsigned short*>(base)[(size_t)sl * fc.a.Fs + fc.a.F - 1];
      else xbits = reinterpret_cast<const unsigned*>(base)[(size_t)sl * fc.a.Fs + fc.a.F - 1];
    } else xbits = 0u;
  }
  template <typename FC>
  __device__ __forceinline__ void load(const FC& fc, const ST* base, const int32_t* scol, int r0, int R) {
    load_rows(fc, base, load_slots(fc, scol, r0, R), r0, R);
  }
  // The same through a buffer resource over the whole store (below 4 GB): a row's address is the resource (SGPRs) +
  // this lane's constant byte offset (one VGPR) + a scalar offset built from the row's slot -- no 64-bit address
  // pair per row in vector registers (thirty of them per frame spilled in the pipelined kernels).
  template <typename FC>
  __device__ __forceinline__ void load_rows_buf(const FC& fc, __amdgpu_buffer_rsrc_t rs, unsigned frame_off, int sl, int R, int r0 = 0) {
    nr = RT > 0 ? RT : (R - r0 < RB ? R - r0 : RB);
    const unsigned rowb = (unsigned)fc.a.Fs * (unsigned)sizeof(ST);
    // (raw bits: the conversion would wait for this load before the row loads below are even issued; x()
    // converts at the first use, a frame later)
    if (fc.has_x) {
      const unsigned o = frame_off + (unsigned)sl * rowb + (unsigned)(fc.a.F - 1) * (unsigned)sizeof(ST);
      if constexpr (sizeof(ST) == 2) xbits = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, o, 0, 0);
      else xbits = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0);
    } else xbits = 0u;
#pragma unroll
    for (int r = 0; r < RB; ++r)
      if (r == 0 || on(r)) {
        const unsigned so = frame_off + (unsigned)__builtin_amdgcn_readlane(sl, r) * rowb;       // uniform
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const unsigned vo = (unsigned)(fc.cv[c] ? fc.f0[c] : 0) * (unsigned)sizeof(ST);
          if constexpr (sizeof(ST) == 2) raw[r][c] = __builtin_bit_cast(raw_t, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0));
          else raw[r][c] = __builtin_bit_cast(raw_t, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
        }
      }
  }
  // f(r) for every row of the batch, r a compile-time constant after unrolling.  Runtime row counts: one uniform
  // branch per CHUNK of rows instead of one per row (90 branches per frame in hg_stream, each ending a scheduling
  // region and a precise vmcnt count); only the last, partial chunk checks row by row.
  template <typename FN>
  __device__ __forceinline__ void for_rows(FN f) const {
    constexpr int CH = VN_ROWCHUNK;
#pragma unroll
    for (int r0 = 0; r0 < RB; r0 += CH) {
      if (RT > 0 ? r0 + CH <= RT : r0 + CH <= nr) {
#pragma unroll
        for (int r = r0; r < r0 + CH && r < RB; ++r) f(r);
      } else {
#pragma unroll
        for (int r = r0; r < r0 + CH && r < RB; ++r)
          if (on(r)) f(r);
      }
    }
  }
  __device__ __forceinline__ float x() const {
    if constexpr (sizeof(ST) == 2) return __builtin_bit_cast(float, xbits << 16);
    else return __builtin_bit_cast(float, xbits);
  }
  // Between two passes over the same batch: without this the compiler keeps the UNPACKED floats of the first pass
  // alive for the next one (common-subexpression elimination of the bf16 -> float conversions: 120 registers).
  __device__ __forceinline__ void repack() {
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(raw[r][c]));
  }
  __device__ __forceinline__ void get(int r, f32x4 (&v)[NCH]) const {     // r: compile-time after unrolling
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = f32x4{(float)raw[r][c][0], (float)raw[r][c][1], (float)raw[r][c][2], (float)raw[r][c][3]};
  }
  // The extra bin is handled across the lanes instead of along the row loop: lane j holds it for row j, so a
  // sum over the samples is one wavefront reduction of a per-lane term (xmask() zeroes the lanes without a row).
  template <typename FC>
  __device__ __forceinline__ float xmask(const FC& fc) const { return (fc.has_x && fc.lane < nr) ? 1.f : 0.f; }
};

// frames of a wavefront: a contiguous block, so consecutive frames share the utterance (W rows stay in L1)
__device__ __forceinline__ void wave_frames(int NT, int& n_beg, int& n_end) {
  const int wpb = blockDim.x >> 6;
  const int gw = blockIdx.x * wpb + (threadIdx.x >> 6), nw = gridDim.x * wpb;
  const int per = (NT + nw - 1) / nw;
  // (wave-uniform by construction; said so, the frame loop and every per-frame address run on the scalar unit)
  n_beg = __builtin_amdgcn_readfirstlane(gw * per);
  n_end = n_beg + per < NT ? n_beg + per : NT;
}

template <int NCH, int KP, typename ST>
__global__ __launch_bounds__(256, (KP <= 8 && NCH == 1) ? VN_WS_WAVES : 2) void wstats_stream_kernel(const StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  FrameCtx<NCH, KP, ST> fc(a, wlds + (KP <= 8 ? (size_t)(threadIdx.x >> 6) * a.Fs * KP : 0));
  fc.stage_block_w();
  using RBt = RowBatch<NCH, ST>;
  int n_beg, n_end;
  wave_frames(a.NT, n_beg, n_end);
  if (a.R <= RBt::RB) {
    // The frame fits one batch: two half batches, each refilled with the NEXT frame's rows as soon as it has been
    // consumed, so that half a frame of loads is in flight while the other half is computed.
    using Half = RowBatch<NCH, ST, 2>;
    constexpr int HB = Half::RB;
    Half h0, h1;
    auto base_of = [&](int n) { return reinterpret_cast<const ST*>(a.VsS) + (size_t)n * a.Rs * a.Fs; };
    const int R0 = a.R < HB ? a.R : HB;              // rows of the first half
    if (n_beg < n_end) {
      h0.load(fc, base_of(n_beg), a.src + n_beg, 0, R0);
      if (a.R > HB) h1.load(fc, base_of(n_beg), a.src + n_beg, HB, a.R); else h1.nr = 0, h1.xbits = 0u;
    }
    for (int n = n_beg; n < n_end; ++n) {
      const int utt = a.frame_utt[n];
      fc.set_utt(utt);
      const float gn = a.g[n];
      float h[KP];
#pragma unroll
      for (int k = 0; k < KP; k += 4) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(a.Ht + (size_t)n * KP + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[k + t] = vn_uniform(hv[t]);      // (wave-uniform: scalar registers)
      }
      f32x4 vb[NCH], x2[NCH], a1[NCH], a2[NCH];
      float vbx, x2x, a1x = 0.f, a2x = 0.f;
      fc.noise_var(utt, h, vb, vbx);
      fc.load_x2(n, x2, x2x);
#pragma unroll
      for (int c = 0; c < NCH; ++c) a1[c] = a2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto consume = [&](const Half& hb) {
#pragma unroll
        for (int r = 0; r < HB; ++r)
          if (r < hb.nr) {
            f32x4 v[NCH];
            hb.get(r, v);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const f32x2 g2 = {gn, gn};
              const f32x2 q0 = rcp2(g2 * v[c].lo + vb[c].lo), q1 = rcp2(g2 * v[c].hi + vb[c].hi);
              a1[c].lo += q0; a1[c].hi += q1;
              a2[c].lo = q0 * q0 + a2[c].lo; a2[c].hi = q1 * q1 + a2[c].hi;
            }
          }
        const float q = fast_rcp(gn * hb.x() + vbx) * hb.xmask(fc);
        a1x += wave_sum(q);
        a2x += wave_sum(q * q);
      };
      consume(h0);
      if (n + 1 < n_end) h0.load(fc, base_of(n + 1), a.src + n + 1, 0, R0);
      if (a.R > HB) {
        consume(h1);
        if (n + 1 < n_end) h1.load(fc, base_of(n + 1), a.src + n + 1, HB, a.R);
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) a2[c][t] *= x2[c][t];
      fc.store_row(a.A1 + (size_t)n * a.Fs, a1, a1x);
      fc.store_row(a.P + (size_t)n * a.Fs, a2, a2x * x2x);
    }
    return;
  }
  for (int n = n_beg; n < n_end; ++n) {
    const ST* base = reinterpret_cast<const ST*>(a.VsS) + (size_t)n * a.Rs * a.Fs;
    const int32_t* srow = a.src + n;
    RBt rb;
    rb.load(fc, base, srow, 0, a.R);                 // (issued first: the longest latency of the frame)
    const int utt = a.frame_utt[n];
    fc.set_utt(utt);
    const float gn = a.g[n];
    float h[KP];
#pragma unroll
    for (int k = 0; k < KP; k += 4) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(a.Ht + (size_t)n * KP + k);
#pragma unroll
      for (int t = 0; t < 4; ++t) h[k + t] = vn_uniform(hv[t]);
    }
    f32x4 vb[NCH], x2[NCH], a1[NCH], a2[NCH];
    float vbx, x2x, a1x = 0.f, a2x = 0.f;
    fc.noise_var(utt, h, vb, vbx);
    fc.load_x2(n, x2, x2x);
#pragma unroll
    for (int c = 0; c < NCH; ++c) a1[c] = a2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < a.R; r0 += RBt::RB) {
      if (r0 > 0) rb.load(fc, base, srow, r0, a.R);
#pragma unroll
      for (int r = 0; r < RBt::RB; ++r)
        if (r < rb.nr) {
          f32x4 v[NCH];
          rb.get(r, v);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const f32x2 g2 = {gn, gn};
            const f32x2 q0 = rcp2(g2 * v[c].lo + vb[c].lo), q1 = rcp2(g2 * v[c].hi + vb[c].hi);
            a1[c].lo += q0; a1[c].hi += q1;
            a2[c].lo = q0 * q0 + a2[c].lo; a2[c].hi = q1 * q1 + a2[c].hi;
          }
        }
      const float q = fast_rcp(gn * rb.x() + vbx) * rb.xmask(fc);
      a1x += wave_sum(q);
      a2x += wave_sum(q * q);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) a2[c][t] *= x2[c][t];
    fc.store_row(a.A1 + (size_t)n * a.Fs, a1, a1x);
    fc.store_row(a.P + (size_t)n * a.Fs, a2, a2x * x2x);
  }
}

// RT > 0: exactly RT samples per frame (no per-row branch: 90 uniform branches per frame otherwise)
template <int NCH, int KP, typename ST, int RT = 0>
__global__ __launch_bounds__(256, (KP <= 8 && NCH == 1 && !(VN_HG_FULLF && sizeof(ST) == 4)) ? VN_HG_WAVES : 2) void hg_stream_kernel(const StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  FrameCtx<NCH, KP, ST> fc(a, wlds + (KP <= 8 ? (size_t)(threadIdx.x >> 6) * a.Fs * KP : 0));
  fc.stage_block_w();
  // (two chunks, rank <= 16: 0.306 -> 0.286 ms on the 1024-pt shape; at rank 32 the registers do not suffice: 0.535 -> 0.608)
  // float rows (bf16x3 mode), one chunk: 32 rows x 4 registers at two wavefronts per SIMD hold the whole frame as well (16 per
  // batch re-read both batches in every pass: 3x the traffic of a kernel that is memory-bound with float rows)
  using RBt = RowBatch<NCH, ST, 1, RT, ((VN_HG_FULL2 && sizeof(ST) == 2 && NCH == 2 && KP <= VN_HG_FULL2_KMAX) || (VN_HG_FULLF && sizeof(ST) == 4 && NCH == 1 && KP <= 8)) ? 32 : 0>;
  int n_beg, n_end;
  wave_frames(a.NT, n_beg, n_end);
#if VN_HG_STAGGER > 0
  {
    // The wavefronts of a SIMD start together and do the same work per frame, so they would all load, then all
    // compute, in convoy (T = T_mem + T_compute).  Stagger them by their hardware wave slot: slot i waits i quarter
    // periods once, and the load phase of one then falls under the arithmetic of the others for the rest of the launch.
    const unsigned slot = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((4 - 1) << 11)) & 3u;     // HW_REG_HW_ID, wave_id[3:0]
    for (unsigned k = 0; k < slot * VN_HG_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
  }
#endif
  const bool one = a.R <= RBt::RB;                    // the frame fits one batch: its rows are read once
  for (int n = n_beg; n < n_end; ++n) {
    const ST* base = reinterpret_cast<const ST*>(a.VsS) + (size_t)n * a.Rs * a.Fs;
    const int32_t* srow = a.src + n;
    RBt rb;
    // Order of the frame's requests (vmcnt counts in order, and behind the per-row branches the compiler can only wait
    // with vmcnt(0)): first every small operand of the frame and the slot map, one round trip together; then, with
    // nothing else outstanding, the rows; nothing is requested behind the rows until they are consumed.
    // (Requesting the operands a frame ahead, so that the rows go out at once, was measured: 0.179 against 0.172 ms.)
    const int utt = a.frame_utt[n];
    const float gn = a.g[n];
    const int sl = RBt::load_slots(fc, srow, 0, a.R);
    f32x4 vb[NCH], x2[NCH];
    float vbx, x2x;
    fc.load_x2(n, x2, x2x);
    float hs[KP];
    if (a.gains_only) {
      fc.ext_var(n, vb, vbx);
    } else {
#pragma unroll
      for (int k = 0; k < KP; k += 4) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(a.Ht + (size_t)n * KP + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) hs[k + t] = hv[t];
      }
    }
    fc.set_utt(utt, a.gains_only ? nullptr : a.normW);
    rb.load_rows(fc, base, sl, 0, a.R);
    const f32x2 gn2 = {gn, gn};
    if (!a.gains_only) {
      // ---- H update (mcem.py:118-121): W already updated and normalised; H carries the pending column norms
#pragma unroll
      // (wave-uniform values, moved to the scalar registers: KP vector registers each for the old and the new activations
      // otherwise -- 64 of them at rank 32, which is what kept a whole frame's rows from fitting beside them)
      for (int k = 0; k < KP; ++k) hs[k] = vn_uniform(hs[k] * fc.nwk(k));
      fc.noise_var(utt, hs, vb, vbx);
      f32x4 a1[NCH], a2[NCH];
      float a1x = 0.f, a2x = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) a1[c] = a2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int r0 = 0; r0 < a.R; r0 += RBt::RB) {
        if (r0 > 0) rb.load(fc, base, srow, r0, a.R);
        auto row1 = [&](int r) {
          f32x4 v[NCH];
          rb.get(r, v);
#if VN_HG_PROBE == 1
#pragma unroll
          for (int c = 0; c < NCH; ++c) { a1[c] += v[c]; a2[c] += v[c]; }
          return;
#endif
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            // two bins per instruction (v_pk_fma / v_pk_add_f32: next to transcendentals a packed instruction costs the
            // issue slot of a plain one -- tools/ubench/overlap.hip)
            const f32x2 q0 = rcp2(gn2 * v[c].lo + vb[c].lo), q1 = rcp2(gn2 * v[c].hi + vb[c].hi);
            a1[c].lo += q0; a1[c].hi += q1;
            a2[c].lo = q0 * q0 + a2[c].lo; a2[c].hi = q1 * q1 + a2[c].hi;
          }
        };
        rb.for_rows(row1);
        const float q = fast_rcp(gn * rb.x() + vbx) * rb.xmask(fc);
        a1x += wave_sum(q);
        a2x += wave_sum(q * q);
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a2[c][t] = fc.cv[c] ? a2[c][t] * x2[c][t] : 0.f;
          a1[c][t] = fc.cv[c] ? a1[c][t] : 0.f;
        }
      const bool lead = fc.lane == 0 && fc.has_x;
      a2x = lead ? a2x * x2x : 0.f;
      a1x = lead ? a1x : 0.f;
      float hn[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        float nu, de;
        fc.w_dot(utt, k, a2, a1, a2x, a1x, nu, de);
        nu = wave_sum(nu);
        de = wave_sum(de);
        hn[k] = vn_uniform(k < a.K ? hs[k] * __builtin_amdgcn_sqrtf(nu * fast_rcp(de)) : 0.f);       // mcem.py:121
      }
      if (fc.lane == 0) {
#pragma unroll
        for (int k = 0; k < KP; k += 4)
          *reinterpret_cast<f32x4*>(a.Ht + (size_t)n * KP + k) = f32x4{hn[k], hn[k + 1], hn[k + 2], hn[k + 3]};
      }
      fc.noise_var(utt, hn, vb, vbx);                                                     // mcem.py:124-125
    }
    // ---- g update (mcem.py:138-142 / :564-568)
    float nu = 0.f, de = 0.f;
    if (one && VN_HG_REPACK) rb.repack();      // (no unpacked floats carried from pass to pass: see RowBatch::repack)
    {
      f32x4 ng[NCH], dg[NCH];
      float ngx = 0.f, dgx = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) ng[c] = dg[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int r0 = 0; r0 < a.R; r0 += RBt::RB) {
        if (!one) rb.load(fc, base, srow, r0, a.R);
        auto row2 = [&](int r) {
          f32x4 v[NCH];
          rb.get(r, v);
#if VN_HG_PROBE == 1
#pragma unroll
          for (int c = 0; c < NCH; ++c) { dg[c] += v[c]; ng[c] += v[c]; }
          return;
#endif
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const f32x2 q0 = rcp2(gn2 * v[c].lo + vb[c].lo), q1 = rcp2(gn2 * v[c].hi + vb[c].hi);
            const f32x2 vq0 = v[c].lo * q0, vq1 = v[c].hi * q1;
            dg[c].lo += vq0; dg[c].hi += vq1;                                   // sum_r Vs / Vx
            ng[c].lo = vq0 * q0 + ng[c].lo; ng[c].hi = vq1 * q1 + ng[c].hi;     // sum_r Vs / Vx^2
          }
        };
        rb.for_rows(row2);
        const float q = fast_rcp(gn * rb.x() + vbx), vq = rb.x() * q * rb.xmask(fc);
        dgx += wave_sum(vq);
        ngx += wave_sum(vq * q);
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (fc.cv[c]) {
#pragma unroll
          for (int t = 0; t < 4; ++t) { nu += x2[c][t] * ng[c][t]; de += dg[c][t]; }
        }
      if (fc.lane == 0 && fc.has_x) { nu += x2x * ngx; de += dgx; }
      nu = wave_sum(nu);
      de = wave_sum(de);
    }
    const float gnew = gn * __builtin_amdgcn_sqrtf(nu * fast_rcp(de));                     // mcem.py:142
    if (fc.lane == 0) a.g[n] = gnew;
    const f32x2 gw2 = {gnew, gnew};
    if (one && VN_HG_REPACK) rb.repack();
    // ---- cost (mcem.py:70) with the refreshed variances (:151-152); samples two at a time:
    // log Vx0 + log Vx1 = log(Vx0 Vx1), 1/Vx0 + 1/Vx1 = (Vx0 + Vx1)/(Vx0 Vx1)
    f32x4 cl[NCH], cx[NCH];
    float clx = 0.f, cxx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) cl[c] = cx[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < a.R; r0 += RBt::RB) {
      if (!one) rb.load(fc, base, srow, r0, a.R);
#pragma unroll
      for (int r = 0; r < RBt::RB; r += 2) {
        if (VN_HG_PROBE == 1) continue;
        if (rb.on(r + 1)) {
          f32x4 v0[NCH], v1[NCH];
          rb.get(r, v0);
          rb.get(r + 1, v1);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const f32x2 xa0 = gw2 * v0[c].lo + vb[c].lo, xa1 = gw2 * v1[c].lo + vb[c].lo;
            const f32x2 xb0 = gw2 * v0[c].hi + vb[c].hi, xb1 = gw2 * v1[c].hi + vb[c].hi;
            const f32x2 pa = xa0 * xa1, pb = xb0 * xb1;
            cl[c].lo += log2_2(pa); cl[c].hi += log2_2(pb);
            cx[c].lo = (xa0 + xa1) * rcp2(pa) + cx[c].lo; cx[c].hi = (xb0 + xb1) * rcp2(pb) + cx[c].hi;
          }
        } else if (rb.on(r)) {
          f32x4 v0[NCH];
          rb.get(r, v0);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const f32x2 xa = gw2 * v0[c].lo + vb[c].lo, xb = gw2 * v0[c].hi + vb[c].hi;
            cl[c].lo += log2_2(xa); cl[c].hi += log2_2(xb);
            cx[c].lo += rcp2(xa); cx[c].hi += rcp2(xb);
          }
        }
      }
      const float xm = rb.xmask(fc), x0 = gnew * rb.x() + vbx;
      clx += wave_sum(fast_log2(x0) * xm);
      cxx += wave_sum(fast_rcp(x0) * xm);
    }
    float cs = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (fc.cv[c]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) cs += cl[c][t] * LN2_F + x2[c][t] * cx[c][t];
      }
    if (fc.lane == 0 && fc.has_x) cs += clx * LN2_F + x2x * cxx;
    const double cd = sum_rows4_d((double)sum_row16(cs));     // rows in fp32 (DPP), then fp64 across the 4 rows
    if (fc.lane == 0) a.cost_frames[n] = cd;
  }
}

template <int NCH, int KP, typename ST>
__global__ __launch_bounds__(256, (KP <= 8 && NCH == 1) ? VN_WS_WAVES : 2) void wf_stream_kernel(const StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  FrameCtx<NCH, KP, ST> fc(a, wlds + (KP <= 8 ? (size_t)(threadIdx.x >> 6) * a.Fs * KP : 0));
  fc.stage_block_w();
  using RBt = RowBatch<NCH, ST>;
  int n_beg, n_end;
  wave_frames(a.NT, n_beg, n_end);
  const float invR = 1.0f / (float)a.R;
  for (int n = n_beg; n < n_end; ++n) {
    const ST* base = reinterpret_cast<const ST*>(a.VsS) + (size_t)n * a.Rs * a.Fs;
    const int32_t* srow = a.src + n;
    RBt rb;
    rb.load(fc, base, srow, 0, a.R);
    const int utt = a.frame_utt[n];
    fc.set_utt(utt);
    const float gn = a.g[n];
    f32x4 vb[NCH];
    float vbx;
    if (a.Vb) {
      fc.ext_var(n, vb, vbx);
    } else {
      float h[KP];
#pragma unroll
      for (int k = 0; k < KP; k += 4) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(a.Ht + (size_t)n * KP + k);
#pragma unroll
        for (int t = 0; t < 4; ++t) h[k + t] = vn_uniform(hv[t]);      // (wave-uniform: scalar registers)
      }
      fc.noise_var(utt, h, vb, vbx);
    }
    f32x4 ws[NCH], wn[NCH];
    float wsx = 0.f, wnx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) ws[c] = wn[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < a.R; r0 += RBt::RB) {
      if (r0 > 0) rb.load(fc, base, srow, r0, a.R);
#pragma unroll
      for (int r = 0; r < RBt::RB; ++r)
        if (r < rb.nr) {
          f32x4 v[NCH];
          rb.get(r, v);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const f32x2 g2 = {gn, gn};
            const f32x2 s0 = g2 * v[c].lo, s1 = g2 * v[c].hi;
            const f32x2 q0 = rcp2(s0 + vb[c].lo), q1 = rcp2(s1 + vb[c].hi);
            ws[c].lo = s0 * q0 + ws[c].lo; ws[c].hi = s1 * q1 + ws[c].hi;                         // g Vs / Vx
            wn[c].lo = vb[c].lo * q0 + wn[c].lo; wn[c].hi = vb[c].hi * q1 + wn[c].hi;             // Vb / Vx
          }
        }
      const float sc = gn * rb.x(), q = fast_rcp(sc + vbx) * rb.xmask(fc);
      wsx += wave_sum(sc * q);
      wnx += wave_sum(vbx * q);
    }
    // S_hat = WFs X, N_hat = WFn X (mcem.py:175-176); bins >= F zeroed
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (fc.cv[c]) {
        const size_t o = (size_t)n * a.Fs + fc.f0[c];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float ms = ws[c][t] * invR, mn = wn[c][t] * invR;
          const float xr = a.X[2 * (o + t)], xi = a.X[2 * (o + t) + 1];
          a.S_hat[2 * (o + t)] = ms * xr;  a.S_hat[2 * (o + t) + 1] = ms * xi;
          a.N_hat[2 * (o + t)] = mn * xr;  a.N_hat[2 * (o + t) + 1] = mn * xi;
          if (a.WFs) a.WFs[o + t] = ms;
          if (a.WFn) a.WFn[o + t] = mn;
        }
      }
    if (a.Fm + fc.lane < a.Fs) {
      const size_t o = (size_t)n * a.Fs + a.Fm + fc.lane;
      const bool lead = fc.lane == 0 && fc.has_x;
      const float ms = lead ? wsx * invR : 0.f, mn = lead ? wnx * invR : 0.f;
      const float xr = a.X[2 * o], xi = a.X[2 * o + 1];
      a.S_hat[2 * o] = ms * xr;  a.S_hat[2 * o + 1] = ms * xi;
      a.N_hat[2 * o] = mn * xr;  a.N_hat[2 * o + 1] = mn * xi;
      if (a.WFs) a.WFs[o] = ms;
      if (a.WFn) a.WFn[o] = mn;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Frame-pipelined forms (the frame's samples fit one register batch: R <= RowBatch::RB).  Two wavefronts per SIMD
// with 256 registers each instead of four with 128: a wavefront holds TWO batches, the frame it computes on and the
// NEXT frame's rows, requested in full before the first arithmetic instruction of the current frame.  Nothing in
// the frame body waits on vmcnt except for operands requested a whole frame earlier, there is no spill (a scratch
// reload would wait, through the in-order vmcnt, for the thirty row loads in flight), and every wavefront has
// 15 KB of loads outstanding all the time.
// ---------------------------------------------------------------------------------------------------------------
template <int NCH, int KP>
struct FrameSmall {            // the per-frame operands besides the rows, requested one frame ahead like the rows
  f32x4 x2[NCH];
  float x2x, g;
  float hl;                  // lane k: H[k, n] (one register instead of KP: two of these structs are live, and a spilled
                             // member's reload, waited for with vmcnt(0), waits for the whole row prefetch behind it)
  int utt;
  __device__ __forceinline__ void get_h(float (&h)[KP]) const {
#pragma unroll
    for (int k = 0; k < KP; ++k) h[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hl), k));
  }
};

template <int NCH, int KP, typename ST, int RT>
__global__ __launch_bounds__(256, 2) void wstats_stream2_kernel(const StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  FrameCtx<NCH, KP, ST> fc(a, wlds + (KP <= 8 ? (size_t)(threadIdx.x >> 6) * a.Fs * KP : 0));
  fc.stage_block_w();
  using RBt = RowBatch<NCH, ST, 1, RT>;
  using FS = FrameSmall<NCH, KP>;
  int n_beg, n_end;
  wave_frames(a.NT, n_beg, n_end);
  if (n_beg >= n_end) return;
  __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.VsS), 0, (int)((unsigned)a.NT * (unsigned)(a.Rs * a.Fs) * (unsigned)sizeof(ST)), 0x00020000);
  // slots are requested TWO frames ahead, rows and operands one frame ahead: the row addresses depend on the slots,
  // and a wait for the slots placed between two frames' row loads would expose a full memory round trip per frame
  auto request = [&](int n, RBt& rb, FS& s, int sl) {
    s.utt = a.frame_utt[n];
    s.g = a.g[n];
    s.hl = a.Ht[(size_t)n * KP + (fc.lane < KP ? fc.lane : 0)];
    fc.load_x2(n, s.x2, s.x2x);
    rb.load_rows_buf(fc, vrs, (unsigned)n * (unsigned)(a.Rs * a.Fs) * (unsigned)sizeof(ST), sl, a.R);
  };
  auto compute = [&](int n, RBt& rb, const FS& s) {
    fc.set_utt(s.utt);
    f32x4 vb[NCH], a1[NCH], a2[NCH];
    float vbx, a1x = 0.f, a2x = 0.f;
    float h[KP];
    s.get_h(h);
    fc.noise_var(s.utt, h, vb, vbx);
#pragma unroll
    for (int c = 0; c < NCH; ++c) a1[c] = a2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RBt::RB; ++r) {
      if (RT > 0 && (r % VN_ROWGRP) == 0) __builtin_amdgcn_sched_barrier(0);     // bound the interleave (registers)
      if (rb.on(r)) {
        f32x4 v[NCH];
        rb.get(r, v);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const f32x2 g2 = {s.g, s.g};
          const f32x2 q0 = rcp2(g2 * v[c].lo + vb[c].lo), q1 = rcp2(g2 * v[c].hi + vb[c].hi);
          a1[c].lo += q0; a1[c].hi += q1;
          a2[c].lo = q0 * q0 + a2[c].lo; a2[c].hi = q1 * q1 + a2[c].hi;
        }
      }
    }
    const float q = fast_rcp(s.g * rb.x() + vbx) * rb.xmask(fc);
    a1x = wave_sum(q);
    a2x = wave_sum(q * q);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) a2[c][t] *= s.x2[c][t];
    fc.store_row(a.A1 + (size_t)n * a.Fs, a1, a1x);
    fc.store_row(a.P + (size_t)n * a.Fs, a2, a2x * s.x2x);
  };
  auto slots = [&](int n) { return n < n_end ? RBt::load_slots(fc, a.src + n, 0, a.R) : 0; };
  RBt rbA, rbB;
  FS sA, sB;
  int sl1 = slots(n_beg + 1);
  request(n_beg, rbA, sA, slots(n_beg));
  for (int n = n_beg; n < n_end; n += 2) {
    int sl2 = 0, sl3 = 0;
    if (n + 1 < n_end) { request(n + 1, rbB, sB, sl1); sl2 = slots(n + 2); }
    compute(n, rbA, sA);
    if (n + 1 < n_end) {
      if (n + 2 < n_end) { request(n + 2, rbA, sA, sl2); sl3 = slots(n + 3); }
      compute(n + 1, rbB, sB);
    }
    sl1 = sl3;
  }
}


// ============================================================================
// Rotating-register forms (bf16 rows, one 256-bin chunk, rank <= 8, exactly RT samples per frame): the bench shapes.
// A wavefront keeps ONE frame's rows in registers (2 per row) and refills each row's registers with the same row of
// the NEXT frame right after its last use, so RT row loads stay in flight per wavefront all the time and the loop is
// straight-line: the compiler's vmcnt counts are exact (wait for the oldest row only), which no form with per-row
// branches or with a second batch of registers achieved (conservative vmcnt(0) there waits for the whole prefetch;
// spilled operands, reloaded through vmcnt as well, do the same).  Small per-frame operands are requested a frame ahead,
// the slot map two frames ahead.
// ============================================================================
template <int KP>
struct RotSmall {              // a frame's operands besides the rows
  f32x4 x2;
  // ONE load with per-lane addresses: lane k < KP: H[k, n]; lane KP: the frame's utterance (int bits); lane KP+1: g[n];
  // lane KP+2: X2 of the extra bin.  Read back with readlane at the first use, two frames later -- a scalar taken at the
  // load (the compiler moves a uniform value to an SGPR at once) would wait there, for every older row load as well.
  float pk;
  __device__ __forceinline__ float lane_f(int l) const { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pk), l)); }
  __device__ __forceinline__ int utt() const { return __builtin_amdgcn_readlane(__builtin_bit_cast(int, pk), KP); }
  __device__ __forceinline__ float g() const { return lane_f(KP + 1); }
  __device__ __forceinline__ float x2x() const { return lane_f(KP + 2); }
};

template <int KP, int RT>
struct RotCtx {
  using FC = FrameCtx<1, KP, __bf16>;
  const StreamArgs& a;
  const FC& fc;
  __amdgpu_buffer_rsrc_t vrs;
  unsigned rowb, frameb, voff, xvoff;
  const char* pk_base;         // per-lane base and stride (bytes per frame) of the packed small-operand load
  unsigned pk_stride;
  // two register sets: the frame in set S is consumed while set 1-S is refilled with the next frame's rows, row by
  // row behind the uses.  (One set refilled in place would do -- a row's registers are free after its last use -- but
  // the register allocator does not coalesce the loop-carried values and copies every row at the back edge, which
  // waits for the whole prefetch.)
  bf16x4 raw[2][RT];
  unsigned xb[2];              // lane j < RT: bf16 bits of the extra bin of row j
  __device__ RotCtx(const StreamArgs& a_, const FC& fc_) : a(a_), fc(fc_) {
    vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.VsS), 0, (int)((unsigned)a.NT * (unsigned)(a.Rs * a.Fs) * 2u), 0x00020000);
    rowb = (unsigned)a.Fs * 2u;
    frameb = (unsigned)a.Rs * rowb;
    voff = (unsigned)(fc.cv[0] ? fc.f0[0] : 0) * 2u;
    xvoff = (unsigned)(a.F - 1) * 2u;
    xb[0] = xb[1] = 0u;
    const int l = fc.lane;
    if (l == KP) { pk_base = reinterpret_cast<const char*>(a.frame_utt); pk_stride = 4u; }
    else if (l == KP + 1) { pk_base = reinterpret_cast<const char*>(a.g); pk_stride = 4u; }
    else if (l == KP + 2 && fc.has_x) { pk_base = reinterpret_cast<const char*>(a.X2 + a.F - 1); pk_stride = (unsigned)a.Fs * 4u; }
    else if (a.Ht) { pk_base = reinterpret_cast<const char*>(a.Ht + (l < KP ? l : 0)); pk_stride = (unsigned)KP * 4u; }
    else { pk_base = reinterpret_cast<const char*>(a.g); pk_stride = 4u; }
  }
  __device__ __forceinline__ int slots(int n) const {       // lane j: slot of row j of frame n (sample-major map)
    return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(a.src + n) + (unsigned)(fc.lane < RT ? fc.lane : 0) * (unsigned)a.NT * 4u);
  }
  __device__ __forceinline__ void req_small(int n, RotSmall<KP>& s) const {
    s.pk = *reinterpret_cast<const float*>(pk_base + (size_t)n * pk_stride);
    f32x4 x2v[1];
    float unused;
    fc.load_x2(n, x2v, unused);
    s.x2 = x2v[0];
  }
  // Row r of frame n into set S.  `after`: a value the last use of the registers' previous content produces -- the
  // load's address is made to depend on it, or the compiler hoists all RT refills above the frame's arithmetic.
  // vbase: voff, or an offset past the buffer (the hardware returns zeros without touching memory: no frame follows)
  template <int S>
  __device__ __forceinline__ void req_row(int n, int sl, int r, float after, unsigned vbase) {
    const unsigned so = (unsigned)n * frameb + (unsigned)__builtin_amdgcn_readlane(sl, r) * rowb;
    unsigned vo = vbase;
    asm volatile("" : "+v"(vo) : "v"(after));
    raw[S][r] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(vrs, vo, so, VN_ROW_AUX));
  }
  template <int S>
  __device__ __forceinline__ void req_x(int n, int sl, bool on) {
    if (fc.has_x) xb[S] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(vrs, on ? (unsigned)sl * rowb + xvoff : 0xF0000000u, (unsigned)n * frameb, 0);
  }
  template <int S>
  __device__ __forceinline__ f32x4 row(int r) const {
    return f32x4{(float)raw[S][r][0], (float)raw[S][r][1], (float)raw[S][r][2], (float)raw[S][r][3]};
  }
  template <int S>
  __device__ __forceinline__ float x() const { return __builtin_bit_cast(float, xb[S] << 16); }
  __device__ __forceinline__ float xmask() const { return (fc.has_x && fc.lane < RT) ? 1.f : 0.f; }
};

template <int KP>
__device__ __forceinline__ void rot_h(float hl, float (&h)[KP]) {
#pragma unroll
  for (int k = 0; k < KP; ++k) h[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hl), k));
}

template <int KP, int RT>
__global__ __launch_bounds__(256, 2) void wstats_rot_kernel(const StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  FrameCtx<1, KP, __bf16> fc(a, wlds + (size_t)(threadIdx.x >> 6) * a.Fs * KP);
  int n_beg, n_end;
  wave_frames(a.NT, n_beg, n_end);
  if (n_beg >= n_end) return;
  RotCtx<KP, RT> rc(a, fc);
  // Small operands TWO frames ahead (cur, nx1, nx2), like the slot map: requested a single frame ahead they are younger
  // than the previous step's refills, and the wait for them (vmcnt counts in order) at the top of a step would be a wait
  // for every row of the frame -- the rows refilled last would get no time at all.
  RotSmall<KP> cur, nx1, nx2;
  int sl[2];                   // sl[S]: slots of the frame that goes into set S next
  const int n_last = n_end - 1;
  auto clampf = [&](int n) { return n < n_last ? n : n_last; };
  {
    sl[0] = rc.slots(n_beg);
    rc.req_small(n_beg, cur);
    rc.req_small(clampf(n_beg + 1), nx1);
    sl[1] = rc.slots(clampf(n_beg + 1));
#pragma unroll
    for (int r = 0; r < RT; ++r) rc.template req_row<0>(n_beg, sl[0], r, 0.f, rc.voff);
    rc.template req_x<0>(n_beg, sl[0], true);
  }
  // frame n from set S; set 1-S gets frame n+1 (nothing behind the last frame: refills past the buffer's end)
  auto step = [&](int n, auto set_c) {
    constexpr int S = decltype(set_c)::value, T = 1 - S;
    const bool more = n < n_last;
    const int nn = clampf(n + 1);
    rc.req_small(clampf(n + 2), nx2);
    sl[S] = rc.slots(clampf(n + 2));
    const int utt = cur.utt();
    const float gn = cur.g();
    fc.set_utt(utt);
    float h[KP];
    rot_h<KP>(cur.pk, h);
    f32x4 vb[1], a1, a2;
    float vbx;
    fc.noise_var(utt, h, vb, vbx);
    a1 = a2 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x2 g2 = {gn, gn};
    const unsigned vnext = more ? rc.voff : 0xF0000000u;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const f32x4 v = rc.template row<S>(r);
      const f32x2 q0 = rcp2(g2 * v.lo + vb[0].lo), q1 = rcp2(g2 * v.hi + vb[0].hi);
      a1.lo += q0; a1.hi += q1;
      a2.lo = q0 * q0 + a2.lo; a2.hi = q1 * q1 + a2.hi;
      rc.template req_row<T>(nn, sl[T], r, a2[3], vnext);
      __builtin_amdgcn_sched_barrier(0);             // one row per region
    }
    const float q = fast_rcp(gn * rc.template x<S>() + vbx) * rc.xmask();
    rc.template req_x<T>(nn, sl[T], more);
    const float a1x = wave_sum(q), a2x = wave_sum(q * q);
    a2 *= cur.x2;
    f32x4 o1[1] = {a1}, o2[1] = {a2};
    fc.store_row(a.A1 + (size_t)n * a.Fs, o1, a1x);
    fc.store_row(a.P + (size_t)n * a.Fs, o2, a2x * cur.x2x());
    cur = nx1;
    nx1 = nx2;
  };
  int n = n_beg;
  for (; n < n_last; n += 2) {
    step(n, std::integral_constant<int, 0>{});
    step(n + 1, std::integral_constant<int, 1>{});
  }
  if (n == n_last) step(n, std::integral_constant<int, 0>{});
}


// ----------------------------------------------------------------------------
// W statistics AND the W update's sums over frames in one pass (mcem.py:107-110): wstats_rot's frame loop, but the
// frame's A1 = sum_r 1/Vx and P = X2 sum_r 1/Vx^2 never leave the registers -- each wavefront adds P[f] H[k, n] and
// A1[f] H[k, n] of its frames to 2 x 4 bins x KP accumulators per lane (the extra bin F-1: lane k keeps rank k), the four
// wavefronts of a workgroup (one tile of <= 64 consecutive frames of ONE utterance, 16 per wavefront) add theirs up
// through LDS in fixed order, and the workgroup writes one partial sum [Fs][2 KP] per tile.  w_update_tiles_kernel
// (aux.hip) adds an utterance's tiles in fixed order.  Against wstats_rot + w_partial: no A1 / P round trip through
// HBM (2 x NT x Fs floats written and read back) and one launch less per EM iteration; sums in a fixed order, so
// results are reproducible run to run.
// ----------------------------------------------------------------------------
constexpr int WF_TILE = 64, WF_WFR = 16;            // frames per workgroup tile / per wavefront
// LDS of a workgroup: W of the tile's utterance, rank-major [KP][Fs] (shared by the four wavefronts: they are in one
// utterance by construction), then per wavefront 2 KP accumulator rows [slot = 2 k + stat][lane][4 bins] (a lane reads and
// writes its 16 bytes of a slot: conflict-free ds_read/write_b128) and 2 x 64 floats for the extra bin.  The accumulators
// live in LDS, not in registers: wstats_rot's two register sets of rows leave no room for 64 more (it spilled 175).
template <int KP>
__host__ __device__ constexpr int wf_acc_floats() { return 2 * KP * 256 + 128; }
template <int KP, int RT>
__global__ __launch_bounds__(256, 2) void wstats_fused_kernel(const StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  FrameCtx<1, KP, __bf16> fc(a, wlds);
  float* acc_all = wlds + (size_t)a.Fs * KP;
  float* acc = acc_all + (size_t)wave * wf_acc_floats<KP>();
  RotCtx<KP, RT> rc(a, fc);
  const unsigned l4 = (unsigned)fc.lane * 4u;
  for (int tile = blockIdx.x; tile < a.n_t64; tile += gridDim.x) {
    const int t_n0 = a.t64_n0[tile], t_cnt = a.t64_cnt[tile];
    const int utt = a.frame_utt[t_n0];
    // ---- W of the utterance into LDS (all four wavefronts), this wavefront's accumulators to zero
    __syncthreads();                                    // the previous tile's readers are done
    if (utt != fc.wutt) {
      const f32x4* src = reinterpret_cast<const f32x4*>(a.W + (size_t)utt * a.Fs * KP);
      for (int e = threadIdx.x; e < a.Fs * KP / 4; e += 256) fc.put_t(e, src[e]);
      fc.wutt = utt;                                    // (set_utt below then only refreshes the extra bin's column)
    }
#pragma unroll
    for (int sidx = 0; sidx < 2 * KP; ++sidx) *reinterpret_cast<f32x4*>(acc + sidx * 256 + l4) = f32x4{0.f, 0.f, 0.f, 0.f};
    float numx = 0.f, denx = 0.f;
    __syncthreads();
    const int n_beg = t_n0 + wave * WF_WFR;
    const int n_end = n_beg + WF_WFR < t_n0 + t_cnt ? n_beg + WF_WFR : t_n0 + t_cnt;
    if (n_beg < n_end) {
      RotSmall<KP> cur, nx1, nx2;
      int sl[2];
      const int n_last = n_end - 1;
      auto clampf = [&](int n) { return n < n_last ? n : n_last; };
      {
        sl[0] = rc.slots(n_beg);
        rc.req_small(n_beg, cur);
        rc.req_small(clampf(n_beg + 1), nx1);
        sl[1] = rc.slots(clampf(n_beg + 1));
#pragma unroll
        for (int r = 0; r < RT; ++r) rc.template req_row<0>(n_beg, sl[0], r, 0.f, rc.voff);
        rc.template req_x<0>(n_beg, sl[0], true);
      }
      auto step = [&](int n, auto set_c) {
        constexpr int S = decltype(set_c)::value, T = 1 - S;
        const bool more = n < n_last;
        const int nn = clampf(n + 1);
        rc.req_small(clampf(n + 2), nx2);
        sl[S] = rc.slots(clampf(n + 2));
        const float gn = cur.g();
        fc.set_utt(utt);
        float h[KP];
        rot_h<KP>(cur.pk, h);
        f32x4 vb[1], a1, a2;
        float vbx;
        fc.noise_var(utt, h, vb, vbx);
        a1 = a2 = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x2 g2 = {gn, gn};
        const unsigned vnext = more ? rc.voff : 0xF0000000u;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const f32x4 v = rc.template row<S>(r);
          const f32x2 q0 = rcp2(g2 * v.lo + vb[0].lo), q1 = rcp2(g2 * v.hi + vb[0].hi);
          a1.lo += q0; a1.hi += q1;
          a2.lo = q0 * q0 + a2.lo; a2.hi = q1 * q1 + a2.hi;
          rc.template req_row<T>(nn, sl[T], r, a2[3], vnext);
          __builtin_amdgcn_sched_barrier(0);             // one row per region
        }
        const float q = fast_rcp(gn * rc.template x<S>() + vbx) * rc.xmask();
        rc.template req_x<T>(nn, sl[T], more);
        const float a1x = wave_sum(q), a2x = wave_sum(q * q);
        a2 *= cur.x2;                                                     // P = X2 sum_r 1/Vx^2   (mcem.py:107)
        // sums over frames of P H^T and A1 H^T (mcem.py:108-109), this wavefront's share, four ranks per round trip to LDS
#pragma unroll
        for (int k0 = 0; k0 < KP; k0 += 4) {
          f32x4 nv[4], dv[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            nv[t] = *reinterpret_cast<const f32x4*>(acc + (2 * (k0 + t)) * 256 + l4);
            dv[t] = *reinterpret_cast<const f32x4*>(acc + (2 * (k0 + t) + 1) * 256 + l4);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            *reinterpret_cast<f32x4*>(acc + (2 * (k0 + t)) * 256 + l4) = a2 * h[k0 + t] + nv[t];
            *reinterpret_cast<f32x4*>(acc + (2 * (k0 + t) + 1) * 256 + l4) = a1 * h[k0 + t] + dv[t];
          }
        }
        const float hl = fc.lane < KP ? cur.pk : 0.f;                      // lane k: H[k, n]
        numx = (a2x * cur.x2x()) * hl + numx;
        denx = a1x * hl + denx;
        cur = nx1;
        nx1 = nx2;
      };
      int n = n_beg;
      for (; n < n_last; n += 2) {
        step(n, std::integral_constant<int, 0>{});
        step(n + 1, std::integral_constant<int, 1>{});
      }
      if (n == n_last) step(n, std::integral_constant<int, 0>{});
    }
    acc[2 * KP * 256 + fc.lane] = numx;
    acc[2 * KP * 256 + 64 + fc.lane] = denx;
    __syncthreads();
    // ---- the four wavefronts' sums added in the fixed order 0, 1, 2, 3; the tile's partial [slot = 2 k + stat][Fs]
    // (a wavefront writes whole 1 KB pieces: this lane's four bins of a slot are 16 contiguous bytes)
    float* dst = a.wpart64 + (size_t)tile * 2 * KP * a.Fs;
    for (int sidx = wave; sidx < 2 * KP; sidx += 4) {
      f32x4 v = *reinterpret_cast<const f32x4*>(acc_all + sidx * 256 + l4);
#pragma unroll
      for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(acc_all + (size_t)w * wf_acc_floats<KP>() + sidx * 256 + l4);
      if (fc.cv[0]) *reinterpret_cast<f32x4*>(dst + (size_t)sidx * a.Fs + fc.f0[0]) = v;
    }
    if (wave == 0 && fc.has_x && fc.lane < KP) {          // the extra bin F-1: lane k holds rank k
      float vn = 0.f, vd = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        vn += acc_all[(size_t)w * wf_acc_floats<KP>() + 2 * KP * 256 + fc.lane];
        vd += acc_all[(size_t)w * wf_acc_floats<KP>() + 2 * KP * 256 + 64 + fc.lane];
      }
      dst[(size_t)(2 * fc.lane) * a.Fs + a.F - 1] = vn;
      dst[(size_t)(2 * fc.lane + 1) * a.Fs + a.F - 1] = vd;
    }
  }
}


// ----------------------------------------------------------------------------
// The same sums for SMALL batches (no more 16-frame groups than CUs: one utterance through the drop-in classes).  In
// wstats_fused_kernel a wavefront walks its 16 frames one after the other -- with 8 workgroups on the chip that is 16 frame
// latencies in a row (44 us per EM iteration for one 4 s utterance).  Here a workgroup owns ONE 16-frame group (a wave tile
// of the chain) and its four wavefronts compute the statistics of 4 frames each side by side; only the rank-K accumulation
// -- the part whose ORDER is the result -- stays serial: wavefront 0's frames, barrier, wavefront 1's, ... into one set of
// LDS accumulators, with the expressions of wstats_fused_kernel.  The group's sums go to part16[group]; w_combine_groups_kernel
// (aux.hip) adds a tile's four groups in the order wstats_fused_kernel adds its four wavefronts, then the tiles: W, H, g and
// the cost come out bit-identical to the large-batch path (tested), so an utterance's result still does not depend on the
// batch it sits in.
// ----------------------------------------------------------------------------
// (ST = float, the bf16x3 mode's rows, was measured too: no gain over wstats_stream + w_partial at any batch size -- 131.9 against
// 132.4 ms per step, 46.0 against 45.5 ms for one utterance -- and is not instantiated.)
template <int KP, int RT, typename ST>
__global__ __launch_bounds__(256, 2) void wstats_group_kernel(const StreamArgs a, const int32_t* __restrict__ wt_n0,
                                                              const int32_t* __restrict__ wt_cnt, int n_groups, float* __restrict__ part16) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  FrameCtx<1, KP, ST> fc(a, wlds);
  float* acc = wlds + (size_t)a.Fs * KP;                  // [slot = 2 k + stat][lane][4 bins], then 2 x 64 floats for the extra bin
  using RBt = RowBatch<1, ST, 1, RT, (sizeof(ST) == 4 ? 32 : 0)>;
  const unsigned l4 = (unsigned)fc.lane * 4u;
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int g_n0 = wt_n0[grp], g_cnt = wt_cnt[grp];
    const int utt = a.frame_utt[g_n0];
    __syncthreads();                                      // the previous group's readers are done
    if (utt != fc.wutt) {
      const f32x4* src = reinterpret_cast<const f32x4*>(a.W + (size_t)utt * a.Fs * KP);
      for (int e = threadIdx.x; e < a.Fs * KP / 4; e += 256) fc.put_t(e, src[e]);
      fc.wutt = utt;
    }
    for (int sidx = wave; sidx < 2 * KP; sidx += 4) *reinterpret_cast<f32x4*>(acc + sidx * 256 + l4) = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave == 0) { acc[2 * KP * 256 + fc.lane] = 0.f; acc[2 * KP * 256 + 64 + fc.lane] = 0.f; }
    __syncthreads();
    // ---- phase 1: A1 = sum_r 1/Vx, P = X2 sum_r 1/Vx^2 of this wavefront's (up to) four frames, kept in registers
    const int n_beg = g_n0 + 4 * wave;
    const int n_end = n_beg + 4 < g_n0 + g_cnt ? n_beg + 4 : g_n0 + g_cnt;
    f32x4 A1[4], A2[4];
    float a1xs[4], pxs[4], hls[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      A1[i] = A2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      a1xs[i] = pxs[i] = hls[i] = 0.f;
      const int n = n_beg + i;
      if (n < n_end) {
        const ST* base = reinterpret_cast<const ST*>(a.VsS) + (size_t)n * a.Rs * a.Fs;
        const int sl = RBt::load_slots(fc, a.src + n, 0, RT);
        const float gn = a.g[n];
        f32x4 x2[1];
        float x2x;
        fc.load_x2(n, x2, x2x);
        float h[KP];
#pragma unroll
        for (int k = 0; k < KP; k += 4) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(a.Ht + (size_t)n * KP + k);
#pragma unroll
          for (int t = 0; t < 4; ++t) h[k + t] = vn_uniform(hv[t]);
        }
        hls[i] = fc.lane < KP ? a.Ht[(size_t)n * KP + (fc.lane < KP ? fc.lane : 0)] : 0.f;      // lane k: H[k, n]
        fc.set_utt(utt);
        RBt rb;
        rb.load_rows(fc, base, sl, 0, RT);
        f32x4 vb[1], a1, a2;
        float vbx;
        fc.noise_var(utt, h, vb, vbx);
        a1 = a2 = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x2 g2 = {gn, gn};
        auto row1 = [&](int r) {
          f32x4 v[1];
          rb.get(r, v);
          const f32x2 q0 = rcp2(g2 * v[0].lo + vb[0].lo), q1 = rcp2(g2 * v[0].hi + vb[0].hi);
          a1.lo += q0; a1.hi += q1;
          a2.lo = q0 * q0 + a2.lo; a2.hi = q1 * q1 + a2.hi;
        };
        rb.for_rows(row1);
        const float q = fast_rcp(gn * rb.x() + vbx) * rb.xmask(fc);
        const float a1x = wave_sum(q), a2x = wave_sum(q * q);
        a2 *= x2[0];                                                        // P = X2 sum_r 1/Vx^2   (mcem.py:107)
        A1[i] = a1; A2[i] = a2;
        a1xs[i] = a1x;
        pxs[i] = a2x * x2x;
      }
    }
    // ---- phase 2: the sums over frames of P H^T and A1 H^T (mcem.py:108-109), frame by frame in the group's order
    for (int r = 0; r < 4; ++r) {
      if (wave == r) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (n_beg + i < n_end) {
            float h[KP];
            rot_h<KP>(hls[i], h);
            const f32x4 a1 = A1[i], a2 = A2[i];
#pragma unroll
            for (int k0 = 0; k0 < KP; k0 += 4) {
              f32x4 nv[4], dv[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                nv[t] = *reinterpret_cast<const f32x4*>(acc + (2 * (k0 + t)) * 256 + l4);
                dv[t] = *reinterpret_cast<const f32x4*>(acc + (2 * (k0 + t) + 1) * 256 + l4);
              }
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                // (explicit fused multiply-adds, the form wstats_fused_kernel's expressions contract to: here the products do not
                // change from round to round, the compiler hoists them out of the round loop and would add them unfused)
                const f32x4 hh = {h[k0 + t], h[k0 + t], h[k0 + t], h[k0 + t]};
                *reinterpret_cast<f32x4*>(acc + (2 * (k0 + t)) * 256 + l4) = __builtin_elementwise_fma(a2, hh, nv[t]);
                *reinterpret_cast<f32x4*>(acc + (2 * (k0 + t) + 1) * 256 + l4) = __builtin_elementwise_fma(a1, hh, dv[t]);
              }
            }
            const float hl = hls[i];
            float numx = acc[2 * KP * 256 + fc.lane], denx = acc[2 * KP * 256 + 64 + fc.lane];
            numx = __builtin_fmaf(pxs[i], hl, numx);
            denx = __builtin_fmaf(a1xs[i], hl, denx);
            acc[2 * KP * 256 + fc.lane] = numx;
            acc[2 * KP * 256 + 64 + fc.lane] = denx;
          }
      }
      __syncthreads();
    }
    // ---- the group's sums, slot-major [2 k + stat][Fs] like a tile's partial
    float* dst = part16 + (size_t)grp * 2 * KP * a.Fs;
    for (int sidx = wave; sidx < 2 * KP; sidx += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(acc + sidx * 256 + l4);
      if (fc.cv[0]) *reinterpret_cast<f32x4*>(dst + (size_t)sidx * a.Fs + fc.f0[0]) = v;
    }
    if (wave == 0 && fc.has_x && fc.lane < KP) {
      dst[(size_t)(2 * fc.lane) * a.Fs + a.F - 1] = acc[2 * KP * 256 + fc.lane];
      dst[(size_t)(2 * fc.lane + 1) * a.Fs + a.F - 1] = acc[2 * KP * 256 + 64 + fc.lane];
    }
  }
}


StreamArgs base_args(const vaenmf_plan* p) {
  StreamArgs a = {};
  a.VsS = p->VsS; a.src = p->src; a.frame_utt = p->d_frame_utt; a.Vb = p->Vb_ext;
  a.store_f32 = p->cfg.precision == VAENMF_PREC_BF16X3;
  a.n_sms = p->n_sms;
  a.NT = p->NT; a.R = p->store_R; a.Rs = p->store_Rs; a.F = p->cfg.F; a.Fm = p->Fm; a.Fs = p->Fs; a.K = p->cfg.K;
  return a;
}

enum { SK_WSTATS, SK_HG, SK_WF };

template <int KIND, int NCH, int KP, typename ST>
int launch_st(StreamArgs a, int grid, hipStream_t st) {
  // rank <= 8: one W[utt] per wavefront; above: one per workgroup when it leaves room for two workgroups per CU
  const size_t one = (size_t)a.Fs * KP * sizeof(float);
  const size_t lds = KP <= 8 ? 4 * one : (one <= 72 * 1024 ? one : 0);
  a.w_blk_lds = KP > 8 && lds > 0;
  const void* fn = KIND == SK_WSTATS ? (const void*)wstats_stream_kernel<NCH, KP, ST>
                   : (KIND == SK_HG ? (const void*)hg_stream_kernel<NCH, KP, ST> : (const void*)wf_stream_kernel<NCH, KP, ST>);
  if (int e = vn_ensure_dyn_lds(fn, 80 * 1024)) return e;
  constexpr bool PIPE_OK = NCH == 1 && KP <= 8;
  // (The same form of the H/g kernel needs two register sets of 60 plus the three passes' working set: it spills at 256
  // registers with 30 samples -- 0.62 ms against 0.17 for the batch form -- and gains nothing with 10; not kept.)
  if (VN_ROT && PIPE_OK && sizeof(ST) == 2 && KIND == SK_WSTATS && (a.R == 30 || a.R == 10)) {
    constexpr int K1 = PIPE_OK ? KP : 8;
    int g4 = a.n_sms * 2;                               // one resident set: 2 workgroups of 4 wavefronts per CU (two per SIMD)
    if (g4 * 4 > a.NT) g4 = (a.NT + 3) / 4;
    if (a.R == 30) {
      if (int e = vn_ensure_dyn_lds((const void*)wstats_rot_kernel<K1, 30>, 80 * 1024)) return e;
      hipLaunchKernelGGL((wstats_rot_kernel<K1, 30>), dim3(g4), dim3(256), lds, st, a);
    } else {
      if (int e = vn_ensure_dyn_lds((const void*)wstats_rot_kernel<K1, 10>, 80 * 1024)) return e;
      hipLaunchKernelGGL((wstats_rot_kernel<K1, 10>), dim3(g4), dim3(256), lds, st, a);
    }
    return 0;
  }
  if (PIPE_OK && VN_STREAM2 && KIND == SK_WSTATS && a.R <= RowBatch<NCH, ST>::RB) {
    // frame-pipelined form: two resident wavefronts per SIMD, one resident set per launch.  (The same form of the H/g
    // kernel was measured twice, 0.23 ms against 0.17-0.20 for the batch-at-a-time one at four wavefronts per SIMD,
    // and is not kept; exact-row-count instantiations of this one spill: 0.29 vs 0.155 ms.)
    int g2 = a.n_sms * 2;
    if (g2 * 4 > a.NT) g2 = (a.NT + 3) / 4;
    constexpr int N1 = PIPE_OK ? NCH : 1, K1 = PIPE_OK ? KP : 8;
#define VN_GO2(RT)                                                                                         \
    do {                                                                                                       \
      if (int e = vn_ensure_dyn_lds((const void*)wstats_stream2_kernel<N1, K1, ST, RT>, 80 * 1024)) return e; \
      hipLaunchKernelGGL((wstats_stream2_kernel<N1, K1, ST, RT>), dim3(g2), dim3(256), lds, st, a);           \
    } while (0)
    constexpr int RBm = RowBatch<N1, ST>::RB;
    if (VN_WS_EXACT && a.R == 30 && RBm >= 30) VN_GO2((RBm >= 30 ? 30 : 0));
    else if (VN_WS_EXACT && a.R == 10) VN_GO2(10);
    else VN_GO2(0);
#undef VN_GO2
    return 0;
  }
  if (KIND == SK_WSTATS) hipLaunchKernelGGL((wstats_stream_kernel<NCH, KP, ST>), dim3(grid), dim3(256), lds, st, a);
  else if (KIND == SK_HG) {
    constexpr int RBm = (VN_HG_FULLF && sizeof(ST) == 4 && NCH == 1 && KP <= 8) ? 32 : RowBatch<NCH, ST>::RB;     // (as hg_stream_kernel's RBt)
    if (VN_HG_EXACT && NCH == 1 && KP <= 8 && a.R == 30 && RBm >= 30) {
      if (int e = vn_ensure_dyn_lds((const void*)hg_stream_kernel<NCH, KP, ST, (RBm >= 30 && NCH == 1 ? 30 : 0)>, 80 * 1024)) return e;
      hipLaunchKernelGGL((hg_stream_kernel<NCH, KP, ST, (RBm >= 30 && NCH == 1 ? 30 : 0)>), dim3(grid), dim3(256), lds, st, a);
    } else if (VN_HG_EXACT && NCH == 1 && KP <= 8 && a.R == 10) {
      if (int e = vn_ensure_dyn_lds((const void*)hg_stream_kernel<NCH, KP, ST, (RBm >= 10 && NCH == 1 ? 10 : 0)>, 80 * 1024)) return e;
      hipLaunchKernelGGL((hg_stream_kernel<NCH, KP, ST, (RBm >= 10 && NCH == 1 ? 10 : 0)>), dim3(grid), dim3(256), lds, st, a);
#if VN_HG_EXACT2
    } else if (NCH == 2 && sizeof(ST) == 2 && KP == 16 && a.R == 30) {
      // two 256-bin chunks (F = 513), rank <= 16, bf16 rows: the whole 30-sample frame sits in one batch of registers
      // (RBX = 32): the exact-count form drops the per-row branches here as well
      constexpr int RT2 = (NCH == 2 && sizeof(ST) == 2 && KP == 16) ? 30 : 0;
      if (int e = vn_ensure_dyn_lds((const void*)hg_stream_kernel<NCH, KP, ST, RT2>, 80 * 1024)) return e;
      hipLaunchKernelGGL((hg_stream_kernel<NCH, KP, ST, RT2>), dim3(grid), dim3(256), lds, st, a);
#endif
    } else hipLaunchKernelGGL((hg_stream_kernel<NCH, KP, ST>), dim3(grid), dim3(256), lds, st, a);
  }
  else hipLaunchKernelGGL((wf_stream_kernel<NCH, KP, ST>), dim3(grid), dim3(256), lds, st, a);
  return 0;
}
template <int KIND, int NCH, int KP>
int launch_one(const StreamArgs& a, int grid, hipStream_t st) {
  if (a.store_f32) return launch_st<KIND, NCH, KP, float>(a, grid, st);
  return launch_st<KIND, NCH, KP, __bf16>(a, grid, st);
}
template <int KIND, int NCH>
int launch_kp(const StreamArgs& a, int Kp, int grid, hipStream_t st) {
  switch (Kp) {
    case 8: return launch_one<KIND, NCH, 8>(a, grid, st);
    case 16: return launch_one<KIND, NCH, 16>(a, grid, st);
    default: return launch_one<KIND, NCH, 32>(a, grid, st);
  }
}
template <int KIND>
int launch_stream(const vaenmf_plan* p, const StreamArgs& a, hipStream_t st) {
  const int nch = (p->Fm + 255) / 256;
  // enough wavefronts in flight to cover the HBM latency: 4 per SIMD, blocks of 4 wavefronts
  // exactly one resident set of wavefronts (4 per SIMD at rank 8): the W-statistics kernel pipelines its loads
  // across the frames of a wavefront and wants long runs (blocks per CU 4: 0.169 ms, 6: 0.220, 8: 0.179, 12: 0.192);
  // H/g is indifferent (4: 0.287, 8: 0.291, 16: 0.305)
  int grid = p->n_sms * 4;
  if (grid * 4 > p->NT) grid = (p->NT + 3) / 4;
  int rc;
  if (nch <= 1) rc = launch_kp<KIND, 1>(a, p->Kp, grid, st);
  else if (nch == 2) rc = launch_kp<KIND, 2>(a, p->Kp, grid, st);
  else rc = launch_kp<KIND, 3>(a, p->Kp, grid, st);
  if (rc) return rc;
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}

int check_store(const vaenmf_plan* p) {
  VN_REQUIRE(p != nullptr && p->have_weights && p->NT > 0, "plan not ready");
  VN_REQUIRE(p->store_R > 0, "the sample store is empty: vaenmf_sample_store(plan, 1), then vaenmf_mh_chain");
  return 0;
}

}  // namespace

// aux.hip
int vn_launch_w_update(const vaenmf_plan* p, float* W, const float* Ht, hipStream_t st);
int vn_launch_w_update_tiles(const vaenmf_plan* p, float* W, hipStream_t st, bool groups);

namespace {
// W statistics + the W update's sums in one kernel: the bench shapes of wstats_rot (bf16 rows, one 256-bin chunk, rank <= 8,
// 30 or 10 samples per frame, NMF noise model).  VAENMF_WFUSED=0 keeps the two-kernel path (A/B runs, tests).
bool w_fused_ok(const vaenmf_plan* p, const StreamArgs& a) {
  if (!(VN_ROT && !a.store_f32 && p->Kp == 8 && (p->Fm + 255) / 256 == 1 && (a.R == 30 || a.R == 10) && !a.gains_only)) return false;
  const char* e = getenv("VAENMF_WFUSED");              // (read per call: tests switch it inside one process)
  return !(e && e[0] == '0');
}
// small batches: one workgroup per 16-frame group (VAENMF_WGROUP=0 keeps the tile kernel: tests)
bool w_group_ok(const vaenmf_plan* p) {
  if (!(p->wpart16 && p->n_wtiles <= p->n_sms)) return false;
  const char* e = getenv("VAENMF_WGROUP");
  return !(e && e[0] == '0');
}
int launch_w_group(const vaenmf_plan* p, StreamArgs a, hipStream_t st) {
  const size_t lds = ((size_t)a.Fs * 8 + (size_t)wf_acc_floats<8>()) * sizeof(float);
  const int grid = p->n_wtiles;
  if (a.R == 30) hipLaunchKernelGGL((wstats_group_kernel<8, 30, __bf16>), dim3(grid), dim3(256), lds, st, a, p->d_wt_n0, p->d_wt_cnt, p->n_wtiles, p->wpart16);
  else hipLaunchKernelGGL((wstats_group_kernel<8, 10, __bf16>), dim3(grid), dim3(256), lds, st, a, p->d_wt_n0, p->d_wt_cnt, p->n_wtiles, p->wpart16);
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_w_fused(const vaenmf_plan* p, StreamArgs a, hipStream_t st) {
  a.t64_n0 = p->d_t64_n0; a.t64_cnt = p->d_t64_cnt; a.n_t64 = p->n_t64; a.wpart64 = p->wpart64;
  const size_t lds = ((size_t)a.Fs * 8 + (size_t)4 * wf_acc_floats<8>()) * sizeof(float);
  int grid = a.n_sms * 2;                               // one resident set: 2 workgroups of 4 wavefronts per CU
  if (const char* e = getenv("VAENMF_WFUSED_GRID")) {   // (tests: a small grid makes every workgroup walk several tiles / utterances)
    const int g = atoi(e);
    if (g > 0 && g < grid) grid = g;
  }
  if (grid > p->n_t64) grid = p->n_t64;
  if (a.R == 30) {
    if (int e = vn_ensure_dyn_lds((const void*)wstats_fused_kernel<8, 30>, 80 * 1024)) return e;
    hipLaunchKernelGGL((wstats_fused_kernel<8, 30>), dim3(grid), dim3(256), lds, st, a);
  } else {
    if (int e = vn_ensure_dyn_lds((const void*)wstats_fused_kernel<8, 10>, 80 * 1024)) return e;
    hipLaunchKernelGGL((wstats_fused_kernel<8, 10>), dim3(grid), dim3(256), lds, st, a);
  }
  VN_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int vaenmf_m_step_stored(vaenmf_plan* p, const float* X2, float* W, float* Ht, float* g, double* cost_frames,
                                    void* stream) {
  if (int e = check_store(p)) return e;
  hipStream_t st = (hipStream_t)stream;
  StreamArgs a = base_args(p);
  a.X2 = X2; a.W = W; a.Ht = Ht; a.g = g; a.A1 = p->A1; a.P = p->P; a.normW = p->normW;
  a.cost_frames = cost_frames ? cost_frames : p->cost_frames;
  if (p->Vb_ext) {                                      // noNMF: only the gains move (mcem.py:543-578)
    a.gains_only = 1;
    ProfScope ps(p, VN_K_HG, st);
    return launch_stream<SK_HG>(p, a, st);
  }
  p->last_w_fused = w_fused_ok(p, a) ? 1 : 0;
  if (p->last_w_fused && w_group_ok(p)) {
    p->last_w_fused = 2;
    { ProfScope ps(p, VN_K_WSTATS, st); if (int e = launch_w_group(p, a, st)) return e; }
    { ProfScope ps(p, VN_K_WUPDATE, st); if (int e = vn_launch_w_update_tiles(p, W, st, true)) return e; }
  } else if (p->last_w_fused) {
    { ProfScope ps(p, VN_K_WSTATS, st); if (int e = launch_w_fused(p, a, st)) return e; }
    { ProfScope ps(p, VN_K_WUPDATE, st); if (int e = vn_launch_w_update_tiles(p, W, st, false)) return e; }
  } else {
    { ProfScope ps(p, VN_K_WSTATS, st); if (int e = launch_stream<SK_WSTATS>(p, a, st)) return e; }
    { ProfScope ps(p, VN_K_WUPDATE, st); if (int e = vn_launch_w_update(p, W, Ht, st)) return e; }
  }
  { ProfScope ps(p, VN_K_HG, st); if (int e = launch_stream<SK_HG>(p, a, st)) return e; }
  return 0;
}

extern "C" int vaenmf_wiener_stored(vaenmf_plan* p, const float* W, const float* Ht, const float* g, const float* X,
                                    float* S_hat, float* N_hat, float* WFs, float* WFn, void* stream) {
  if (int e = check_store(p)) return e;
  VN_REQUIRE(X && S_hat && N_hat, "vaenmf_wiener_stored: null spectrogram / outputs");
  StreamArgs a = base_args(p);
  a.W = W; a.Ht = const_cast<float*>(Ht); a.g = const_cast<float*>(g); a.X = X;
  a.S_hat = S_hat; a.N_hat = N_hat; a.WFs = WFs; a.WFn = WFn;
  ProfScope ps(p, VN_K_WF, (hipStream_t)stream);
  return launch_stream<SK_WF>(p, a, (hipStream_t)stream);
}
