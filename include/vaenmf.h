/*
 * vaenmf.h -- C ABI of libvaenmf.so: the MI355X (gfx950) engine for the VAE-NMF
 * "reconstruct" hot path of sp-uhh/guided-vae-nmf.
 *
 * The reference has no FFI boundary (it is pure Python on torch); the boundary it
 * does have is the Python object surface that scripts/evaluate_M1.py:111-166 and
 * scripts/evaluate_M2_vad.py:95-166 touch (python/models/mcem.py MCEM_M1/MCEM_M2,
 * python/models/models.py, python/processing/stft.py).  Each entry point below
 * names the reference lines it replaces; guided-vae-nmf_amd/vaenmf binds them with
 * ctypes and re-creates that Python surface on top.
 *
 * Conventions
 *   - every function returns 0 on success or a negative code; vaenmf_last_error()
 *     returns a thread-local message.  Nothing throws.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  All work is
 *     enqueued asynchronously; no call synchronises the device unless stated.
 *   - pointers marked DEV are device pointers owned by the caller (e.g. the
 *     PyTorch-ROCm allocator), 16-byte aligned, never freed by the callee.
 *     Pointers marked HOST are host memory read during the call.
 *   - no allocation happens after vaenmf_plan_create / vaenmf_set_decoder_weights /
 *     vaenmf_bind_batch.
 *
 * Data layout in HBM (all "frame-major": one row per STFT frame, frames of all
 * utterances of the bound batch concatenated, NT = total frames)
 *   Fs  = F rounded up to 16        (query VAENMF_Q_FS)    feature row stride
 *   Kp  = 8, 16 or 32 (>= K)        (query VAENMF_Q_KP)    padded NMF rank
 *   X2  float  [NT][Fs]   mixture power spectrogram |X|^2      (mcem.py:47, transposed)
 *   X   float2 [NT][Fs]   mixture STFT, complex64              (mcem.py:46, transposed)
 *   W   float  [U][Fs][Kp] NMF dictionary per utterance        (mcem.py:48)  pad = 0
 *   Ht  float  [NT][Kp]   NMF activations, transposed          (mcem.py:49)  pad = 0
 *   g   float  [NT]       per-frame gain                       (mcem.py:51)
 *   Z   float  [NT][L]    last draw of the latent variables    (mcem.py:368, transposed)
 *   Zs  float  [NT][Rcap][L] posterior samples                 (mcem.py:386)
 *   B1  float  [NT][H1]   per-frame first-layer bias b1 + W1[:,L:] y_n  (M2: the label
 *                         part of decoder(cat([Z,y])) folded once, mcem.py:242); NULL = M1
 */
#ifndef VAENMF_H
#define VAENMF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vaenmf_plan vaenmf_plan;

enum { VAENMF_PREC_BF16X3 = 0,   /* bf16 MFMA, 3-term hi/lo split, fp32 accumulate (~fp32 accuracy) */
       VAENMF_PREC_BF16   = 1 }; /* plain bf16 MFMA, fp32 accumulate                                   */

enum { VAENMF_RNG_REPLAY = 0,    /* caller supplies the normal / uniform draws (parity runs)          */
       VAENMF_RNG_DEVICE = 1 };  /* counter-seeded xoshiro128+ / Box-Muller streams on the device     */

enum { VAENMF_Q_FS = 0, VAENMF_Q_KP = 1, VAENMF_Q_TILES = 2, VAENMF_Q_NT = 3, VAENMF_Q_NUTT = 4,
       VAENMF_Q_MSTEP_PATH = 5,  /* M-step path of the last vaenmf_em_run: 1 = streaming the sample store, 2 = decoding */
       VAENMF_Q_WTILES = 6,      /* 16-frame wave tiles of the bound batch */
       VAENMF_Q_EM_GRAPH = 7,    /* 1 when the last vaenmf_em_run was launched as a captured HIP graph, 0 when launch by launch */
       VAENMF_Q_W_FUSED = 9,     /* 1 when the last stored M-step ran the W statistics fused with the W update's sums over frames
                                    (2: per 16-frame group, the small-batch form; the same bits) */
       VAENMF_Q_CHAIN_KERNEL = 10, /* kernel of the last MH chain: 0 = team kernel (64-bit addresses), 1 = one wavefront per 16 frames,
                                      2 = four wavefronts per 16 frames (bench shape, batches of at most one wave tile per CU) */
       VAENMF_Q_DEV_ALLOCS = 8 };/* device allocations the library has made in this process so far (any plan): a caller that
                                    reuses a plan can check that a call allocated nothing */

enum { VAENMF_ACT_NONE = 0, VAENMF_ACT_TANH = 1, VAENMF_ACT_RELU = 2, VAENMF_ACT_SIGMOID = 3,
       VAENMF_ACT_STEP = 4 };   /* 1 if x > 0 else 0: sigmoid(x) > 0.5, scripts/evaluate_M2_vad.py:131 */

typedef struct {
  int32_t F;          /* frequency bins, n_fft/2+1 (<= 640)                         */
  int32_t K;          /* NMF rank (<= 32)                                           */
  int32_t L;          /* latent dimension (this build: 32)                          */
  int32_t H1, H2;     /* decoder hidden sizes, first and second layer (this build: 128,128) */
  int32_t max_frames; /* capacity: total frames of a bound batch                    */
  int32_t max_utts;   /* capacity: utterances of a bound batch                      */
  int32_t precision;  /* VAENMF_PREC_*                                              */
} vaenmf_config;

typedef struct {
  int32_t  mode;        /* VAENMF_RNG_*                                             */
  uint32_t call;        /* chain-invocation counter (EM iteration index; WF = niter) */
  const float* eps;     /* DEV [S][NT][L] N(0,1) draws, step-major   (REPLAY only; mcem.py:407) */
  const float* u;       /* DEV [S][NT]    U(0,1) draws               (REPLAY only; mcem.py:420) */
} vaenmf_rng;

const char* vaenmf_last_error(void);

/* Plan = shapes + device-resident decoder weights + batch tiling + workspace. */
int  vaenmf_plan_create(const vaenmf_config* cfg, vaenmf_plan** out);
void vaenmf_plan_destroy(vaenmf_plan* p);
int  vaenmf_plan_query(const vaenmf_plan* p, int32_t what);

/* Decoder weights, HOST, nn.Linear layout [out][in] row-major as in the state_dict
 * keys decoder.hidden.{0,1}.{weight,bias}, decoder.reconstruction.{weight,bias}
 * (models.py:107-121).  in1 = L + Dy; the first L columns of W1 feed the MFMA path,
 * the remaining Dy columns are kept for vaenmf_layer1_bias. */
int vaenmf_set_decoder_weights(vaenmf_plan* p, const float* W1, int32_t in1, const float* b1,
                               const float* W2, const float* b2, const float* W3, const float* b3);

/* Bind a batch: frame_offsets HOST [n_utt+1] (frames of utterance u are rows
 * frame_offsets[u] .. frame_offsets[u+1]-1); utt_seeds HOST [n_utt] or NULL (device RNG
 * streams are keyed by (utt seed, frame index within the utterance), so results do not
 * depend on how utterances are batched). */
int vaenmf_bind_batch(vaenmf_plan* p, int32_t n_utt, const int32_t* frame_offsets,
                      const uint64_t* utt_seeds);
/* The same without waiting for the GPU when the batch's frame structure repeats the bound one:
 * the frame tables stay, the seeds go up stream-ordered from pinned memory.  A caller that never
 * synchronises (results collected later) can prepare batch k+1 while batch k runs.  A batch with
 * another frame structure synchronises `stream` and uploads the tables as vaenmf_bind_batch does.
 * STREAM CONTRACT: the seed upload is ordered on `stream` only -- the calls that consume the batch (vaenmf_mh_chain,
 * vaenmf_em_run, ...) must be enqueued on the same stream, or after an event / synchronisation that orders them behind it. */
int vaenmf_bind_batch_async(vaenmf_plan* p, int32_t n_utt, const int32_t* frame_offsets,
                            const uint64_t* utt_seeds, void* stream);

/* EM.init_parameters (mcem.py:42-44, :51) for the bound batch on the device: W = max(U(0,1), eps) (F x K per utterance),
 * H = max(U(0,1), eps) (K x N), g = 1; padding zero.  Counter-based draws keyed by (utterance seed of vaenmf_bind_batch,
 * salt, element): an utterance's initialisation does not depend on the batch it sits in.  W DEV [U][Fs][Kp], Ht DEV
 * [NT][Kp], g DEV [NT].  (The drop-in classes draw W0 / H0 on the host in the reference's order instead.) */
int vaenmf_init_nmf(vaenmf_plan* p, float* W, float* Ht, float* g, uint64_t salt, float eps, void* stream);

/* B1[n][h] = b1[h] + sum_d W1[h][L+d] * y[n][d]   (label half of mcem.py:242/261/283).
 * y DEV [NT][Dy]. */
int vaenmf_layer1_bias(vaenmf_plan* p, const float* y, int32_t Dy, float* B1, void* stream);

/* Fixed noise variance for the *_noNMF variants (EM_noNMF / MCEM_M2_noNMF, mcem.py:493-760): Vb DEV [NT][Fs]
 * (caller-owned, must stay valid) replaces W H in every later call and vaenmf_m_step then updates the gains only
 * (mcem.py:543-578); NULL restores the NMF noise model. */
int vaenmf_set_noise_psd(vaenmf_plan* p, const float* Vb);

/* Metropolis-Hastings chain of one E-step / Wiener phase -- replaces
 * MCEM_M1.sample_posterior (mcem.py:371-441) and MCEM_M2.sample_posterior (:218-294):
 * nsamples+burnin random-walk steps per frame, samples after burn-in to Zs[:, 0..nsamples-1, :].
 * update_Z != 0: Z is overwritten with the last draw, as E_step does (mcem.py:466);
 * update_Z == 0: Z is only read, as compute_WF does (mcem.py:477-478).  acc_out (DEV
 * [S][NT], may be NULL) receives the log-acceptance of every step (mcem.py:415-417).
 * Zs may be NULL where the wave-private chain kernels run (vaenmf_wchain_addressable; every shape but F > 528 and bf16x3 with
 * F > 272): the samples are then not recorded -- with the sample-variance store on nothing reads them (vaenmf_em_run does
 * this for its E-steps). */
int vaenmf_mh_chain(vaenmf_plan* p, const float* X2, const float* W, const float* Ht, const float* g,
                    float* Z, int32_t update_Z, const float* B1, float* Zs, int32_t Rcap,
                    int32_t nsamples, int32_t burnin, float var_rw,
                    const vaenmf_rng* rng, float* acc_out, void* stream);

/* 1 when the wave-private chain kernel (32-bit buffer offsets per lane) can address every buffer of a batch of NT frames
 * (Zs [NT][Rcap][L], the replay draws [steps][NT][L], X2 [NT][Fs], W [n_utt][Fs][Kp], ...), 0 when vaenmf_mh_chain runs
 * the 64-bit team kernel for it instead.  Pure host arithmetic (no GPU needed). */
int vaenmf_wchain_addressable(int64_t NT, int32_t Rcap, int32_t steps, int32_t Fs, int32_t Kp, int32_t n_utt, int32_t replay);

/* The draws the DEVICE generator hands to step s of chain invocation rng->call:
 * eps_out DEV [S][NT][L], u_out DEV [S][NT].  Test/debug aid: a REPLAY run fed with
 * these buffers is bit-identical to the DEVICE run. */
int vaenmf_rng_fill(vaenmf_plan* p, uint32_t call, int32_t S, float* eps_out, float* u_out, void* stream);

/* Vs = decoder(Z_samples): replaces compute_Vs (mcem.py:444-454 / :297-307).
 * Vs_out DEV [NT][R][Fs] (the reference's (R,F,N) tensor, frame-major). */
int vaenmf_decode(vaenmf_plan* p, const float* Zs, int32_t Rcap, int32_t R, const float* B1,
                  float* Vs_out, void* stream);

/* One M-step -- replaces EM.M_step (mcem.py:90-152) plus
 * compute_expected_neg_log_like (:68-70): multiplicative updates of W, H (exponent 1/2),
 * L1 column normalisation, gain update.  The R posterior samples are re-decoded on
 * chip instead of being streamed from HBM.  W, Ht, g are updated in place.
 * cost_frames DEV [NT] double: sum_{r,f}(log Vx + X2/Vx) per frame (after the update);
 * the mean over (R,F,N) of an utterance is mcem.py:70. */
int vaenmf_m_step(vaenmf_plan* p, const float* X2, float* W, float* Ht, float* g,
                  const float* Zs, int32_t Rcap, int32_t R, const float* B1,
                  double* cost_frames, void* stream);

/* Wiener filter from the R samples in Zs -- replaces compute_WF's averaging
 * (mcem.py:486-488) and the complex products of EM.run (:175-176):
 * S_hat = mean_r(g Vs/Vx) * X,  N_hat = mean_r(Vb/Vx) * X.   X, S_hat, N_hat DEV
 * complex64 [NT][Fs] (interleaved re,im); WFs/WFn DEV [NT][Fs] optional (NULL). */
int vaenmf_wiener(vaenmf_plan* p, const float* X2, const float* W, const float* Ht, const float* g,
                  const float* Zs, int32_t Rcap, int32_t R, const float* B1,
                  const float* X, float* S_hat, float* N_hat, float* WFs, float* WFn, void* stream);

/* Fused driver -- replaces EM.run (mcem.py:155-178) for the whole bound batch with no
 * host synchronisation per iteration: niter x (E-step, M-step, cost), then the Wiener
 * chain and filter.  cost DEV [n_utt][niter] double.  (nsE, biE) / (nsWF, biWF) are the
 * EFFECTIVE sample/burn-in counts (the caller applies MCEM_M1's positional-shift quirk,
 * mcem.py:461-462).  Device RNG only (rng_mode REPLAY is served step by step by the
 * calls above).
 * A call whose signature (buffers, shapes, counts) repeats the previous call's is captured into a
 * HIP graph once and replayed from then on -- one launch per call instead of ~600; the batch's
 * contents (spectrogram, seeds, frame tables) sit behind the same pointers and are read at run
 * time.  Profiling (vaenmf_profile_*) and VAENMF_GRAPH=0 keep the launch-by-launch path;
 * results are the same either way. */
int vaenmf_em_run(vaenmf_plan* p, const float* X2, float* W, float* Ht, float* g, float* Z,
                  const float* B1, float* Zs, int32_t Rcap, int32_t niter,
                  int32_t nsE, int32_t biE, int32_t nsWF, int32_t biWF, float var_rw,
                  const float* X, float* S_hat, float* N_hat, double* cost, void* stream);

/* Dense layer Y = act(X Wt^T + b): encoder / classifier forwards
 * (models.py:101-104, 33-38, 57-62).  X DEV [M][in], Wt DEV [out][in], b DEV [out],
 * Y DEV [M][ldy] (first `out` columns written). */
int vaenmf_dense(const float* X, int32_t M, int32_t in, int32_t ldx, const float* Wt, const float* b,
                 int32_t out, int32_t act, float* Y, int32_t ldy, void* stream);

/* Sample-variance store.  With the store enabled, vaenmf_mh_chain also writes the decoded
 * variances Vs = exp(decoder(Z')) of every post-burn-in proposal (they are in registers
 * anyway) to a plan-owned buffer VsS DEV [NT][Rs][Fs], Rs = nsamples + 1 -- float rows in
 * bf16x3 mode, bf16 rows in bf16 mode (half the traffic; rounding of the size the bf16
 * decoder products carry anyway) -- and a map
 * src DEV int32 [Rs][NT]: the variances of sample r of frame n (the state after post-burn-in
 * step r, mcem.py:429-437) are the row VsS[n][src[r][n]].  The M-step and the Wiener
 * filter can then stream the samples' variances from HBM instead of decoding Zs again
 * (vaenmf_m_step_stored, vaenmf_wiener_stored; vaenmf_em_run does so by itself).  The store
 * describes the most recent vaenmf_mh_chain call only.  vaenmf_sample_store_gather copies
 * the samples' rows out densely, Vs_out DEV float [NT][nsamples][Fs] (what vaenmf_decode
 * computes from Zs; bins >= F unspecified).
 * vaenmf_sample_store(plan, max_samples): max_samples > 0 switches the store on and sizes it -- an ALLOCATING call,
 * like vaenmf_plan_create -- for the bound batch (call it after vaenmf_bind_batch; before any batch is bound: for
 * the plan's frame capacity) and chains of up to max_samples samples per frame; 0 switches it off (memory kept).
 * vaenmf_mh_chain never allocates: it fails with a message if the store is too small for its batch. */
int vaenmf_sample_store(vaenmf_plan* plan, int32_t max_samples);
int vaenmf_sample_store_gather(vaenmf_plan* plan, float* Vs_out, void* stream);
/* vaenmf_m_step / vaenmf_wiener over the store of the most recent chain (same updates, same
 * outputs; the samples are the store's, so no Zs / B1 arguments). */
int vaenmf_m_step_stored(vaenmf_plan* plan, const float* X2, float* W, float* Ht, float* g,
                         double* cost_frames, void* stream);
int vaenmf_wiener_stored(vaenmf_plan* plan, const float* W, const float* Ht, const float* g,
                         const float* X, float* S_hat, float* N_hat, float* WFs, float* WFn,
                         void* stream);

/* |X|^2 (mcem.py:47): X DEV complex64 [n], X2 DEV float [n]. */
int vaenmf_power_spec(const float* X, float* X2, int64_t n, void* stream);

/* STFT front end -- replaces python/processing/stft.py:16-63 (librosa.core.stft with
 * center=True, reflect padding, periodic Hann).  vaenmf_stft_num_frames applies the
 * integer-window check (:37-38) and the end-pad rule (:48-53) on the host and returns
 * n_fft, hop, the frame count and the padded length of one utterance.
 * vaenmf_stft_batch: wav DEV float [sum T]; sample_offsets DEV int64 [n_utt+1];
 * frame_offsets DEV int32 [n_utt+1]; frame_utt DEV int32 [NT]; padded_len DEV int32
 * [n_utt]; X DEV complex64 [NT][Fs] (bins >= F zeroed).  nfft: power of two <= 2048.
 * The transform runs in float64 and is rounded once to complex64 like the reference. */
int vaenmf_stft_num_frames(int64_t n_samples, double fs, double wlen_sec, double hop_percent,
                           int32_t* nfft, int32_t* hop, int32_t* n_frames, int32_t* n_padded);
int vaenmf_stft_batch(const float* wav, int32_t n_frames_total, const int64_t* sample_offsets,
                      const int32_t* frame_offsets, const int32_t* frame_utt,
                      const int32_t* padded_len, int32_t nfft, int32_t hop, int32_t Fs,
                      float* X, void* stream);
/* iSTFT back end -- replaces stft.py:66-102 (librosa.core.istft with length=max_len,
 * window-sum-square normalised overlap-add): S DEV complex64 [NT][Fs] -> wav_out DEV
 * float [sum T] (T = sample_offsets[u+1]-sample_offsets[u] = max_len of utterance u);
 * work DEV float [NT][nfft] scratch. */
int vaenmf_istft_batch(const float* S, int32_t n_utt, int32_t n_frames_total,
                       const int64_t* sample_offsets, const int32_t* frame_offsets,
                       int32_t nfft, int32_t hop, int32_t Fs, float* work, float* wav_out,
                       void* stream);

/* Label / guide front-ends of the M2 path -- replace python/processing/target.py.
 * vaenmf_lorenz_labels: clean_speech_IBM (target.py:7-28, mode VAENMF_LABEL_IBM) and
 * clean_speech_VAD (:30-50, mode VAENMF_LABEL_VAD) for a batch of utterances: power =
 * |X|^2 (VAD: summed over the bins of a frame), descending sort per utterance, Lorenz
 * curve cumsum/sum, threshold = last sorted power whose Lorenz value is < quantile_fraction,
 * label = power > threshold ? hi : lo, where lo/hi are the caller's softened and rounded
 * values round(0.5 -/+ 0.5 quantile_weight) (:24-27).  The float32 arithmetic follows
 * numpy's operation order, so the 0/1 decisions are those of the reference bit for bit.
 * X DEV complex64 [NT][Fs]; frame_offsets HOST int32 [n_utt+1]; labels DEV float
 * [NT][ld] (IBM, bins < F written) or [NT] (VAD); thr_out DEV float [n_utt] or null;
 * work DEV scratch of vaenmf_lorenz_work_bytes() bytes.  Synchronises the stream; an
 * utterance with no Lorenz value below the fraction is the reference's IndexError. */
enum { VAENMF_LABEL_IBM = 0, VAENMF_LABEL_VAD = 1 };
int64_t vaenmf_lorenz_work_bytes(int32_t n_frames_total, int32_t F, int32_t n_utt, int32_t mode);
int vaenmf_lorenz_labels(const float* X, int32_t n_utt, const int32_t* frame_offsets, int32_t F,
                         int32_t Fs, int32_t mode, float quantile_fraction, float lo, float hi,
                         float* labels, int32_t ld, float* thr_out, void* work,
                         int64_t work_bytes, void* stream);
/* ideal_wiener_mask (target.py:104-116): |S|^2 / (|S|^2 + |N|^2 + eps), S, N DEV
 * complex64 [n], mask DEV float [n]. */
int vaenmf_wiener_mask(const float* S, const float* N, int64_t n, float eps, float* mask,
                       void* stream);
/* Supervised Wiener-mask baseline, scripts/evaluate_wiener_filter.py:99: S_hat = mask * X.
 * X, S_hat DEV complex64 [NT][Fs]; mask DEV float [NT][ldm] (bins < F read; bins >= F of
 * S_hat zeroed). */
int vaenmf_apply_mask(const float* X, const float* mask, int32_t ldm, int32_t NT, int32_t F,
                      int32_t Fs, float* S_hat, void* stream);

/* SPP-based speech-presence / noise-PSD estimator -- replaces SPPNoiseEstimator.update
 * driven frame by frame (python/models/spp_estimation.py:17-160; timo_mask_estimation
 * :163-183 is spp_out, the noise PSD the first return value).  per DEV float [NT][ld]
 * noisy periodogram |Y|^2, frame_offsets DEV int32 [n_utt+1]; the recursion runs in
 * float64 per (utterance, bin) and restarts at every utterance.  spp_out / psd_out DEV
 * float [NT][ldo], either may be null.  Defaults of the reference: fixed_smooth 0.8,
 * prob_smooth 0.9, prior 0.5, snr_opt_db 15, num_frames_init 10 (:10-14).
 * vaenmf_spp_noise_given: the v_spp_in branch (:145-153, timo_noise_estimation
 * :218-235), elementwise over n entries. */
int vaenmf_spp_estimate(const float* per, int32_t ld, int32_t n_utt, const int32_t* frame_offsets,
                        int32_t F, double fixed_smooth, double prob_smooth, double prior,
                        double snr_opt_db, int32_t num_frames_init, float* spp_out,
                        float* psd_out, int32_t ldo, void* stream);
int vaenmf_spp_noise_given(const float* per, const float* spp_in, int64_t n, double fixed_smooth,
                           float* psd_out, void* stream);

/* SI-SDR sufficient statistics -- python/metrics.py:12-60: per utterance the Gram
 * matrix of (s_hat, s, n) in float64: out DEV [n_utt][6] =
 * {<sh,sh>, <sh,s>, <sh,n>, <s,s>, <s,n>, <n,n>}; sample_offsets DEV int64 [n_utt+1]. */
int vaenmf_gram3_batch(const float* s_hat, const float* s, const float* n, int32_t n_utt,
                       const int64_t* sample_offsets, double* out, void* stream);

/* Per-kernel device timing with HIP events recorded on the launch stream around every
 * hot-path launch (kinds: 0 mh_chain, 1 decode+W-statistics, 2 W update, 3 decode+H/g/cost,
 * 4 decode+Wiener).  enable(max_launches>0) pre-creates the events (no allocation at
 * launch time); read() synchronises, returns summed milliseconds and launch counts per
 * kind (arrays of 5) and resets. */
int vaenmf_profile_enable(vaenmf_plan* p, int32_t max_launches);
int vaenmf_profile_read(vaenmf_plan* p, double* ms, int64_t* counts);

/* Measurement aid of bench.py (no reference counterpart; not on the hot path): the rate in GB/s at which the GPU delivers
 * a plain streaming read of `bytes` bytes of buf (DEV, 16-byte aligned) to a hand-written kernel (16 bytes per lane, 8
 * wavefronts per SIMD, grid = resident set) -- the measured ceiling the streaming M-step kernels are compared with.
 * `reps` timed sweeps after one untimed one; sink DEV (4 bytes, never written in practice).  Synchronises the stream. */
int vaenmf_hbm_read_probe(const void* buf, int64_t bytes, int32_t reps, void* sink, double* gbps_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VAENMF_H */
