"""Batched reconstruct pipeline (the build's counterpart of scripts/evaluate_M1.py:
111-177 process_utt/process_sublist, many utterances per launch): device-resident
waveforms -> STFT -> |X|^2 -> encoder -> fused EM (MH E-steps, multiplicative-update
M-steps) -> Wiener filter -> iSTFT -> enhanced waveforms, plus utterance sharding over
ranks (np.array_split, evaluate_M1.py:203-206) and the final metrics all-reduce."""
import numpy as np
import torch

from . import stft as vstft
from . import metrics as vmet
from .engine import BatchEngine, decoder_params_from_state, latent_dim_from_state
from .mcem import _encoder_params


def shard(items, world_size, rank):
    """Contiguous split of the sorted list over ranks, exactly np.array_split
    (scripts/evaluate_M1.py:203)."""
    return list(np.array_split(np.asarray(items, dtype=object), world_size)[rank])


class Reconstructor:
    def __init__(self, state_dict, x_dim, nmf_rank, niter=100, nsamples_E_step=10, burnin_E_step=30,
                 nsamples_WF=25, burnin_WF=75, var_RW=0.01, model="M1", reference_compat=True,
                 fs=16000, wlen_sec=64e-3, hop_percent=0.25, eps=1e-8, precision="bf16x3", device="cuda:0",
                 max_frames=1 << 16, max_utts=128, store=None):
        self.store = store          # sample-variance store in the fused run: None = the engine's default (on in bf16 mode)
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(v)) for k, v in state_dict.items()}
        self.enc = _encoder_params(sd)
        self.F, self.K, self.niter, self.var_RW, self.eps = int(x_dim), int(nmf_rank), int(niter), var_RW, eps
        self.fs, self.wlen_sec, self.hop_percent = fs, wlen_sec, hop_percent
        if model == "M1" and reference_compat:          # mcem.py:461-462 / :477-478 positional shift
            self.nsE, self.biE, self.nsW, self.biW = burnin_E_step, 30, burnin_WF, 30
        else:
            self.nsE, self.biE, self.nsW, self.biW = nsamples_E_step, burnin_E_step, nsamples_WF, burnin_WF
        self.device = torch.device(device)
        self.eng = BatchEngine(self.F, self.K, decoder_params_from_state(sd), precision=precision, device=device,
                               max_frames=max_frames, max_utts=max_utts, z_dim=latent_dim_from_state(sd))
        self.model = model

    def enhance(self, wav, sample_counts, seeds=None, init_seed=0, y=None, classifier=None, mean=None, std=None):
        """wav: device float32 [sum T].  Returns (s_hat, n_hat) device float32 [sum T], cost [U,niter] (device).
        M2: give the labels `y` (device [NT,Dy]) or a classifier [(W,b)...] (+ optional mean/std, (F,1)):
        labels = classifier(normalised |X|^2) > 0.5 as in scripts/evaluate_M2_vad.py:122-131."""
        eng = self.eng
        X, fc = vstft.stft_batch(wav, sample_counts, self.fs, self.wlen_sec, self.hop_percent, Fs=eng.Fs, device=self.device)
        eng.bind(fc, Rcap=max(self.nsE, self.nsW), seeds=seeds)
        eng.set_spectrogram(X)
        # W = max(rand(F,K), eps), H = max(rand(K,N), eps), g = 1 (mcem.py:42-44, :51): the library's generator, keyed by
        # the utterance seeds (an utterance's start does not depend on the batch it sits in)
        eng.init_nmf_device(salt=init_seed, eps=self.eps)
        if classifier is not None:
            self.y_soft, y = eng.classify(classifier, mean, std, self.eps)
            self.y_hard = y
        if y is not None:
            eng.set_labels(y)
        eng.encode(self.enc, y)
        cost, S, N = eng.run(self.niter, self.nsE, self.biE, self.nsW, self.biW, self.var_RW, store=self.store)
        nfft, hop = vstft.frame_geometry(sample_counts[0], self.fs, self.wlen_sec, self.hop_percent)[:2]
        s_hat = vstft.istft_batch(S, fc, sample_counts, nfft, hop, device=self.device)
        n_hat = vstft.istft_batch(N, fc, sample_counts, nfft, hop, device=self.device)
        self.frame_counts = fc
        return s_hat, n_hat, cost


class MaskEnhancer:
    """Supervised Wiener-mask baseline, the build's counterpart of scripts/evaluate_wiener_filter.py:71-113
    (process_utt), batched: soft mask = classifier(normalised |X|^2) (ReLU hidden layers, sigmoid output of
    F units, python/models/models.py:160-185), S_hat = mask * X, iSTFT with max_len = T_orig.
    classifier = [(W, b) hidden..., (W, b) output] float32 numpy; mean/std (F,1) or None (std_norm False)."""

    def __init__(self, classifier, x_dim, mean=None, std=None, fs=16000, wlen_sec=64e-3, hop_percent=0.25, eps=1e-8,
                 device="cuda:0"):
        self.F, self.fs, self.wlen_sec, self.hop_percent, self.eps = int(x_dim), fs, wlen_sec, hop_percent, eps
        self.device = torch.device(device)
        layers = [(np.asarray(w, np.float64), np.asarray(b, np.float64)) for w, b in classifier]
        if mean is not None:                 # (x - mean) / (std + eps) folded into the first layer (:88-90)
            m = np.asarray(mean, np.float64).reshape(-1)
            s = np.asarray(std, np.float64).reshape(-1) + eps
            w0, b0 = layers[0]
            layers[0] = (w0 / s[None, :], b0 - (w0 / s[None, :]) @ m)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        self.layers = [(t(w), t(b)) for w, b in layers]

    def enhance(self, wav, sample_counts):
        """wav device float32 [sum T] -> (s_hat device float32 [sum T], soft mask device float32 [NT][F])."""
        from ._lib import lib, check, ACT_RELU, ACT_SIGMOID
        ptr = lambda x: x.data_ptr()
        st = torch.cuda.current_stream().cuda_stream
        Fs = (self.F + 15) // 16 * 16
        X, fc = vstft.stft_batch(wav, sample_counts, self.fs, self.wlen_sec, self.hop_percent, Fs=Fs, device=self.device)
        NT = X.shape[0]
        Xr = X                                            # [NT][Fs][2] float32 = complex64
        X2 = torch.empty(NT, Fs, dtype=torch.float32, device=self.device)
        check(lib().vaenmf_power_spec(ptr(Xr), ptr(X2), NT * Fs, st))                        # :84
        h, ld = X2, Fs
        for i, (w, b) in enumerate(self.layers):
            out = w.shape[0]
            y = torch.empty(NT, out, dtype=torch.float32, device=self.device)
            act = ACT_SIGMOID if i == len(self.layers) - 1 else ACT_RELU
            check(lib().vaenmf_dense(ptr(h), NT, w.shape[1], ld, ptr(w), ptr(b), out, act, ptr(y), out, st))
            h, ld = y, out
        if h.shape[1] != self.F:
            raise ValueError("the mask classifier must end in F = %d units, got %d" % (self.F, h.shape[1]))
        S = torch.empty_like(Xr)
        check(lib().vaenmf_apply_mask(ptr(Xr), ptr(h), self.F, NT, self.F, Fs, ptr(S), st))  # :99
        nfft, hop = vstft.frame_geometry(sample_counts[0], self.fs, self.wlen_sec, self.hop_percent)[:2]
        s_hat = vstft.istft_batch(S, fc, sample_counts, nfft, hop, device=self.device)
        self.frame_counts = fc
        return s_hat, h


def allreduce_stats(stats, device):
    """Sum the metric sufficient statistics over ranks (RCCL when the process group is
    'nccl'); the only collective of the whole job."""
    import torch.distributed as dist
    t = torch.as_tensor(stats, dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():          # (also with one rank: the same call path as the N-rank job)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
