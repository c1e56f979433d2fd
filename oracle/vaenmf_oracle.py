"""CPU oracle for the VAE-NMF reconstruct hot path (numpy, float32).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline.  The product
path (``guided-vae-nmf_amd/``) never imports this module and has no CPU fallback.

It restates, op for op, the algorithm of the reference's hot path
(``python/models/mcem.py``, ``python/models/models.py``,
``python/processing/stft.py``, ``python/metrics.py`` of sp-uhh/guided-vae-nmf);
every function cites the reference file:line it follows.

Pinning: ``tests/golden/make_golden.py`` imports the reference itself (in the
build container only), runs it on seeded weights with *recorded* random draws and
commits inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against every one of those vectors.  STFT/iSTFT: the
reference delegates to librosa (absent here, unpinned version); the restatement
follows librosa's published algorithm and is pinned by round-trip identity plus
the reference-committed processed wavs -- its only golden spectrogram fixtures
are pickles, which are not loaded (unsafe loader) => STFT is "parity unpinned"
against librosa output itself.

All arithmetic is float32 like the reference (torch default dtype), except
``cost`` (float64 array, mcem.py:157) and the SI-SDR metrics (float64).
"""
import math

import numpy as np

f32 = np.float32


# ----------------------------------------------------------------------------
# Random sources.  The reference consumes torch's global CPU generator in this
# order per utterance (mcem.py:42-43, models.py:10, mcem.py:407/420):
#   rand(F,K), rand(K,N), randn(N,L) [encoder reparametrisation, unused],
#   then per MH step randn(L,N), rand(N).
# ----------------------------------------------------------------------------
class ReplayRNG:
    """Replays recorded draws (list of float32 arrays) in order."""

    def __init__(self, draws):
        self.draws = list(draws)
        self.pos = 0

    def _next(self, shape):
        a = np.asarray(self.draws[self.pos], dtype=f32)
        self.pos += 1
        assert tuple(a.shape) == tuple(shape), (a.shape, shape, self.pos - 1)
        return a

    def rand(self, *shape):
        return self._next(shape)

    def randn(self, *shape):
        return self._next(shape)


class NumpyRNG:
    """Seeded numpy generator with the same call surface (cpu_baseline, tests)."""

    def __init__(self, seed=0):
        self.g = np.random.default_rng(seed)

    def rand(self, *shape):
        return self.g.random(shape, dtype=f32)

    def randn(self, *shape):
        return self.g.standard_normal(shape, dtype=f32)


# ----------------------------------------------------------------------------
# MLPs (python/models/models.py)
# ----------------------------------------------------------------------------
def linear(x, w, b):
    """nn.Linear: y = x W^T + b (models.py:96,114,116)."""
    return (x @ w.T + b).astype(f32, copy=False)


def n_hidden(params, prefix):
    i = 0
    while "%s.hidden.%d.weight" % (prefix, i) in params:
        i += 1
    return i


def encoder_forward(params, x, eps=None):
    """Encoder.forward + GaussianSample (models.py:101-104, 33-38, 9-22).

    Returns (z, mu, log_var); z = mu + exp(0.5*log_var)*eps if eps is given
    (the reference always draws eps = randn(N,L), models.py:10), else None.
    """
    h = np.asarray(x, dtype=f32)
    for i in range(n_hidden(params, "encoder")):
        h = np.tanh(linear(h, params["encoder.hidden.%d.weight" % i],
                           params["encoder.hidden.%d.bias" % i]))
    mu = linear(h, params["encoder.sample.mu.weight"], params["encoder.sample.mu.bias"])
    log_var = linear(h, params["encoder.sample.log_var.weight"],
                     params["encoder.sample.log_var.bias"])
    z = None
    if eps is not None:
        z = (mu + np.exp(f32(0.5) * log_var) * eps).astype(f32)
    return z, mu, log_var


def decoder_forward(params, z):
    """Decoder.forward (models.py:118-121): tanh(Linear)* -> exp(Linear)."""
    h = np.asarray(z, dtype=f32)
    for i in range(n_hidden(params, "decoder")):
        h = np.tanh(linear(h, params["decoder.hidden.%d.weight" % i],
                           params["decoder.hidden.%d.bias" % i]))
    return np.exp(linear(h, params["decoder.reconstruction.weight"],
                         params["decoder.reconstruction.bias"]))


def classifier_forward(params, x, two_classes=False, bn_eps=1e-5):
    """Classifier.forward (models.py:57-62): relu(layer) for EVERY module of `hidden` -- Linear and, with batch_norm=True,
    the BatchNorm1d behind it (models.py:50-52: relu(BN(relu(Linear)))), in eval mode (running statistics, as
    scripts/reconstruct_dnn_classif.py:129 runs it) -- then sigmoid(output_layer).  two_classes: Classifier2Classes
    (models.py:64-88): softmax over the two classes of output_layer(x).view(-1, 2, y_dim); returns (N, 2, y_dim)."""
    h = np.asarray(x, dtype=f32)
    i = 0
    while "hidden.%d.weight" % i in params:
        w = params["hidden.%d.weight" % i]
        if w.ndim == 2:
            h = np.maximum(linear(h, w, params["hidden.%d.bias" % i]), f32(0))
        else:                                  # BatchNorm1d, eval mode
            inv = f32(1) / np.sqrt(params["hidden.%d.running_var" % i].astype(f32) + f32(bn_eps))
            h = np.maximum((h - params["hidden.%d.running_mean" % i].astype(f32)) * inv * w.astype(f32) + params["hidden.%d.bias" % i].astype(f32), f32(0))
        i += 1
    o = linear(h, params["output_layer.weight"], params["output_layer.bias"])
    if two_classes:
        o = o.reshape(o.shape[0], 2, -1)
        e = np.exp(o - o.max(1, keepdims=True))
        return (e / e.sum(1, keepdims=True)).astype(f32)
    return (f32(1) / (f32(1) + np.exp(-o))).astype(f32)


def classifier_labels(params, x_pow, mean=None, std=None, eps=1e-8):
    """evaluate_M2_vad.py:122-131: optional (x-mean^T)/(std+eps)^T, classifier,
    hard threshold 0.5.  x_pow (N,F); mean/std (F,1).  Returns (soft, hard)."""
    x = np.asarray(x_pow, dtype=f32)
    if mean is not None:
        x = (x - mean.T.astype(f32)) / (std.astype(f32) + f32(eps)).T
    soft = classifier_forward(params, x.astype(f32))
    return soft, (soft > 0.5).astype(f32)


def xavier_normal_params(dims, seed=0, y_dim=0, bias_std=0.0):
    """Seeded Xavier-normal weights with zero biases in the reference's
    state_dict key layout (models.py:136-140, 193-197).
    dims = [x_dim, z_dim, h_dim list]; y_dim>0 gives the M2 (DeepGenerativeModel)
    shapes: encoder in = x+y, decoder in = z+y (models.py:189-190)."""
    x_dim, z_dim, h_dim = dims
    g = np.random.default_rng(seed)

    def lin(o, i):
        std = math.sqrt(2.0 / (i + o))
        w = (g.standard_normal((o, i)) * std).astype(f32)
        return w, (g.standard_normal(o) * bias_std).astype(f32)   # bias_std=0: reference init

    p = {}
    enc = [x_dim + y_dim, *h_dim]
    for i in range(1, len(enc)):
        p["encoder.hidden.%d.weight" % (i - 1)], p["encoder.hidden.%d.bias" % (i - 1)] = lin(enc[i], enc[i - 1])
    p["encoder.sample.mu.weight"], p["encoder.sample.mu.bias"] = lin(z_dim, h_dim[-1])
    p["encoder.sample.log_var.weight"], p["encoder.sample.log_var.bias"] = lin(z_dim, h_dim[-1])
    dec = [z_dim + y_dim, *reversed(h_dim)]
    for i in range(1, len(dec)):
        p["decoder.hidden.%d.weight" % (i - 1)], p["decoder.hidden.%d.bias" % (i - 1)] = lin(dec[i], dec[i - 1])
    p["decoder.reconstruction.weight"], p["decoder.reconstruction.bias"] = lin(x_dim, dec[-1])
    return p


def xavier_normal_classifier(dims, seed=0, bias_std=0.0):
    """Classifier([x_dim, h_dim, y_dim]) key layout (models.py:44-55)."""
    x_dim, h_dim, y_dim = dims
    g = np.random.default_rng(seed)
    p = {}
    neurons = [x_dim, *h_dim]
    for i in range(1, len(neurons)):
        std = math.sqrt(2.0 / (neurons[i] + neurons[i - 1]))
        p["hidden.%d.weight" % (i - 1)] = (g.standard_normal((neurons[i], neurons[i - 1])) * std).astype(f32)
        p["hidden.%d.bias" % (i - 1)] = (g.standard_normal(neurons[i]) * bias_std).astype(f32)
    std = math.sqrt(2.0 / (h_dim[-1] + y_dim))
    p["output_layer.weight"] = (g.standard_normal((y_dim, h_dim[-1])) * std).astype(f32)
    p["output_layer.bias"] = (g.standard_normal(y_dim) * bias_std).astype(f32)
    return p


# ----------------------------------------------------------------------------
# MCEM (python/models/mcem.py)
# ----------------------------------------------------------------------------
def _inv(x):
    return (f32(1) / x).astype(f32, copy=False)


class MCEMOracle:
    """EM base + MCEM_M1 / MCEM_M2 (mcem.py:8-178, 181-345, 348-490).

    ``model`` is "M1" or "M2".  For M1, ``reference_compat=True`` reproduces the
    positional-argument shift of mcem.py:461-462/477-478 (sample_posterior is
    declared (Z, y, nsamples, burnin) at :371 but called with (Z, nsamples,
    burnin)): E-step runs nsamples=burnin_E_step, burnin=30 (default) and the
    Wiener chain nsamples=burnin_WF, burnin=30.  M2 runs as documented.
    """

    def __init__(self, model, niter, nsamples_E_step=10, burnin_E_step=30,
                 nsamples_WF=25, burnin_WF=75, var_RW=0.01, reference_compat=True):
        assert model in ("M1", "M2")
        self.model = model
        self.niter = niter
        self.nsamples_E_step = nsamples_E_step
        self.burnin_E_step = burnin_E_step
        self.nsamples_WF = nsamples_WF
        self.burnin_WF = burnin_WF
        self.var_RW = var_RW
        self.reference_compat = reference_compat

    # -- effective (nsamples, burnin) per phase ------------------------------
    def e_step_counts(self):
        if self.model == "M1" and self.reference_compat:
            return self.burnin_E_step, 30          # mcem.py:461-462 + default :371
        return self.nsamples_E_step, self.burnin_E_step

    def wf_counts(self):
        if self.model == "M1" and self.reference_compat:
            return self.burnin_WF, 30              # mcem.py:477-478 + default :371
        return self.nsamples_WF, self.burnin_WF

    # -- mcem.py:36-57, 361-369 (M1), 207-216 (M2) -----------------------------
    def init_parameters(self, X, params, nmf_rank, eps, rng, y=None, W0=None, H0=None):
        """X complex64 (N,F); y float32 (N,Dy) for M2.  Draw order: rand(F,K),
        rand(K,N) (mcem.py:42-43), then the encoder's randn(N,L) (models.py:10)."""
        N, F = X.shape
        self.rng = rng
        self.params = params
        if W0 is None:
            W0 = np.maximum(rng.rand(F, nmf_rank), f32(eps))
            H0 = np.maximum(rng.rand(nmf_rank, N), f32(eps))
        self.W = np.asarray(W0, f32).copy()
        self.H = np.asarray(H0, f32).copy()
        self.g = np.ones(N, f32)
        self.X = X.T                                   # (F,N) mcem.py:46
        self.X_abs_2 = (np.abs(X.T) ** 2).astype(f32)  # mcem.py:47
        self.compute_Vb()
        self.Vs = self.Vs_scaled = self.Vx = None
        if self.model == "M2":
            self.y = np.asarray(y, f32).T              # (Dy,N) mcem.py:213
            enc_in = np.concatenate([self.X_abs_2, self.y], 0).T
        else:
            self.y = None
            enc_in = self.X_abs_2.T
        L = params["encoder.sample.mu.weight"].shape[0]
        self.L = L
        eps_enc = rng.randn(N, L)                      # drawn, result unused (mcem.py:367)
        _, mu, _ = encoder_forward(params, enc_in, eps_enc)
        self.Z = mu.T.copy()                           # (L,N) posterior MEAN mcem.py:367-368

    def compute_Vb(self):
        self.Vb = (self.W @ self.H).astype(f32)        # mcem.py:81-82

    def _dec(self, Z_LN):
        """decoder(cat([Z,y]).T).T -> (F,N)  (mcem.py:392 / :242)."""
        zin = Z_LN if self.y is None else np.concatenate([Z_LN, self.y], 0)
        return decoder_forward(self.params, zin.T).T

    # -- mcem.py:371-441 (M1) / 218-294 (M2) -----------------------------------
    def sample_posterior(self, Z, nsamples, burnin, trace=None):
        F, N = self.X.shape
        L = self.L
        sd = np.sqrt(f32(self.var_RW))                 # torch.sqrt(var_RM_t)
        Zs = np.zeros((N, nsamples, L), f32)
        Z_t = Z.copy()
        Vs_t = self._dec(Z_t)
        g_t = self.g.copy()
        Vb_t = self.Vb.copy()
        Vx_t = g_t * Vs_t + Vb_t
        cpt = 0
        for m in range(nsamples + burnin):
            Zp = (Z_t + sd * self.rng.randn(L, N)).astype(f32)          # :407
            Vsp = self._dec(Zp)                                         # :410
            Vxp = g_t * Vsp + Vb_t                                      # :411-412
            acc = (np.sum(np.log(Vx_t) - np.log(Vxp)
                          + (_inv(Vx_t) - _inv(Vxp)) * self.X_abs_2, 0)
                   + f32(.5) * np.sum(Z_t ** 2 - Zp ** 2, 0)).astype(f32)  # :415-417
            u = self.rng.rand(N)
            is_acc = np.log(u) < acc                                    # :420
            if trace is not None:
                trace.append(dict(acc=acc.copy(), is_acc=is_acc.copy()))
            Z_t[:, is_acc] = Zp[:, is_acc]                              # :429
            Vs_t = self._dec(Z_t)                                       # :432
            Vx_t = g_t * Vs_t + Vb_t
            if m > burnin - 1:                                          # :435-437
                Zs[:, cpt, :] = Z_t.T
                cpt += 1
        return Zs

    # -- mcem.py:444-454 / 297-307 ---------------------------------------------
    def compute_Vs(self, Zs):
        N, R, L = Zs.shape
        if self.y is not None:
            zin = np.concatenate([Zs, np.broadcast_to(self.y.T[:, None, :], (N, R, self.y.shape[0]))], 2)
        else:
            zin = Zs
        Vs = decoder_forward(self.params, zin.reshape(N * R, -1)).reshape(N, R, -1)
        self.Vs = np.ascontiguousarray(np.moveaxis(Vs, 0, -1))          # (R,F,N)

    def compute_Vs_scaled(self):
        self.Vs_scaled = self.g * self.Vs                               # :75-76

    def compute_Vx(self):
        self.Vx = self.Vs_scaled + self.Vb                              # :78-79

    # -- mcem.py:456-471 / 309-325 ---------------------------------------------
    def E_step(self):
        ns, bi = self.e_step_counts()
        Zs = self.sample_posterior(self.Z, ns, bi)
        self.Z_samples = Zs
        self.Z = Zs[:, -1, :].T.copy()
        self.compute_Vs(Zs)
        self.compute_Vs_scaled()
        self.compute_Vx()

    # -- mcem.py:90-152 ----------------------------------------------------------
    def M_step(self):
        X2 = self.X_abs_2
        iv = _inv(self.Vx)
        num = (X2 * np.sum(iv * iv, 0)) @ self.H.T                      # :107
        den = np.sum(iv, 0) @ self.H.T                                  # :109
        self.W = (self.W * np.sqrt(num / den)).astype(f32)              # :110
        self.compute_Vb(); self.compute_Vx()                            # :113-114
        iv = _inv(self.Vx)
        num = self.W.T @ (X2 * np.sum(iv * iv, 0))                      # :118
        den = self.W.T @ np.sum(iv, 0)                                  # :120
        self.H = (self.H * np.sqrt(num / den)).astype(f32)              # :121
        self.compute_Vb(); self.compute_Vx()                            # :124-125
        norm_col_W = np.sum(np.abs(self.W), 0)                          # :129
        self.W = self.W / norm_col_W[None, :]                           # :131
        self.H = self.H * norm_col_W[:, None]                           # :133
        iv = _inv(self.Vx)
        num = np.sum(X2 * np.sum(self.Vs * iv * iv, 0), 0)              # :138
        den = np.sum(np.sum(self.Vs * iv, 0), 0)                        # :141
        self.g = (self.g * np.sqrt(num / den)).astype(f32)              # :142
        self.compute_Vs_scaled(); self.compute_Vx()                     # :151-152

    def compute_expected_neg_log_like(self):
        return np.mean(np.log(self.Vx) + self.X_abs_2 / self.Vx, dtype=f32)  # :70

    # -- mcem.py:473-490 / 327-345 ---------------------------------------------
    def compute_WF(self, sample=False):
        if sample:
            ns, bi = self.wf_counts()
            Zs = self.sample_posterior(self.Z, ns, bi)
            self.Z_samples_WF = Zs
            self.compute_Vs(Zs)
            self.compute_Vs_scaled()
            self.compute_Vx()
        WFs = np.mean(self.Vs_scaled / self.Vx, 0, dtype=f32)
        WFn = np.mean(self.Vb / self.Vx, 0, dtype=f32)
        return WFs, WFn

    # -- mcem.py:155-178 ---------------------------------------------------------
    def run(self):
        cost = np.zeros(self.niter)
        for n in range(self.niter):
            self.E_step()
            self.M_step()
            cost[n] = self.compute_expected_neg_log_like()
        WFs, WFn = self.compute_WF(sample=True)
        self.WFs, self.WFn = WFs, WFn
        self.S_hat = WFs * self.X                                       # :175
        self.N_hat = WFn * self.X                                       # :176
        return cost


class MCEMOracleNoNMF(MCEMOracle):
    """EM_noNMF + MCEM_M2_noNMF (mcem.py:493-760): the noise variance Vb is given and fixed, only the
    gains g are updated.  Constructor-style API like the reference (:608-629): X complex (N,F), Vb (N,F),
    g (N,), Z (N,L), y (N,Dy)."""

    def __init__(self, X, Vb, g, Z, y, params, niter, rng, nsamples_E_step=10, burnin_E_step=30,
                 nsamples_WF=25, burnin_WF=75, var_RW=0.01):
        super().__init__("M2", niter, nsamples_E_step, burnin_E_step, nsamples_WF, burnin_WF, var_RW)
        self.rng, self.params = rng, params
        self.X = X.T                                             # mcem.py:503
        self.X_abs_2 = (np.abs(X.T) ** 2).astype(f32)            # :504
        self.Vb = np.asarray(Vb, f32).T.copy()                   # :507
        self.g = np.asarray(g, f32).copy()                       # :508
        self.Z = np.asarray(Z, f32).T.copy()                     # :617
        self.y = np.asarray(y, f32).T.copy()                     # :618
        self.L = self.Z.shape[0]
        self.Vs = self.Vs_scaled = self.Vx = None

    def M_step(self):                                            # mcem.py:543-578
        self.compute_Vx()
        iv = _inv(self.Vx)
        num = np.sum(self.X_abs_2 * np.sum(self.Vs * iv * iv, 0), 0)
        den = np.sum(np.sum(self.Vs * iv, 0), 0)
        self.g = (self.g * np.sqrt(num / den)).astype(f32)
        self.compute_Vs_scaled(); self.compute_Vx()


# ----------------------------------------------------------------------------
# STFT / iSTFT (python/processing/stft.py -> librosa.core.stft/istft)
# ----------------------------------------------------------------------------
def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) as librosa uses."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))


def stft(x, fs=16e3, wlen_sec=50e-3, win="hann", hop_percent=0.25, center=True,
         pad_mode="reflect", pad_at_end=True, dtype="complex64"):
    """stft.py:16-63.  End-pad rule (:48-53), then librosa.core.stft: reflect
    pad n_fft/2 both sides, periodic Hann, frames at hop, rfft -> (F, n_frames)."""
    if wlen_sec * fs != int(wlen_sec * fs):
        raise ValueError("wlen_sample of STFT is not an integer.")       # :37-38
    assert win == "hann" and center and pad_mode == "reflect"
    nfft = int(wlen_sec * fs)
    hop = int(hop_percent * nfft)
    x_ = np.asarray(x)
    if pad_at_end:
        utt_len = len(x) / fs
        if math.ceil(utt_len / wlen_sec / hop_percent) != int(utt_len / wlen_sec / hop_percent):
            x_ = np.pad(x_, (0, hop), mode="constant")                    # :51
    y = np.pad(x_, nfft // 2, mode="reflect")
    n_frames = 1 + (len(y) - nfft) // hop
    w = hann_periodic(nfft)
    idx = np.arange(nfft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = y[idx] * w[None, :]
    return np.fft.rfft(frames, axis=1).T.astype(dtype)


def istft(Sxx, fs=16000, wlen_sec=50e-3, win="hann", hop_percent=0.25, center=True,
          dtype="float32", max_len=None):
    """stft.py:66-102 -> librosa.core.istft: irfft, synthesis Hann, overlap-add,
    divide by window-sum-square where > tiny, strip centre pad, fix to length."""
    if wlen_sec * fs != int(wlen_sec * fs):
        raise ValueError("wlen_sample of iSTFT is not an integer.")       # :87-88
    nfft = int(wlen_sec * fs)
    hop = int(hop_percent * nfft)
    n_frames = Sxx.shape[1]
    w = hann_periodic(nfft)
    ytmp = np.fft.irfft(Sxx.T, n=nfft, axis=1) * w[None, :]
    n_out = nfft + hop * (n_frames - 1)
    y = np.zeros(n_out)
    wss = np.zeros(n_out)
    for i in range(n_frames):
        y[i * hop:i * hop + nfft] += ytmp[i]
        wss[i * hop:i * hop + nfft] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    y = y[nfft // 2:]                                   # center=True
    if max_len is not None:
        if len(y) >= max_len:
            y = y[:max_len]
        else:
            y = np.pad(y, (0, max_len - len(y)))
    else:
        y = y[:len(y) - nfft // 2]
    return y.astype(dtype)


# ----------------------------------------------------------------------------
# Metrics (python/metrics.py)
# ----------------------------------------------------------------------------
def si_sdr_components(s_hat, s, n):
    """metrics.py:12-37."""
    alpha_s = np.dot(s_hat, s) / np.linalg.norm(s) ** 2
    s_target = alpha_s * s
    alpha_n = np.dot(s_hat, n) / np.linalg.norm(n) ** 2
    e_noise = alpha_n * n
    e_art = s_hat - s_target - e_noise
    return s_target, e_noise, e_art


def energy_ratios(s_hat, s, n):
    """metrics.py:39-60 -> (si_sdr, si_sir, si_sar) in dB."""
    s_target, e_noise, e_art = si_sdr_components(s_hat, s, n)
    st = np.linalg.norm(s_target) ** 2
    si_sdr = 10 * np.log10(st / np.linalg.norm(e_noise + e_art) ** 2)
    si_sir = 10 * np.log10(st / np.linalg.norm(e_noise) ** 2)
    si_sar = 10 * np.log10(st / np.linalg.norm(e_art) ** 2)
    return si_sdr, si_sir, si_sar


def mean_confidence_interval(data, confidence=0.95):
    """metrics.py:5-10 (t-distribution CI)."""
    import scipy.stats
    a = 1.0 * np.array(data)
    n = len(a)
    m, se = np.mean(a), scipy.stats.sem(a)
    h = se * scipy.stats.t.ppf((1 + confidence) / 2., n - 1)
    return np.round(m, 3), np.round(h, 3)


# ----------------------------------------------------------------------------
# Synthetic workload (mixing recipe of scripts/create_test_set.py:74-103)
# ----------------------------------------------------------------------------
def synth_utterance(seed, n_samples=64000, fs=16000):
    """Seeded speech-like + noise mixture.  Returns (s, n, x, snr_db) float64."""
    g = np.random.default_rng(1000 + seed)
    from scipy.signal import lfilter
    e = g.standard_normal(n_samples)
    s = lfilter([1.0], [1.0, -1.6, 0.81], e)              # AR(2) resonance
    t = np.arange(n_samples) / fs
    env = 0.5 * (1 + np.sign(np.sin(2 * np.pi * (1.5 + 0.5 * g.random()) * t + g.random() * 6.28)))
    env = np.convolve(env, np.ones(400) / 400, mode="same")
    s = s * env
    s = s / np.max(np.abs(s))
    nz = g.standard_normal(n_samples)
    nz = np.convolve(nz, [1.0, 0.6, 0.3], mode="same")
    snr_db = [-5.0, 0.0, 5.0][int(np.random.RandomState(seed).randint(3))]
    k = np.sum(s ** 2) * 10 ** (-snr_db / 10) / np.sum(nz ** 2)
    nz = nz * np.sqrt(k)
    norm = np.max(np.abs(np.concatenate([s, nz, s + nz])))
    return s / norm, nz / norm, (s + nz) / norm, snr_db


# ----------------------------------------------------------------------------
# Label / guide front-ends (python/processing/target.py)
# ----------------------------------------------------------------------------
def pairwise_sum_f32(a):
    """NumPy's float32 add-reduce over a 1-D run, restated (numpy/_core/src/umath/loops_utils.h.src,
    pairwise_sum; numpy 2.2 is this image's pinned version): < 8 elements a running sum; <= 128 eight
    interleaved partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the tail; else split at
    n/2 rounded down to a multiple of 8.  Pins the order the HIP label kernels must reproduce bit for bit."""
    a = np.asarray(a, dtype=np.float32)
    n = len(a)
    f = np.float32
    if n < 8:
        r = f(0.0)
        for x in a:
            r = f(r + x)
        return r
    if n <= 128:
        r = [f(a[i]) for i in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = f(r[k] + a[i + k])
            i += 8
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        while i < n:
            res = f(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f(pairwise_sum_f32(a[:n2]) + pairwise_sum_f32(a[n2:]))


def power_c64(obs):
    """abs(obs * obs.conj()) of a complex64 array as this image's numpy evaluates it (target.py:16, :37):
    the product's real part is fma(re, re, round(im*im)), its imaginary part exactly 0."""
    re = obs.real.astype(np.float64)
    im2 = (obs.imag.astype(np.float32) * obs.imag.astype(np.float32)).astype(np.float64)
    return (re * re + im2).astype(np.float32)


def lorenz_threshold(power, quantile_fraction):
    """target.py:18-22 / :39-42: descending sort, Lorenz curve cumsum/sum in float32 (running cumsum,
    pairwise total), threshold = last sorted value whose Lorenz value is below the fraction."""
    srt = np.sort(np.asarray(power, np.float32), axis=None)[::-1]
    total = pairwise_sum_f32(srt)
    run = np.float32(0.0)
    thr = None
    q = np.float32(quantile_fraction)
    for v in srt:
        run = np.float32(run + v)
        if np.float32(run / total) < q:
            thr = v
        else:
            break                      # the curve is non-decreasing: nothing further qualifies
    if thr is None:
        raise IndexError("index -1 is out of bounds for axis 0 with size 0")   # what target.py:22 raises
    return thr


def _soften(mask, quantile_weight):
    """target.py:24-27: 0.5 + w (mask - 0.5), rounded half-to-even, float32."""
    return np.float32(np.round(0.5 + quantile_weight * (mask.astype(np.float64) - 0.5)))


def clean_speech_IBM(observations, quantile_fraction=0.98, quantile_weight=0.999):
    """target.py:7-28.  observations complex64 (F, N) -> float32 (F, N) in {0,1}."""
    power = power_c64(observations)
    return _soften(power > lorenz_threshold(power, quantile_fraction), quantile_weight)


def frame_power(observations):
    """target.py:38: power.sum(axis=0) -- for the (F, N) Fortran-ordered STFT the reference's stft returns,
    each frame is a contiguous run of F values reduced pairwise."""
    power = power_c64(observations)
    return np.array([pairwise_sum_f32(power[:, n]) for n in range(power.shape[1])], dtype=np.float32)


def clean_speech_VAD(observations, quantile_fraction=0.98, quantile_weight=0.999):
    """target.py:30-50 -> float32 (1, N)."""
    p = frame_power(observations)
    return _soften(p > lorenz_threshold(p, quantile_fraction), quantile_weight)[None]


def noise_robust_clean_speech_VAD(observations, quantile_fraction_begin=0.93, quantile_fraction_end=0.99, quantile_weight=0.999):
    """target.py:52-76: active from the first frame of the strict VAD up to (excluding) the last frame of the
    lenient one."""
    vad = clean_speech_VAD(observations, quantile_fraction_begin, quantile_weight)[0]
    end = clean_speech_VAD(observations, quantile_fraction_end, quantile_weight)[0]
    b, e = np.nonzero(vad)[0][0], np.nonzero(end)[0][-1]
    vad[b:e] = 1
    return vad[None]


def noise_robust_clean_speech_IBM(observations, vad_quantile_fraction_begin=0.93, vad_quantile_fraction_end=0.99,
                                  ibm_quantile_fraction=0.999, quantile_weight=0.999):
    """target.py:78-102."""
    vad = noise_robust_clean_speech_VAD(observations, vad_quantile_fraction_begin, vad_quantile_fraction_end, quantile_weight)
    return clean_speech_IBM(observations, ibm_quantile_fraction, quantile_weight) * vad


def ideal_wiener_mask(speech_tf, noise_tf, eps=1e-8):
    """target.py:104-116."""
    sp = np.power(abs(speech_tf), 2)
    npow = np.power(abs(noise_tf), 2)
    return sp / (sp + npow + eps)


# ----------------------------------------------------------------------------
# SPP-based speech-presence / noise-PSD estimator (python/models/spp_estimation.py)
# ----------------------------------------------------------------------------
def spp_recursion(per, fixed_smooth=0.8, prob_smooth=0.9, prior=0.5, snr_opt_db=15, num_frames_init=10):
    """SPPNoiseEstimator.update applied frame by frame (spp_estimation.py:92-143), all bins at once.
    per (frames, bins) -> (noise_psd, spp) float64 (frames, bins)."""
    per = np.asarray(per)
    nfr, nb = per.shape
    snr = 10.0 ** (snr_opt_db / 10.0)                                    # :76
    k_glr = (1 - prior) / prior * (1.0 + snr)                            # :85
    k_exp = snr / (1.0 + snr)                                            # :86
    old = np.zeros(nb)
    smooth = np.zeros(nb)
    psd_out = np.zeros((nfr, nb))
    spp_out = np.zeros((nfr, nb))
    for i in range(nfr):
        y = per[i]
        if i < num_frames_init:                                          # :105-117: running mean; returns the periodogram
            old = old + y / num_frames_init
            psd_out[i] = y
            continue
        inv_glr = k_glr * np.exp(-y / (old + 1e-8) * k_exp)              # :120-121
        spp = 1.0 / (1.0 + inv_glr)                                      # :124
        smooth = (1 - prob_smooth) * spp + prob_smooth * smooth          # :128-129
        stuck = smooth > 0.99
        spp[stuck] = np.minimum(spp[stuck], 0.99)                        # :130-131
        nper = (1.0 - spp) * y + spp * old                               # :135-136
        old = (1.0 - fixed_smooth) * nper + fixed_smooth * old           # :138-142
        psd_out[i] = old
        spp_out[i] = spp
    return psd_out, spp_out


def timo_mask_estimation(spectrogram):
    """spp_estimation.py:163-183: (bins, frames) power spectrogram -> SPP mask, same shape and dtype."""
    return spp_recursion(np.asarray(spectrogram).T)[1].T.astype(np.asarray(spectrogram).dtype)


def timo_vad_estimation(spectrogram):
    """spp_estimation.py:185-214."""
    s = np.asarray(spectrogram).sum(axis=0)
    return spp_recursion(s[:, None])[1][:, 0].astype(s.dtype)


def timo_noise_estimation(spectrogram, mask, fixed_smooth=0.8):
    """spp_estimation.py:218-235: the v_spp_in branch of update (:145-153) returns before the old PSD is stored,
    so the old PSD is zero throughout."""
    sp = np.asarray(spectrogram)
    nper = (np.float32(1.0) - np.asarray(mask, np.float32)) * sp.astype(np.float32)     # float32 product (:147), then float64
    return ((1.0 - fixed_smooth) * nper.astype(np.float64)).astype(sp.dtype)
