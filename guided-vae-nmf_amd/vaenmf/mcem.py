"""MCEM_M1 / MCEM_M2 with the reference's call surface (python/models/mcem.py:348-490,
181-345): ctor, init_parameters(...), run() -> float64 cost[niter], result attributes
S_hat / N_hat (numpy complex64 (F,N)) and the inspectable state W, H, g, Z, Vb, Vs,
Vs_scaled, Vx, X, X_abs_2.  One utterance per object like the reference; everything
numerical runs in libvaenmf.so through BatchEngine (batch of one).  For throughput use
vaenmf.pipeline (many utterances per launch).

rng="replay" draws the reference's random numbers from torch's global CPU generator in
the reference's order (mcem.py:42-43, models.py:10, mcem.py:407/420) and replays them
on the device; rng="device" uses the on-device generator.
reference_compat=True reproduces MCEM_M1's positional-argument shift
(mcem.py:371 vs :461-462, :477-478): E-step = burnin_E_step samples after 30 burn-in
steps, Wiener chain = burnin_WF samples after 30."""
import numpy as np
import torch

from .engine import BatchEngine, decoder_params_from_state, LAT


def _state(vae):
    if not hasattr(vae, "state_dict"):
        raise TypeError("vae must expose state_dict() in the reference key layout "
                        "(encoder.hidden.*, encoder.sample.mu, decoder.hidden.*, decoder.reconstruction)")
    return {k: v.detach().cpu() for k, v in vae.state_dict().items()}


def _weights_key(vae):
    """Identity of a model's weights: storage address, in-place version counter (load_state_dict, an optimiser step or
    any other in-place write bumps it) and shape of every state_dict entry."""
    try:
        sd = vae.state_dict(keep_vars=True)
    except TypeError:
        sd = vae.state_dict()
    return (id(vae),) + tuple((k, v.data_ptr(), v._version, tuple(v.shape)) for k, v in sd.items())


def _encoder_params(sd):
    enc, i = [], 0
    while "encoder.hidden.%d.weight" % i in sd:
        enc.append((sd["encoder.hidden.%d.weight" % i].numpy(), sd["encoder.hidden.%d.bias" % i].numpy()))
        i += 1
    enc.append((sd["encoder.sample.mu.weight"].numpy(), sd["encoder.sample.mu.bias"].numpy()))
    return enc


class _MCEM:
    model = None

    def __init__(self, niter, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, var_RW=0.01,
                 rng="replay", precision="bf16x3", reference_compat=True, fused_store=True):
        self.niter = niter
        # rng="device": run() is ONE call of the fused driver; fused_store=True lets the chain keep its samples' variances
        # in HBM for the streaming M-step (one 4 s utterance: 69 -> ~40 ms); False decodes the samples again like the
        # step-wise E_step / M_step calls do (bit-equal to them)
        self.fused_store = fused_store
        self.nsamples_E_step, self.burnin_E_step = nsamples_E_step, burnin_E_step
        self.nsamples_WF, self.burnin_WF = nsamples_WF, burnin_WF
        self.var_RW = var_RW
        self.rng, self.precision, self.reference_compat = rng, precision, reference_compat
        self.Vs = self.Vs_scaled = self.Vx = None
        self._eng = self._eng_key = self._enc = None
        self._call = 0

    # picklable like the reference object (sent through a spawn Pool, evaluate_M1.py:206-216)
    def __getstate__(self):
        d = dict(self.__dict__)
        for k in ("_eng", "vae", "_eng_key", "_enc"):
            d[k] = None
        return {k: v for k, v in d.items() if not isinstance(v, torch.Tensor) or not v.is_cuda}

    def e_step_counts(self):
        if self.model == "M1" and self.reference_compat:
            return self.burnin_E_step, 30
        return self.nsamples_E_step, self.burnin_E_step

    def wf_counts(self):
        if self.model == "M1" and self.reference_compat:
            return self.burnin_WF, 30
        return self.nsamples_WF, self.burnin_WF

    def _init(self, X, y, vae, nmf_rank, eps, device):
        if type(vae).__name__ == "RVAE":
            raise NameError("MCEM algorithm only valid for FFNN VAE")         # mcem.py:362-363
        dev = torch.device(device if device not in (None, "cpu") else "cuda:0")
        N, F = X.shape
        L = getattr(vae, "latent_dim", None) or getattr(vae, "z_dim")
        if L not in (16, LAT):
            raise NotImplementedError("latent dim %d: this build supports 16 and %d" % (L, LAT))
        self._L = L
        ns_e, _ = self.e_step_counts()
        ns_w, _ = self.wf_counts()
        self.device, self.vae = dev, vae
        # The reference calls init_parameters once per utterance on ONE object (scripts/evaluate_M1.py:111-166): the engine
        # (plan, packed decoder weights, device buffers, encoder weights on the device) is kept across calls while the model,
        # its weights (identity and in-place version of every tensor), F, K, precision and device stay the same, and grows
        # when an utterance has more frames than any before it.
        key = (F, int(nmf_rank), self.precision, str(dev), max(ns_e, ns_w), _weights_key(vae))
        eng = self._eng
        if eng is None or self._eng_key != key or N > eng._max_frames:
            sd = _state(vae)
            cap = N if (eng is None or self._eng_key != key) else max(N, 2 * eng._max_frames)
            if eng is not None:
                eng.close()
            eng = BatchEngine(F, nmf_rank, decoder_params_from_state(sd), precision=self.precision, device=dev, max_frames=cap, max_utts=1, z_dim=L)
            self._enc, self._eng_key = _encoder_params(sd), key
        eng.bind([N], Rcap=max(ns_e, ns_w))
        self._eng, self._N, self._F, self._K = eng, N, F, nmf_rank
        # draw order of the reference: rand(F,K), rand(K,N) (mcem.py:42-43)
        W0 = torch.max(torch.rand(F, nmf_rank), eps * torch.ones(F, nmf_rank)).numpy()
        H0 = torch.max(torch.rand(nmf_rank, N), eps * torch.ones(nmf_rank, N)).numpy()
        self.X = X.T                                                           # mcem.py:46
        eng.set_spectrogram([np.asarray(X, dtype=np.complex64)])
        eng.init_nmf([W0], [H0])
        yy = None
        if y is not None:
            yy = torch.as_tensor(y, dtype=torch.float32).to(dev).reshape(N, -1).contiguous()
            eng.set_labels(yy)
            self.y = torch.t(yy)                                               # mcem.py:213
        torch.randn(N, L)                      # the encoder's reparametrisation draw (models.py:10), unused
        eng.encode(self._enc, yy)
        self._call = 0
        self.Vs = self.Vs_scaled = self.Vx = None

    # ---- state views in the reference's shapes --------------------------------
    @property
    def X_abs_2(self):
        return self._eng.X2[:, :self._F].T
    @property
    def W(self):
        return self._eng.W[0, :self._F, :self._K]
    @property
    def H(self):
        return self._eng.Ht[:, :self._K].T
    @property
    def g(self):
        return self._eng.g
    @property
    def Z(self):
        return self._eng.Z[:, :self._L].T
    @property
    def Vb(self):
        return self._eng.Vb(0)

    def compute_Vs_scaled(self):
        self.Vs_scaled = self.g * self.Vs

    def compute_Vx(self):
        self.Vx = self.Vs_scaled + self.Vb

    def _refresh(self, R):
        """Materialise (R,F,N) views of the sample variances (not needed by run())."""
        Vs = self._eng.decode(R)                                               # [N,R,Fs]
        self.Vs = Vs[:, :, :self._F].permute(1, 2, 0)
        self.compute_Vs_scaled()
        self.compute_Vx()

    def _chain(self, nsamples, burnin, want_acc=False, update_Z=True):
        eng, N = self._eng, self._N
        eps = u = None
        if self.rng == "replay":
            S = nsamples + burnin
            L = self._L
            e = torch.zeros(S, N, LAT)                                         # (latent dim 16: columns 16..31 stay zero)
            uu = torch.empty(S, N)
            for m in range(S):                                                 # mcem.py:407, :420
                e[m, :, :L] = torch.randn(L, N).T
                uu[m] = torch.rand(N)
            eps, u = e.to(self.device), uu.to(self.device)
        acc = eng.mh_chain(nsamples, burnin, self.var_RW, call=self._call, eps=eps, u=u, want_acc=want_acc,
                           update_Z=update_Z)
        self._call += 1
        return acc

    def E_step(self):
        ns, bi = self.e_step_counts()
        self._chain(ns, bi)
        self._R = ns

    def M_step(self):
        self._eng.m_step(self._R)

    def compute_expected_neg_log_like(self):
        return float(self._eng.cost_from_frames(self._R)[0])

    def compute_WF(self, sample=False):
        if sample:
            ns, bi = self.wf_counts()
            self._chain(ns, bi, update_Z=False)                                # mcem.py:477-478: self.Z untouched
            self._R = ns
        S, Nn, WFs, WFn = self._eng.wiener(self._R, want_masks=True)
        self._S_dev, self._N_dev = S, Nn
        return WFs[:, :self._F].T, WFn[:, :self._F].T

    def run(self):
        cost = np.zeros(self.niter)
        if self.rng == "device":
            ns, bi = self.e_step_counts()
            nw, bw = self.wf_counts()
            c, S, Nn = self._eng.run(self.niter, ns, bi, nw, bw, self.var_RW, store=bool(self.fused_store))
            cost[:] = c[0].cpu().numpy()
            self._R = nw
        else:
            for n in range(self.niter):                                        # mcem.py:159-165
                self.E_step()
                self.M_step()
                cost[n] = self.compute_expected_neg_log_like()
            self.compute_WF(sample=True)                                       # mcem.py:173
            S, Nn = self._S_dev, self._N_dev
        F = self._F
        to_c = lambda t: np.ascontiguousarray(t[:, :F].cpu().numpy()).view(np.complex64).reshape(self._N, F).T
        self.S_hat = to_c(S)                                                   # mcem.py:175
        self.N_hat = to_c(Nn)                                                  # mcem.py:176
        return cost


class MCEM_M1(_MCEM):
    model = "M1"

    def init_parameters(self, X, vae, nmf_rank, eps, device):
        self._init(X, None, vae, nmf_rank, eps, device)


class MCEM_M2(_MCEM):
    model = "M2"

    def init_parameters(self, X, y, vae, nmf_rank, eps, device):
        self._init(X, y, vae, nmf_rank, eps, device)


class EM_noNMF(_MCEM):
    """Reference surface of EM_noNMF (mcem.py:493-604): "meant to be an abstract class that should not be instantiated
    but only inherited" -- constructor (X, Vb, g, vae, niter, device), fixed noise variance Vb, gains-only M-step
    (:543-578), run() (:580-604).  The sampler comes from the subclass (MCEM_M2_noNMF)."""
    model = "M2"

    def __init__(self, X, Vb, g, vae, niter=100, device="cpu", nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25,
                 burnin_WF=75, var_RW=0.01, rng="replay", precision="bf16x3", fused_store=True):
        super().__init__(niter, nsamples_E_step, burnin_E_step, nsamples_WF, burnin_WF, var_RW, rng=rng, precision=precision,
                         fused_store=fused_store)
        dev = torch.device(device if device not in (None, "cpu") else "cuda:0")
        N, F = X.shape
        sd = _state(vae)
        ns_e, _ = self.e_step_counts()
        ns_w, _ = self.wf_counts()
        self.device, self.vae = dev, vae
        self._L = getattr(vae, "latent_dim", None) or getattr(vae, "z_dim")
        eng = BatchEngine(F, 1, decoder_params_from_state(sd), precision=precision, device=dev, max_frames=N, max_utts=1, z_dim=self._L)
        eng.bind([N], Rcap=max(ns_e, ns_w))
        self._eng, self._N, self._F, self._K = eng, N, F, 1
        self.X = X.T                                                           # mcem.py:503
        eng.set_spectrogram([np.asarray(X, dtype=np.complex64)])
        vb = torch.zeros(N, eng.Fs, dtype=torch.float32)
        vb[:, :F] = torch.as_tensor(np.asarray(Vb), dtype=torch.float32)
        eng.set_noise_psd(vb)                                                  # mcem.py:507
        eng.g.copy_(torch.as_tensor(g, dtype=torch.float32).reshape(N))        # mcem.py:508
        self._call = 0

    @property
    def Vb(self):
        return self._eng._Vb_ext[:, :self._F].T


class MCEM_M2_noNMF(EM_noNMF):
    """Reference surface of MCEM_M2_noNMF (mcem.py:606-760): constructor-style, the noise variance Vb is given and
    fixed, only the gains are updated.  X complex (N,F), Vb (N,F), g (N,), Z (N,L), y (N,Dy).  rng="device" runs the
    fused driver (vaenmf_em_run serves the fixed-noise model too: the chain reads the given Vb, the M-step is gains-only)."""

    def __init__(self, X, Vb, g, Z, y, vae, niter, device, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25,
                 burnin_WF=75, var_RW=0.01, rng="replay", precision="bf16x3", fused_store=True):
        if type(vae).__name__ == "RVAE":
            raise NameError("MCEM algorithm only valid for FFNN VAE")          # mcem.py:614-615
        super().__init__(X, Vb, g, vae, niter, device, nsamples_E_step, burnin_E_step, nsamples_WF, burnin_WF, var_RW,
                         rng=rng, precision=precision, fused_store=fused_store)
        eng, N = self._eng, self._N
        eng.Z.zero_()
        eng.Z[:, :self._L].copy_(torch.as_tensor(Z, dtype=torch.float32).reshape(N, self._L))   # mcem.py:617
        yy = torch.as_tensor(y, dtype=torch.float32).to(self.device).reshape(N, -1).contiguous()
        eng.set_labels(yy)
        self.y = torch.t(yy)                                                   # mcem.py:618
