"""PyTorch-CPU restatement of the reconstruct loop (EM.run of python/models/mcem.py) -- the CPU BASELINE.

THIS IS TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path never does and has no CPU fallback.

Why it exists beside ``vaenmf_oracle.py`` (numpy): SURVEY 8(d) asks for the CPU baseline to be the reference's
own kind of program -- float32 torch tensors on the host cores, dense GEMMs through torch's BLAS, the same
sequence of tensor operations as mcem.py -- because the reference's Python cannot travel to the GPU box.  This
file is that program, written from the maths (mcem.py:36-178 EM base, :348-490 MCEM_M1, :181-345 MCEM_M2;
models.py:90-121 encoder / decoder), and pinned by the same reference-recorded golden runs as the numpy oracle
(tests/test_oracle_golden.py::test_torch_cpu_restatement_*).  Every method cites the reference lines it follows.
"""
import numpy as np
import torch


class ReplayDraws:
    """Hands out recorded draws (float32 numpy arrays) in order: the parity runs."""

    def __init__(self, draws):
        self.draws, self.pos = list(draws), 0

    def _next(self, shape):
        a = torch.from_numpy(np.ascontiguousarray(self.draws[self.pos], dtype=np.float32))
        self.pos += 1
        assert tuple(a.shape) == tuple(shape), (tuple(a.shape), tuple(shape))
        return a

    def rand(self, *shape):
        return self._next(shape)

    def randn(self, *shape):
        return self._next(shape)


class TorchDraws:
    """torch's own CPU generator (the timed baseline runs)."""

    def __init__(self, seed=0):
        self.g = torch.Generator()
        self.g.manual_seed(seed)

    def rand(self, *shape):
        return torch.rand(*shape, generator=self.g)

    def randn(self, *shape):
        return torch.randn(*shape, generator=self.g)


def _lin(p, name, x):
    return torch.addmm(p[name + ".bias"], x, p[name + ".weight"].t())


class TorchMCEM:
    """M1 / M2 Monte-Carlo EM on float32 CPU tensors.  params: state_dict layout of the reference's
    VariationalAutoencoder / DeepGenerativeModel (numpy or torch values)."""

    def __init__(self, model, niter, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75,
                 var_RW=0.01, reference_compat=True):
        assert model in ("M1", "M2")
        self.model, self.niter, self.var_RW = model, int(niter), float(var_RW)
        if model == "M1" and reference_compat:      # positional shift of mcem.py:461-462 / :477-478 against :371
            self.e_counts, self.wf_counts = (burnin_E_step, 30), (burnin_WF, 30)
        else:
            self.e_counts, self.wf_counts = (nsamples_E_step, burnin_E_step), (nsamples_WF, burnin_WF)

    # ---- models.py:107-121 / :90-104
    def _n_hidden(self, prefix):
        n = 0
        while "%s.hidden.%d.weight" % (prefix, n) in self.p:
            n += 1
        return n

    def decode(self, zin):                       # (M, L+Dy) -> (M, F)
        h = zin
        for i in range(self._n_hidden("decoder")):
            h = torch.tanh(_lin(self.p, "decoder.hidden.%d" % i, h))
        return torch.exp(_lin(self.p, "decoder.reconstruction", h))

    def encode_mean(self, x):
        h = x
        for i in range(self._n_hidden("encoder")):
            h = torch.tanh(_lin(self.p, "encoder.hidden.%d" % i, h))
        return _lin(self.p, "encoder.sample.mu", h)

    def _dec_cols(self, Z):                      # (L,N) -> (F,N): decoder(cat([Z,y]).T).T, mcem.py:392 / :242
        zin = Z if self.y is None else torch.cat([Z, self.y], 0)
        return self.decode(zin.t()).t()

    # ---- mcem.py:36-57, :361-369, :207-216
    def init_parameters(self, X, params, nmf_rank, eps, rng, y=None):
        self.p = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))) for k, v in params.items()}
        self.rng = rng
        N, F = X.shape
        self.W = torch.clamp_min(rng.rand(F, nmf_rank), eps)            # :42
        self.H = torch.clamp_min(rng.rand(nmf_rank, N), eps)            # :43
        self.g = torch.ones(N)                                          # :51
        self.X = np.asarray(X).T                                        # (F,N) complex64, :46
        self.X2 = torch.from_numpy((np.abs(self.X) ** 2).astype(np.float32))   # :47
        self.Vb = self.W @ self.H                                       # :82
        self.y = None if y is None else torch.as_tensor(np.asarray(y, np.float32)).t().contiguous()   # (Dy,N) :213
        enc_in = self.X2.t() if self.y is None else torch.cat([self.X2, self.y], 0).t()
        L = self.p["encoder.sample.mu.weight"].shape[0]
        rng.randn(N, L)                                                 # the reparametrisation draw (models.py:10), unused
        self.Z = self.encode_mean(enc_in).t().contiguous()              # (L,N) posterior mean, :367-368
        self.L = L

    # ---- mcem.py:371-441 / :218-294
    def sample_posterior(self, Z, nsamples, burnin):
        F, N = self.X2.shape
        sd = float(np.sqrt(np.float32(self.var_RW)))
        Zs = torch.zeros(N, nsamples, self.L)
        Z_t = Z.clone()
        Vx_t = self.g * self._dec_cols(Z_t) + self.Vb
        k = 0
        for m in range(nsamples + burnin):
            Zp = Z_t + sd * self.rng.randn(self.L, N)                   # :407
            Vxp = self.g * self._dec_cols(Zp) + self.Vb                 # :410-412
            acc = torch.sum(torch.log(Vx_t) - torch.log(Vxp) + (1.0 / Vx_t - 1.0 / Vxp) * self.X2, 0) \
                + 0.5 * torch.sum(Z_t ** 2 - Zp ** 2, 0)                # :415-417
            ok = torch.log(self.rng.rand(N)) < acc                      # :420
            Z_t[:, ok] = Zp[:, ok]                                      # :429
            Vx_t = self.g * self._dec_cols(Z_t) + self.Vb               # :432-433
            if m >= burnin:                                             # :435-437
                Zs[:, k, :] = Z_t.t()
                k += 1
        return Zs

    # ---- mcem.py:444-454 / :297-307, :75-79
    def compute_Vs(self, Zs):
        N, R, _ = Zs.shape
        zin = Zs if self.y is None else torch.cat([Zs, self.y.t()[:, None, :].expand(N, R, self.y.shape[0])], 2)
        self.Vs = self.decode(zin.reshape(N * R, -1)).reshape(N, R, -1).permute(1, 2, 0).contiguous()   # (R,F,N)

    def _refresh(self):
        self.Vs_scaled = self.g * self.Vs
        self.Vx = self.Vs_scaled + self.Vb

    def E_step(self):                                                    # :456-471 / :309-325
        Zs = self.sample_posterior(self.Z, *self.e_counts)
        self.Z = Zs[:, -1, :].t().contiguous()
        self.compute_Vs(Zs)
        self._refresh()

    def M_step(self):                                                    # :90-152
        X2 = self.X2
        iv = 1.0 / self.Vx
        self.W = self.W * torch.sqrt(((X2 * (iv * iv).sum(0)) @ self.H.t()) / (iv.sum(0) @ self.H.t()))     # :107-110
        self.Vb = self.W @ self.H
        self.Vx = self.Vs_scaled + self.Vb                               # :113-114
        iv = 1.0 / self.Vx
        self.H = self.H * torch.sqrt((self.W.t() @ (X2 * (iv * iv).sum(0))) / (self.W.t() @ iv.sum(0)))    # :118-121
        self.Vb = self.W @ self.H
        self.Vx = self.Vs_scaled + self.Vb                               # :124-125
        nrm = self.W.abs().sum(0)                                        # :129
        self.W = self.W / nrm[None, :]
        self.H = self.H * nrm[:, None]                                   # :131-133
        iv = 1.0 / self.Vx
        self.g = self.g * torch.sqrt((X2 * (self.Vs * iv * iv).sum(0)).sum(0) / (self.Vs * iv).sum(0).sum(0))   # :138-142
        self._refresh()                                                  # :151-152

    def cost(self):                                                      # :68-70
        return float(torch.mean(torch.log(self.Vx) + self.X2 / self.Vx))

    def compute_WF(self):                                                # :473-490 / :327-345 (sample=True)
        Zs = self.sample_posterior(self.Z, *self.wf_counts)
        self.compute_Vs(Zs)
        self._refresh()
        return torch.mean(self.Vs_scaled / self.Vx, 0), torch.mean(self.Vb / self.Vx, 0)

    def run(self):                                                       # :155-178
        cost = np.zeros(self.niter)
        with torch.no_grad():
            for n in range(self.niter):
                self.E_step()
                self.M_step()
                cost[n] = self.cost()
            WFs, WFn = self.compute_WF()
        self.S_hat = WFs.numpy() * self.X                                # :175
        self.N_hat = WFn.numpy() * self.X                                # :176
        return cost
