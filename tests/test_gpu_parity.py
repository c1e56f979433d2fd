"""GPU parity tests: the HIP path (through the C ABI, libvaenmf.so) against the numpy
oracle and the golden vectors generated from the reference.

Stated tolerances (float32 unless noted):
  * decoder variances Vs, bf16x3 mode: 2e-4 relative (split-bf16 MFMA products carry
    ~2^-17 relative error per product; fp32 reference ~2^-24);  bf16 mode: 5e-2.
  * MH log-acceptance (mcem.py:415-417): 2e-3 absolute (sum of ~F terms).
  * W, H, g after an M-step from identical samples: 5e-4 relative; cost: 1e-4 relative.
  * trajectories with replayed noise: decisions identical on the golden cases (they were
    chosen with decision margins >= 1e-3), final S_hat 2e-3 relative (L2).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vaenmf_oracle as orc
from helpers import load_case, rel_err, nrm_err, GOLDEN


def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def dec_list(params):
    if "decoder.hidden.1.weight" not in params:          # one hidden layer (h_dim = [128])
        return [params["decoder.hidden.0.weight"], params["decoder.hidden.0.bias"], params["decoder.reconstruction.weight"], params["decoder.reconstruction.bias"]]
    return [params["decoder.hidden.0.weight"], params["decoder.hidden.0.bias"], params["decoder.hidden.1.weight"],
            params["decoder.hidden.1.bias"], params["decoder.reconstruction.weight"], params["decoder.reconstruction.bias"]]


def make_engine(params, F, K, counts_N, Rcap, precision="bf16x3", seeds=None):
    from vaenmf.engine import BatchEngine
    z_dim = int(params["encoder.sample.mu.weight"].shape[0]) if "encoder.sample.mu.weight" in params else 32
    eng = BatchEngine(F, K, dec_list(params), precision=precision, max_frames=sum(counts_N), max_utts=len(counts_N), z_dim=z_dim)
    eng.bind(counts_N, Rcap=Rcap, seeds=seeds)
    return eng


@pytest.mark.parametrize("F,precision,tol", [(65, "bf16x3", 2e-4), (257, "bf16x3", 2e-4), (513, "bf16x3", 2e-4),
                                             (257, "bf16", 5e-2)])
def test_decoder_forward(F, precision, tol):
    """compute_Vs (mcem.py:444-454): Vs = decoder(Z_samples), ragged batch, R not a multiple of 16."""
    need_gpu()
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=3, bias_std=0.1)
    counts, R = [5, 33, 17], 21
    eng = make_engine(params, F, 4, counts, Rcap=24, precision=precision)
    g = np.random.default_rng(0)
    Zs = g.standard_normal((sum(counts), 24, 32)).astype(np.float32)
    eng.Zs.copy_(torch.from_numpy(Zs))
    Vs = eng.decode(R).cpu().numpy()
    ref = orc.decoder_forward(params, Zs[:, :R].reshape(-1, 32)).reshape(sum(counts), R, F)
    assert rel_err(Vs[:, :, :F], ref) < tol
    assert np.all(Vs[:, :, F:] == 0)


def test_decoder_forward_m2_label_bias():
    """decoder(cat([Z, y])) (mcem.py:242): the label half folded into a per-frame bias."""
    need_gpu()
    F, Dy = 65, 65
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=4, y_dim=Dy, bias_std=0.1)
    counts, R = [19, 40], 10
    eng = make_engine(params, F, 4, counts, Rcap=R)
    g = np.random.default_rng(1)
    NT = sum(counts)
    Zs = g.standard_normal((NT, R, 32)).astype(np.float32)
    y = (g.random((NT, Dy)) > 0.5).astype(np.float32)
    eng.Zs.copy_(torch.from_numpy(Zs))
    eng.set_labels(torch.from_numpy(y))
    Vs = eng.decode(R).cpu().numpy()
    zin = np.concatenate([Zs, np.broadcast_to(y[:, None, :], (NT, R, Dy))], 2)
    ref = orc.decoder_forward(params, zin.reshape(NT * R, -1)).reshape(NT, R, F)
    assert rel_err(Vs[:, :, :F], ref) < 2e-4


def setup_from_case(name, model, precision="bf16x3"):
    z, params, draws, meta = load_case(name)
    nsE, biE, nsW, biW = meta["counts"]
    o = orc.MCEMOracle(model, meta["niter"], nsE, biE, nsW, biW, 0.01, reference_compat=True)
    rng = orc.ReplayRNG(draws)
    y = z["y"] if model == "M2" else None
    o.init_parameters(z["X"], params, meta["K"], 1e-8, rng, y=y)
    ns, bi = o.e_step_counts()
    nw, bw = o.wf_counts()
    eng = make_engine(params, meta["F"], meta["K"], [meta["N"]], Rcap=max(ns, nw), precision=precision)
    eng.set_spectrogram([z["X"]])
    eng.init_nmf([z["W0"]], [z["H0"]])
    if y is not None:
        eng.set_labels(torch.from_numpy(y))
    eng.Z.zero_()
    eng.Z[:, :meta["L"]].copy_(torch.from_numpy(np.ascontiguousarray(z["Z0"].T)))
    return z, params, meta, o, rng, eng


def replay_buffers(rng, S, N, L, dev):
    """Take the next S (randn(L,N), rand(N)) pairs of the recorded stream (a 16-dimensional latent space: the recorded
    draws fill columns 0..15 of the engine's 32-wide rows, the padding columns stay zero)."""
    eps = np.zeros((S, N, 32), np.float32)
    u = np.empty((S, N), np.float32)
    for m in range(S):
        eps[m, :, :rng.draws[rng.pos].shape[0]] = rng.draws[rng.pos].T
        u[m] = rng.draws[rng.pos + 1]
        rng.pos += 2
    return torch.from_numpy(eps).to(dev), torch.from_numpy(u).to(dev)


CASES = [("m1_f65", "M1"), ("m2_vad_f65", "M2"), ("m2_ibm_f65", "M2"), ("m1_f257", "M1"),
         # decoder shapes beside 32 -> 128 -> 128 -> F that the reference's scripts list (scripts/evaluate_M1.py:44-85): latent
         # dimension 16 (zero-padded first layer, no random walk on the padding), ONE hidden layer (layer 2 skipped), both + M2
         ("m1_f65_z16", "M1"), ("m1_f65_h128", "M1"), ("m2_vad_f65_z16_h128", "M2")]


@pytest.mark.parametrize("name,model", CASES)
def test_encoder_init(name, model):
    """Z = mu_enc(|X|^2 [cat y]) (mcem.py:367-368 / :214-215) via vaenmf_dense."""
    need_gpu()
    z, params, meta, o, rng, eng = setup_from_case(name, model)
    enc = [(params["encoder.hidden.%d.weight" % i], params["encoder.hidden.%d.bias" % i]) for i in range(2) if "encoder.hidden.%d.weight" % i in params]
    enc.append((params["encoder.sample.mu.weight"], params["encoder.sample.mu.bias"]))
    y = torch.from_numpy(z["y"]).to(eng.device) if model == "M2" else None
    eng.encode(enc, y)
    L = meta["L"]
    assert np.max(np.abs(eng.Z[:, :L].cpu().numpy().T - z["Z0"])) < 2e-5 and float(eng.Z[:, L:].abs().max() if L < 32 else 0.0) == 0.0
    assert rel_err(eng.X2[:, :meta["F"]].cpu().numpy().T, o.X_abs_2) < 1e-6


@pytest.mark.parametrize("name,model", CASES)
def test_first_em_iteration(name, model):
    """E-step chain (every log-acceptance, every decision, the samples), then the M-step,
    against the reference's recorded first iteration."""
    need_gpu()
    z, params, meta, o, rng, eng = setup_from_case(name, model)
    ns, bi = o.e_step_counts()
    S, N, F, K = ns + bi, meta["N"], meta["F"], meta["K"]
    pos0 = rng.pos
    eps, u = replay_buffers(rng, S, N, 32, eng.device)
    acc = eng.mh_chain(ns, bi, 0.01, eps=eps, u=u, want_acc=True).cpu().numpy()
    ref_acc = z["acc"][:S]
    assert np.max(np.abs(acc - ref_acc)) < 2e-3
    dec_gpu = np.log(u.cpu().numpy()) < acc
    dec_ref = np.log(u.cpu().numpy()) < ref_acc
    assert np.array_equal(dec_gpu, dec_ref)
    # oracle chain on the same draws
    rng.pos = pos0
    Zs_ref = o.sample_posterior(o.Z, ns, bi)
    L = meta["L"]
    assert np.max(np.abs(eng.Zs[:, :ns, :L].cpu().numpy() - Zs_ref)) < 5e-6   # fma contraction of z + sd*eps
    assert L == 32 or float(eng.Zs[:, :ns, L:].abs().max()) == 0.0         # the padding latents never move
    assert np.max(np.abs(eng.Z[:, :L].cpu().numpy().T - z["E1_Z"])) < 1e-5
    Vs = eng.decode(ns).cpu().numpy()[:, :, :F]                    # [N,R,F]
    assert rel_err(np.moveaxis(Vs, 0, -1), z["E1_Vs"]) < 2e-4
    # M-step
    eng.m_step(ns)
    W = eng.W[0, :F, :K].cpu().numpy()
    H = eng.Ht[:, :K].cpu().numpy().T
    assert rel_err(W, z["M1_W"]) < 5e-4
    assert rel_err(H, z["M1_H"]) < 5e-4
    assert rel_err(eng.g.cpu().numpy(), z["M1_g"]) < 5e-4
    assert rel_err(eng.Vb(0).cpu().numpy(), z["M1_Vb"]) < 5e-4
    cost = eng.cost_from_frames(ns)[0]
    assert abs(cost - z["cost"][0]) / abs(z["cost"][0]) < 1e-4
    # padding stays clean
    assert float(eng.W[0, F:].abs().max()) == 0 if eng.Fs > F else True
    assert float(eng.Ht[:, K:].abs().max()) == 0 if eng.Kp > K else True


@pytest.mark.parametrize("name,model", CASES)
def test_full_run_replay(name, model):
    """EM.run (mcem.py:155-178) step by step with the reference's recorded noise."""
    need_gpu()
    z, params, meta, o, rng, eng = setup_from_case(name, model)
    ns, bi = o.e_step_counts()
    nw, bw = o.wf_counts()
    N, F = meta["N"], meta["F"]
    cost = np.zeros(meta["niter"])
    for it in range(meta["niter"]):
        eps, u = replay_buffers(rng, ns + bi, N, 32, eng.device)
        eng.mh_chain(ns, bi, 0.01, eps=eps, u=u)
        eng.m_step(ns)
        cost[it] = eng.cost_from_frames(ns)[0]
    eps, u = replay_buffers(rng, nw + bw, N, 32, eng.device)
    eng.mh_chain(nw, bw, 0.01, eps=eps, u=u, update_Z=False)
    S, Nn, WFs, WFn = eng.wiener(nw, want_masks=True)
    assert rng.pos == len(rng.draws)
    assert np.max(np.abs(cost - z["cost"]) / np.abs(z["cost"])) < 2e-4
    to_c = lambda t: np.ascontiguousarray(t[:, :F].cpu().numpy()).view(np.complex64).reshape(N, F).T
    assert nrm_err(to_c(S), z["S_hat"]) < 2e-3
    assert nrm_err(to_c(Nn), z["N_hat"]) < 2e-3
    assert rel_err(WFs[:, :F].cpu().numpy().T, z["WFs"]) < 5e-3
    assert np.max(np.abs(eng.Z[:, :meta["L"]].cpu().numpy().T - z["Z"])) < 1e-5
    assert rel_err(eng.W[0, :F, :meta["K"]].cpu().numpy(), z["W"]) < 2e-3
    assert rel_err(eng.g.cpu().numpy(), z["g"]) < 2e-3


def test_mirror_classes_match_reference_surface():
    """MCEM_M1 drop-in surface (mcem.py:350-369, :155-178) with rng='replay': draws come
    from torch's global generator in the reference's order."""
    need_gpu()
    import vaenmf
    z, params, draws, meta = load_case("m1_f65")
    nsE, biE, nsW, biW = meta["counts"]
    vae = vaenmf.VariationalAutoencoder([meta["F"], meta["L"], [128, 128]])
    vae.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    m = vaenmf.MCEM_M1(niter=meta["niter"], nsamples_E_step=nsE, burnin_E_step=biE, nsamples_WF=nsW, burnin_WF=biW,
                       var_RW=0.01)
    torch.manual_seed(int(z["seed"]))          # the seed the golden script used for the reference run
    m.init_parameters(X=z["X"], vae=vae, nmf_rank=meta["K"], eps=1e-8, device="cuda:0")
    assert rel_err(m.W.cpu().numpy(), z["W0"]) == 0        # same generator, same draw order
    cost = m.run()
    assert cost.dtype == np.float64 and cost.shape == (meta["niter"],)
    assert np.max(np.abs(cost - z["cost"]) / np.abs(z["cost"])) < 2e-4
    assert m.S_hat.dtype == np.complex64 and m.S_hat.shape == (meta["F"], meta["N"])
    assert nrm_err(m.S_hat, z["S_hat"]) < 2e-3
    assert nrm_err(m.N_hat, z["N_hat"]) < 2e-3
    with pytest.raises(NameError):
        RVAE = type("RVAE", (), {})
        m.init_parameters(X=z["X"], vae=RVAE(), nmf_rank=4, eps=1e-8, device="cuda:0")


def test_device_rng_streams():
    """On-device generator: (i) rng_fill is reproducible and batching-independent,
    (ii) moments are right, (iii) a REPLAY run fed with rng_fill's buffers is
    bit-identical to the DEVICE run."""
    need_gpu()
    z, params, draws, meta = load_case("m1_f257")
    F, K = meta["F"], meta["K"]
    counts = [40, 70, 33]
    seeds = [11, 22, 33]
    g = np.random.default_rng(5)
    Xs = [(g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))).astype(np.complex64) for n in counts]

    def fresh(cnts, sds, xs):
        eng = make_engine(params, F, K, cnts, Rcap=8, seeds=sds)
        eng.set_spectrogram(xs)
        eng.init_nmf([np.maximum(g2.random((F, K)), 1e-8).astype(np.float32) for _ in cnts],
                     [np.maximum(g2.random((K, n)), 1e-8).astype(np.float32) for n in cnts])
        return eng

    g2 = np.random.default_rng(6)
    eng = fresh(counts, seeds, Xs)
    S = 20
    eps, u = eng.rng_fill(3, S)
    e = eps.cpu().numpy()
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1) < 0.01 and abs(u.cpu().numpy().mean() - 0.5) < 0.02
    assert abs((e ** 4).mean() - 3.0) < 0.15
    # batching independence: utterance 1 alone gets the same streams
    g2 = np.random.default_rng(6)
    eng1 = fresh([counts[1]], [seeds[1]], [Xs[1]])
    eps1, u1 = eng1.rng_fill(3, S)
    sl = eng.utt_slice(1)
    assert torch.equal(eps[:, sl], eps1) and torch.equal(u[:, sl], u1)
    # device == replay(rng_fill)
    Z0 = eng.Z.clone()
    eng.mh_chain(8, 12, 0.01, call=3)
    Zs_dev, Z_dev = eng.Zs.clone(), eng.Z.clone()
    eng.Z.copy_(Z0)
    eng.mh_chain(8, 12, 0.01, eps=eps, u=u)
    assert torch.equal(eng.Zs, Zs_dev) and torch.equal(eng.Z, Z_dev)
    assert float((Z_dev - Z0).abs().max()) > 0      # chains moved


def test_fused_run_equals_stepwise_and_batches_are_independent():
    """vaenmf_em_run (no host sync) == the same kernels called step by step; an utterance
    gives the same result alone and inside a ragged batch (device RNG keyed per utterance)."""
    need_gpu()
    z, params, draws, meta = load_case("m1_f257")
    F, K = meta["F"], meta["K"]
    counts, seeds = [37, 64, 50], [5, 6, 7]
    g = np.random.default_rng(8)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (1 + 3 * np.exp(-np.arange(F) / 40.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    enc = [(params["encoder.hidden.0.weight"], params["encoder.hidden.0.bias"]),
           (params["encoder.hidden.1.weight"], params["encoder.hidden.1.bias"]),
           (params["encoder.sample.mu.weight"], params["encoder.sample.mu.bias"])]

    def prep(idx):
        eng = make_engine(params, F, K, [counts[i] for i in idx], Rcap=12, seeds=[seeds[i] for i in idx])
        eng.set_spectrogram([Xs[i] for i in idx])
        eng.init_nmf([W0[i] for i in idx], [H0[i] for i in idx])
        eng.encode(enc)
        return eng

    niter, nsE, biE, nsW, biW = 3, 6, 5, 12, 7
    eng = prep([0, 1, 2])
    cost, S, N = eng.run(niter, nsE, biE, nsW, biW, 0.01, store=False)
    cost = cost.cpu().numpy()
    assert np.all(np.isfinite(cost)) and np.all(np.isfinite(S.cpu().numpy()))
    # step by step
    eng2 = prep([0, 1, 2])
    c2 = np.zeros((3, niter))
    for it in range(niter):
        eng2.mh_chain(nsE, biE, 0.01, call=it)
        eng2.m_step(nsE)
        c2[:, it] = eng2.cost_from_frames(nsE)
    eng2.mh_chain(nsW, biW, 0.01, call=niter, update_Z=False)
    S2, N2, _, _ = eng2.wiener(nsW)
    assert torch.equal(S, S2) and torch.equal(N, N2)
    assert np.max(np.abs(c2 - cost) / np.abs(cost)) < 1e-12
    # utterance 1 alone
    eng3 = prep([1])
    cost3, S3, N3 = eng3.run(niter, nsE, biE, nsW, biW, 0.01, store=False)
    sl = eng.utt_slice(1)
    assert torch.equal(S[sl], S3) and np.array_equal(cost[1], cost3.cpu().numpy()[0])
    # the same with the sample-variance store (the default of run()): fused == step-wise stored calls, bit for bit
    eng4 = prep([0, 1, 2])
    cost4, S4, N4 = eng4.run(niter, nsE, biE, nsW, biW, 0.01)
    from vaenmf import _lib
    assert _lib.lib().vaenmf_plan_query(eng4._plan, _lib.Q_MSTEP_PATH) == 1
    eng5 = prep([0, 1, 2])
    eng5.sample_store(True)
    c5 = np.zeros((3, niter))
    for it in range(niter):
        eng5.mh_chain(nsE, biE, 0.01, call=it)
        eng5.m_step_stored()
        c5[:, it] = eng5.cost_from_frames(nsE)
    eng5.mh_chain(nsW, biW, 0.01, call=niter, update_Z=False)
    S5, N5, _, _ = eng5.wiener_stored()
    assert torch.equal(S4, S5) and torch.equal(N4, N5)
    assert np.max(np.abs(c5 - cost4.cpu().numpy()) / np.abs(c5)) < 1e-12


def test_stft_istft_and_metrics():
    """stft.py:16-63 / :66-102 on the GPU vs the oracle, on reference-committed audio;
    metrics.py:39-60 known answer (-6.2 / -4.3 / -1.9 dB)."""
    need_gpu()
    from vaenmf import stft as vstft, metrics as vmet
    zz = np.load(GOLDEN + "/metrics_dummy_m2.npz")
    x = zz["a_s"] / 32768.0
    for wlen in (64e-3, 32e-3):
        X = vstft.stft(x, fs=16000, wlen_sec=wlen, hop_percent=0.25)
        Xo = orc.stft(x, fs=16000, wlen_sec=wlen, hop_percent=0.25)
        assert X.shape == Xo.shape and X.dtype == np.complex64
        assert np.max(np.abs(X - Xo)) < 2e-6 * np.max(np.abs(Xo))
        xr = vstft.istft(X, fs=16000, wlen_sec=wlen, hop_percent=0.25, max_len=len(x))
        assert xr.dtype == np.float32 and len(xr) == len(x)
        assert np.max(np.abs(xr - x)) < 2e-6
        xo = orc.istft(Xo, fs=16000, wlen_sec=wlen, hop_percent=0.25, max_len=len(x) + 700)
        xr2 = vstft.istft(Xo, fs=16000, wlen_sec=wlen, hop_percent=0.25, max_len=len(x) + 700)
        assert np.max(np.abs(xr2 - xo)) < 2e-6
    with pytest.raises(ValueError):
        vstft.stft(x, fs=16000, wlen_sec=50.01e-3)
    r = vmet.energy_ratios(zz["a_s_est"] / 32768.0, zz["a_s"] / 32768.0, zz["a_n"] / 32768.0)
    assert np.allclose(r, zz["a_ratios"], atol=1e-6)


def test_stft_against_reference_known_answer():
    """The HIP STFT against the reference's OWN committed output (data/subset/pickle/CSR-1-WSJ-0/si_et_05_frames.p,
    |stft(x)|^2 of its raw utterance 440c020a; fixture tests/golden/stft_frames.npz, extracted without unpickling):
    first / last 96 frames and every frame's sum over the bins, same bounds as the oracle's CPU test."""
    need_gpu()
    from vaenmf import stft as vstft
    z = np.load(GOLDEN + "/stft_frames.npz")
    x = z["pcm_a"].astype(np.float64) / 32768.0
    x = x[int(0.1 * 16000):]
    x = x / np.max(np.abs(x))                                       # tests/dataset/test_csr1_wjs0_dataset.py:40-47
    X = vstft.stft(x, fs=16000, wlen_sec=64e-3, win="hann", hop_percent=0.25)
    P = np.power(np.abs(X), 2)
    n0 = int(z["frame_counts"][0])
    assert P.shape == (513, n0)
    scale = np.max(z["head"])
    assert np.max(np.abs(P[:, :96] - z["head"])) < 2e-7 * scale and np.max(np.abs(P[:, n0 - 96:] - z["tail"])) < 2e-7 * scale
    big = z["head"] > 1e-6 * scale
    assert np.max(np.abs(P[:, :96][big] / z["head"][big] - 1)) < 1e-5
    assert np.max(np.abs(P.sum(0, dtype=np.float64) / z["col_sums"] - 1)) < 1e-6


def test_stft_against_the_reference_train_and_validation_known_answers():
    """The HIP STFT against the reference's two other committed golden outputs (si_tr_s_frames.p, si_dt_05_frames.p;
    tests/golden/stft_frames_tr_dt.npz): two utterances of each set, the bounds of the oracle's CPU test."""
    need_gpu()
    from vaenmf import stft as vstft
    from test_oracle_golden import _check_stft_tr_dt
    _check_stft_tr_dt(lambda x: vstft.stft(x, fs=16000, wlen_sec=64e-3, win="hann", hop_percent=0.25), 1e-5)


@pytest.mark.parametrize("F,K,R,model", [(513, 10, 30, "M1"), (513, 32, 10, "M2"), (257, 32, 30, "M1"), (129, 16, 40, "M1")])
def test_m_step_and_chain_other_shapes(F, K, R, model):
    """M-step (mcem.py:90-152) + cost + one MH chain + Wiener filter at the shapes the golden runs do not
    cover: F=513 (n_fft 1024, the reference scripts' STFT; one team of 8 waves), NMF rank 10 (the scripts'
    default), 16 and 32 (padded-rank code paths), R > 32 (chunked sample decode), ragged batch of 3."""
    need_gpu()
    Dy = 1 if model == "M2" else 0
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=7, y_dim=Dy, bias_std=0.05)
    counts = [21, 40, 9]
    g = np.random.default_rng(F + K)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (0.5 + 3 * np.exp(-np.arange(F) / 60.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    ys = [(g.random((n, Dy)) > 0.5).astype(np.float32) for n in counts] if Dy else [None] * 3
    eng = make_engine(params, F, K, counts, Rcap=R)
    eng.set_spectrogram(Xs)
    eng.init_nmf(W0, H0)
    if Dy:
        eng.set_labels(torch.from_numpy(np.concatenate(ys)))
    NT = sum(counts)
    Zs = (0.7 * g.standard_normal((NT, R, 32))).astype(np.float32)
    gains = (0.5 + g.random(NT)).astype(np.float32)
    eng.Zs.copy_(torch.from_numpy(Zs))
    eng.g.copy_(torch.from_numpy(gains))
    # oracle objects per utterance, driven from the same state
    oracles = []
    for u, n in enumerate(counts):
        o = orc.MCEMOracle(model, 1)
        o.init_parameters(Xs[u], params, K, 1e-8, orc.NumpyRNG(0), y=ys[u], W0=W0[u], H0=H0[u])
        sl = eng.utt_slice(u)
        o.g = gains[sl].copy()
        o.compute_Vs(Zs[sl]); o.compute_Vs_scaled(); o.compute_Vx()
        oracles.append(o)
    # Wiener filter from the same samples (before the update)
    S, Nn, WFs, WFn = eng.wiener(R, want_masks=True)
    for u, o in enumerate(oracles):
        ws, wn = o.compute_WF(sample=False)
        sl = eng.utt_slice(u)
        assert rel_err(WFs[sl, :F].cpu().numpy().T, ws) < 5e-4 and rel_err(WFn[sl, :F].cpu().numpy().T, wn) < 5e-4
    eng.m_step(R)
    cost = eng.cost_from_frames(R)
    for u, o in enumerate(oracles):
        o.M_step()
        sl = eng.utt_slice(u)
        assert rel_err(eng.W[u, :F, :K].cpu().numpy(), o.W) < 1e-3
        assert rel_err(eng.Ht[sl, :K].cpu().numpy().T, o.H) < 1e-3
        assert rel_err(eng.g[sl].cpu().numpy(), o.g) < 1e-3
        assert abs(cost[u] - o.compute_expected_neg_log_like()) / abs(cost[u]) < 2e-4
    assert float(eng.W[:, F:].abs().max() if eng.Fs > F else 0) == 0
    assert float(eng.Ht[:, K:].abs().max() if eng.Kp > K else 0) == 0
    # one replayed MH chain from this state: log-acceptances against the oracle
    S_steps, ns = 6, 3
    eps = g.standard_normal((S_steps, NT, 32)).astype(np.float32)
    uu = g.random((S_steps, NT)).astype(np.float32)
    Z0 = (0.5 * g.standard_normal((NT, 32))).astype(np.float32)
    eng.Z.copy_(torch.from_numpy(Z0))
    acc = eng.mh_chain(ns, S_steps - ns, 0.01, eps=torch.from_numpy(eps).to(eng.device), u=torch.from_numpy(uu).to(eng.device),
                       want_acc=True).cpu().numpy()
    for u, o in enumerate(oracles):
        sl = eng.utt_slice(u)
        o.W = eng.W[u, :F, :K].cpu().numpy(); o.H = eng.Ht[sl, :K].cpu().numpy().T.copy(); o.g = eng.g[sl].cpu().numpy()
        o.compute_Vb()
        draws = []
        for m in range(S_steps):
            draws += [eps[m, sl].T.copy(), uu[m, sl].copy()]
        o.rng = orc.ReplayRNG(draws)
        tr = []
        Zs_ref = o.sample_posterior(Z0[sl].T.copy(), ns, S_steps - ns, trace=tr)
        ref_acc = np.stack([t["acc"] for t in tr])
        margin = np.abs(np.log(uu[:, sl]) - ref_acc)
        assert np.max(np.abs(acc[:, sl] - ref_acc)) < 3e-3
        if margin.min() > 1e-2:                     # decisions are only comparable away from the threshold
            assert np.max(np.abs(eng.Zs[sl, :ns].cpu().numpy() - Zs_ref)) < 1e-5


@pytest.mark.parametrize("F,model", [(257, "M1"), (513, "M1"), (257, "M2"), (65, "M1")])
def test_wave_chain_bf16_mode(F, model):
    """The wave-private chain kernel in bf16 mode (8 wavefronts per workgroup at F=257; 4 with four W3 tiles streamed
    from L2 at F=513; 4 with the per-frame layer-1 bias for M2) on a ragged batch with replayed noise:
    every log-acceptance (mcem.py:415-417) within 0.5 of the fp32 oracle's (bf16 products: ~1 % on each of F
    variances) and within 0.1 of the round-1 team kernel's (same bf16 operands, other summation order; its odd last bin
    is an fp32 dot product); where both kernels decide every step alike the samples agree to fma rounding."""
    need_gpu()
    Dy = 1 if model == "M2" else 0
    K, ns, S_steps = 8, 3, 6
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=11, y_dim=Dy, bias_std=0.05)
    counts = [21, 40, 9]
    g = np.random.default_rng(F + 3)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (0.5 + 3 * np.exp(-np.arange(F) / 60.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    ys = [(g.random((n, Dy)) > 0.5).astype(np.float32) for n in counts] if Dy else [None] * 3
    NT = sum(counts)
    eps = g.standard_normal((S_steps, NT, 32)).astype(np.float32)
    uu = g.random((S_steps, NT)).astype(np.float32)
    Z0 = (0.5 * g.standard_normal((NT, 32))).astype(np.float32)
    gains = (0.5 + g.random(NT)).astype(np.float32)
    out = {}
    for kern in ("wave", "team"):
        os.environ["VAENMF_TEAM_CHAIN"] = "1" if kern == "team" else "0"
        try:
            eng = make_engine(params, F, K, counts, Rcap=ns, precision="bf16")
            eng.set_spectrogram(Xs)
            eng.init_nmf(W0, H0)
            if Dy:
                eng.set_labels(torch.from_numpy(np.concatenate(ys)))
            eng.g.copy_(torch.from_numpy(gains))
            eng.Z.copy_(torch.from_numpy(Z0))
            acc = eng.mh_chain(ns, S_steps - ns, 0.01, eps=torch.from_numpy(eps).to(eng.device), u=torch.from_numpy(uu).to(eng.device),
                               want_acc=True).cpu().numpy()
            out[kern] = (acc, eng.Zs[:, :ns].cpu().numpy().copy(), eng.Z.cpu().numpy().copy())
        finally:
            os.environ.pop("VAENMF_TEAM_CHAIN", None)
    assert np.all(np.isfinite(out["wave"][0]))
    # (M2 at 8 wavefronts keeps the per-frame layer-1 bias rows as bf16 in LDS -- one more 2^-9 rounding on the layer-1
    # pre-activations, the size of the bf16 activation rounding the mode has anyway; the team kernel adds them in fp32)
    assert np.max(np.abs(out["wave"][0] - out["team"][0])) < (0.2 if model == "M2" else 0.1)
    same = np.all((np.log(uu) < out["wave"][0]) == (np.log(uu) < out["team"][0]), axis=0)
    assert same.mean() > 0.9
    assert np.max(np.abs(out["wave"][1][same] - out["team"][1][same])) < 1e-5 and np.max(np.abs(out["wave"][2][same] - out["team"][2][same])) < 1e-5
    off = np.concatenate([[0], np.cumsum(counts)])
    for u, n in enumerate(counts):
        sl = slice(off[u], off[u + 1])
        o = orc.MCEMOracle(model, 1)
        o.init_parameters(Xs[u], params, K, 1e-8, orc.NumpyRNG(0), y=ys[u], W0=W0[u], H0=H0[u])
        o.g = gains[sl].copy()
        draws = []
        for m in range(S_steps):
            draws += [eps[m, sl].T.copy(), uu[m, sl].copy()]
        o.rng = orc.ReplayRNG(draws)
        tr = []
        o.sample_posterior(Z0[sl].T.copy(), ns, S_steps - ns, trace=tr)
        ref = np.stack([t["acc"] for t in tr])
        # (only the first step is comparable for every frame: afterwards a chain whose decision differs is in another state)
        assert np.max(np.abs(out["wave"][0][0, sl] - ref[0])) < 0.5


def test_classifier_labels():
    """scripts/evaluate_M2_vad.py:122-131: (x-mean^T)/(std+eps)^T -> Classifier -> > 0.5."""
    need_gpu()
    F, Dy = 129, 5
    z = np.load(GOLDEN + "/mlp_forward.npz")
    cp = {k.split(":p:")[1]: z[k] for k in z.files if k.startswith("clf:p:")}
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=3, y_dim=Dy)
    counts = [11, 30]
    g = np.random.default_rng(2)
    Xs = [(g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))).astype(np.complex64) for n in counts]
    eng = make_engine(params, F, 4, counts, Rcap=4)
    eng.set_spectrogram(Xs)
    clf = [(cp["hidden.0.weight"], cp["hidden.0.bias"]), (cp["hidden.1.weight"], cp["hidden.1.bias"]),
           (cp["output_layer.weight"], cp["output_layer.bias"])]
    mean = g.random((F, 1)).astype(np.float32) * 2
    std = (0.5 + g.random((F, 1))).astype(np.float32)
    x_pow = np.concatenate([np.abs(x) ** 2 for x in Xs]).astype(np.float32)
    for mm, ss in ((None, None), (mean, std)):
        soft, hard = eng.classify(clf, mm, ss)
        rs, rh = orc.classifier_labels(cp, x_pow, mm, ss)
        assert np.max(np.abs(soft.cpu().numpy() - rs)) < 2e-5
        sure = np.abs(rs - 0.5) > 1e-4
        assert np.array_equal(hard.cpu().numpy()[sure], rh[sure])
    # golden forward of the reference Classifier itself
    eng2 = make_engine(params, F, 4, [z["clf_x"].shape[0]], Rcap=4)
    y = eng2.dense(torch.from_numpy(z["clf_x"]).to(eng2.device), *[torch.from_numpy(a).to(eng2.device) for a in clf[0]], 2)
    y = eng2.dense(y, *[torch.from_numpy(a).to(eng2.device) for a in clf[1]], 2)
    y = eng2.dense(y, *[torch.from_numpy(a).to(eng2.device) for a in clf[2]], 3)
    assert np.max(np.abs(y.cpu().numpy() - z["clf_y"])) < 1e-5


def test_file_driver_ragged_batch(tmp_path):
    """The evaluate_M1.py-style driver (read _x.wav -> enhance -> write _s_est/_n_est.wav) on a ragged batch:
    outputs keep each utterance's length (istft max_len = T_orig) and do not depend on batching."""
    need_gpu()
    import os
    from vaenmf import wavio
    from vaenmf.driver import evaluate, speech_list
    from vaenmf.pipeline import Reconstructor
    z = np.load(GOLDEN + "/metrics_dummy_m2.npz")
    x = z["a_s"] / 32768.0 + 0.5 * z["a_n"] / 32768.0
    root = str(tmp_path) + "/"
    raw, proc, out = root + "raw/", root + "processed/", root + "out/"
    lens = {"440c020a": 20000, "440c020b": 31111, "440c020c": 12345}
    for name, T in lens.items():
        for base in (raw, proc):
            os.makedirs(base + "CSR-1-WSJ-0/WAV/wsj0/si_et_05/440/", exist_ok=True)
        wavio.write(raw + "CSR-1-WSJ-0/WAV/wsj0/si_et_05/440/%s.wav" % name, x[:T], 16000)
        wavio.write(proc + "CSR-1-WSJ-0/WAV/wsj0/si_et_05/440/%s_x.wav" % name, x[1000:1000 + T], 16000)
    files = speech_list(raw, "test")
    assert [os.path.basename(f) for f in files] == ["440c020a.wav", "440c020b.wav", "440c020c.wav"]
    params = orc.xavier_normal_params([513, 32, [128, 128]], seed=0)
    rec = Reconstructor(params, 513, 10, niter=3, fs=16000, wlen_sec=64e-3, precision="bf16x3", max_frames=400, max_utts=4)
    w_all = evaluate(rec, files, proc, out + "all/", batch_size=8)
    w_one = evaluate(rec, files, proc, out + "one/", batch_size=1)
    assert len(w_all) == 3
    for (sa, na), (so, no), name in zip(w_all, w_one, sorted(lens)):
        a, fs = wavio.read(sa)
        assert fs == 16000 and len(a) == lens[name] == len(wavio.read(na)[0])
        assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    # same utterance seed => identical result whatever the batch composition: the chains' generator streams AND the NMF
    # initialisation (vaenmf_init_nmf) are keyed by the utterance, not by its place in a batch
    for (sa, _), (so, _) in zip(w_all, w_one):
        assert np.array_equal(wavio.read(sa)[0], wavio.read(so)[0])
    # M2 (VAD-guided, bf16 mode with the sample store): the driver also dumps the classifier's labels per utterance
    # under the reference's file names (evaluate_M2_vad.py:165-166; the blank in the soft file's name is the reference's)
    p2 = orc.xavier_normal_params([513, 32, [128, 128]], seed=1, y_dim=1)
    cp = orc.xavier_normal_classifier([513, [128, 128], 1], seed=2)
    clf = [(cp["hidden.0.weight"], cp["hidden.0.bias"]), (cp["hidden.1.weight"], cp["hidden.1.bias"]),
           (cp["output_layer.weight"], cp["output_layer.bias"])]
    rec2 = Reconstructor(p2, 513, 10, niter=2, model="M2", fs=16000, wlen_sec=64e-3, precision="bf16", max_frames=400, max_utts=4)
    w2 = evaluate(rec2, files, proc, out + "m2/", batch_size=8, classifier=clf)
    for (sa, na), name in zip(w2, sorted(lens)):
        stem = sa[:-len("_s_est.wav")]
        soft = torch.load(stem + " _ibm_soft_est.pt", weights_only=True)
        hard = torch.load(stem + "_ibm_hard_est.pt", weights_only=True)
        nfr = 1 + (lens[name] + (256 if lens[name] % 256 else 0)) // 256
        assert soft.shape == hard.shape == (nfr, 1) and torch.equal(hard, (soft > 0.5).float())
        assert len(wavio.read(sa)[0]) == lens[name] and np.all(np.isfinite(wavio.read(sa)[0]))


def _processed_subset_tree(tmp_path):
    """The reference's committed processed utterances si_et_05/440/440c020{a,b,c}_{s,n,x}.wav, cropped to 1.25 s
    (tests/golden/processed_subset.npz: int16 samples, data only), written back as the tree the drivers read."""
    from vaenmf import wavio
    z = np.load(GOLDEN + "/processed_subset.npz")
    root = str(tmp_path) + "/"
    raw, proc = root + "raw/", root + "processed/"
    for u, rel in zip("abc", z["rel"]):
        for base in (raw, proc):
            os.makedirs(os.path.dirname(base + str(rel)), exist_ok=True)
        wavio.write(raw + str(rel), z[u + "_x"] / 32768.0, 16000)
        for k in "snx":
            wavio.write(proc + os.path.splitext(str(rel))[0] + "_%s.wav" % k, z["%s_%s" % (u, k)] / 32768.0, 16000)
    return z, root, raw, proc


def test_driver_and_metrics_on_reference_utterances(tmp_path):
    """Rows f1 / f2 on the reference's OWN files: evaluate_M1-style driver -> run_metrics_M1-style harness on a cropped
    subset of data/subset/processed, with the input SNRs of its si_et_05_snr_db.p (tests/golden/snr_db.npz).
    Checks: (1) the harness' SI-SDR / SI-SIR / SI-SAR equal the oracle's energy_ratios on the very files the driver
    wrote (1e-6 dB); the per-SNR table groups like metrics.py:93-104; (2) the 'oracle' and 'timo' label sources of
    evaluate_M2_ibm.py:132-141 dump exactly the labels the oracle computes from the same wavs; (3) one utterance
    through the drop-in MCEM_M1 with REPLAYED noise against the oracle run on the same draws: same cost trajectory,
    SI-SDR of the enhanced signal within 0.02 dB."""
    need_gpu()
    import vaenmf
    from vaenmf import wavio, run_metrics
    from vaenmf.driver import evaluate, speech_list
    from vaenmf.pipeline import Reconstructor
    z, root, raw, proc = _processed_subset_tree(tmp_path)
    files = speech_list(raw, "test")
    assert files == [str(r) for r in z["rel"]]
    snr = np.load(GOLDEN + "/snr_db.npz")["processed__CSR-1-WSJ-0__si_et_05_snr_db"]
    F, K = 513, 10
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
    rec = Reconstructor(params, F, K, niter=5, fs=16000, wlen_sec=64e-3, precision="bf16x3", max_frames=400, max_utts=4)
    out = root + "models/M1/"
    evaluate(rec, files, proc, out, batch_size=8)
    all_metrics, st = run_metrics.main(files, proc, out, snr)
    for fp, m in zip(files, all_metrics):
        stem = os.path.splitext(fp)[0]
        ref = orc.energy_ratios(wavio.read(out + stem + "_s_est.wav")[0], wavio.read(proc + stem + "_s.wav")[0], wavio.read(proc + stem + "_n.wav")[0])
        assert np.max(np.abs(np.asarray(m) - np.asarray(ref))) < 1e-6
    assert st[0, 0, 0] == 3 and st[1, 0, 0] == 1 and st[2, 0, 0] == 2          # all, SNR -5 (1 utterance), SNR 0 (2)
    assert abs(st[2, 0, 1] - (all_metrics[1][0] + all_metrics[2][0])) < 1e-9
    # (2) label sources
    p2 = orc.xavier_normal_params([F, 32, [128, 128]], seed=1, y_dim=F)
    rec2 = Reconstructor(p2, F, K, niter=2, model="M2", fs=16000, wlen_sec=64e-3, precision="bf16", max_frames=400, max_utts=4)
    for src in ("oracle", "timo"):
        evaluate(rec2, files, proc, root + "models/M2_%s/" % src, batch_size=8, label_source=src, label_type="ibm", quantile_fraction=0.999, quantile_weight=0.999)
        for fp in files:
            stem = os.path.splitext(fp)[0]
            hard = torch.load(root + "models/M2_%s/" % src + stem + "_ibm_hard_est.pt", weights_only=True).numpy()
            if src == "oracle":
                S = orc.stft(wavio.read(proc + stem + "_s.wav")[0], fs=16000, wlen_sec=64e-3)
                assert np.array_equal(hard, orc.clean_speech_IBM(S, 0.999, 0.999).T)
            else:
                X = orc.stft(wavio.read(proc + stem + "_x.wav")[0], fs=16000, wlen_sec=64e-3)
                want = (orc.timo_mask_estimation(np.power(np.abs(X), 2)) > 0.5).T
                assert np.mean(hard != want) < 1e-4          # (float64 recursion on |X|^2 of two STFTs that differ in the last bit)
            assert np.all(np.isfinite(wavio.read(root + "models/M2_%s/" % src + stem + "_s_est.wav")[0]))
    # (3) replayed noise on real audio, drop-in class against the oracle
    stem = os.path.splitext(files[1])[0]
    x, s, n = (wavio.read(proc + stem + "_%s.wav" % k)[0] for k in "xsn")
    X = orc.stft(x, fs=16000, wlen_sec=64e-3).T
    g = orc.NumpyRNG(5)
    draws = []
    class Rec:
        def rand(self, *sh): a = g.rand(*sh); draws.append(a); return a
        def randn(self, *sh): a = g.randn(*sh); draws.append(a); return a
    o = orc.MCEMOracle("M1", 2)
    o.init_parameters(X, params, K, 1e-8, Rec())
    c_ref = o.run()
    vae = vaenmf.VariationalAutoencoder([F, 32, [128, 128]])
    vae.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    m = vaenmf.MCEM_M1(2, rng="replay")
    it = iter(draws)
    _r, _n = torch.rand, torch.randn
    torch.rand = lambda *a, **k: torch.tensor(next(it))
    torch.randn = lambda *a, **k: torch.tensor(next(it))
    try:
        m.init_parameters(X=X, vae=vae, nmf_rank=K, eps=1e-8, device="cuda:0")
        c = m.run()
    finally:
        torch.rand, torch.randn = _r, _n
    assert np.max(np.abs(c - c_ref) / np.abs(c_ref)) < 1e-3
    sd_ref = orc.energy_ratios(orc.istft(o.S_hat, fs=16000, wlen_sec=64e-3, max_len=len(x)).astype(np.float64), s, n)[0]
    sd_gpu = orc.energy_ratios(orc.istft(m.S_hat, fs=16000, wlen_sec=64e-3, max_len=len(x)).astype(np.float64), s, n)[0]
    print("replayed run on 440c020b: SI-SDR oracle %.4f dB, HIP %.4f dB" % (sd_ref, sd_gpu))
    assert abs(sd_ref - sd_gpu) < 0.02


def test_nonmf_variant_against_reference():
    """MCEM_M2_noNMF drop-in (mcem.py:606-760) with replayed noise against the reference's recorded run."""
    need_gpu()
    import vaenmf
    z, params, draws, meta = load_case("m2_nonmf_f65")
    nsE, biE, nsW, biW = meta["counts"]
    vae = vaenmf.DeepGenerativeModel([meta["F"], meta["Dy"], meta["L"], [128, 128]], None)
    vae.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    it = iter(draws)
    _r, _n = torch.rand, torch.randn
    torch.rand = lambda *s, **k: torch.tensor(next(it))
    torch.randn = lambda *s, **k: torch.tensor(next(it))
    try:
        m = vaenmf.MCEM_M2_noNMF(X=z["X"], Vb=z["Vb"], g=torch.tensor(z["g0"]), Z=torch.tensor(z["Z0"]), y=torch.tensor(z["y"]),
                                 vae=vae, niter=meta["niter"], device="cuda:0", nsamples_E_step=nsE, burnin_E_step=biE,
                                 nsamples_WF=nsW, burnin_WF=biW, var_RW=0.01)
        cost = m.run()
    finally:
        torch.rand, torch.randn = _r, _n
    assert np.max(np.abs(cost - z["cost"]) / np.abs(z["cost"])) < 2e-4
    assert rel_err(m.g.cpu().numpy(), z["g"]) < 2e-3
    assert np.max(np.abs(m.Z.cpu().numpy() - z["Z"])) < 1e-5
    assert nrm_err(m.S_hat, z["S_hat"]) < 2e-3 and nrm_err(m.N_hat, z["N_hat"]) < 2e-3


def test_nonmf_fused_run_equals_stepwise():
    """The fused driver (vaenmf_em_run, device generator) with a fixed noise PSD -- the *_noNMF model -- equals the
    same run stepped through E_step / M_step / compute_WF with the device generator: bit for bit."""
    need_gpu()
    import vaenmf
    z, params, draws, meta = load_case("m2_nonmf_f65")
    nsE, biE, nsW, biW = meta["counts"]
    vae = vaenmf.DeepGenerativeModel([meta["F"], meta["Dy"], meta["L"], [128, 128]], None)
    vae.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    assert issubclass(vaenmf.MCEM_M2_noNMF, vaenmf.EM_noNMF)
    outs = []
    for fused in (True, False):
        m = vaenmf.MCEM_M2_noNMF(X=z["X"], Vb=z["Vb"], g=torch.tensor(z["g0"]), Z=torch.tensor(z["Z0"]), y=torch.tensor(z["y"]),
                                 vae=vae, niter=3, device="cuda:0", nsamples_E_step=nsE, burnin_E_step=biE,
                                 nsamples_WF=nsW, burnin_WF=biW, var_RW=0.01, rng="device", fused_store=False)
        if fused:
            cost = m.run()
        else:
            cost = np.zeros(3)
            for n in range(3):
                m.E_step(); m.M_step(); cost[n] = m.compute_expected_neg_log_like()
            m.compute_WF(sample=True)
            F = meta["F"]
            m.S_hat = np.ascontiguousarray(m._S_dev[:, :F].cpu().numpy()).view(np.complex64).reshape(m._N, F).T
        outs.append((cost, m.S_hat, m.g.cpu().numpy()))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-12) and np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert np.all(np.isfinite(outs[0][0])) and outs[0][0][-1] < outs[0][0][0] * 1.5


@pytest.mark.parametrize("prec", ["bf16", "bf16x3"])
def test_em_run_graph_replay_equals_launch_by_launch(prec):
    """vaenmf_em_run captures a call whose signature repeats into a HIP graph and replays it.  The same batch with the
    same seeds through the pipeline four times: call 1 runs launch by launch, call 2 captures and launches the graph,
    calls 3 and 4 replay it -- every output bit for bit equal to call 1's, a different seed gives a different result
    through the same graph (the batch's contents are read at run time, not baked in), a second batch shape alternating
    with the first gets a graph of its own, and VAENMF_Q_EM_GRAPH says which path ran."""
    need_gpu()
    from vaenmf.pipeline import Reconstructor
    from vaenmf import _lib
    F, K, T, U = 257, 8, 4000, 3
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=5)
    dev = torch.device("cuda:0")
    wav = torch.from_numpy(np.concatenate([orc.synth_utterance(u, T)[2] for u in range(U)]).astype(np.float32)).to(dev)
    rec = Reconstructor(params, F, K, niter=4, fs=16000, wlen_sec=0.032, precision=prec, device=dev, max_frames=U * (T // 128 + 8), max_utts=U)
    q = lambda: _lib.lib().vaenmf_plan_query(rec.eng._plan, _lib.Q_EM_GRAPH)
    outs, paths = [], []
    for call in range(4):
        s_hat, n_hat, cost = rec.enhance(wav, [T] * U, seeds=[11, 12, 13], init_seed=3)
        outs.append((s_hat.cpu().numpy().copy(), n_hat.cpu().numpy().copy(), cost.cpu().numpy().copy()))
        paths.append(q())
    assert paths == [0, 1, 1, 1], paths
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2])
    s2, _, c2 = rec.enhance(wav, [T] * U, seeds=[21, 22, 23], init_seed=4)        # same shapes: still the graph
    assert q() == 1
    assert not np.array_equal(s2.cpu().numpy(), outs[0][0]) and np.all(np.isfinite(c2.cpu().numpy()))
    # a job that alternates two batch shapes (config 4: 63 / 62 utterances) keeps one graph per signature
    wavB, cntB = wav[:2 * T].contiguous(), [T] * 2
    resB, pathsB = [], []
    for call in range(3):
        sB, _, _ = rec.enhance(wavB, cntB, seeds=[31, 32], init_seed=9)
        resB.append(sB.cpu().numpy().copy()); pathsB.append(q())
        sA, _, _ = rec.enhance(wav, [T] * U, seeds=[11, 12, 13], init_seed=3)
        assert q() == 1 and np.array_equal(sA.cpu().numpy(), outs[0][0])
    assert pathsB == [0, 1, 1], pathsB
    assert np.array_equal(resB[1], resB[0]) and np.array_equal(resB[2], resB[0])


def test_label_front_ends_bit_exact():
    """vaenmf.target (csrc/labels.hip) against the reference's own outputs (tests/golden/labels_f257.npz,
    generated by importing python/processing/target.py): every 0/1 label identical, thresholds identical."""
    need_gpu()
    from vaenmf import target
    z = np.load(os.path.join(GOLDEN, "labels_f257.npz"))
    for u in (0, 1):
        S, N = z["S%d" % u], z["N%d" % u]
        for got, key in ((target.clean_speech_IBM(S, 0.999, 0.999), "ibm%d"), (target.clean_speech_IBM(S), "ibm98_%d"),
                         (target.clean_speech_VAD(S, 0.999, 0.999), "vad%d"), (target.clean_speech_VAD(S), "vad98_%d"),
                         (target.noise_robust_clean_speech_VAD(S), "nrvad%d"), (target.noise_robust_clean_speech_IBM(S), "nribm%d")):
            ref = z[key % u]
            assert got.shape == ref.shape and got.dtype == ref.dtype, key
            assert np.array_equal(got, ref), (key, int((got != ref).sum()))
        # soft mask: float32 tolerance.  (numpy's vectorised abs(complex64) is not the correctly rounded hypot -- it
        # differs from numpy's own scalar abs by 1 ulp in ~1 % of the entries -- so bit equality is not defined here)
        assert np.max(np.abs(target.ideal_wiener_mask(S, N) - z["iwm%d" % u])) < 3e-7
    # the batched entry point: both utterances in one call, thresholds against the oracle
    Xs = [np.ascontiguousarray(z["S%d" % u].T) for u in (0, 1)]
    F = Xs[0].shape[1]
    X = torch.from_numpy(np.concatenate(Xs)).cuda()
    Xp = torch.zeros(X.shape[0], 272, dtype=torch.complex64, device="cuda")
    Xp[:, :F] = X
    for mode, key in (("ibm", "ibm98_%d"), ("vad", "vad98_%d")):
        y, thr = target.lorenz_labels_batch(Xp, [x.shape[0] for x in Xs], F, mode, want_thresholds=True)
        y, thr = y.cpu().numpy(), thr.cpu().numpy()
        o = 0
        for u, x in enumerate(Xs):
            ref = z[key % u]
            p = orc.power_c64(z["S%d" % u])
            t_ref = orc.lorenz_threshold(p if mode == "ibm" else orc.frame_power(z["S%d" % u]), 0.98)
            assert thr[u] == t_ref
            got = y[o:o + x.shape[0]]
            assert np.array_equal(got.T if mode == "ibm" else got[None], ref)
            o += x.shape[0]
    # an utterance whose strongest entry alone exceeds the fraction: the reference raises IndexError
    one = np.zeros((5, 4), np.complex64)
    one[2, 1] = 3.0
    with pytest.raises(RuntimeError, match="index -1 is out of bounds"):
        target.clean_speech_IBM(one, 0.5)


def test_mask_baseline_against_oracle():
    """MaskEnhancer (scripts/evaluate_wiener_filter.py:71-113) against the numpy restatement: classifier mask,
    S_hat = mask * X, iSTFT."""
    need_gpu()
    from vaenmf.pipeline import MaskEnhancer
    F, T = 257, 9000
    clf = orc.xavier_normal_classifier([F, [128, 128, 128, 128, 128], F], seed=4)
    layers = [(clf["hidden.%d.weight" % i], clf["hidden.%d.bias" % i]) for i in range(5)] + [(clf["output_layer.weight"], clf["output_layer.bias"])]
    g = np.random.default_rng(2)
    mean, std = g.random((F, 1)) * 0.1, 0.5 + g.random((F, 1))
    xs = [orc.synth_utterance(s, n_samples=T + 500 * s)[2] for s in (1, 2)]
    counts = [len(x) for x in xs]
    enh = MaskEnhancer(layers, F, mean, std, wlen_sec=32e-3, device="cuda:0")
    s_hat, mask = enh.enhance(torch.from_numpy(np.concatenate(xs).astype(np.float32)).cuda(), counts)
    s_hat, mask = s_hat.cpu().numpy(), mask.cpu().numpy()
    o = n0 = 0
    for x, c in zip(xs, counts):
        X = orc.stft(x.astype(np.float32), fs=16000, wlen_sec=32e-3, hop_percent=0.25)          # (F, N)
        xin = ((np.abs(X.T) ** 2).astype(np.float32) - mean.T) / (std + 1e-8).T
        m_ref = orc.classifier_forward(clf, xin.astype(np.float32))
        N = X.shape[1]
        assert np.max(np.abs(mask[n0:n0 + N] - m_ref)) < 2e-5
        s_ref = orc.istft((m_ref * X.T).T, fs=16000, wlen_sec=32e-3, hop_percent=0.25, max_len=c)
        assert nrm_err(s_hat[o:o + c], s_ref) < 2e-5
        o += c
        n0 += N


def test_spp_estimator_against_reference():
    """vaenmf.spp_estimation (csrc/labels.hip) against the reference's outputs (tests/golden/spp_f257.npz):
    soft SPP within 2e-6 (float64 recursion with the device's exp, stored as float32), the hard labels
    `mask > 0.5` of scripts/evaluate_M2_ibm.py:139 identical, the mask-driven noise PSD exact."""
    need_gpu()
    from vaenmf import spp_estimation as spp
    z = np.load(os.path.join(GOLDEN, "spp_f257.npz"))
    for u in (0, 1):
        P = z["P%d" % u]
        m = spp.timo_mask_estimation(P)
        assert m.shape == P.shape and m.dtype == P.dtype
        assert np.max(np.abs(m - z["mask%d" % u])) < 2e-6
        assert np.array_equal(m > 0.5, z["mask%d" % u] > 0.5)
        assert np.max(np.abs(spp.timo_vad_estimation(P) - z["vad%d" % u])) < 2e-6
        assert np.array_equal(spp.timo_noise_estimation(P, z["mask%d" % u]), z["psd%d" % u])
    # batch entry: both utterances in one launch, state restarted per utterance
    per = torch.from_numpy(np.concatenate([np.ascontiguousarray(z["P%d" % u].T) for u in (0, 1)])).cuda()
    s, psd = spp.spp_batch(per, [z["P0"].shape[1], z["P1"].shape[1]], 257, want_psd=True)
    s, psd = s.cpu().numpy(), psd.cpu().numpy()
    n0 = z["P0"].shape[1]
    assert np.max(np.abs(s[:n0].T - z["mask0"])) < 2e-6 and np.max(np.abs(s[n0:].T - z["mask1"])) < 2e-6
    ref_psd = orc.spp_recursion(z["P1"].T)[0]
    assert np.max(np.abs(psd[n0:] - ref_psd) / (np.abs(ref_psd) + 1e-12)) < 1e-6


def test_sample_store_holds_the_samples_variances():
    """Sample-variance store: after a chain, row src[n][r] of the store is the decoded variance of sample r --
    the same numbers vaenmf_decode computes from Zs (both modes; ragged batch; burn-in shorter and longer than
    the sample count) -- and the chain itself is unchanged by storing."""
    need_gpu()
    z, params, draws, meta = load_case("m1_f257")
    F, K = meta["F"], meta["K"]
    counts, seeds = [37, 64, 70], [5, 6, 7]
    g = np.random.default_rng(8)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (1 + 3 * np.exp(-np.arange(F) / 40.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    for prec in ("bf16x3", "bf16"):
        for (ns, bi) in ((6, 9), (10, 3), (5, 0)):
            eng = make_engine(params, F, K, counts, Rcap=12, seeds=seeds, precision=prec)
            eng.set_spectrogram(Xs)
            eng.init_nmf(W0, H0)
            Z0 = eng.Z.clone()
            eng.mh_chain(ns, bi, 0.01, call=2)
            Zs_plain, Z_plain = eng.Zs.clone(), eng.Z.clone()
            eng.Z.copy_(Z0)
            eng.sample_store(True)
            eng.mh_chain(ns, bi, 0.01, call=2)
            assert torch.equal(eng.Zs, Zs_plain) and torch.equal(eng.Z, Z_plain)
            got = eng.stored_variances(ns)[:, :, :F].cpu().numpy()
            ref = eng.decode(ns)[:, :, :F].cpu().numpy()
            assert np.all(np.isfinite(got))
            # same MFMA products, accumulated with the operands in swapped roles: equal to rounding; bf16 mode
            # stores bf16 rows (8 significant bits: relative rounding up to 2^-8); the odd last bin is an fp32 dot
            # product in the decoding kernel and a bf16 MFMA product in the chain (the mode's own tolerance)
            rel = np.abs(got - ref) / ref
            assert np.max(rel[:, :, :F - 1]) < (2e-5 if prec == "bf16x3" else 4e-3), (prec, ns, bi)
            assert np.max(rel[:, :, F - 1]) < (2e-5 if prec == "bf16x3" else 5e-2), (prec, ns, bi)
            eng.sample_store(False)


@pytest.mark.parametrize("F,K,prec", [(257, 8, "bf16x3"), (257, 8, "bf16"), (513, 10, "bf16x3"), (65, 4, "bf16x3"), (257, 32, "bf16"),
                                       (513, 32, "bf16"), (513, 32, "bf16x3")])   # the last two: BASELINE config 5 (stress) instantiations
def test_stored_m_step_and_wiener_match_the_decoding_ones(F, K, prec):
    """vaenmf_m_step_stored / vaenmf_wiener_stored (streaming the chain's stored variances) against
    vaenmf_m_step / vaenmf_wiener (decoding Zs again) from the same state and the same chain: the same W, H, g,
    cost and Wiener outputs up to summation order (bf16x3 mode, float rows: 2e-5 relative on the updates, 1e-6 on
    the cost; bf16 mode: bf16 rows with up to 2^-8 relative rounding per stored variance, and the odd last bin comes
    from the chain's bf16 MFMA products in the store but from the decoding kernel's fp32 dot product: 1e-2 / 5e-4)."""
    need_gpu()
    tu, tc, tw = (2e-5, 1e-6, 2e-5) if prec == "bf16x3" else (1e-2, 5e-4, 1e-2)
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=3, bias_std=0.1)
    counts, seeds = [37, 64, 70], [5, 6, 7]
    g = np.random.default_rng(8)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (1 + 3 * np.exp(-np.arange(F) / 40.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]

    def prep():
        eng = make_engine(params, F, K, counts, Rcap=40, seeds=seeds, precision=prec)
        eng.set_spectrogram(Xs)
        eng.init_nmf(W0, H0)
        eng.sample_store(True)
        return eng

    for (ns, bi) in ((10, 6), (7, 0), (27, 3), (40, 2)):     # 40 samples: more than one register batch of rows
        ea, eb = prep(), prep()
        for it in range(2):
            for e in (ea, eb):
                e.mh_chain(ns, bi, 0.01, call=it)
            assert torch.equal(ea.Zs, eb.Zs)
            ca = ea.m_step(ns).clone()
            cb = eb.m_step_stored().clone()
            for name in ("W", "Ht", "g"):
                x, y = getattr(ea, name).cpu().numpy(), getattr(eb, name).cpu().numpy()
                assert np.max(np.abs(x - y) / (np.abs(x) + 1e-20)) < tu, (name, ns, bi, it)
            assert np.max(np.abs(ca.cpu().numpy() - cb.cpu().numpy()) / np.abs(ca.cpu().numpy())) < tc
            # keep the two engines on the same trajectory
            for name in ("W", "Ht", "g"):
                getattr(eb, name).copy_(getattr(ea, name))
        for e in (ea, eb):
            e.mh_chain(ns, bi, 0.01, call=9, update_Z=False)
        Sa, Na, WFsa, WFna = ea.wiener(ns, want_masks=True)
        Sb, Nb, WFsb, WFnb = eb.wiener_stored(want_masks=True)
        assert float((WFsa[:, :F] - WFsb[:, :F]).abs().max()) < tw and float((WFna[:, :F] - WFnb[:, :F]).abs().max()) < tw
        assert nrm_err(Sb.cpu().numpy(), Sa.cpu().numpy()) < tw and nrm_err(Nb.cpu().numpy(), Na.cpu().numpy()) < tw
        assert float(Sb[:, F:].abs().max()) == 0.0 and float(WFsb[:, F:].abs().max()) == 0.0


@pytest.mark.parametrize("variant", ["m2_labels", "nonmf_gains_only"])
def test_stored_path_with_labels_and_fixed_noise(variant):
    """The streaming M-step / Wiener filter in the two configurations the parametrised test above does not reach:
    M2 (per-frame layer-1 bias B1 from the labels, mcem.py:242) and the *_noNMF variants (fixed noise variance, gains
    only, mcem.py:543-578), bf16 mode, against the decoding kernels from the same chain."""
    need_gpu()
    F, K, Dy = 257, 8, 1
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=4, y_dim=Dy if variant == "m2_labels" else 0, bias_std=0.1)
    counts, seeds = [37, 64, 70], [5, 6, 7]
    g = np.random.default_rng(9)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (1 + 3 * np.exp(-np.arange(F) / 40.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    y = torch.from_numpy((g.random((sum(counts), Dy)) > 0.5).astype(np.float32))

    def prep():
        eng = make_engine(params, F, K, counts, Rcap=12, seeds=seeds, precision="bf16")
        eng.set_spectrogram(Xs)
        eng.init_nmf(W0, H0)
        if variant == "m2_labels":
            eng.set_labels(y.cuda())
        else:
            Vb = torch.zeros(eng.NT, eng.Fs, device="cuda")
            Vb[:, :F] = torch.from_numpy(g.random((sum(counts), F)).astype(np.float32) + 0.1).cuda()
            eng.set_noise_psd(Vb)
        eng.sample_store(True)
        return eng

    g = np.random.default_rng(10)          # (both engines draw the same noise PSD)
    ea = prep()
    g = np.random.default_rng(10)
    eb = prep()
    for it in range(2):
        for e in (ea, eb):
            e.mh_chain(10, 4, 0.01, call=it)
        assert torch.equal(ea.Zs, eb.Zs)
        ca, cb = ea.m_step(10).clone(), eb.m_step_stored().clone()
        for name in ("W", "Ht", "g"):
            x, yv = getattr(ea, name).cpu().numpy(), getattr(eb, name).cpu().numpy()
            assert np.max(np.abs(x - yv) / (np.abs(x) + 1e-20)) < 1e-2, (name, it)
        assert np.max(np.abs(ca.cpu().numpy() - cb.cpu().numpy()) / np.abs(ca.cpu().numpy())) < 5e-4
        for name in ("W", "Ht", "g"):
            getattr(eb, name).copy_(getattr(ea, name))
    for e in (ea, eb):
        e.mh_chain(10, 4, 0.01, call=9, update_Z=False)
    Sa, Na, _, _ = ea.wiener(10)
    Sb, Nb, _, _ = eb.wiener_stored()
    assert nrm_err(Sb.cpu().numpy(), Sa.cpu().numpy()) < 1e-2 and nrm_err(Nb.cpu().numpy(), Na.cpu().numpy()) < 1e-2


@pytest.mark.parametrize("F,K", [(257, 8), (513, 10)])
def test_bench_mode_m_step_against_the_oracle(F, K):
    """The bench mode's M-step -- bf16 MFMA chain, sample variances stored as bf16 rows, streaming W / H / g / cost
    kernels -- against the fp32 ORACLE's M-step (mcem.py:90-152) from the same posterior samples (the chain's own Zs,
    decoded by the oracle in fp32).  Stated bounds: W, H, g within 3e-2 relative (bf16 products carry ~1 % on each
    variance, bf16 storage 0.4 %; every update is a ratio of sums over 30 samples x F bins), cost within 3e-3."""
    need_gpu()
    R, bi = 30, 4
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=5, bias_std=0.05)
    counts = [21, 40, 9]
    g = np.random.default_rng(F * 3 + K)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (0.5 + 3 * np.exp(-np.arange(F) / 60.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    eng = make_engine(params, F, K, counts, Rcap=R, precision="bf16", seeds=[3, 4, 5])
    eng.set_spectrogram(Xs)
    eng.init_nmf(W0, H0)
    gains = (0.5 + g.random(sum(counts))).astype(np.float32)
    eng.g.copy_(torch.from_numpy(gains))
    eng.Z.copy_(torch.from_numpy((0.5 * g.standard_normal((sum(counts), 32))).astype(np.float32)))
    eng.sample_store(True)
    eng.mh_chain(R, bi, 0.01, call=0)
    Zs = eng.Zs[:, :R].cpu().numpy()
    eng.m_step_stored()
    cost = eng.cost_from_frames(R)
    for u, n in enumerate(counts):
        sl = eng.utt_slice(u)
        o = orc.MCEMOracle("M1", 1)
        o.init_parameters(Xs[u], params, K, 1e-8, orc.NumpyRNG(0), W0=W0[u], H0=H0[u])
        o.g = gains[sl].copy()
        o.compute_Vs(Zs[sl]); o.compute_Vs_scaled(); o.compute_Vx()
        o.M_step()
        assert rel_err(eng.W[u, :F, :K].cpu().numpy(), o.W) < 3e-2
        assert rel_err(eng.Ht[sl, :K].cpu().numpy().T, o.H) < 3e-2
        assert rel_err(eng.g[sl].cpu().numpy(), o.g) < 3e-2
        assert abs(cost[u] - o.compute_expected_neg_log_like()) / abs(cost[u]) < 3e-3


def test_fused_run_with_the_sample_store():
    """vaenmf_em_run with the store on (chain stores, streaming M-step and Wiener filter) against the run that
    decodes the samples again: same device RNG streams; the trajectories agree to rounding for the first
    iterations (they are chaotic beyond: a flipped accept decision changes a frame's chain)."""
    need_gpu()
    z, params, draws, meta = load_case("m1_f257")
    F, K = meta["F"], meta["K"]
    counts, seeds = [37, 64, 50], [5, 6, 7]
    g = np.random.default_rng(8)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (1 + 3 * np.exp(-np.arange(F) / 40.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]

    def run(store, niter):
        eng = make_engine(params, F, K, counts, Rcap=12, seeds=seeds)
        eng.set_spectrogram(Xs)
        eng.init_nmf(W0, H0)
        return eng.run(niter, 6, 5, 12, 7, 0.01, store=store)

    ca, Sa, Na = run(False, 1)
    cb, Sb, Nb = run(True, 1)
    assert np.max(np.abs(ca.cpu().numpy() - cb.cpu().numpy()) / np.abs(ca.cpu().numpy())) < 1e-6
    assert nrm_err(Sb.cpu().numpy(), Sa.cpu().numpy()) < 1e-3
    ca, Sa, Na = run(False, 4)
    cb, Sb, Nb = run(True, 4)
    assert np.max(np.abs(ca.cpu().numpy() - cb.cpu().numpy()) / np.abs(ca.cpu().numpy())) < 5e-3
    assert np.all(np.isfinite(Sb.cpu().numpy())) and np.all(np.isfinite(cb.cpu().numpy()))


def test_drop_in_object_reuses_its_engine_across_utterances():
    """scripts/evaluate_M1.py:111-166 calls init_parameters once per utterance on ONE MCEM object.  The second call with
    the same model reuses the plan, the packed weights and every device buffer (no device allocation by the library, none
    by torch's allocator), gives the results a fresh object gives, and a model whose weights were written in place gets a
    new engine."""
    need_gpu()
    import vaenmf
    from vaenmf import _lib
    z, params, draws, meta = load_case("m1_f65")
    nsE, biE, nsW, biW = meta["counts"]
    vae = vaenmf.VariationalAutoencoder([meta["F"], meta["L"], [128, 128]])
    vae.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    mk = lambda: vaenmf.MCEM_M1(niter=meta["niter"], nsamples_E_step=nsE, burnin_E_step=biE, nsamples_WF=nsW, burnin_WF=biW, var_RW=0.01)
    m = mk()
    runs = []
    for rep in range(3):
        torch.manual_seed(int(z["seed"]))
        if rep:
            torch.cuda.synchronize()
            a0 = _lib.lib().vaenmf_plan_query(m._eng._plan, _lib.Q_DEV_ALLOCS)
            r0, eng0, plan0 = torch.cuda.memory_reserved(), m._eng, m._eng._plan.value
        m.init_parameters(X=z["X"], vae=vae, nmf_rank=meta["K"], eps=1e-8, device="cuda:0")
        if rep:
            assert m._eng is eng0 and m._eng._plan.value == plan0
            assert _lib.lib().vaenmf_plan_query(m._eng._plan, _lib.Q_DEV_ALLOCS) == a0        # the library allocated nothing
            assert torch.cuda.memory_reserved() == r0                                         # nor did torch's allocator
        c = m.run()
        runs.append((c.copy(), m.S_hat.copy()))
    assert np.max(np.abs(runs[0][0] - z["cost"]) / np.abs(z["cost"])) < 2e-4
    for c, s in runs[1:]:                                   # the reused engine gives the first run's result bit for bit
        assert np.array_equal(c, runs[0][0]) and np.array_equal(s, runs[0][1])
    # a shorter utterance on the same object: still the same engine; a longer one grows it
    eng0 = m._eng
    torch.manual_seed(1)
    m.init_parameters(X=z["X"][:10], vae=vae, nmf_rank=meta["K"], eps=1e-8, device="cuda:0")
    assert m._eng is eng0 and m.run().shape == (meta["niter"],) and m.S_hat.shape == (meta["F"], 10)
    Xl = np.concatenate([z["X"], z["X"]], 0)
    m.init_parameters(X=Xl, vae=vae, nmf_rank=meta["K"], eps=1e-8, device="cuda:0")
    assert m._eng is not eng0 and m._eng._max_frames >= Xl.shape[0]
    assert np.all(np.isfinite(m.run())) and m.S_hat.shape == (meta["F"], Xl.shape[0])
    # weights written in place (load_state_dict copies in place): the engine is rebuilt with the new weights
    eng1 = m._eng
    with torch.no_grad():
        vae.decoder.reconstruction.bias.add_(0.25)
    torch.manual_seed(int(z["seed"]))
    m.init_parameters(X=z["X"], vae=vae, nmf_rank=meta["K"], eps=1e-8, device="cuda:0")
    assert m._eng is not eng1
    c2 = m.run()
    assert np.max(np.abs(c2 - runs[0][0])) > 1e-3          # other weights, other costs
    import pickle
    m2 = pickle.loads(pickle.dumps(m))                     # still picklable (spawn Pool, evaluate_M1.py:206-216)
    assert m2._eng is None and m2.niter == m.niter


@pytest.mark.parametrize("R,counts", [(30, [21, 64, 130, 9, 65]), (10, [70, 5, 64])])
def test_fused_w_statistics_equal_the_two_kernel_path(R, counts):
    """wstats_fused_kernel (W statistics and the W update's sums over frames in one pass, one partial per <= 64-frame tile,
    mcem.py:107-110) against wstats_rot + w_partial + w_update from the same store: the same W, normalisation and --
    through the H/g kernel that follows -- H, g, cost, up to the order of the float sums over frames (2e-5); ragged
    utterances: tiles of 64, 2, 1 frames, wavefronts without frames.  The fused path is reproducible bit for bit."""
    need_gpu()
    from vaenmf import _lib
    F, K = 257, 8
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=5, bias_std=0.05)
    g = np.random.default_rng(R + len(counts))
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (0.5 + 3 * np.exp(-np.arange(F) / 60.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    gains = (0.5 + g.random(sum(counts))).astype(np.float32)

    def run(fused):
        os.environ["VAENMF_WFUSED"] = "1" if fused else "0"
        os.environ["VAENMF_WGROUP"] = "0"                  # (this test is about the TILE kernel; small batches would take the group kernel)
        try:
            eng = make_engine(params, F, K, counts, Rcap=R, precision="bf16", seeds=list(range(len(counts))))
            eng.set_spectrogram(Xs)
            eng.init_nmf(W0, H0)
            eng.g.copy_(torch.from_numpy(gains))
            eng.sample_store(True)
            eng.mh_chain(R, 3, 0.01, call=0)
            c = eng.m_step_stored().clone()
            assert _lib.lib().vaenmf_plan_query(eng._plan, _lib.Q_W_FUSED) == (1 if fused else 0)
            return [t.cpu().numpy().copy() for t in (eng.W, eng.Ht, eng.g, c)]
        finally:
            os.environ.pop("VAENMF_WFUSED", None)
            os.environ.pop("VAENMF_WGROUP", None)

    a, b, a2 = run(True), run(False), run(True)
    for x, y, name in zip(a, b, ("W", "Ht", "g", "cost")):
        assert np.max(np.abs(x - y) / (np.abs(y) + 1e-20)) < 2e-5, name
    for x, y in zip(a, a2):
        assert np.array_equal(x, y)
    # a grid smaller than the tile count (long batches on the real grid): every workgroup walks several tiles and
    # utterances in its grid-stride loop -- the same sums, bit for bit (a tile's partial does not depend on who computes it)
    os.environ["VAENMF_WFUSED_GRID"] = "2"
    try:
        a3 = run(True)
    finally:
        os.environ.pop("VAENMF_WFUSED_GRID", None)
    for x, y in zip(a, a3):
        assert np.array_equal(x, y)
    assert float(np.abs(a[0][:, F:]).max()) == 0.0 and float(np.abs(a[0][:, :, K:]).max()) == 0.0 if a[0].shape[2] > K else True


def test_driver_and_metrics_on_all_nine_reference_utterances_ragged(tmp_path):
    """Rows f1 / f2 on ALL the utterances the reference commits (data/subset/processed: si_tr_s/011, si_dt_05/050,
    si_et_05/440, three each), cropped to NINE DIFFERENT lengths (tests/golden/processed_subset9.npz, made by
    tests/golden/make_processed_subset.py): speech_list per subset in the reference's order
    (python/dataset/csr1_wjs0_dataset.py:19-54), the evaluate_M1-style driver on one ragged batch of nine (bench mode:
    bf16, sample store) and on batches of four, run_metrics per subset with the reference's own input-SNR lists, the
    oracle's energy_ratios on the written files, and the 'oracle' IBM label source on the ragged batch."""
    need_gpu()
    from vaenmf import wavio, run_metrics
    from vaenmf.driver import evaluate, speech_list
    from vaenmf.pipeline import Reconstructor
    from vaenmf import _lib
    z = np.load(GOLDEN + "/processed_subset9.npz")
    root = str(tmp_path) + "/"
    raw, proc = root + "raw/", root + "processed/"
    lens = {}
    for i, rel in enumerate(z["rel"]):
        rel = str(rel)
        for base in (raw, proc):
            os.makedirs(os.path.dirname(base + rel), exist_ok=True)
        wavio.write(raw + rel, z["u%d_x" % i] / 32768.0, 16000)
        for k in "snx":
            wavio.write(proc + os.path.splitext(rel)[0] + "_%s.wav" % k, z["u%d_%s" % (i, k)] / 32768.0, 16000)
        lens[rel] = len(z["u%d_x" % i])
    assert len(set(lens.values())) == 9                                   # ragged: nine different lengths
    files = []
    for subset in ("train", "validation", "test"):
        fl = speech_list(raw, subset)
        assert fl == [str(r) for r, s in zip(z["rel"], z["subset"]) if str(s) == subset]
        files += fl
    F, K = 513, 10
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
    rec = Reconstructor(params, F, K, niter=4, fs=16000, wlen_sec=64e-3, precision="bf16", max_frames=1100, max_utts=9)
    out9, out4 = root + "models/M1_b9/", root + "models/M1_b4/"
    w9 = evaluate(rec, files, proc, out9, batch_size=16)
    assert _lib.lib().vaenmf_plan_query(rec.eng._plan, _lib.Q_MSTEP_PATH) == 1
    nfr = [1 + (lens[f] + (256 if lens[f] % 256 else 0)) // 256 for f in files]
    assert list(rec.frame_counts) == nfr and len(set(nfr)) >= 8           # the batch the engine saw was ragged
    w4 = evaluate(rec, files, proc, out4, batch_size=4)
    assert len(w9) == len(w4) == 9
    for (sa, na), fp in zip(w9, files):
        a, fs = wavio.read(sa)
        assert fs == 16000 and len(a) == lens[fp] == len(wavio.read(na)[0])
        assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    for (s9, _), (s4, _) in zip(w9, w4):                                        # batching does not change any utterance's result
        assert np.array_equal(wavio.read(s9)[0], wavio.read(s4)[0])
    # run_metrics per subset against the oracle on the written files; per-SNR tables from the reference's own SNR lists
    snr_z = np.load(GOLDEN + "/snr_db.npz")
    for subset, key in (("validation", "processed__CSR-1-WSJ-0__si_dt_05_snr_db"), ("test", "processed__CSR-1-WSJ-0__si_et_05_snr_db")):
        fl = speech_list(raw, subset)
        snr = snr_z[key]
        all_metrics, st = run_metrics.main(fl, proc, out9, snr)
        for fp, m in zip(fl, all_metrics):
            stem = os.path.splitext(fp)[0]
            ref = orc.energy_ratios(wavio.read(out9 + stem + "_s_est.wav")[0], wavio.read(proc + stem + "_s.wav")[0], wavio.read(proc + stem + "_n.wav")[0])
            assert np.max(np.abs(np.asarray(m) - np.asarray(ref))) < 1e-6
        assert st[0, 0, 0] == 3 and st[1, 0, 0] == 1 and st[2, 0, 0] == 2      # all; SNR -5 dB: 1 utterance; the other bin: 2
        assert abs(st[0, 0, 1] - sum(m[0] for m in all_metrics)) < 1e-9
    mt = run_metrics.compute_metrics(speech_list(raw, "train"), proc, out9)
    assert len(mt) == 3 and np.all(np.isfinite(np.asarray(mt)))
    # 'oracle' IBM labels (evaluate_M2_ibm.py:132-134) on the ragged batch: exactly the oracle's labels per utterance
    p2 = orc.xavier_normal_params([F, 32, [128, 128]], seed=1, y_dim=F)
    rec2 = Reconstructor(p2, F, K, niter=2, model="M2", fs=16000, wlen_sec=64e-3, precision="bf16", max_frames=1100, max_utts=9)
    evaluate(rec2, files, proc, root + "models/M2/", batch_size=16, label_source="oracle", label_type="ibm", quantile_fraction=0.999, quantile_weight=0.999)
    for fp, n in zip(files, nfr):
        stem = os.path.splitext(fp)[0]
        hard = torch.load(root + "models/M2/" + stem + "_ibm_hard_est.pt", weights_only=True).numpy()
        S = orc.stft(wavio.read(proc + stem + "_s.wav")[0], fs=16000, wlen_sec=64e-3)
        assert hard.shape == (n, F) and np.array_equal(hard, orc.clean_speech_IBM(S, 0.999, 0.999).T)
        assert np.all(np.isfinite(wavio.read(root + "models/M2/" + stem + "_s_est.wav")[0]))


def test_classifier_batch_norm_and_two_class_on_the_device():
    """Classifier(batch_norm=True) (eval mode: relu(BN(relu(Linear))), models.py:50-52, scripts/reconstruct_dnn_classif.py:
    80, 125-129) and Classifier2Classes (models.py:64-88) through vaenmf_dense with the layers
    vaenmf.engine.classifier_layers_from_state folds, against forwards of the reference classes
    (tests/golden/mlp_forward_variants.npz): soft outputs 1e-5."""
    need_gpu()
    from vaenmf import _lib
    from vaenmf.engine import classifier_layers_from_state
    z = np.load(GOLDEN + "/mlp_forward_variants.npz")
    params = orc.xavier_normal_params([129, 32, [128, 128]], seed=3)
    eng = make_engine(params, 129, 4, [z["x"].shape[0]], Rcap=4)
    dev = eng.device
    for tag, two, want in (("bn", False, z["bn_y"]), ("c2", True, z["c2_y"][:, 0, :])):
        p = {k.split(":p:")[1]: z[k] for k in z.files if k.startswith(tag + ":p:")}
        layers = classifier_layers_from_state(p, two_classes=two)
        h = torch.from_numpy(z["x"]).to(dev)
        for w, b in layers[:-1]:
            h = eng.dense(h, torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), _lib.ACT_RELU)
        w, b = layers[-1]
        y = eng.dense(h, torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), _lib.ACT_SIGMOID).cpu().numpy()
        assert y.shape == want.shape and np.max(np.abs(y - want)) < 1e-5, tag


def test_stress_shape_at_its_own_iteration_count():
    """BASELINE config 5's shape at its own 500 EM iterations (1024-pt STFT, F=513, NMF rank 32) on a small ragged batch
    through the whole pipeline in the bench mode: the fused driver's chunked cost reduction over 20 chunks of 25 iterations,
    the graph replay of a 2 500-launch call (second and third call of the same signature), finite outputs, and the cost --
    the expected negative log-likelihood the EM iterations decrease in expectation (mcem.py:70, :165) -- far below its start."""
    need_gpu()
    from vaenmf.pipeline import Reconstructor
    from vaenmf import _lib
    F, K, NITER = 513, 32, 500
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
    T = [9000, 12000, 7000]
    sig = [orc.synth_utterance(u, t) for u, t in enumerate(T)]
    dev = torch.device("cuda:0")
    wav = torch.from_numpy(np.concatenate([s[2] for s in sig]).astype(np.float32)).to(dev)
    rec = Reconstructor(params, F, K, niter=NITER, fs=16000, wlen_sec=64e-3, precision="bf16", device=dev, max_frames=200, max_utts=3)
    outs = []
    for rep in range(3):
        s_hat, n_hat, cost = rec.enhance(wav, T, seeds=[11, 12, 13], init_seed=4)
        outs.append((s_hat.cpu().numpy(), cost.cpu().numpy()))
    assert _lib.lib().vaenmf_plan_query(rec.eng._plan, _lib.Q_EM_GRAPH) == 1 and _lib.lib().vaenmf_plan_query(rec.eng._plan, _lib.Q_MSTEP_PATH) == 1
    s0, c0 = outs[0]
    assert c0.shape == (3, NITER) and np.all(np.isfinite(c0)) and np.all(np.isfinite(s0)) and np.abs(s0).max() > 0
    assert np.all(c0[:, -50:].mean(1) < c0[:, :5].mean(1) - 0.1)             # the EM iterations did their work
    assert np.all(np.abs(np.diff(c0[:, 100:], axis=1)) < 0.05)               # no jump anywhere behind the transient (a lost chunk would show)
    for s, c in outs[1:]:                                                     # eager, captured and replayed calls agree bit for bit
        assert np.array_equal(s, s0) and np.array_equal(c, c0)


@pytest.mark.parametrize("nfft,K,prec,store", [(512, 8, "bf16", None), (1024, 10, "bf16", None), (1024, 32, "bf16", None),
                                               (1024, 10, "bf16x3", False), (512, 8, "bf16x3", None)])
def test_an_utterances_result_does_not_depend_on_its_batch(nfft, K, prec, store):
    """The reference processes one utterance at a time (scripts/evaluate_M1.py:176-177); here utterances share launches, so
    an utterance's enhanced signal and cost must not depend on which batch it sits in: generator streams and the NMF
    initialisation are keyed by the utterance, every per-frame / per-utterance sum has a fixed order (round 3 fixed the one
    exception: the rank > 8 noise variance summed its ranks in another order when the utterance's W was not the copy staged
    in LDS).  Seven ragged utterances together, three of them, one alone -- bit for bit, in every mode the bench reports."""
    need_gpu()
    from vaenmf.pipeline import Reconstructor
    dev = torch.device("cuda:0")
    T = [20000, 27111, 16384, 24000, 31000, 18000, 22222]
    sig = [orc.synth_utterance(u, t) for u, t in enumerate(T)]
    F = nfft // 2 + 1
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
    rec = Reconstructor(params, F, K, niter=4, fs=16000, wlen_sec=nfft / 16000, precision=prec, device=dev, max_frames=1500, max_utts=8, store=store)

    def run(idx):
        wav = torch.from_numpy(np.concatenate([sig[i][2] for i in idx]).astype(np.float32)).to(dev)
        s, n, c = rec.enhance(wav, [T[i] for i in idx], seeds=[100 + i for i in idx], init_seed=0)
        off = np.concatenate([[0], np.cumsum([T[i] for i in idx])])
        s, n = s.cpu().numpy(), n.cpu().numpy()
        return {i: (s[off[k]:off[k + 1]], n[off[k]:off[k + 1]], c[k].cpu().numpy()) for k, i in enumerate(idx)}
    a, b, c = run([0, 1, 2, 3, 4, 5, 6]), run([4, 5, 6]), run([6])
    for i in (4, 5, 6):
        for x, y in zip(a[i], b[i]):
            assert np.array_equal(x, y), i
    for x, y in zip(a[6], c[6]):
        assert np.array_equal(x, y)


class _RecordingRNG:
    """Seeded numpy generator (the oracle's NumpyRNG) that keeps what it drew, in order."""

    def __init__(self, seed):
        self.g, self.draws = orc.NumpyRNG(seed), []

    def rand(self, *shape):
        self.draws.append(self.g.rand(*shape))
        return self.draws[-1]

    def randn(self, *shape):
        self.draws.append(self.g.randn(*shape))
        return self.draws[-1]


def _oracle_one_iteration(X, params, K, seed0, min_margin=5e-4):
    """One EM iteration + the Wiener chain of the oracle on a recorded numpy stream; the stream's seed is advanced until every
    accept / reject decision of both chains sits at least `min_margin` from its threshold (a replayed trajectory is only
    comparable while the decisions agree; the reference-generated goldens were selected the same way, at 1e-3; the HIP
    path's log-acceptances are within 1e-4 of the oracle's on these sizes)."""
    for k in range(200):
        o = orc.MCEMOracle("M1", 1, 10, 30, 25, 75, 0.01, reference_compat=True)
        r = _RecordingRNG(seed0 + 1000 * k)
        o.init_parameters(X, params, K, 1e-8, r)
        out = dict(o=o, W0=o.W.copy(), H0=o.H.copy(), Z0=o.Z.copy())
        ns, bi = o.e_step_counts()
        nw, bw = o.wf_counts()
        tr, p0 = [], len(r.draws)
        Zs = o.sample_posterior(o.Z, ns, bi, trace=tr)
        out["e_draws"], out["acc"], out["Zs"] = r.draws[p0:], np.stack([t["acc"] for t in tr]), Zs
        o.Z = Zs[:, -1, :].T.copy()
        o.compute_Vs(Zs); o.compute_Vs_scaled(); o.compute_Vx()
        o.M_step()
        out["W"], out["H"], out["g"], out["cost"] = o.W.copy(), o.H.copy(), o.g.copy(), float(o.compute_expected_neg_log_like())
        tr2, p1 = [], len(r.draws)
        Zw = o.sample_posterior(o.Z, nw, bw, trace=tr2)
        out["w_draws"] = r.draws[p1:]
        o.compute_Vs(Zw); o.compute_Vs_scaled(); o.compute_Vx()
        out["WFs"], out["WFn"] = o.compute_WF(sample=False)
        margins = [np.abs(np.log(d[2 * m + 1]) - t["acc"]).min() for d, trc in ((out["e_draws"], tr), (out["w_draws"], tr2)) for m, t in enumerate(trc)]
        if min(margins) > min_margin:
            return out
    raise AssertionError("no stream with clear decisions found")


def _replay_tensors(outs, key, S, dev):
    # (ascontiguousarray: stacking transposed views keeps THEIR memory order, and the C ABI takes dense row-major buffers)
    eps = np.ascontiguousarray(np.concatenate([np.stack([o[key][2 * m].T for m in range(S)]) for o in outs], 1))      # [S, NT, L]
    u = np.ascontiguousarray(np.concatenate([np.stack([o[key][2 * m + 1] for m in range(S)]) for o in outs], 1))      # [S, NT]
    return torch.from_numpy(eps).to(dev), torch.from_numpy(u).to(dev), u


@pytest.mark.parametrize("counts,prec", [([1], "bf16x3"), ([3], "bf16x3"), ([17], "bf16x3"), ([1, 2, 19, 16, 1], "bf16x3"), ([1, 2, 19, 16, 1], "bf16")])
def test_tiny_and_ragged_utterances_against_the_oracle(counts, prec):
    """Edge sizes: utterances of ONE frame, of fewer frames than a wavefront's 16, of 16 k + 1 and 16 k + 3 frames, alone and
    in one ragged batch -- one EM iteration (chain with every log-acceptance and decision, samples, M-step, cost) and the
    Wiener filter against the oracle run per utterance on the same draws (mcem.py:371-441, :90-152, :473-490).  bf16x3:
    the golden-run tolerances; bf16 (wave chain + sample store): finite, the same decisions where the margin is wide."""
    need_gpu()
    F, K, L = 65, 4, 32
    params = orc.xavier_normal_params([F, L, [128, 128]], seed=3, bias_std=0.05)
    gx = np.random.RandomState(11)
    Xs = [((gx.randn(n, F) + 1j * gx.randn(n, F)) * np.exp(0.5 * gx.randn(n, 1))).astype(np.complex64) for n in counts]
    outs = [_oracle_one_iteration(X, params, K, 100 + i) for i, X in enumerate(Xs)]
    ns, bi = outs[0]["o"].e_step_counts()
    nw, bw = outs[0]["o"].wf_counts()
    NT = sum(counts)
    eng = make_engine(params, F, K, counts, Rcap=max(ns, nw), precision=prec)
    dev = eng.device
    eng.set_spectrogram(Xs)
    eng.init_nmf([o["W0"] for o in outs], [o["H0"] for o in outs])
    eng.Z.zero_()
    eng.Z[:, :L].copy_(torch.from_numpy(np.ascontiguousarray(np.concatenate([o["Z0"].T for o in outs], 0))))
    S = ns + bi
    eps, u_d, u = _replay_tensors(outs, "e_draws", S, dev)
    acc_ref = np.concatenate([o["acc"] for o in outs], 1)
    acc = eng.mh_chain(ns, bi, 0.01, eps=eps, u=u_d, want_acc=True).cpu().numpy()
    assert acc.shape == (S, NT) and np.all(np.isfinite(acc))
    if prec != "bf16x3":
        # bf16 products: the trajectories part ways at the first narrow decision; until then the decisions agree
        wide = np.abs(np.log(u) - acc_ref) > 0.5
        first = np.argmax(~np.equal(np.log(u) < acc, np.log(u) < acc_ref), axis=0)        # per frame: first step that differs (0 if none)
        for n in range(NT):
            m = first[n] if (np.log(u[:, n]) < acc[:, n])[first[n]] != (np.log(u[:, n]) < acc_ref[:, n])[first[n]] else S
            assert m == S or not wide[m, n], (n, m)
        eng.m_step(ns)
        for t in (eng.W, eng.Ht, eng.g):
            assert bool(torch.isfinite(t).all())
        assert np.all(np.isfinite(eng.cost_from_frames(ns)))
        return
    assert np.max(np.abs(acc - acc_ref)) < 2e-3
    assert np.array_equal(np.log(u) < acc, np.log(u) < acc_ref)
    Zs_ref = np.concatenate([o["Zs"] for o in outs], 0)
    assert np.max(np.abs(eng.Zs[:, :ns, :L].cpu().numpy() - Zs_ref)) < 5e-6
    eng.m_step(ns)
    cost = eng.cost_from_frames(ns)
    off = np.concatenate([[0], np.cumsum(counts)])
    for i, o in enumerate(outs):
        sl = slice(off[i], off[i + 1])
        assert rel_err(eng.W[i, :F, :K].cpu().numpy(), o["W"]) < 5e-4
        assert rel_err(eng.Ht[sl, :K].cpu().numpy().T, o["H"]) < 5e-4
        assert rel_err(eng.g[sl].cpu().numpy(), o["g"]) < 5e-4
        assert abs(cost[i] - o["cost"]) / abs(o["cost"]) < 1e-4
    # Wiener filter with the samples of a second chain
    eps, u_d, _ = _replay_tensors(outs, "w_draws", nw + bw, dev)
    eng.mh_chain(nw, bw, 0.01, eps=eps, u=u_d, update_Z=False)
    Sh, Nh, WFs, WFn = eng.wiener(nw, want_masks=True)
    for i, o in enumerate(outs):
        sl = slice(off[i], off[i + 1])
        assert rel_err(WFs[sl, :F].cpu().numpy().T, o["WFs"]) < 5e-3
        sh = np.ascontiguousarray(Sh[sl, :F].cpu().numpy()).view(np.complex64).reshape(counts[i], F).T
        assert nrm_err(sh, o["WFs"] * o["o"].X) < 2e-3


@pytest.mark.parametrize("F,K,model,rng,zdim,hdim", [(257, 8, "M1", "device", 32, [128, 128]), (257, 8, "M2", "device", 32, [128, 128]),
                                                     (257, 8, "M1", "replay", 32, [128, 128]), (513, 10, "M1", "device", 32, [128, 128]),
                                                     (513, 10, "M2", "device", 32, [128, 128]), (513, 32, "M1", "replay", 32, [128, 128]),
                                                     # decoder shapes of section 3.8: one hidden layer (two barriers per evaluation), 16 latents
                                                     (257, 8, "M1", "device", 16, [128]), (513, 10, "M2", "replay", 16, [128])])
def test_four_wavefront_chain_equals_the_wave_chain_bit_for_bit(F, K, model, rng, zdim, hdim):
    """Small batches of the bench shape (at most one 16-frame wave tile per CU: one utterance through the drop-in classes)
    run wchain4_kernel -- four wavefronts per tile, each owning two of the output layer's eight bin-tile pairs, the pair
    energies exchanged through LDS and added in the one-wavefront kernel's order.  Same proposals, same log-acceptances,
    same decisions, same samples, same stored rows as wchain_kernel, bit for bit (mcem.py:371-441) -- ragged utterances
    (tiles of 16, 1, 5 frames), store on, a chain with burn-in and one without, then the M-step over both stores; the bench
    shape (17 bin tiles) and the reference scripts' 1024-pt STFT (33 bin tiles, four pairs per wavefront)."""
    need_gpu()
    from vaenmf import _lib
    R, BI = 30, 30
    counts = [33, 17, 5]
    NT = sum(counts)
    ydim = F if model == "M2" else 0
    params = orc.xavier_normal_params([F, zdim, hdim], seed=5, y_dim=ydim, bias_std=0.05)
    g = np.random.default_rng(7)
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (0.5 + 3 * np.exp(-np.arange(F) / 60.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    gains = (0.5 + g.random(NT)).astype(np.float32)
    Z0 = (0.5 * g.standard_normal((NT, 32))).astype(np.float32)
    Z0[:, zdim:] = 0.0
    y = (g.random((NT, F)) > 0.5).astype(np.float32)
    eps = g.standard_normal((R + BI, NT, 32)).astype(np.float32)
    eps[:, :, zdim:] = 0.0
    u = g.random((R + BI, NT)).astype(np.float32)

    def run(four):
        os.environ["VAENMF_WCHAIN4"] = "1" if four else "0"
        try:
            eng = make_engine(params, F, K, counts, Rcap=R, precision="bf16", seeds=[11, 12, 13])
            dev = eng.device
            eng.set_spectrogram(Xs)
            eng.init_nmf(W0, H0)
            eng.g.copy_(torch.from_numpy(gains))
            if model == "M2":
                eng.set_labels(torch.from_numpy(y))
            eng.Z.copy_(torch.from_numpy(Z0))
            eng.sample_store(True)
            kw = dict(eps=torch.from_numpy(eps).to(dev), u=torch.from_numpy(u).to(dev)) if rng == "replay" else dict(call=3)
            out = []
            acc = eng.mh_chain(R, BI, 0.01, want_acc=True, **kw)
            assert _lib.lib().vaenmf_plan_query(eng._plan, _lib.Q_CHAIN_KERNEL) == (2 if four else 1)
            out += [acc.cpu().numpy().copy(), eng.Zs[:, :R].cpu().numpy().copy(), eng.Z.cpu().numpy().copy(), eng.stored_variances(R).cpu().numpy().copy()]
            c = eng.m_step_stored().clone()
            out += [t.cpu().numpy().copy() for t in (eng.W, eng.Ht, eng.g, c)]
            # a chain without burn-in (slot R holds the initial state), Z not updated: the Wiener chain's form
            kw2 = dict(eps=kw["eps"][:R], u=kw["u"][:R]) if rng == "replay" else dict(call=4)
            eng.mh_chain(R, 0, 0.01, update_Z=False, **kw2)
            out += [eng.Zs[:, :R].cpu().numpy().copy(), eng.Z.cpu().numpy().copy(), eng.stored_variances(R).cpu().numpy().copy()]
            return out
        finally:
            os.environ.pop("VAENMF_WCHAIN4", None)

    a, b = run(True), run(False)
    names = ("acc", "Zs", "Z", "rows", "W", "Ht", "g", "cost", "Zs2", "Z2", "rows2")
    assert np.abs(a[1] - Z0[:, None, :]).max() > 0.05            # the chains moved
    for x, yv, name in zip(a, b, names):
        assert np.array_equal(x, yv), name


@pytest.mark.parametrize("R,counts", [(30, [33, 17, 5, 70, 64]), (10, [129, 1, 16]), (30, [501])])
def test_group_w_statistics_equal_the_tile_kernel_bit_for_bit(R, counts):
    """Small batches run wstats_group_kernel: one workgroup per 16-frame group, the four wavefronts compute the statistics of
    4 frames each side by side, the rank-K sums over frames are accumulated in the group's frame order, and the W update
    rebuilds every 64-frame tile from its groups in the order wstats_fused_kernel adds its four wavefronts (mcem.py:107-110).
    W, normalisation and -- through the H/g kernel -- H, g and the cost are bit-identical to the tile kernel's: ragged
    utterances with full tiles, tiles of 1, 2, 3 groups, groups of 1 / 5 / 6 frames."""
    need_gpu()
    from vaenmf import _lib
    F, K = 257, 8
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=5, bias_std=0.05)
    g = np.random.default_rng(R + len(counts))
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (0.5 + 3 * np.exp(-np.arange(F) / 60.0))).astype(np.complex64) for n in counts]
    W0 = [np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts]
    H0 = [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts]
    gains = (0.5 + g.random(sum(counts))).astype(np.float32)

    def run(group):
        os.environ["VAENMF_WGROUP"] = "1" if group else "0"
        try:
            eng = make_engine(params, F, K, counts, Rcap=R, precision="bf16", seeds=list(range(len(counts))))
            eng.set_spectrogram(Xs)
            eng.init_nmf(W0, H0)
            eng.g.copy_(torch.from_numpy(gains))
            eng.sample_store(True)
            out = []
            for it in range(2):                                      # the second iteration starts from the first one's W, H, g
                eng.mh_chain(R, 3, 0.01, call=it)
                c = eng.m_step_stored().clone()
                assert _lib.lib().vaenmf_plan_query(eng._plan, _lib.Q_W_FUSED) == (2 if group else 1)
                out += [t.cpu().numpy().copy() for t in (eng.W, eng.Ht, eng.g, c)]
            return out
        finally:
            os.environ.pop("VAENMF_WGROUP", None)

    a, b = run(True), run(False)
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), ("W", "Ht", "g", "cost")[i % 4] + " of iteration %d" % (i // 4)
    assert np.all(np.isfinite(a[0])) and float(np.abs(a[4] - a[0]).max()) > 0


@pytest.mark.parametrize("nfft,K,NU", [(512, 8, 24), (1024, 10, 48)])
def test_small_batch_kernels_give_the_large_batch_result(nfft, K, NU):
    """A batch with no more 16-frame wave tiles than CUs runs wchain4_kernel and wstats_group_kernel, a larger one
    wchain_kernel and wstats_fused_kernel.  The same utterance alone (small-batch kernels) and inside a batch of 24 / 48
    (large-batch kernels; checked through the plan queries): enhanced signals and cost bit for bit -- through the whole
    fused pipeline (STFT, device NMF initialisation, graph-replayed EM loop, Wiener filter, iSTFT)."""
    need_gpu()
    from vaenmf import _lib
    from vaenmf.pipeline import Reconstructor
    dev = torch.device("cuda:0")
    n_sms = torch.cuda.get_device_properties(0).multi_processor_count
    T = [20000 + 997 * (i % 11) for i in range(NU)]
    sig = [orc.synth_utterance(u, t) for u, t in enumerate(T)]
    F = nfft // 2 + 1
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
    rec = Reconstructor(params, F, K, niter=3, fs=16000, wlen_sec=nfft / 16000, precision="bf16", device=dev, max_frames=NU * 260, max_utts=NU)

    def run(idx):
        wav = torch.from_numpy(np.concatenate([sig[i][2] for i in idx]).astype(np.float32)).to(dev)
        s, n, c = rec.enhance(wav, [T[i] for i in idx], seeds=[100 + i for i in idx], init_seed=0)
        off = np.concatenate([[0], np.cumsum([T[i] for i in idx])])
        s = s.cpu().numpy()
        kern = _lib.lib().vaenmf_plan_query(rec.eng._plan, _lib.Q_CHAIN_KERNEL)
        tiles = _lib.lib().vaenmf_plan_query(rec.eng._plan, _lib.Q_WTILES)
        return {i: (s[off[k]:off[k + 1]], c[k].cpu().numpy()) for k, i in enumerate(idx)}, kern, tiles

    big, kb, tb = run(list(range(NU)))
    for i in (5, 17):
        one, k1, t1 = run([i])
        assert t1 <= n_sms and k1 == 2, (t1, k1)
        for x, y in zip(big[i], one[i]):
            assert np.array_equal(x, y), i
    if tb > n_sms:
        assert kb == 1, kb
    else:
        pytest.skip("this GPU has %d CUs: the %d-utterance batch (%d wave tiles) still counts as small" % (n_sms, NU, tb))
