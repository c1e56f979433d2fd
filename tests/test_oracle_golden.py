"""CPU: pin the numpy oracle against golden vectors produced by the reference
itself (tests/golden/make_golden.py).  Tolerances are float32 round-off of two
different BLAS/libm stacks (torch vs numpy): 2e-5 relative on variances."""
import os

import numpy as np
import pytest

import vaenmf_oracle as orc
from helpers import load_case, rel_err, nrm_err, GOLDEN

CASES = [("m1_f65", "M1"), ("m2_vad_f65", "M2"), ("m2_ibm_f65", "M2"), ("m1_f257", "M1"),
         ("m1_f65_z16", "M1"), ("m1_f65_h128", "M1"), ("m2_vad_f65_z16_h128", "M2")]      # decoder shapes of scripts/evaluate_M1.py:44-85: z_dim 16, h_dim [128]


def run_oracle(name, model):
    z, params, draws, meta = load_case(name)
    nsE, biE, nsW, biW = meta["counts"]
    m = orc.MCEMOracle(model, meta["niter"], nsE, biE, nsW, biW, 0.01, reference_compat=True)
    rng = orc.ReplayRNG(draws)
    y = z["y"] if model == "M2" else None
    m.init_parameters(z["X"], params, meta["K"], 1e-8, rng, y=y)
    return z, m, rng


@pytest.mark.parametrize("name,model", CASES)
def test_init_matches_reference(name, model):
    z, m, _ = run_oracle(name, model)
    assert rel_err(m.W, z["W0"]) == 0 and rel_err(m.H, z["H0"]) == 0
    assert np.max(np.abs(m.Z - z["Z0"])) < 2e-5


@pytest.mark.parametrize("name,model", CASES)
def test_first_iteration_steps(name, model):
    """E-step (MH log-acceptance of every step, decisions, Vs, Vx) then M-step."""
    z, m, rng = run_oracle(name, model)
    trace = []
    ns, bi = m.e_step_counts()
    Zs = m.sample_posterior(m.Z, ns, bi, trace=trace)
    n1 = int(z["E1_nacc"])
    assert len(trace) == n1 == ns + bi
    acc = np.stack([t["acc"] for t in trace])
    assert np.max(np.abs(acc - z["acc"][:n1])) < 2e-3 * max(1.0, np.abs(z["acc"][:n1]).max() * 1e-2)
    m.Z = Zs[:, -1, :].T.copy()
    m.compute_Vs(Zs); m.compute_Vs_scaled(); m.compute_Vx()
    assert np.max(np.abs(m.Z - z["E1_Z"])) < 1e-5
    assert rel_err(m.Vs, z["E1_Vs"]) < 2e-5
    assert rel_err(m.Vx, z["E1_Vx"]) < 2e-5
    m.M_step()
    for k, v in (("M1_W", m.W), ("M1_H", m.H), ("M1_g", m.g), ("M1_Vb", m.Vb), ("M1_Vx", m.Vx)):
        assert rel_err(v, z[k]) < 5e-5, k


@pytest.mark.parametrize("name,model", CASES)
def test_full_run(name, model):
    z, m, rng = run_oracle(name, model)
    cost = m.run()
    assert rng.pos == len(rng.draws)          # same number and order of random draws
    assert cost.dtype == np.float64 and cost.shape == (m.niter,)
    assert np.max(np.abs(cost - z["cost"]) / np.abs(z["cost"])) < 2e-5
    assert tuple(m.Vs.shape) == tuple(z["Vs_shape"])
    for k, v in (("W", m.W), ("H", m.H), ("g", m.g)):
        assert rel_err(v, z[k]) < 2e-4, k
    assert np.max(np.abs(m.Z - z["Z"])) < 2e-5
    assert rel_err(m.WFs, z["WFs"]) < 2e-4 and rel_err(m.WFn, z["WFn"]) < 2e-4
    assert nrm_err(m.S_hat, z["S_hat"]) < 1e-5 and nrm_err(m.N_hat, z["N_hat"]) < 1e-5
    assert m.S_hat.dtype == np.complex64


def test_quirk_counts():
    """mcem.py:371 vs :461-462/:477-478 -- M1's positional shift."""
    q = np.load(GOLDEN + "/quirk_counts.npz")
    m1 = orc.MCEMOracle("M1", 1)
    m2 = orc.MCEMOracle("M2", 1)
    (r, b), (rw, bw) = m1.e_step_counts(), m1.wf_counts()
    assert (r + b, r, rw + bw, rw) == tuple(q["M1"])
    (r, b), (rw, bw) = m2.e_step_counts(), m2.wf_counts()
    assert (r + b, r, rw + bw, rw) == tuple(q["M2"])
    assert orc.MCEMOracle("M1", 1, reference_compat=False).e_step_counts() == (10, 30)


def test_mlp_forward():
    z = np.load(GOLDEN + "/mlp_forward.npz")
    for tag in ("m1", "m2"):
        p = {k.split(":p:")[1]: z[k] for k in z.files if k.startswith(tag + ":p:")}
        zz, mu, lv = orc.encoder_forward(p, z[tag + "_x"], z[tag + "_eps"])
        assert np.max(np.abs(mu - z[tag + "_mu"])) < 1e-5
        assert np.max(np.abs(lv - z[tag + "_lv"])) < 1e-5
        assert np.max(np.abs(zz - z[tag + "_zz"])) < 1e-5
        assert rel_err(orc.decoder_forward(p, z[tag + "_z"]), z[tag + "_dec"]) < 1e-5
    p = {k.split(":p:")[1]: z[k] for k in z.files if k.startswith("clf:p:")}
    assert np.max(np.abs(orc.classifier_forward(p, z["clf_x"]) - z["clf_y"])) < 1e-6


def test_energy_ratios_known_answer():
    """python/metrics.py:39-60 on the reference-committed dummy-M2 wavs; the
    reference's own figure title for 440c020a reads -6.2 / -4.3 / -1.9 dB."""
    z = np.load(GOLDEN + "/metrics_dummy_m2.npz")
    r = orc.energy_ratios(z["a_s_est"] / 32768.0, z["a_s"] / 32768.0, z["a_n"] / 32768.0)
    assert np.allclose(r, z["a_ratios"], rtol=0, atol=1e-9)
    assert [round(float(v), 1) for v in r] == [-6.2, -4.3, -1.9]


def test_stft_roundtrip_and_shapes():
    """stft.py:48-53 end-pad rule + librosa framing; istft(stft(x)) == x."""
    z = np.load(GOLDEN + "/metrics_dummy_m2.npz")
    x = z["a_s"] / 32768.0
    X = orc.stft(x, fs=16000, wlen_sec=64e-3, hop_percent=0.25)
    assert X.dtype == np.complex64 and X.shape[0] == 513
    T = len(x)
    need_pad = int(np.ceil(T / 16000 / 64e-3 / 0.25)) != int(T / 16000 / 64e-3 / 0.25)
    assert X.shape[1] == 1 + (T + (256 if need_pad else 0)) // 256
    xr = orc.istft(X, fs=16000, wlen_sec=64e-3, hop_percent=0.25, max_len=T)
    assert xr.dtype == np.float32 and len(xr) == T
    assert np.max(np.abs(xr - x)) < 1e-6
    Xa = orc.stft(np.zeros(64000), fs=16000, wlen_sec=32e-3)
    assert Xa.shape == (257, 501)
    with pytest.raises(ValueError):
        orc.stft(x, fs=16000, wlen_sec=50.01e-3)


def test_nonmf_variant_full_run():
    """MCEM_M2_noNMF (mcem.py:606-760): fixed noise variance, gain-only M-step (:543-578)."""
    z, params, draws, meta = load_case("m2_nonmf_f65")
    nsE, biE, nsW, biW = meta["counts"]
    rng = orc.ReplayRNG(draws)
    m = orc.MCEMOracleNoNMF(z["X"], z["Vb"], z["g0"], z["Z0"], z["y"], params, meta["niter"], rng, nsE, biE, nsW, biW, 0.01)
    cost = m.run()
    assert rng.pos == len(rng.draws)
    assert np.max(np.abs(cost - z["cost"]) / np.abs(z["cost"])) < 2e-5
    assert rel_err(m.g, z["g"]) < 2e-4 and np.max(np.abs(m.Z - z["Z"])) < 2e-5
    assert nrm_err(m.S_hat, z["S_hat"]) < 1e-5 and nrm_err(m.N_hat, z["N_hat"]) < 1e-5


def test_label_front_ends_against_reference():
    """python/processing/target.py:7-116 restated with explicit float32 operation order (pairwise total,
    running cumsum, fma power) -- the order the HIP kernels reproduce -- against the reference's outputs."""
    z = np.load(os.path.join(GOLDEN, "labels_f257.npz"))
    for u in (0, 1):
        S, N = z["S%d" % u], z["N%d" % u]
        assert np.array_equal(orc.power_c64(S), abs(S * S.conj()))
        assert np.array_equal(orc.clean_speech_IBM(S, 0.999, 0.999), z["ibm%d" % u])
        assert np.array_equal(orc.clean_speech_IBM(S), z["ibm98_%d" % u])
        assert np.array_equal(orc.clean_speech_VAD(S, 0.999, 0.999), z["vad%d" % u])
        assert np.array_equal(orc.clean_speech_VAD(S), z["vad98_%d" % u])
        assert np.array_equal(orc.noise_robust_clean_speech_VAD(S), z["nrvad%d" % u])
        assert np.array_equal(orc.noise_robust_clean_speech_IBM(S), z["nribm%d" % u])
        assert np.array_equal(orc.ideal_wiener_mask(S, N), z["iwm%d" % u])
    g = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 127, 128, 129, 257, 1000, 24415):
        a = (g.random(n) ** 4 * 1000).astype(np.float32)
        assert orc.pairwise_sum_f32(a) == np.sum(a), n


def test_spp_estimator_against_reference():
    """python/models/spp_estimation.py:163-235 restated (all bins per frame at once) against the reference's outputs."""
    z = np.load(os.path.join(GOLDEN, "spp_f257.npz"))
    for u in (0, 1):
        P = z["P%d" % u]
        assert np.array_equal(orc.timo_mask_estimation(P), z["mask%d" % u])
        assert np.array_equal(orc.timo_vad_estimation(P), z["vad%d" % u])
        assert np.array_equal(orc.timo_noise_estimation(P, z["mask%d" % u]), z["psd%d" % u])


def _stft_fixture_power():
    """|STFT|^2 of the reference's raw utterance 440c020a as tests/dataset/test_csr1_wjs0_dataset.py:40-66 prepares it
    (drop the first 0.1 s, divide by the peak), with the oracle's STFT; and the reference's own values for it."""
    z = np.load(os.path.join(GOLDEN, "stft_frames.npz"))
    x = z["pcm_a"].astype(np.float64) / 32768.0                     # sf.read of 16-bit PCM
    x = x[int(0.1 * 16000):]
    x = x / np.max(np.abs(x))
    return z, x


def test_stft_against_reference_known_answer():
    """STFT pinned against the reference's OWN committed output: data/subset/pickle/CSR-1-WSJ-0/si_et_05_frames.p
    (extracted without unpickling by tests/golden/extract_ref_pickles.py) holds |stft(x)|^2 produced by the reference
    (python/processing/stft.py:16-63 -> librosa).  First / last 96 frames (reflect padding at both ends, end-pad rule)
    and the per-frame sums over the bins of every frame."""
    z, x = _stft_fixture_power()
    X = orc.stft(x, fs=16000, wlen_sec=64e-3, win="hann", hop_percent=0.25)
    P = np.power(np.abs(X), 2)
    n0 = int(z["frame_counts"][0])
    assert P.shape == (513, n0)
    scale = np.max(z["head"])
    assert np.max(np.abs(P[:, :96] - z["head"])) < 2e-7 * scale and np.max(np.abs(P[:, n0 - 96:] - z["tail"])) < 2e-7 * scale
    big = z["head"] > 1e-6 * scale
    assert np.max(np.abs(P[:, :96][big] / z["head"][big] - 1)) < 5e-6
    assert np.max(np.abs(P.sum(0, dtype=np.float64) / z["col_sums"] - 1)) < 1e-6


@pytest.mark.parametrize("name,model", CASES)
def test_torch_cpu_restatement_full_run(name, model):
    """oracle/vaenmf_torch_cpu.py (the PyTorch-CPU program timed as bench.py's cpu_baseline) against the
    reference-recorded runs: same draws in the same order, same cost trajectory, same outputs."""
    import torch
    import vaenmf_torch_cpu as tc
    torch.set_num_threads(1)
    z, params, draws, meta = load_case(name)
    nsE, biE, nsW, biW = meta["counts"]
    m = tc.TorchMCEM(model, meta["niter"], nsE, biE, nsW, biW, 0.01, reference_compat=True)
    rng = tc.ReplayDraws(draws)
    m.init_parameters(z["X"], params, meta["K"], 1e-8, rng, y=z["y"] if model == "M2" else None)
    assert np.max(np.abs(m.Z.numpy() - z["Z0"])) < 2e-5
    cost = m.run()
    assert rng.pos == len(rng.draws)
    assert np.max(np.abs(cost - z["cost"]) / np.abs(z["cost"])) < 2e-5
    for k, v in (("W", m.W), ("H", m.H), ("g", m.g)):
        assert rel_err(v.numpy(), z[k]) < 2e-4, k
    assert nrm_err(m.S_hat, z["S_hat"]) < 1e-5 and nrm_err(m.N_hat, z["N_hat"]) < 1e-5


def _check_stft_tr_dt(stft_fn, tol_rel):
    """The reference's two other golden STFT outputs (si_tr_s_frames.p (513, 972), si_dt_05_frames.p (513, 976): same test
    recipe, tests/dataset/test_csr1_wjs0_dataset.py:17-83): first two utterances of each set."""
    z = np.load(os.path.join(GOLDEN, "stft_frames_tr_dt.npz"))
    for tag in ("tr", "dt"):
        cn = z[tag + "_frame_counts"]
        for which, n_fr in (("a", int(cn[0])), ("b", int(cn[1]))):
            x = z["%s_pcm_%s" % (tag, which)].astype(np.float64) / 32768.0
            x = x[int(0.1 * 16000):]
            x = x / np.max(np.abs(x))
            P = np.power(np.abs(stft_fn(x)), 2)
            assert P.shape == (513, n_fr), (tag, which, P.shape)
            if which == "a":
                head, tail, cs = z[tag + "_head"], z[tag + "_tail"], z[tag + "_col_sums"]
                scale = np.max(head)
                assert np.max(np.abs(P[:, :48] - head)) < 2e-7 * scale and np.max(np.abs(P[:, n_fr - 48:] - tail)) < 2e-7 * max(scale, np.max(tail))
                big = head > 1e-6 * scale
                assert np.max(np.abs(P[:, :48][big] / head[big] - 1)) < tol_rel
            else:
                fb, cs = z[tag + "_first_b"], z[tag + "_col_sums_b"]
                assert np.max(np.abs(P[:, :4] - fb)) < 2e-7 * max(np.max(fb), 1e-30) + 1e-12
            assert np.max(np.abs(P.sum(0, dtype=np.float64) / cs - 1)) < 1e-6


def test_stft_against_the_reference_train_and_validation_known_answers():
    _check_stft_tr_dt(lambda x: orc.stft(x, fs=16000, wlen_sec=64e-3, win="hann", hop_percent=0.25), 5e-6)


def test_classifier_batch_norm_and_two_class_forwards():
    """Classifier(batch_norm=True) in eval mode and Classifier2Classes (models.py:41-88) against forwards of the reference
    classes (tests/golden/mlp_forward_variants.npz): the oracle's restatement, the mirror torch modules (same state_dict
    keys), and the folded layer list the HIP path runs (vaenmf.engine.classifier_layers_from_state) evaluated in numpy."""
    import torch
    import vaenmf
    from vaenmf.engine import classifier_layers_from_state
    z = np.load(os.path.join(GOLDEN, "mlp_forward_variants.npz"))
    pb = {k.split(":p:")[1]: z[k] for k in z.files if k.startswith("bn:p:")}
    p2 = {k.split(":p:")[1]: z[k] for k in z.files if k.startswith("c2:p:")}
    x = z["x"]
    assert np.max(np.abs(orc.classifier_forward(pb, x) - z["bn_y"])) < 1e-6
    assert np.max(np.abs(orc.classifier_forward(p2, x, two_classes=True) - z["c2_y"])) < 1e-6
    m = vaenmf.Classifier([129, [128, 128], 5], batch_norm=True)
    m.load_state_dict({k: torch.tensor(v) for k, v in pb.items()}); m.eval()
    m2 = vaenmf.Classifier2Classes([129, [128, 128], 5])
    m2.load_state_dict({k: torch.tensor(v) for k, v in p2.items()}); m2.eval()
    with torch.no_grad():
        assert np.max(np.abs(m(torch.tensor(x)).numpy() - z["bn_y"])) < 1e-6
        assert np.max(np.abs(m2(torch.tensor(x)).numpy() - z["c2_y"])) < 1e-6

    def run(layers):
        h = x.astype(np.float64)
        for w, b in layers[:-1]:
            h = np.maximum(h @ w.astype(np.float64).T + b, 0)
        w, b = layers[-1]
        return 1 / (1 + np.exp(-(h @ w.astype(np.float64).T + b)))
    assert np.max(np.abs(run(classifier_layers_from_state(pb)) - z["bn_y"])) < 2e-6
    assert np.max(np.abs(run(classifier_layers_from_state(p2, two_classes=True)) - z["c2_y"][:, 0, :])) < 2e-6
