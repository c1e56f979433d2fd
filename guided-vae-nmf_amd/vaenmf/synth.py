"""Seeded synthetic workload (no dataset on the box): speech-like bursts mixed with
coloured noise at SNR in {-5,0,5} dB following the mixing recipe of the reference's
scripts/create_test_set.py:74-103 (noise gain from the power ratio, normalisation by
max |s|,|n|,|s+n|).  Seeded Xavier-normal weights in the reference's state_dict layout
(python/models/models.py:136-140).  The test suite checks that these generators equal
the oracle's copies."""
import math

import numpy as np


def synth_utterance(seed, n_samples=64000, fs=16000):
    """Returns (s, n, x, snr_db), float64 arrays of length n_samples."""
    from scipy.signal import lfilter
    g = np.random.default_rng(1000 + seed)
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


def xavier_normal_params(dims, seed=0, y_dim=0, bias_std=0.0):
    x_dim, z_dim, h_dim = dims
    g = np.random.default_rng(seed)

    def lin(o, i):
        std = math.sqrt(2.0 / (i + o))
        w = (g.standard_normal((o, i)) * std).astype(np.float32)
        return w, (g.standard_normal(o) * bias_std).astype(np.float32)

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
    """Classifier([x_dim, h_dim, y_dim]) key layout (python/models/models.py:44-55)."""
    x_dim, h_dim, y_dim = dims
    g = np.random.default_rng(seed)
    p = {}
    neurons = [x_dim, *h_dim]
    for i in range(1, len(neurons)):
        std = math.sqrt(2.0 / (neurons[i] + neurons[i - 1]))
        p["hidden.%d.weight" % (i - 1)] = (g.standard_normal((neurons[i], neurons[i - 1])) * std).astype(np.float32)
        p["hidden.%d.bias" % (i - 1)] = (g.standard_normal(neurons[i]) * bias_std).astype(np.float32)
    std = math.sqrt(2.0 / (h_dim[-1] + y_dim))
    p["output_layer.weight"] = (g.standard_normal((y_dim, h_dim[-1])) * std).astype(np.float32)
    p["output_layer.bias"] = (g.standard_normal(y_dim) * bias_std).astype(np.float32)
    return p
