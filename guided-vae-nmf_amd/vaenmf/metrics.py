"""SI-SDR / SI-SIR / SI-SAR (python/metrics.py:12-60) from float64 Gram sums computed
on the GPU (vaenmf_gram3_batch), the statistics table (metrics.py:5-10, 70-108) and
the sufficient statistics that are all-reduced over ranks."""
import numpy as np
import torch

from ._lib import check, lib
from .engine import _ptr, _stream

METRIC_KEYS = ("SI-SDR", "SI-SIR", "SI-SAR")


def ratios_from_gram(G):
    """G [...,6] = <sh,sh>,<sh,s>,<sh,n>,<s,s>,<s,n>,<n,n>  ->  (si_sdr, si_sir, si_sar) dB.
    s_target = a_s s, e_noise = a_n n with a_s = <sh,s>/<s,s>, a_n = <sh,n>/<n,n>
    (metrics.py:26-35), energies expanded through the Gram matrix."""
    G = np.asarray(G, dtype=np.float64)
    hh, hs, hn, ss, sn, nn = [G[..., i] for i in range(6)]
    a_s, a_n = hs / ss, hn / nn
    e_t = a_s ** 2 * ss                                            # |s_target|^2
    e_n = a_n ** 2 * nn                                            # |e_noise|^2
    e_r = hh - 2 * a_s * hs + e_t                                  # |sh - s_target|^2 = |e_noise + e_art|^2
    e_a = hh + e_t + e_n - 2 * a_s * hs - 2 * a_n * hn + 2 * a_s * a_n * sn   # |e_art|^2
    return 10 * np.log10(e_t / e_r), 10 * np.log10(e_t / e_n), 10 * np.log10(e_t / e_a)


_SOFF = {}


def gram3_batch_device(s_hat, s, n, sample_counts):
    """Device float32 [sum T] x3 -> DEVICE float64 [U,6] (no synchronisation: a job can collect the Gram sums of many
    batches and read them back once)."""
    dev = s_hat.device
    key = (tuple(int(t) for t in sample_counts), str(dev))
    soff = _SOFF.get(key)
    if soff is None:
        if len(_SOFF) >= 16:
            _SOFF.pop(next(iter(_SOFF)))
        soff = _SOFF[key] = torch.tensor(np.concatenate([[0], np.cumsum(sample_counts)]), dtype=torch.int64, device=dev)
    out = torch.empty(len(sample_counts), 6, device=dev, dtype=torch.float64)
    check(lib().vaenmf_gram3_batch(_ptr(s_hat), _ptr(s), _ptr(n), len(sample_counts), _ptr(soff), _ptr(out), _stream()))
    return out


def gram3_batch(s_hat, s, n, sample_counts):
    """Device float32 [sum T] x3 -> numpy float64 [U,6]."""
    return gram3_batch_device(s_hat, s, n, sample_counts).cpu().numpy()


def energy_ratios(s_hat, s, n):
    """Reference signature (metrics.py:39): numpy time signals -> (si_sdr, si_sir, si_sar)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    G = gram3_batch(t(s_hat), t(s), t(n), [len(s)])
    r = ratios_from_gram(G[0])
    return float(r[0]), float(r[1]), float(r[2])


def mean_confidence_interval(data, confidence=0.95, round=3):
    """metrics.py:5-10."""
    import scipy.stats
    a = 1.0 * np.array(data)
    n = len(a)
    m, se = np.mean(a), scipy.stats.sem(a)
    h = se * scipy.stats.t.ppf((1 + confidence) / 2., n - 1)
    return np.round(m, 3), np.round(h, 3)


def sufficient_stats(values, snr_db, snr_bins=(-5.0, 0.0, 5.0)):
    """values [U,3] (SI-SDR,SI-SIR,SI-SAR), snr_db [U] -> float64 [(1+len(bins)),3,3] of
    (count, sum, sum of squares): what the ranks all-reduce (sum) so that rank 0 can print
    mean +- t-CI overall and per input SNR (metrics.py:70-108)."""
    values = np.asarray(values, dtype=np.float64).reshape(-1, 3)
    snr_db = np.asarray(snr_db, dtype=np.float64)
    out = np.zeros((1 + len(snr_bins), 3, 3))
    groups = [np.ones(len(values), bool)] + [snr_db == b for b in snr_bins]
    for gi, m in enumerate(groups):
        v = values[m]
        out[gi, :, 0] = len(v)
        out[gi, :, 1] = v.sum(0)
        out[gi, :, 2] = (v ** 2).sum(0)
    return out


def stats_table(st, snr_bins=(-5.0, 0.0, 5.0), confidence=0.95):
    """mean and t-CI half width per metric from all-reduced sufficient statistics."""
    import scipy.stats
    rows = {}
    for gi, name in enumerate(["all"] + ["snr=%g" % b for b in snr_bins]):
        for k, key in enumerate(METRIC_KEYS):
            n, s1, s2 = st[gi, k]
            if n < 1:
                continue
            m = s1 / n
            h = float("nan")
            if n > 1:
                var = max((s2 - n * m * m) / (n - 1), 0.0)
                h = np.sqrt(var / n) * scipy.stats.t.ppf((1 + confidence) / 2., n - 1)
            rows[(name, key)] = (np.round(m, 3), np.round(h, 3), int(n))
    return rows


def compute_stats(metrics_keys, all_metrics, all_snr_db, model_data_dir=None, confidence=0.95):
    """metrics.py:70-108: prints the same table (overall, then per input SNR)."""
    metrics = {key: [j[i] for j in all_metrics] for i, key in enumerate(metrics_keys)}
    print("{:<10} {:<10} {:<10}".format('METRIC', 'AVERAGE', 'CONF. INT.'))
    for key, metric in metrics.items():
        m, h = mean_confidence_interval(metric, confidence=confidence)
        print("{:<10} {:<10} {:<10}".format(key, m, h))
    print('\n')
    all_snr_db = np.asarray(all_snr_db)
    for snr_db in np.unique(all_snr_db):
        print('Input SNR = {:.2f}'.format(snr_db))
        print("{:<10} {:<10} {:<10}".format('METRIC', 'AVERAGE', 'CONF. INT.'))
        for key, metric in metrics.items():
            subset = np.array(metric)[np.where(all_snr_db == snr_db)]
            m, h = mean_confidence_interval(subset, confidence=confidence)
            print("{:<10} {:<10} {:<10}".format(key, m, h))
        print('\n')
