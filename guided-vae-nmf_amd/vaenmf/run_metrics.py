"""The statistics harness: the build's counterpart of scripts/run_metrics_M1.py:63-176 (compute_metrics_utt / main)
for the metrics that are in scope -- SI-SDR, SI-SIR, SI-SAR (python/metrics.py:12-60); ESTOI / PESQ / POLQA and the
figures stay external (third-party packages).

  per utterance   read <processed>/<rel>_s.wav, _n.wav, _x.wav and <model dir>/<rel>_s_est.wav, the float64 Gram
                  sums of (s_est, s, n) on the device (vaenmf_gram3_batch), closed-form ratios
  table           compute_stats (metrics.py:70-108): mean +- t-CI overall and per input SNR; the SNR list is what
                  read_dataset(processed_data_dir, dataset_type, 'snr_db') returns in the reference (run_metrics_M1.py:149)
"""
import os

import numpy as np
import torch

from . import wavio
from . import metrics as vmet


def compute_metrics(file_paths, processed_data_dir, model_data_dir, device="cuda:0", batch_size=64):
    """-> list of [si_sdr, si_sir, si_sar] per utterance, in file order (run_metrics_M1.py:63-98)."""
    out = []
    for b0 in range(0, len(file_paths), batch_size):
        files = file_paths[b0:b0 + batch_size]
        sig = {k: [] for k in ("s", "n", "e")}
        counts = []
        for fp in files:
            stem = os.path.splitext(fp)[0]
            s, fs = wavio.read(processed_data_dir + stem + "_s.wav")
            n, _ = wavio.read(processed_data_dir + stem + "_n.wav")
            e, fs_e = wavio.read(model_data_dir + stem + "_s_est.wav")
            if fs_e != fs or not (len(s) == len(n) == len(e)):
                raise ValueError("length / rate mismatch for " + fp)
            sig["s"].append(s); sig["n"].append(n); sig["e"].append(e)
            counts.append(len(s))
        t = lambda l: torch.from_numpy(np.concatenate(l).astype(np.float32)).to(device)
        G = vmet.gram3_batch(t(sig["e"]), t(sig["s"]), t(sig["n"]), counts)
        r = np.stack(vmet.ratios_from_gram(G), 1)
        out += [list(map(float, row)) for row in r]
    return out


def main(file_paths, processed_data_dir, model_data_dir, all_snr_db, confidence=0.95, device="cuda:0"):
    """run_metrics_M1.py:147-176: per-utterance metrics, then the printed table; returns (all_metrics, sufficient statistics)."""
    all_metrics = compute_metrics(file_paths, processed_data_dir, model_data_dir, device)
    vmet.compute_stats(metrics_keys=list(vmet.METRIC_KEYS), all_metrics=all_metrics, all_snr_db=np.asarray(all_snr_db),
                       model_data_dir=model_data_dir, confidence=confidence)
    bins = tuple(float(b) for b in np.unique(np.asarray(all_snr_db, dtype=np.float64)))
    return all_metrics, vmet.sufficient_stats(np.asarray(all_metrics), all_snr_db, snr_bins=bins)
