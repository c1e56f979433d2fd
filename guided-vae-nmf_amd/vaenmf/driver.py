"""File-tree driver: the build's counterpart of scripts/evaluate_M1.py:111-222 and
scripts/evaluate_M2_vad.py (process_utt / process_sublist / main), batched.

  file list     speech_list(input_speech_dir, dataset_type): sorted recursive glob of
                CSR-1-WSJ-0/WAV/wsj0/{si_tr_s,si_dt_05,si_et_05}/**/*.wav, paths relative to the input dir
                (python/dataset/csr1_wjs0_dataset.py:19-54)
  sharding      np.array_split(file_paths, world_size)[rank]               (evaluate_M1.py:203-206)
  per utterance read <processed>/<rel>_x.wav, STFT, MCEM, iSTFT with max_len = T_orig,
                write <output>/<rel>_s_est.wav and _n_est.wav             (evaluate_M1.py:114-166)
  M2            with a classifier, the soft and hard labels of every utterance are saved as
                '<rel> _ibm_soft_est.pt' (sic: the reference's file name has the blank) and
                '<rel>_ibm_hard_est.pt', (frames, label dim) tensors      (evaluate_M2_vad.py:165-166)
  label sources (scripts/evaluate_M2_ibm.py:121-141, `classif_type`): 'dnn' -- the classifier on the (optionally
                normalised) power spectrogram, hard labels = soft > 0.5; 'oracle' -- clean_speech_IBM / clean_speech_VAD of
                <processed>/<rel>_s.wav (hard = soft, as the reference does); 'timo' -- the SPP estimator on |X|^2,
                hard = soft > 0.5.  All on the device, for the whole batch.
Utterances of different length are batched together (ragged frame counts)."""
import glob
import os

import numpy as np
import torch

from . import wavio
from . import stft as vstft
from .pipeline import Reconstructor, shard

_SUBDIR = {"train": "si_tr_s/", "validation": "si_dt_05/", "test": "si_et_05/"}


def speech_list(input_speech_dir, dataset_type="train"):
    data_dir = input_speech_dir + "CSR-1-WSJ-0/WAV/wsj0/" + _SUBDIR.get(dataset_type, "")
    paths = sorted(glob.glob(data_dir + "**/*.wav", recursive=True))
    return [os.path.relpath(p, input_speech_dir) for p in paths]


def _read_batch(files, processed_data_dir, suffix, fs):
    wavs, counts = [], []
    for fp in files:
        x, fs_x = wavio.read(processed_data_dir + os.path.splitext(fp)[0] + suffix)
        if fs_x != fs:
            raise ValueError("Unexpected sampling rate")
        wavs.append(x)
        counts.append(len(x))
    return wavs, counts


def evaluate(rec: Reconstructor, file_paths, processed_data_dir, output_data_dir, batch_size=64,
             world_size=1, rank=0, classifier=None, mean=None, std=None, seed=0, label_source="dnn", label_type="ibm",
             quantile_fraction=0.999, quantile_weight=0.999):
    """Enhance this rank's shard of `file_paths`; returns the list of written (s_est, n_est) paths.
    M2: labels from `label_source` ('dnn' needs `classifier`; 'oracle' reads the clean speech; 'timo' the SPP
    estimator), `label_type` 'ibm' (y_dim F) or 'vad' (y_dim 1); M1 ignores them."""
    from . import target as vtarget
    from . import spp_estimation as vspp
    if label_source not in ("dnn", "oracle", "timo"):
        raise ValueError("label_source must be 'dnn', 'oracle' or 'timo'")
    m2 = rec.model == "M2"
    mine = shard(file_paths, world_size, rank)
    written = []
    for b0 in range(0, len(mine), batch_size):
        files = mine[b0:b0 + batch_size]
        wavs, counts = _read_batch(files, processed_data_dir, "_x.wav", rec.fs)
        wav = torch.from_numpy(np.concatenate(wavs).astype(np.float32)).to(rec.device)
        seeds = [seed * 1000003 + (b0 + i) for i in range(len(files))]
        y = y_soft = None
        if m2 and label_source == "oracle":                  # evaluate_M2_ibm.py:132-134
            swavs, scounts = _read_batch(files, processed_data_dir, "_s.wav", rec.fs)
            if scounts != counts:
                raise ValueError("clean speech and mixture differ in length")
            S, fc = vstft.stft_batch(torch.from_numpy(np.concatenate(swavs).astype(np.float32)).to(rec.device), counts, rec.fs,
                                     rec.wlen_sec, rec.hop_percent, Fs=rec.eng.Fs, device=rec.device)
            y_soft = vtarget.lorenz_labels_batch(S, fc, rec.F, label_type, quantile_fraction, quantile_weight)
            y_soft = y_soft.reshape(S.shape[0], -1)
            y = y_soft
        elif m2 and label_source == "timo":                  # evaluate_M2_ibm.py:136-141
            X, fc = vstft.stft_batch(wav, counts, rec.fs, rec.wlen_sec, rec.hop_percent, Fs=rec.eng.Fs, device=rec.device)
            P = (torch.view_as_complex(X.contiguous()).abs() ** 2) if not X.is_complex() else X.abs() ** 2
            if label_type == "vad":
                y_soft = vspp.spp_batch(P[:, :rec.F].sum(1, keepdim=True).contiguous(), fc, 1)
            else:
                y_soft = vspp.spp_batch(P.contiguous(), fc, rec.F)
            y = (y_soft > 0.5).float()
        elif m2 and classifier is None:
            raise ValueError("label_source='dnn' needs a classifier")
        s_hat, n_hat, _ = rec.enhance(wav, counts, seeds=seeds, init_seed=seed, y=y,
                                      classifier=classifier if (m2 and label_source == "dnn") else None, mean=mean, std=std)
        if m2 and label_source == "dnn":
            y_soft, y = rec.y_soft, rec.y_hard
        s_hat, n_hat = s_hat.cpu().numpy(), n_hat.cpu().numpy()
        off = np.concatenate([[0], np.cumsum(counts)])
        foff = np.concatenate([[0], np.cumsum(rec.frame_counts)])
        for i, fp in enumerate(files):
            out = os.path.splitext(output_data_dir + fp)[0]
            os.makedirs(os.path.dirname(out), exist_ok=True)
            wavio.write(out + "_s_est.wav", s_hat[off[i]:off[i + 1]], rec.fs)
            wavio.write(out + "_n_est.wav", n_hat[off[i]:off[i + 1]], rec.fs)
            if m2:
                torch.save(y_soft[foff[i]:foff[i + 1]].cpu(), out + " _ibm_soft_est.pt")
                torch.save(y[foff[i]:foff[i + 1]].cpu(), out + "_ibm_hard_est.pt")
            written.append((out + "_s_est.wav", out + "_n_est.wav"))
    return written
