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
Utterances of different length are batched together (ragged frame counts)."""
import glob
import os

import numpy as np
import torch

from . import wavio
from .pipeline import Reconstructor, shard

_SUBDIR = {"train": "si_tr_s/", "validation": "si_dt_05/", "test": "si_et_05/"}


def speech_list(input_speech_dir, dataset_type="train"):
    data_dir = input_speech_dir + "CSR-1-WSJ-0/WAV/wsj0/" + _SUBDIR.get(dataset_type, "")
    paths = sorted(glob.glob(data_dir + "**/*.wav", recursive=True))
    return [os.path.relpath(p, input_speech_dir) for p in paths]


def evaluate(rec: Reconstructor, file_paths, processed_data_dir, output_data_dir, batch_size=64,
             world_size=1, rank=0, classifier=None, mean=None, std=None, seed=0):
    """Enhance this rank's shard of `file_paths`; returns the list of written (s_est, n_est) paths."""
    mine = shard(file_paths, world_size, rank)
    written = []
    for b0 in range(0, len(mine), batch_size):
        files = mine[b0:b0 + batch_size]
        wavs, counts = [], []
        for fp in files:
            x, fs = wavio.read(processed_data_dir + os.path.splitext(fp)[0] + "_x.wav")
            if fs != rec.fs:
                raise ValueError("Unexpected sampling rate")
            wavs.append(x)
            counts.append(len(x))
        wav = torch.from_numpy(np.concatenate(wavs).astype(np.float32)).to(rec.device)
        seeds = [seed * 1000003 + (b0 + i) for i in range(len(files))]
        s_hat, n_hat, _ = rec.enhance(wav, counts, seeds=seeds, init_seed=seed + b0, classifier=classifier, mean=mean, std=std)
        s_hat, n_hat = s_hat.cpu().numpy(), n_hat.cpu().numpy()
        off = np.concatenate([[0], np.cumsum(counts)])
        foff = np.concatenate([[0], np.cumsum(rec.frame_counts)])
        for i, fp in enumerate(files):
            out = os.path.splitext(output_data_dir + fp)[0]
            os.makedirs(os.path.dirname(out), exist_ok=True)
            wavio.write(out + "_s_est.wav", s_hat[off[i]:off[i + 1]], rec.fs)
            wavio.write(out + "_n_est.wav", n_hat[off[i]:off[i + 1]], rec.fs)
            if classifier is not None:
                torch.save(rec.y_soft[foff[i]:foff[i + 1]].cpu(), out + " _ibm_soft_est.pt")
                torch.save(rec.y_hard[foff[i]:foff[i + 1]].cpu(), out + "_ibm_hard_est.pt")
            written.append((out + "_s_est.wav", out + "_n_est.wav"))
    return written
