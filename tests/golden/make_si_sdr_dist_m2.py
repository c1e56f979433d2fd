#!/usr/bin/env python3
"""Reference-generated outcome distribution of the GUIDED model (MCEM_M2, BASELINE config 3) for
tests/test_gpu_statistical.py -- the counterpart of make_si_sdr_dist.py (MCEM_M1).

IMPORTS THE REFERENCE (python.models.mcem.MCEM_M2, python.models.models.DeepGenerativeModel, python.metrics.
energy_ratios -- build container only, it never travels) and runs its unmodified EM over U short synthetic
utterances x S seeds of torch's global generator with IBM labels (y_dim = F) computed from the clean signal by the
oracle's clean_speech_IBM (the labels are an INPUT of the path; the reference's own target.py is pinned separately by
labels_f257.npz).  The oracle's STFT / iSTFT stand in for librosa, as in make_si_sdr_dist.py.
Committed output (data only): tests/golden/si_sdr_dist_m2.npz -- per (utterance, seed) SI-SDR / SI-SIR / SI-SAR (dB) and
the final EM cost, the labels, the configuration.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_si_sdr_dist_m2.py [threads]
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np
import torch

import vaenmf_oracle as orc
from python.models import mcem as ref_mcem
from python.models import models as ref_models
from python import metrics as ref_metrics

F, K, NITER, FS, WLEN = 257, 8, 20, 16000, 32e-3
UTTS, SEEDS, T = 8, int(os.environ.get("SI_SDR_SEEDS", "192")), 16000


def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0, y_dim=F)
    vae = ref_models.DeepGenerativeModel([F, F, 32, [128, 128]], None)
    vae.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    vae.eval()
    for p in vae.parameters():
        p.requires_grad = False
    out = np.zeros((UTTS, SEEDS, 4))
    labels = []
    t0 = time.time()
    for u in range(UTTS):
        s, n, x, _ = orc.synth_utterance(u, T)
        X = orc.stft(x, fs=FS, wlen_sec=WLEN).T                      # (N, F) complex64
        S = orc.stft(s, fs=FS, wlen_sec=WLEN)                        # (F, N)
        y = (orc.clean_speech_IBM(S, 0.999, 0.999) > 0.5).astype(np.float32).T        # (N, F) hard IBM of the clean speech
        labels.append(y)
        yt = torch.from_numpy(y)
        for sd in range(SEEDS):
            torch.manual_seed(10000 * u + sd)
            m = ref_mcem.MCEM_M2(niter=NITER, nsamples_E_step=10, burnin_E_step=30, nsamples_WF=25, burnin_WF=75, var_RW=0.01)
            m.init_parameters(X=X, y=yt, vae=vae, nmf_rank=K, eps=1e-8, device="cpu")
            cost = m.run()
            s_hat = orc.istft(m.S_hat, fs=FS, wlen_sec=WLEN, max_len=len(x))
            sdr, sir, sar = ref_metrics.energy_ratios(s_hat=s_hat.astype(np.float64), s=s, n=n)
            out[u, sd] = (sdr, sir, sar, float(cost[-1]))
        print("utt %d: SI-SDR %.3f +- %.3f (seed std), cost %.5f +- %.5f   [%.0f s]"
              % (u, out[u, :, 0].mean(), out[u, :, 0].std(ddof=1), out[u, :, 3].mean(), out[u, :, 3].std(ddof=1), time.time() - t0), flush=True)
    sem = np.sqrt(np.sum(out[:, :, 0].var(1, ddof=1) / SEEDS)) / UTTS
    print("mean SI-SDR %.4f dB, s.e. of the mean (seed spread) %.4f dB" % (out[:, :, 0].mean(), sem))
    np.savez_compressed(os.path.join(HERE, "si_sdr_dist_m2.npz"), results=out, labels=np.stack(labels).astype(np.uint8), F=F, K=K, niter=NITER,
                        fs=FS, wlen=WLEN, utts=UTTS, seeds=SEEDS, T=T, columns=np.array(["si_sdr_db", "si_sir_db", "si_sar_db", "final_cost"]))


if __name__ == "__main__":
    main()
