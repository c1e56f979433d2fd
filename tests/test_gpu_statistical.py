"""End-to-end statistical parity with the on-device generator (no replay): the MH sampler is
chaotic, so trajectories cannot match the CPU path bit for bit; what must match is the
distribution of the outcome.  Short synthetic utterances go through the WHOLE pipeline
(STFT -> EM -> Wiener -> iSTFT -> SI-SDR) on the GPU in both precision modes and through the
numpy oracle with several seeds each.

Stated tolerances: mean SI-SDR over (utterances x seeds): |GPU - oracle| <= 0.05 dB + 3 standard
errors of the oracle's own seed-to-seed spread; final EM cost per utterance (mean over seeds):
2e-3 relative (bf16: 1e-2) + 3 standard errors of the oracle's seed-to-seed spread of that mean."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import vaenmf_oracle as orc

F, K, NITER, FS, WLEN = 257, 8, 12, 16000, 32e-3
UTTS, SEEDS, T = 6, 3, 12000


def oracle_run(x, s, n, params, seed):
    X = orc.stft(x, fs=FS, wlen_sec=WLEN).T
    m = orc.MCEMOracle("M1", NITER)
    m.init_parameters(X, params, K, 1e-8, orc.NumpyRNG(seed))
    cost = m.run()
    s_hat = orc.istft(m.S_hat, fs=FS, wlen_sec=WLEN, max_len=len(x))
    return orc.energy_ratios(s_hat.astype(np.float64), s, n)[0], cost[-1]


def test_si_sdr_and_cost_distribution_match_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vaenmf.pipeline import Reconstructor
    from vaenmf import metrics as vm
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
    sig = [orc.synth_utterance(u, T) for u in range(UTTS)]
    ref = np.array([[oracle_run(sig[u][2], sig[u][0], sig[u][1], params, 100 * u + sd) for sd in range(SEEDS)] for u in range(UTTS)])
    dev = torch.device("cuda:0")
    to_dev = lambda i: torch.from_numpy(np.concatenate([sg[i] for sg in sig]).astype(np.float32)).to(dev)
    wav_x, wav_s, wav_n = to_dev(2), to_dev(0), to_dev(1)
    res = {}
    # "bf16": the bench path (sample-variance store with bf16 rows feeding the M-step); "bf16 decode": the same
    # precision with the M-step decoding the samples again
    for name, prec, store in (("bf16x3", "bf16x3", None), ("bf16", "bf16", None), ("bf16 decode", "bf16", False)):
        rec = Reconstructor(params, F, K, niter=NITER, fs=FS, wlen_sec=WLEN, precision=prec, device=dev,
                            max_frames=UTTS * 120, max_utts=UTTS, store=store)
        out = []
        for sd in range(SEEDS):
            s_hat, n_hat, cost = rec.enhance(wav_x, [T] * UTTS, seeds=[1000 * sd + u for u in range(UTTS)], init_seed=sd)
            G = vm.gram3_batch(s_hat, wav_s, wav_n, [T] * UTTS)
            out.append(np.stack([vm.ratios_from_gram(G)[0], cost[:, -1].cpu().numpy()], 1))
        res[name] = np.stack(out, 1)                       # [U, SEEDS, 2]
    sem = ref[:, :, 0].std(1, ddof=1).mean() / np.sqrt(UTTS * SEEDS)
    csem = ref[:, :, 1].std(1, ddof=1) / np.abs(ref[:, :, 1].mean(1)) * np.sqrt(2.0 / SEEDS)   # rel. s.e. of a difference of means
    for prec, ctol in (("bf16x3", 2e-3), ("bf16", 1e-2), ("bf16 decode", 1e-2)):
        d_sdr = res[prec][:, :, 0].mean() - ref[:, :, 0].mean()
        d_cost = np.abs(res[prec][:, :, 1].mean(1) / ref[:, :, 1].mean(1) - 1)
        print("%s: mean SI-SDR gpu %.3f dB, oracle %.3f dB (diff %.3f, oracle seed sem %.3f); rel cost diff per utt %s (seed s.e. %s)"
              % (prec, res[prec][:, :, 0].mean(), ref[:, :, 0].mean(), d_sdr, sem, np.round(d_cost, 4), np.round(csem, 4)))
        assert abs(d_sdr) <= 0.05 + 3 * sem
        assert np.all(d_cost <= ctol + 3 * csem)
