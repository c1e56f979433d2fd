"""dev aid (GPU box): where one bench step spends HOST time (each phase closed with a device synchronize)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "guided-vae-nmf_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import numpy as np, torch
import vaenmf_oracle as orc
from vaenmf.pipeline import Reconstructor
from vaenmf import stft as vstft, metrics as vmet
dev = torch.device("cuda:0")
F, K, U, T = 257, 8, 64, 64000
params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
rec = Reconstructor(params, F, K, niter=100, fs=16000, wlen_sec=0.032, precision="bf16", device=dev, max_frames=U * 520, max_utts=U)
g = np.random.default_rng(0)
wav = torch.from_numpy(g.standard_normal(U * T).astype(np.float32) * 0.1).to(dev)
counts = [T] * U
def sync(): torch.cuda.synchronize()
for it in range(4):
    t = {}
    sync(); t0 = time.perf_counter()
    eng = rec.eng
    X, fc = vstft.stft_batch(wav, counts, rec.fs, rec.wlen_sec, rec.hop_percent, Fs=eng.Fs, device=dev); sync(); t1 = time.perf_counter(); t["stft"] = t1 - t0
    eng.bind(fc, Rcap=max(rec.nsE, rec.nsW), seeds=list(range(U))); sync(); t2 = time.perf_counter(); t["bind"] = t2 - t1
    eng.set_spectrogram(X); sync(); t3 = time.perf_counter(); t["set_spec"] = t3 - t2
    gen = torch.Generator(device=dev); gen.manual_seed(it)
    eng.W.zero_(); eng.W[:, :F, :K] = torch.rand(eng.U, F, K, device=dev, generator=gen).clamp_min(1e-8)
    eng.Ht.zero_(); eng.Ht[:, :K] = torch.rand(eng.NT, K, device=dev, generator=gen).clamp_min(1e-8); eng.g.fill_(1.0); sync(); t4 = time.perf_counter(); t["init"] = t4 - t3
    eng.encode(rec.enc, None); sync(); t5 = time.perf_counter(); t["encode"] = t5 - t4
    cost, S, N = eng.run(rec.niter, rec.nsE, rec.biE, rec.nsW, rec.biW, rec.var_RW, store=rec.store); t6q = time.perf_counter(); sync(); t6 = time.perf_counter(); t["run_queue"] = t6q - t5; t["run_total"] = t6 - t5
    nfft, hop = vstft.frame_geometry(counts[0], rec.fs, rec.wlen_sec, rec.hop_percent)[:2]
    s_hat = vstft.istft_batch(S, fc, counts, nfft, hop, device=dev); n_hat = vstft.istft_batch(N, fc, counts, nfft, hop, device=dev); sync(); t7 = time.perf_counter(); t["istft"] = t7 - t6
    G = vmet.gram3_batch(s_hat, wav, wav, counts); r = vmet.ratios_from_gram(G); sync(); t8 = time.perf_counter(); t["gram"] = t8 - t7
    print("iter", it, {k: round(v * 1e3, 2) for k, v in t.items()}, "total", round((t8 - t0) * 1e3, 2), flush=True)
