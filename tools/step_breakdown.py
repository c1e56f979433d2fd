"""dev aid: wall-clock breakdown of one bench step (host + device) by pipeline stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
import numpy as np, torch
from vaenmf.pipeline import Reconstructor
from vaenmf import stft as vstft, metrics as vmet
from vaenmf.synth import synth_utterance, xavier_normal_params
dev = torch.device("cuda:0")
U, T, F = 64, 64000, 257
sig = [synth_utterance(i % 16) for i in range(U)]
wav = torch.from_numpy(np.concatenate([s[2] for s in sig]).astype(np.float32)).to(dev)
ws = torch.from_numpy(np.concatenate([s[0] for s in sig]).astype(np.float32)).to(dev)
wn = torch.from_numpy(np.concatenate([s[1] for s in sig]).astype(np.float32)).to(dev)
rec = Reconstructor(xavier_normal_params([F, 32, [128, 128]], seed=0), F, 8, niter=100, wlen_sec=32e-3, device=dev,
                    max_frames=U * 520, max_utts=U, precision="bf16")
counts = [T] * U
rec.enhance(wav, counts)
torch.cuda.synchronize()
def tick(label, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); print("%-28s %8.3f ms" % (label, (t - t0) * 1e3)); return t
for rep in range(2):
    eng = rec.eng
    t = time.perf_counter(); t00 = t
    X, fc = vstft.stft_batch(wav, counts, rec.fs, rec.wlen_sec, rec.hop_percent, Fs=eng.Fs, device=dev); t = tick("stft_batch", t)
    eng.bind(fc, Rcap=max(rec.nsE, rec.nsW), seeds=list(range(U))); t = tick("bind", t)
    eng.set_spectrogram(X); t = tick("set_spectrogram", t)
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    eng.W.zero_(); eng.W[:, :F, :8] = torch.rand(U, F, 8, device=dev, generator=gen).clamp_min(1e-8)
    eng.Ht.zero_(); eng.Ht[:, :8] = torch.rand(eng.NT, 8, device=dev, generator=gen).clamp_min(1e-8); eng.g.fill_(1.0); t = tick("init W,H,g", t)
    eng.encode(rec.enc, None); t = tick("encode", t)
    cost, S, N = eng.run(rec.niter, rec.nsE, rec.biE, rec.nsW, rec.biW, rec.var_RW); t = tick("em_run", t)
    nfft, hop = vstft.frame_geometry(T, rec.fs, rec.wlen_sec, rec.hop_percent)[:2]
    s_hat = vstft.istft_batch(S, fc, counts, nfft, hop, device=dev); n_hat = vstft.istft_batch(N, fc, counts, nfft, hop, device=dev); t = tick("istft x2", t)
    G = vmet.gram3_batch(s_hat, ws, wn, counts); t = tick("gram3 + D2H", t)
    print("total %.3f ms" % ((t - t00) * 1e3))
