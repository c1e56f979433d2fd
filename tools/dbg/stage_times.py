"""dev aid: host/GPU time of each stage of Reconstructor.enhance for one 64-utterance batch (synchronised between stages)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
import numpy as np, torch
from vaenmf.pipeline import Reconstructor
from vaenmf import stft as vstft, metrics as vmet
from vaenmf.synth import synth_utterance, xavier_normal_params
dev = torch.device("cuda:0")
U, T, F = 64, 64000, 257
base = [synth_utterance(k) for k in range(16)]
wav = torch.from_numpy(np.concatenate([base[i % 16][2] for i in range(U)]).astype(np.float32)).to(dev)
ws = torch.from_numpy(np.concatenate([base[i % 16][0] for i in range(U)]).astype(np.float32)).to(dev)
wn = torch.from_numpy(np.concatenate([base[i % 16][1] for i in range(U)]).astype(np.float32)).to(dev)
rec = Reconstructor(xavier_normal_params([F, 32, [128, 128]], seed=0), F, 8, niter=100, wlen_sec=32e-3, precision="bf16", device=dev, max_frames=U * 520, max_utts=U)
rec.enhance(wav, [T] * U, seeds=list(range(U)))
torch.cuda.synchronize()
eng = rec.eng
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); print("%-28s %7.2f ms" % (name, (t - t0) * 1e3)); return t
for rep in range(2):
    t = time.perf_counter(); t00 = t
    X, fc = vstft.stft_batch(wav, [T] * U, rec.fs, rec.wlen_sec, rec.hop_percent, Fs=eng.Fs, device=dev); t = tick("stft_batch", t)
    import vaenmf.engine as E
    _z = torch.zeros
    def zz(*a, **k):
        t0 = time.perf_counter(); r = _z(*a, **k); torch.cuda.synchronize(); print("      zeros%s %.2f ms" % (tuple(a), (time.perf_counter() - t0) * 1e3)); return r
    torch.zeros = zz
    t0 = time.perf_counter()
    eng.bind(fc, Rcap=max(rec.nsE, rec.nsW), seeds=list(range(U)))
    torch.zeros = _z
    t = tick("bind (+allocs)", t)
    eng.set_spectrogram(X); t = tick("set_spectrogram", t)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    eng.W.zero_(); eng.W[:, :F, :8] = torch.rand(eng.U, F, 8, device=dev, generator=gen).clamp_min(1e-8)
    eng.Ht.zero_(); eng.Ht[:, :8] = torch.rand(eng.NT, 8, device=dev, generator=gen).clamp_min(1e-8); eng.g.fill_(1.0); t = tick("W/H init", t)
    eng.encode(rec.enc, None); t = tick("encode", t)
    t1 = time.perf_counter()
    cost, S, N = eng.run(rec.niter, rec.nsE, rec.biE, rec.nsW, rec.biW, rec.var_RW); print("   (em_run host enqueue %.2f ms)" % ((time.perf_counter() - t1) * 1e3)); t = tick("em_run", t)
    nfft, hop = vstft.frame_geometry(T, rec.fs, rec.wlen_sec, rec.hop_percent)[:2]
    s_hat = vstft.istft_batch(S, fc, [T] * U, nfft, hop, device=dev); n_hat = vstft.istft_batch(N, fc, [T] * U, nfft, hop, device=dev); t = tick("istft x2", t)
    G = vmet.gram3_batch(s_hat, ws, wn, [T] * U); t = tick("gram3 + D2H", t)
    print("total %.2f ms" % ((t - t00) * 1e3))
