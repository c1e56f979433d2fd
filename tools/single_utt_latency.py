"""dev aid: seconds per utterance of the drop-in path (one 4 s utterance, MCEM_M1.init_parameters + run, device generator)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
import numpy as np, torch
import vaenmf
from vaenmf import stft as vstft
from vaenmf.synth import synth_utterance, xavier_normal_params
NFFT = int(os.environ.get("NFFT", "512"))
F, K, NITER = NFFT // 2 + 1, int(os.environ.get("RANK", "8")), 100
vae = vaenmf.VariationalAutoencoder([F, 32, [128, 128]])
vae.load_state_dict({k: torch.tensor(v) for k, v in xavier_normal_params([F, 32, [128, 128]], seed=0).items()})
x = synth_utterance(0)[2]
X = vstft.stft(x, fs=16000, wlen_sec=NFFT / 16000, hop_percent=0.25).T
m = vaenmf.MCEM_M1(niter=NITER, rng="device", precision=sys.argv[1] if len(sys.argv) > 1 else "bf16", fused_store=(os.environ.get("FUSED_STORE", "1") == "1"))
ts = []
for k in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.init_parameters(X=X, vae=vae, nmf_rank=K, eps=1e-8, device="cuda:0")
    c = m.run()
    ts.append(time.perf_counter() - t0)
print("seconds per utterance:", [round(t, 4) for t in ts], "final cost %.4f" % c[-1])
