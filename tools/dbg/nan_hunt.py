import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
import numpy as np, torch
from vaenmf.pipeline import Reconstructor
from vaenmf.synth import synth_utterance, xavier_normal_params
dev = torch.device("cuda:0")
U, T, F = 6, 12000, 257
wav = torch.from_numpy(np.concatenate([synth_utterance(i)[2][:T] for i in range(U)]).astype(np.float32)).to(dev)
for prec in ("bf16", "bf16x3"):
    for store in (True, False):
        for niter in (1, 2, 3, 12):
            rec = Reconstructor(xavier_normal_params([F, 32, [128, 128]], seed=0), F, 8, niter=niter, wlen_sec=32e-3, device=dev,
                                max_frames=U * 520, max_utts=U, precision=prec, store=store)
            s, n, cost = rec.enhance(wav, [T] * U, seeds=list(range(U)))
            c = cost.cpu().numpy()
            eng = rec.eng
            print(prec, "store", store, "niter", niter, "cost[:,last]", np.round(c[:, -1], 4), "nan in W/H/g/Z:",
                  [bool(torch.isnan(t).any()) for t in (eng.W, eng.Ht, eng.g, eng.Z)], "s nan", bool(torch.isnan(s).any()))
