"""dev aid: acceptance statistics of the MH chain on the bench workload (distinct samples per frame)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
import numpy as np, torch
from vaenmf.pipeline import Reconstructor
from vaenmf.synth import synth_utterance, xavier_normal_params
dev = torch.device("cuda:0")
U, T, F = 8, 64000, 257
wav = torch.from_numpy(np.concatenate([synth_utterance(i)[2] for i in range(U)]).astype(np.float32)).to(dev)
rec = Reconstructor(xavier_normal_params([F, 32, [128, 128]], seed=0), F, 8, niter=int(sys.argv[1]) if len(sys.argv) > 1 else 20,
                    wlen_sec=32e-3, device=dev, max_frames=U * 520, max_utts=U)
s, n, cost = rec.enhance(wav, [T] * U)
Zs = rec.eng.Zs[:, :rec.nsW].cpu().numpy()          # samples of the Wiener chain
same = np.all(Zs[:, 1:] == Zs[:, :-1], axis=2)
print("WF chain: fraction of steps without a move: %.3f ; mean distinct samples per frame: %.1f of %d"
      % (same.mean(), (1 + (~same).sum(1)).mean(), rec.nsW))
print("cost first/last:", cost[0, 0].item(), cost[0, -1].item())
