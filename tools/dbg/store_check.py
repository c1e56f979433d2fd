import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from vaenmf.engine import BatchEngine
from vaenmf.synth import xavier_normal_params
from vaenmf.engine import decoder_params_from_state
F, K = 257, 8
params = xavier_normal_params([F, 32, [128, 128]], seed=0)
sd = {k: torch.as_tensor(v) for k, v in params.items()}
g = np.random.default_rng(8)
for counts, ns, bi in (([94] * 6, 30, 30), ([37, 64, 70], 10, 3), ([94] * 6, 12, 30), ([94]*6, 30, 3)):
    eng = BatchEngine(F, K, decoder_params_from_state(sd), precision="bf16", max_frames=4096, max_utts=16)
    eng.bind(counts, Rcap=max(ns, 75), seeds=list(range(len(counts))))
    Xs = [((g.standard_normal((n, F)) + 1j * g.standard_normal((n, F))) * (1 + 3 * np.exp(-np.arange(F) / 40.0))).astype(np.complex64) for n in counts]
    eng.set_spectrogram(Xs)
    eng.init_nmf([np.maximum(g.random((F, K)), 1e-8).astype(np.float32) for _ in counts], [np.maximum(g.random((K, n)), 1e-8).astype(np.float32) for n in counts])
    eng.sample_store(True)
    eng.mh_chain(ns, bi, 0.01, call=2)
    got = eng.stored_variances(ns)[:, :, :F].cpu().numpy()
    ref = eng.decode(ns)[:, :, :F].cpu().numpy()
    rel = np.abs(got - ref) / ref
    bad = np.argwhere(~(rel < 0.01))
    print(counts, ns, bi, "max rel", np.nanmax(rel), "n bad", len(bad), "nan", np.isnan(got).sum())
    if len(bad):
        print(" bad frames", np.unique(bad[:, 0])[:40], "\n bad samples", np.unique(bad[:, 1]), "\n bad bins", np.unique(bad[:, 2])[:64])
        n, r, f = bad[0]
        print(" first:", n, r, f, got[n, r, f], ref[n, r, f])
