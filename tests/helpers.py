"""Shared helpers for the tests (fixture loading)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    params = {k[2:]: z[k] for k in z.files if k.startswith("p:")}
    draws = [z[k] for k in sorted(k for k in z.files if k[0] == "d" and k[1:].isdigit())]
    F, N, K, L, Dy, niter, nsE, biE, nsW, biW = [int(v) for v in z["meta"]]
    meta = dict(F=F, N=N, K=K, L=L, Dy=Dy, niter=niter, counts=(nsE, biE, nsW, biW))
    return z, params, draws, meta


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-30)))


def nrm_err(a, b):
    a = np.asarray(a).astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b).astype(a.dtype)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
