"""dev aid: per-phase cycle shares of decode_kernel<MODE_HG> from a -DVN_STAMP build (libvaenmf_dbg.so)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
from vaenmf import _lib
_lib.LIB_PATH = os.path.join(ROOT, "guided-vae-nmf_amd", "vaenmf", "libvaenmf_dbg.so")
import numpy as np, torch
from vaenmf.pipeline import Reconstructor
from vaenmf.synth import synth_utterance, xavier_normal_params
dev = torch.device("cuda:0")
U, T, F = 64, 64000, 257
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
wav = torch.from_numpy(np.concatenate([synth_utterance(i % 8)[2] for i in range(U)]).astype(np.float32)).to(dev)
rec = Reconstructor(xavier_normal_params([F, 32, [128, 128]], seed=0), F, 8, niter=4, wlen_sec=32e-3, device=dev,
                    max_frames=U * 520, max_utts=U, precision=prec)
rec.enhance(wav, [T] * U)
torch.cuda.synchronize()
lib = _lib.lib()
lib.vaenmf_debug_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_longlong * 64)()
lib.vaenmf_debug_stamps(buf, 1)
names = ["pre-decode", "decode", "stageA", "H partial sums+shuffles", "barrier H", "hn+store", "vb2+stageB+shuffles", "barrier g", "cost", "barrier c"]
tot = sum(buf[i] for i in range(10))
for i, nme in enumerate(names):
    print("%-26s %12d cycles  %5.1f%%  (%d visits, %.0f cyc/visit)" % (nme, buf[i], 100.0 * buf[i] / max(tot, 1), buf[32 + i], buf[i] / max(buf[32 + i], 1)))
