"""dev aid: per-phase tick shares of wchain_kernel (workgroup 0, wave 0) from a -DVN_STAMP build (libvaenmf_dbg.so)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
from vaenmf import _lib
_lib.LIB_PATH = os.path.join(ROOT, "guided-vae-nmf_amd", "vaenmf", "libvaenmf_%s.so" % (sys.argv[2] if len(sys.argv) > 2 else "dbg"))
import numpy as np, torch
from vaenmf.pipeline import Reconstructor
from vaenmf.synth import synth_utterance, xavier_normal_params
dev = torch.device("cuda:0")
U, T, F = 64, 64000, 257
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
wav = torch.from_numpy(np.concatenate([synth_utterance(i % 8)[2] for i in range(U)]).astype(np.float32)).to(dev)
rec = Reconstructor(xavier_normal_params([F, 32, [128, 128]], seed=0), F, 8, niter=6, wlen_sec=32e-3, device=dev,
                    max_frames=U * 520, max_utts=U, precision=prec)
rec.enhance(wav, [T] * U)
torch.cuda.synchronize()
lib = _lib.lib()
lib.vaenmf_debug_stamps_wc.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_longlong * 32)()
lib.vaenmf_debug_stamps_wc(buf, 1)
names = ["rng+proposal", "layer 1", "layer 2", "layer 3 + energy", "stores + loop end", "sum over q of E", "pr, log u sums", "decide + select"]
tot = sum(buf[i] for i in range(8))
for i, nme in enumerate(names):
    print("%-20s %12d ticks  %5.1f%%  (%d visits, %.0f ticks/visit)" % (nme, buf[i], 100.0 * buf[i] / max(tot, 1), buf[16 + i], buf[i] / max(buf[16 + i], 1)))
print("sum per evaluation: %.0f ticks (s_memtime)" % (tot / max(buf[16 + 3], 1)))
