import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
import numpy as np, torch, ctypes as C
from vaenmf.engine import BatchEngine, decoder_params_from_state
from vaenmf._lib import lib, check
from vaenmf.synth import xavier_normal_params
dev = torch.device("cuda:0")
F, U = 257, 64
sd = {k: torch.as_tensor(v) for k, v in xavier_normal_params([F, 32, [128, 128]], seed=0).items()}
eng = BatchEngine(F, 8, decoder_params_from_state(sd), precision="bf16", device=dev, max_frames=U * 520, max_utts=U)
fc = [501] * U
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.bind(fc, Rcap=75, seeds=list(range(U)))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    off = eng.frame_off; sdn = np.arange(U, dtype=np.uint64)
    check(lib().vaenmf_bind_batch(eng._plan, U, off.ctypes.data, sdn.ctypes.data))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    a = torch.zeros(eng.NT, 75, 32, device=dev); torch.cuda.synchronize(); t3 = time.perf_counter()
    b = torch.zeros(eng.NT, eng.Fs, 2, device=dev); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("bind %.2f ms | C vaenmf_bind_batch %.2f ms | zeros Zs %.2f ms | zeros X %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
