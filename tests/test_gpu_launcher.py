"""bench.py's own launcher on real kernels: `--gpus 2` starts two ranks, each shards the fixed utterance set
(np.array_split, scripts/evaluate_M1.py:203), runs its batches through the HIP pipeline, the metric statistics are
all-reduced and rank 0 prints one line.  A one-GPU box cannot give each rank a device, so the rehearsal mode
(VAENMF_BENCH_SHARE_GPU=1 with --backend gloo: both ranks compute on GPU 0, the <1 KB collectives go through gloo on
host tensors) is used; the measured multi-GPU configuration is the same code with one device per rank and RCCL."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_strong_scaling_job_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, VAENMF_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--total-utts", "7", "--niter", "3",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-parity-mode", "--no-configs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2
    assert d["config"]["utterances_total"] == 7 and d["config"]["utterances_rank0"] == 4      # array_split(7, 2) = 4 + 3
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - 7 * 501 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]           # whole-job frames over the max-over-ranks time
    assert -40.0 < d["si_sdr_mean_db"] < 10.0                                                   # the all-reduced statistics of all 7 utterances
