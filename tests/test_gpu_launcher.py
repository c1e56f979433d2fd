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


_RCCL_ONE_RANK = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "guided-vae-nmf_amd"))
from vaenmf.pipeline import allreduce_stats
from vaenmf import metrics as vm
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # exactly bench.py's call
assert dist.get_backend() == "nccl"
g = np.random.default_rng(5)
st = vm.sufficient_stats(g.normal(2, 3, (11, 3)), g.choice([-5.0, 0.0, 5.0], 11))
tot = allreduce_stats(np.stack([st, 2 * st]), dev)                             # float64 SUM on a device tensor through RCCL
assert tot.dtype == np.float64 and np.array_equal(tot[0], st) and np.array_equal(tot[1], 2 * st)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                       # the max-over-ranks clock
assert float(t.item()) == 1.25
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def test_rccl_allreduce_of_the_metric_statistics_on_one_rank():
    """The job's only collective through the backend the measured configuration uses: init_process_group("nccl",
    device_id=cuda:0) with world_size 1, allreduce_stats on a float64 DEVICE tensor, the MAX reduction of the clock, the
    barrier.  Proves librccl loads and float64 SUM / MAX run on this image before the 8-GPU run depends on it."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_bench_one_rank_under_an_external_launcher_uses_rccl():
    """bench.py as the driver starts it (WORLD_SIZE / RANK / MASTER_* from the environment), one rank, backend nccl: the
    process group is initialised and the statistics all-reduce goes through RCCL on the device."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29742",
               VAENMF_BENCH_FORCE_PG="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--utts", "4", "--niter", "3", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-parity-mode", "--no-configs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["collective_backend"] == "nccl" and d["value"] > 0
