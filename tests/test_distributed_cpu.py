"""Multi-process path on CPU (gloo, world_size 2): utterance sharding + the final
metrics all-reduce, the only collective of the job."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vaenmf import metrics as vm
    from vaenmf.pipeline import shard, allreduce_stats
    ids = list(range(11))                                  # 11 utterances over 2 ranks: 6 + 5
    mine = shard(ids, world, rank)
    g = np.random.default_rng(123)
    vals_all = g.normal(2, 3, (11, 3))
    snr_all = g.choice([-5.0, 0.0, 5.0], 11)
    st = vm.sufficient_stats(vals_all[mine], snr_all[mine])
    tot = allreduce_stats(st, "cpu")
    t = torch.tensor([float(len(mine))])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)               # the bench's max-over-ranks pattern
    q.put((rank, mine, tot, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_allreduce_world2():
    sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
    from vaenmf import metrics as vm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 1, 2, 3, 4, 5] and res[1][1] == [6, 7, 8, 9, 10]
    g = np.random.default_rng(123)
    vals_all = g.normal(2, 3, (11, 3))
    snr_all = g.choice([-5.0, 0.0, 5.0], 11)
    ref = vm.sufficient_stats(vals_all, snr_all)
    for r in res:
        assert np.allclose(r[2], ref) and r[3] == 6.0
    tab = vm.stats_table(res[0][2])
    assert tab[("all", "SI-SDR")][2] == 11


def _run_bench(extra, env=None):
    import json
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, ([json.loads(l) for l in lines]), p.stderr


def test_bench_launcher_starts_the_ranks_itself():
    """`bench.py --gpus N` must run N ranks or fail: here N=2 over gloo on the CPU, in the rehearsal mode that does
    everything but the hot path (spawn with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*, the fixed 1000-utterance set split
    like scripts/evaluate_M1.py:203, batches of <= 64, the statistics all-reduce, one JSON line from rank 0)."""
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--backend", "gloo"])
    assert rc == 0, err[-2000:]
    assert len(out) == 1
    o = out[0]
    assert o["n_gpus"] == 2 and o["scaling"] == "strong" and o["utterances_total"] == 1000 and o["utterances_rank0"] == 500
    assert o["batches_rank0"] == [63, 63, 63, 63, 62, 62, 62, 62] and o["stats_count"] == 1000
    # 8 ranks of the real run: 125 utterances per GPU in waves of 63 + 62
    sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
    from vaenmf.pipeline import shard
    mine = shard(list(range(1000)), 8, 3)
    assert len(mine) == 125 and mine[0] == 375
    assert [len(b) for b in np.array_split(np.asarray(mine), 2)] == [63, 62]


def test_bench_refuses_a_world_size_that_disagrees():
    rc, out, err = _run_bench(["--gpus", "8", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and not out and "disagrees" in err


def test_bench_single_rank_dry_run():
    rc, out, err = _run_bench(["--dry-run"])
    assert rc == 0 and out[0]["n_gpus"] == 1 and out[0]["scaling"] == "weak" and out[0]["utterances_total"] == 64


def test_launcher_fails_fast_when_a_rank_dies_before_the_collective():
    """Rank 1 exits with code 3 before init_process_group: rank 0 would wait in the rendezvous for the collective's own
    timeout (minutes); the launcher must notice the dead rank, stop rank 0 and return its code within seconds."""
    import time
    t0 = time.monotonic()
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--backend", "gloo"], env={"VAENMF_BENCH_TEST_FAULT": "1:exit3"})
    dt = time.monotonic() - t0
    assert rc == 3 and not out, (rc, out, err[-1000:])
    assert "rank 1 exited with code 3" in err
    assert dt < 60, dt


def test_launcher_wall_clock_limit_stops_a_hung_rank():
    import time
    t0 = time.monotonic()
    rc, out, err = _run_bench(["--gpus", "2", "--dry-run", "--backend", "gloo"],
                              env={"VAENMF_BENCH_TEST_FAULT": "0:hang", "VAENMF_BENCH_RANK_TIMEOUT": "20"})
    dt = time.monotonic() - t0
    assert rc == 124 and not out, (rc, out, err[-1000:])
    assert dt < 90, dt
