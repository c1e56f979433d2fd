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
