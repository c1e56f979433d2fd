#!/usr/bin/env python3
"""bench.py -- STFT-frames/s through the VAE-NMF reconstruct loop (BASELINE.json metric).

A "step" is one pass of the whole hot path over synthetic utterances resident in HBM:
waveforms -> STFT -> |X|^2 -> encoder -> 100 x (MH E-step, M-step) -> Wiener chain + filter -> iSTFT ->
SI-SDR sufficient statistics -> all-reduce of the metric statistics over ranks (the job's only collective).

  N = 1 (default)   BASELINE.json configs[1]: one 64-utterance batch per step (64 x 4 s @16 kHz, 512-pt STFT,
                    F=257, 501 frames each, M1, NMF rank 8, 100 EM iterations, reference-faithful MH counts
                    60/30 per E-step and 105/75 for the Wiener chain, bf16 decoder MFMAs with fp32 accumulate).
  N > 1             BASELINE.json configs[3]: the fixed 1000-utterance synthetic set, split over the ranks exactly
                    like scripts/evaluate_M1.py:203 (np.array_split), each rank working through its shard in
                    batches of <= 64 (125 per GPU at N=8: 63 + 62); one step = the whole set; "scaling": "strong".
                    `--total-utts 1000` runs the same job at N=1.

`python bench.py --gpus N` starts the N ranks itself (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, before anything touches the GPU) unless it already runs under a launcher (WORLD_SIZE set, e.g.
torch.distributed.run), in which case --gpus must agree with WORLD_SIZE.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_BF16_DENSE = 2.5e15     # FLOP/s, MI355X_MICROARCH.md (dense bf16 MFMA)
PEAK_HBM = 8.0e12            # B/s spec (6.29e12 measured achievable)
# transcendental issue on gfx950, measured (tools/ubench/trans16.hip): one v_exp/v_log/v_rcp_f32 wave-instruction
# occupies its SIMD for 8.2 cycles, whatever the number of resident waves
TRANS_CYCLES = 8.2
BATCH = 64


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)     # (eager call, then the graph capture: both outside the timed steps)
    ap.add_argument("--utts", type=int, default=BATCH, help="utterances per batch (N=1 default workload: one batch per step)")
    ap.add_argument("--total-utts", type=int, default=0,
                    help="fixed utterance set sharded over the ranks (strong scaling; default 1000 when --gpus > 1)")
    ap.add_argument("--niter", type=int, default=100)
    ap.add_argument("--nfft", type=int, default=512)
    ap.add_argument("--rank-k", type=int, default=8)
    ap.add_argument("--precision", default="bf16", choices=["bf16x3", "bf16"],
                    help="decoder MFMA mode: bf16 (BASELINE config 2) or bf16x3 (3-term split, ~fp32 accuracy)")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the extra bf16x3 measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-slice-utts", type=int, default=16, help="utterances of the config-4 slice of the CPU baseline (0: skip)")
    ap.add_argument("--no-configs", action="store_true", help="skip the one-step measurements of the other BASELINE configs")
    ap.add_argument("--force-store", action="store_true", help="sample store also in bf16x3 mode (float rows)")
    ap.add_argument("--no-store", action="store_true",
                    help="M-step / Wiener filter decode the samples again instead of streaming the chain's stored variances")
    ap.add_argument("--model", default="M1", choices=["M1", "M2vad", "M2ibm"],
                    help="M1 (BASELINE config 2, default) or the guided M2 variants of config 3 (labels from a classifier)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="collective backend (gloo: CPU rehearsal of the launcher)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / sharding / all-reduce rehearsal without a GPU (tests): no hot path, synthetic statistics")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """Start args.gpus child processes of this script, one per GPU; returns the exit code (non-zero if any rank failed)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VAENMF_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))     # fresh children, no re-exec
    # Poll all ranks together: the first rank that exits non-zero, or a rank still running at the wall-clock limit, takes
    # the others down with it (they would otherwise sit in init_process_group / all_reduce until the collective's own
    # timeout, 10-30 minutes) and the launcher returns non-zero -- never a half-reported run.
    limit = float(os.environ.get("VAENMF_BENCH_RANK_TIMEOUT", "1500"))
    t0 = time.monotonic()
    rc = 0
    while rc == 0:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = bad[0][1] if bad[0][1] > 0 else 1
            print("[bench] rank %d exited with code %d: stopping the other ranks" % bad[0], file=sys.stderr, flush=True)
        elif all(c == 0 for c in codes):
            return 0
        elif time.monotonic() - t0 > limit:
            rc = 124
            print("[bench] ranks still running after %.0f s: stopping them" % limit, file=sys.stderr, flush=True)
        else:
            time.sleep(0.05)
    for p in procs:                                     # exactly the processes started above
        if p.poll() is None:
            p.terminate()
    t1 = time.monotonic()
    for p in procs:
        try:
            p.wait(timeout=max(0.1, 5.0 - (time.monotonic() - t1)))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc


def note(msg):
    """progress on stderr (stdout carries the one JSON line)"""
    if os.environ.get("RANK", "0") == "0":
        print("[bench] " + msg, file=sys.stderr, flush=True)


def algorithmic(F, niter, nsE, nsW):
    """minimal-pass bytes per frame B_utt / N (fp32 sample tensor materialised once per iteration) and
    decoder flops per decoder row."""
    bytes_per_frame = niter * 4 * F * (5 * nsE + 6) + 4 * F * (5 * nsW + 6) + 24 * F
    flop_row = 2 * (32 * 128 + 128 * 128 + 128 * F)
    return bytes_per_frame, flop_row


def _cpu_utt(a):
    """One utterance of the CPU baseline in a worker process (one torch thread): seconds of TorchMCEM.run()."""
    u, F, K, niter = a
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vaenmf_oracle as orc
    import vaenmf_torch_cpu as tc
    from vaenmf.synth import synth_utterance
    s, n, x, _ = synth_utterance(u)
    X = orc.stft(x, fs=16000, wlen_sec=32e-3 if F == 257 else 64e-3, hop_percent=0.25).T
    m = tc.TorchMCEM("M1", niter)
    m.init_parameters(X, orc.xavier_normal_params([F, 32, [128, 128]], seed=0), K, 1e-8, tc.TorchDraws(u))
    t0 = time.perf_counter()
    m.run()
    return X.shape[0], time.perf_counter() - t0


def cpu_baseline(F, K, niter, slice_utts=16):
    """The PyTorch-CPU restatement of EM.run (oracle/vaenmf_torch_cpu.py: float32 torch tensors, the reference's own
    sequence of tensor operations, pinned by the reference-recorded golden runs), on the host cores of this box:
      (a) BASELINE config 1: ONE 4 s utterance, M1, `niter` EM iterations + Wiener chain, all threads on the one
          utterance (the reference's single-process run), median of 3 runs;
      (b) a 16-utterance slice of BASELINE config 4, run once: the utterances spread over a pool of
          single-thread worker processes, the way the reference spreads utterances over processes
          (scripts/evaluate_M1.py:203-216)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vaenmf_oracle as orc
    import vaenmf_torch_cpu as tc
    from vaenmf.synth import synth_utterance
    # the GPU box gives one GPU's job 16 host cores; more torch threads than that only add synchronisation time
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    s, n, x, _ = synth_utterance(0)
    X = orc.stft(x, fs=16000, wlen_sec=32e-3 if F == 257 else 64e-3, hop_percent=0.25).T
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
    times = []
    for rep in range(3):
        m = tc.TorchMCEM("M1", niter)
        m.init_parameters(X, params, K, 1e-8, tc.TorchDraws(rep))
        t0 = time.perf_counter()
        m.run()
        times.append(time.perf_counter() - t0)
        note("cpu baseline run %d: %.1f s" % (rep, times[-1]))
    t = sorted(times)[1]
    out = {"value": X.shape[0] / t, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": "PyTorch-CPU restatement of EM.run (oracle/vaenmf_torch_cpu.py), BASELINE config 1: 1 utterance (%d frames, "
                     "F=%d, K=%d), %d EM iterations + Wiener chain, torch.set_num_threads(%d), median of 3 runs "
                     "(%.2f / %.2f / %.2f s)" % (X.shape[0], F, K, niter, threads, *sorted(times))}
    if slice_utts:
        import multiprocessing as mp
        nproc = min(threads, slice_utts)
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(nproc) as pool:
            pool.map(_cpu_utt, [(0, F, K, 1)] * nproc)                     # (workers up, torch imported: outside the clock)
            t0 = time.perf_counter()
            res = pool.map(_cpu_utt, [(u, F, K, niter) for u in range(slice_utts)], chunksize=1)
            wall = time.perf_counter() - t0
        frames = sum(r[0] for r in res)
        note("cpu baseline, %d-utterance slice over %d single-thread workers: %.1f s" % (slice_utts, nproc, wall))
        out["config4_slice"] = {"value": frames / wall, "unit": "frames/s", "cores": nproc, "kind": "port", "wall_s": wall,
                                "sample": "%d utterances of BASELINE config 4 (%d frames, F=%d, K=%d, %d EM iterations + Wiener chain each), one "
                                          "pass, %d single-thread worker processes in parallel (per-utterance run() %.1f-%.1f s)"
                                          % (slice_utts, frames, F, K, niter, nproc, min(r[1] for r in res), max(r[1] for r in res))}
    return out


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))                     # before any GPU call
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d disagrees with WORLD_SIZE=%d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    fault = os.environ.get("VAENMF_BENCH_TEST_FAULT", "")          # launcher tests: "<rank>:exit3" / "<rank>:hang", before any collective
    if fault and int(fault.split(":")[0]) == rank:
        if fault.endswith("exit3"):
            sys.exit(3)
        time.sleep(3600)

    import numpy as np
    import torch
    import torch.distributed as dist
    from vaenmf.pipeline import shard, allreduce_stats
    from vaenmf import metrics as vmet

    if not args.dry_run and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # rehearsal aid for one-GPU boxes: VAENMF_BENCH_SHARE_GPU=1 with --backend gloo runs every rank's compute on GPU 0 and
    # the (tiny) collectives through gloo on host tensors -- the launcher, the sharding and the max-over-ranks clock on real
    # kernels; never the measured configuration (RCCL refuses two ranks on one device)
    share = os.environ.get("VAENMF_BENCH_SHARE_GPU") == "1" and args.backend == "gloo"
    dev = torch.device("cpu") if args.dry_run else torch.device("cuda", 0 if share else local_rank)
    comm_dev = torch.device("cpu") if (args.dry_run or args.backend == "gloo") else dev
    if not args.dry_run:
        torch.cuda.set_device(dev.index)
    # one rank under an external launcher (WORLD_SIZE=1 in the environment) may ask for the process group too
    # (VAENMF_BENCH_FORCE_PG=1): the same RCCL path as the N-rank job, on one device
    use_pg = world > 1 or (env_world is not None and os.environ.get("VAENMF_BENCH_FORCE_PG") == "1")
    backend = None
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.backend == "nccl" and not args.dry_run:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        backend = dist.get_backend()

    total = args.total_utts or (1000 if world > 1 else 0)
    strong = total > 0
    fs, nfft = 16000, args.nfft
    F = nfft // 2 + 1
    T = 64000
    # utterance ids of this rank: the whole set split like scripts/evaluate_M1.py:203, or one batch
    ids = [int(i) for i in shard(list(range(total)), world, rank)] if strong else [rank * args.utts + i for i in range(args.utts)]
    nb = max(1, -(-len(ids) // BATCH))
    batches = [list(b) for b in np.array_split(np.asarray(ids, dtype=np.int64), nb)] if ids else []   # 125 -> 63 + 62

    def barrier():
        if use_pg:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    if args.dry_run:
        # rehearsal of everything around the hot path: shard sizes, the statistics all-reduce, the max-over-ranks clock
        snr = [[-5.0, 0.0, 5.0][i % 3] for i in ids]
        r = np.stack([np.full(len(ids), -25.0), np.zeros(len(ids)), np.zeros(len(ids))], 1)
        barrier()
        t0 = time.perf_counter()
        st = allreduce_stats(vmet.sufficient_stats(r, snr), dev)
        barrier()
        tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        cnt = torch.tensor([float(len(ids))], dtype=torch.float64)
        if use_pg:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "scaling": "strong" if strong else "weak", "utterances_total": int(cnt.item()),
                              "utterances_rank0": len(ids), "batches_rank0": [len(b) for b in batches], "stats_count": float(st[0, 0, 0])}))
        if use_pg:
            dist.destroy_process_group()
        return

    from vaenmf import _lib
    from vaenmf.pipeline import Reconstructor
    from vaenmf.synth import synth_utterance, xavier_normal_params, xavier_normal_classifier
    import ctypes as C

    base = [synth_utterance(k) for k in range(16)]          # 16 distinct signals, cycled (host generation time)
    to_dev = lambda l: torch.from_numpy(np.concatenate(l).astype(np.float32)).to(dev)

    def make_batch(b):
        return (to_dev([base[u % 16][2] for u in b]), to_dev([base[u % 16][0] for u in b]), to_dev([base[u % 16][1] for u in b]),
                [base[u % 16][3] for u in b], [T] * len(b))
    data = [make_batch(b) for b in batches]
    umax = max(len(b) for b in batches)

    def make_rec(model, nfft_, K, precision, store, niter=None):
        F_ = nfft_ // 2 + 1
        Dy = {"M1": 0, "M2vad": 1, "M2ibm": F_}[model]
        params = xavier_normal_params([F_, 32, [128, 128]], seed=0, y_dim=Dy)
        clf = None
        if Dy:
            cp = xavier_normal_classifier([F_, [128, 128], Dy], seed=1)
            clf = [(cp["hidden.0.weight"], cp["hidden.0.bias"]), (cp["hidden.1.weight"], cp["hidden.1.bias"]),
                   (cp["output_layer.weight"], cp["output_layer.bias"])]
        rec = Reconstructor(params, F_, K, niter=niter or args.niter, model="M1" if not Dy else "M2", reference_compat=True, fs=fs,
                            wlen_sec=nfft_ / fs, precision=precision, device=dev, max_frames=umax * 520, max_utts=umax, store=store)
        return rec, clf

    store = False if args.no_store else (True if args.force_store else None)
    rec, clf = make_rec(args.model, nfft, args.rank_k, args.precision, store)
    nsE, biE, nsW, biW = rec.nsE, rec.biE, rec.nsW, rec.biW

    def enqueue(i, r=None, c=None, dd=None):
        """One step's batches onto the GPU; nothing is read back (the host prepares the next step while this one runs).
        Returns the per-batch Gram sums (device) and the last batch's cost (device)."""
        r, c, dd = r or rec, c if r else clf, dd or data
        grams, cost = [], None
        for bi, (wx, ws, wn, snr, counts) in enumerate(dd):
            uids = batches[bi] if dd is data else list(range(len(counts)))
            s_hat, n_hat, cost = r.enhance(wx, counts, seeds=[1000 * i + int(u) for u in uids], init_seed=i * 131 + bi, classifier=c)
            grams.append((vmet.gram3_batch_device(s_hat, ws, wn, counts), snr))      # 6 doubles per utterance, left on the device
        return grams, cost

    def collect(steps_grams):
        """Read the Gram sums of the given steps back (one synchronisation), metric statistics per step, one all-reduce
        (<1 KB per step) over the ranks.  Returns the statistics of the LAST step."""
        accs = []
        for grams in steps_grams:
            acc = None
            for G, snr in grams:
                st = vmet.sufficient_stats(np.stack(vmet.ratios_from_gram(G.cpu().numpy()), 1), snr)
                acc = st if acc is None else acc + st
            accs.append(acc)
        allr = allreduce_stats(np.stack(accs), comm_dev)              # RCCL all-reduce, once per timed region
        return allr[-1]

    def step(i, r=None, c=None, dd=None):
        grams, cost = enqueue(i, r, c, dd)
        return collect([grams]), cost

    note("workload ready: %d utterances on rank 0 in %d batch(es); warm-up" % (len(ids), len(batches)))
    for i in range(args.warmup):
        step(i)
    barrier()
    note("timing %d step(s)" % args.steps)
    t0 = time.perf_counter()
    pending = []
    for i in range(args.steps):                          # K steps enqueued back to back; their results are read inside the region
        grams, cost = enqueue(args.warmup + i)
        pending.append(grams)
    st = collect(pending)
    barrier()
    dt = time.perf_counter() - t0
    note("timed region: %.1f ms per step" % (dt / args.steps * 1e3))
    tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
    nutt = torch.tensor([float(len(ids))], dtype=torch.float64, device=comm_dev)
    if use_pg:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nutt, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    frames_per_utt = rec.frame_counts[0]
    m_step_path = {1: "stored sample variances (rows written by the chain, streamed by the M-step and the Wiener filter)",
                   2: "decode (the M-step decodes the samples again)"}.get(_lib.lib().vaenmf_plan_query(rec.eng._plan, _lib.Q_MSTEP_PATH), "?")

    lib = _lib.lib()
    graph_flag = lambda r: int(lib.vaenmf_plan_query(r.eng._plan, _lib.Q_EM_GRAPH))
    main_graph = graph_flag(rec)
    KNAMES = ["mh_chain", "m_wstats", "w_update", "m_hg", "wiener"]

    def profiled_step(r, c, dd, i, niter):
        """One more step OUTSIDE any timed region with HIP events on the launch stream around every hot-path launch
        (vaenmf_profile_*; launch by launch, the graph path is bypassed while profiling): per-kind ms and launch counts."""
        _lib.check(lib.vaenmf_profile_enable(r.eng._plan, len(dd) * (4 * niter + 8) + 16))
        step(i, r, c, dd)
        ms, cn = (C.c_double * 5)(), (C.c_int64 * 5)()
        _lib.check(lib.vaenmf_profile_read(r.eng._plan, ms, cn))
        _lib.check(lib.vaenmf_profile_enable(r.eng._plan, 0))
        return list(ms), list(cn)

    def chain_roofline(ms, cn, F_, fpu, counts_list, niter, r, trans=True):
        """roofline of the MH-chain kernel from a profiled step: algorithmic decoder flops per launch / HIP-event time"""
        _, flop_row = algorithmic(F_, niter, r.nsE, r.nsW)
        chain_ms = ms[0] / max(cn[0], 1)
        rows = []
        for counts in counts_list:
            rows += [len(counts) * fpu * (r.nsE + r.biE)] * niter + [len(counts) * fpu * (r.nsW + r.biW)]
        rows_avg = float(np.mean(rows))
        achieved = flop_row * rows_avg / (chain_ms * 1e-3) / 1e12 if chain_ms > 0 else 0.0
        trans_per_row = 2 * 256 + 2 * F_
        t_trans = trans_per_row * rows_avg / 64.0 * TRANS_CYCLES / (1024 * 2.4e9)      # 1024 SIMDs at the 2.4 GHz peak clock
        return {"bound": "mfma", "kernel": "wchain_kernel (MH chain)", "achieved": achieved, "peak": PEAK_BF16_DENSE / 1e12,
                "unit": "TFLOP/s", "frac": achieved * 1e12 / PEAK_BF16_DENSE, "avg_launch_ms": chain_ms, "flop_per_row": flop_row,
                "rows_per_launch": rows_avg,
                "valu_issue": {"transcendentals_per_row": trans_per_row, "cycles_per_wave_instruction": TRANS_CYCLES,
                               "floor_ms": t_trans * 1e3, "frac": t_trans * 1e3 / chain_ms if chain_ms > 0 else None}}

    def committed_traffic(tag):
        """HBM bytes per launch per kernel from the committed rocprofv3 PMC passes of this command (profiles/, newest round
        first; FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE as is).  Returns (kernels, source, error)."""
        for rnd in ("round3", "round2", "round1"):
            fn = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (rnd, tag))
            if os.path.exists(fn):
                tj = json.load(open(fn))
                if "kernels" not in tj:
                    return None, fn, "profiles/%s has no 'kernels' key" % os.path.basename(fn)
                return tj["kernels"], "profiles/" + os.path.basename(fn), None
        return None, None, "no profiles/round*_%s_traffic.json committed" % tag

    def traffic_of(kern, sub, src):
        hit = [v["hbm_bytes_per_launch"] for k, v in kern.items() if sub in k]
        if not hit:
            raise KeyError("%s: no kernel matching '%s' (a renamed kernel? re-run tools/profile.sh and commit the summary)" % (src, sub))
        return hit[0]

    ms, cn = profiled_step(rec, clf, data, args.warmup + args.steps, args.niter)

    # ---- the other BASELINE configs (rank 0 of a 1-GPU run), one 64-utterance batch each: two untimed calls (the eager
    # one and the graph capture), then the median of three timed, synchronised calls on the REPLAYED graph; then a
    # profiled call for the kernel breakdown and the chain's roofline
    configs = []
    par = None
    cfg1_gpu = None
    if world == 1 and not strong and rank == 0:
        del rec
        torch.cuda.empty_cache()
        b64 = [make_batch(list(range(BATCH)))]
        umax = BATCH
        todo = []
        if not args.no_parity_mode and args.precision == "bf16" and args.model == "M1":
            todo.append(("parity-grade mode: config 2 in bf16x3 (3-term split products)", "M1", nfft, args.rank_k, "bf16x3", None))
        if not args.no_configs:
            todo += [("config 3: M2 guided, VAD label (y_dim 1) from a classifier", "M2vad", 512, 8, args.precision, None),
                     ("config 3: M2 guided, IBM labels (y_dim F) from a classifier", "M2ibm", 512, 8, args.precision, None),
                     ("reference scripts' STFT: 1024-pt (F=513), rank 10", "M1", 1024, 10, args.precision, None),
                     ("config 5 shape (stress): 1024-pt STFT, rank 32 (100 of its 500 iterations)", "M1", 1024, 32, args.precision, None)]
        for name, model, nf, K, prec, st_ in todo:
            note("config: " + name)
            r2, c2 = make_rec(model, nf, K, prec, st_)
            for w in range(2):
                step(w, r2, c2, b64)
            torch.cuda.synchronize()
            ts = []
            for k in range(3):
                t1 = time.perf_counter()
                _, cst = step(2 + k, r2, c2, b64)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            t2 = sorted(ts)[1]
            gflag = graph_flag(r2)
            F2, fpu = nf // 2 + 1, r2.frame_counts[0]
            ms2, cn2 = profiled_step(r2, c2, b64, 9, args.niter)
            bpf2, _ = algorithmic(F2, args.niter, r2.nsE, r2.nsW)
            fps = BATCH * fpu / t2
            e = {"name": name, "dtype": prec, "frames_per_s": fps, "ms_per_step": t2 * 1e3, "ms_per_step_all": [round(t * 1e3, 3) for t in ts],
                 "timing": "median of 3 synchronised calls after 2 untimed ones", "em_graph_replayed": gflag,
                 "final_cost_mean": float(cst[:, -1].mean().item()),
                 "m_step_path": {1: "stored", 2: "decode"}.get(lib.vaenmf_plan_query(r2.eng._plan, _lib.Q_MSTEP_PATH), "?"),
                 "kernels": {k: {"ms_total": round(ms2[i], 3), "launches": int(cn2[i])} for i, k in enumerate(KNAMES)},
                 "roofline": chain_roofline(ms2, cn2, F2, fpu, [b64[0][4]], args.niter, r2),
                 "hbm_equiv": {"algorithmic_bytes_per_frame": bpf2, "frac_of_8TBps": fps * bpf2 / PEAK_HBM}}
            if name.startswith("parity-grade"):
                par = dict(e, value=fps, unit="frames/s")
            else:
                configs.append(e)
            del r2
            torch.cuda.empty_cache()

        # ---- BASELINE config 1 on the GPU: ONE 4 s utterance through the drop-in class (the reference's call pattern,
        # scripts/evaluate_M1.py:111-166: init_parameters + run per utterance on one object), device generator, same
        # workload as cpu_baseline; the engine is built by the first call and reused by the others
        if not args.no_configs and args.model == "M1":
            import vaenmf
            from vaenmf import stft as vstft
            note("config 1 on the GPU: one utterance through MCEM_M1.init_parameters / run")
            x1 = base[0][2]

            def one_utt(nfft1, K1, title):
                F1 = nfft1 // 2 + 1
                vae = vaenmf.VariationalAutoencoder([F1, 32, [128, 128]])
                vae.load_state_dict({k: torch.tensor(v) for k, v in xavier_normal_params([F1, 32, [128, 128]], seed=0).items()})
                X1 = vstft.stft(x1, fs=fs, wlen_sec=nfft1 / fs, hop_percent=0.25).T
                m1 = vaenmf.MCEM_M1(niter=args.niter, rng="device", precision=args.precision)
                ts, allocs = [], []
                for k in range(4):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    m1.init_parameters(X=X1, vae=vae, nmf_rank=K1, eps=1e-8, device=str(dev))
                    c1 = m1.run()
                    ts.append(time.perf_counter() - t1)
                    allocs.append(int(lib.vaenmf_plan_query(m1._eng._plan, _lib.Q_DEV_ALLOCS)))
                t1u = sorted(ts[1:])[1]
                return {"workload": "%s: one 4 s utterance (%d frames, F=%d, K=%d), %d EM iterations + Wiener chain through "
                                    "MCEM_M1.init_parameters + run (rng='device', %s, batch of one, sample store on)" % (title, X1.shape[0], F1, K1, args.niter, args.precision),
                        "value": X1.shape[0] / t1u, "unit": "frames/s", "seconds_per_utterance": t1u,
                        "seconds_all_calls": [round(t, 4) for t in ts], "timing": "median of calls 2-4 (call 1 builds the engine)",
                        "device_allocations_after_each_call": allocs, "final_cost": float(c1[-1]),
                        "chain_kernel": {0: "team kernel", 1: "one wavefront per 16 frames", 2: "four wavefronts per 16 frames (wchain4_kernel)"}[
                            int(lib.vaenmf_plan_query(m1._eng._plan, _lib.Q_CHAIN_KERNEL))]}

            cfg1_gpu = one_utt(nfft, args.rank_k, "BASELINE config 1")
            if nfft == 512 and args.rank_k == 8:
                # the shape every script of the reference runs (scripts/evaluate_M1.py:77-92: 64 ms window, rank 10)
                cfg1_gpu["reference_scripts_shape"] = one_utt(1024, 10, "the reference scripts' own shape")

    if rank == 0:
        n_total = int(nutt.item())
        frames = n_total * frames_per_utt * args.steps
        value = frames / dt
        bpf, flop_row = algorithmic(F, args.niter, nsE, nsW)
        rl = chain_roofline(ms, cn, F, frames_per_utt, [d_[4] for d_ in data], args.niter,
                            type("R", (), {"nsE": nsE, "biE": biE, "nsW": nsW, "biW": biW}))
        tag = args.precision if args.model == "M1" and nfft == 512 and args.rank_k == 8 else "%s_%s_f%d_k%d" % (args.precision, args.model, F, args.rank_k)
        kern, traffic_src, terr = committed_traffic(tag)
        rl["traffic"] = traffic_of(kern, "chain_kernel", traffic_src) if kern else None
        rl["traffic_source"] = ("%s (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes of this command; not measured in this run)" % traffic_src) if kern else terr
        rl["note"] = ("algorithmic decoder flops (1 proposal decode per MH step, %d flop/row) / HIP-event launch time of a profiled step "
                      "outside the timed region; the bf16x3 mode issues 3 MFMAs per algorithmic product; valu_issue = the bound of the "
                      "kernel's own instruction stream (v_exp/v_log/v_rcp_f32: 8.2 cycles per wave-instruction per SIMD, 1024 SIMDs, 2.4 GHz)" % flop_row)
        kernels = {k: {"ms_total": round(ms[i], 3), "launches": int(cn[i])} for i, k in enumerate(KNAMES)}
        iter_ms = sum(ms[i] / max(cn[i], 1) for i in range(4))
        workload = (("fixed %d-utterance set sharded over %d GPU(s) in batches of <= %d" % (n_total, world, BATCH)) if strong
                    else ("%d-utterance batch per GPU" % args.utts))
        out = {
            "metric": "STFT-frames/sec through VAE-NMF reconstruct loop; SI-SDR parity vs ref",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "bf16x3 (bf16 MFMA, 3-term hi/lo split, fp32 accumulate)" if args.precision == "bf16x3" else "bf16",
            "data": "synthetic",
            "config": {"workload": "%s, %s reconstruct, %d-pt STFT (F=%d, %d frames/utt), NMF rank %d, %d EM iters, MH %d/%d per E-step + %d/%d Wiener chain"
                                   % (workload, args.model, nfft, F, frames_per_utt, args.rank_k, args.niter, nsE + biE, nsE, nsW + biW, nsW),
                       "utterances_total": n_total, "utterances_rank0": len(ids), "batches_rank0": [len(b) for b in batches],
                       "parallelism": "utterance-shard x%d (np.array_split, no collective in the loop)" % world},
            "collective_backend": backend,
            "em_graph_replayed": main_graph,
            "roofline": rl,
            "hbm_equiv": {"algorithmic_bytes_per_frame": bpf, "achieved_GBps": value * bpf / 1e9, "frac_of_8TBps": value * bpf / PEAK_HBM,
                          "note": "SURVEY 8(d) B_utt CREDIT (fp32 sample variances written once, read by W-, H-, g-update and cost per "
                                  "iteration): a credit, not a bandwidth -- the build moves fewer real bytes, see measured_hbm"},
            "m_step_path": m_step_path,
            "kernels": kernels,
            "kernels_note": "HIP events of rank 0's profiled step" + (" (the other ranks run the same kernels on their shards)" if world > 1 else ""),
            "si_sdr_mean_db": float(st[0, 0, 1] / max(st[0, 0, 0], 1)),
            "final_cost_mean": float(cost[:, -1].mean().item()),
        }
        if kern:        # real HBM bytes per EM iteration from the committed PMC passes, against this run's kernel time per iteration
            subs = [("chain_kernel",), ("wstats_stream", "wstats_rot", "wstats_fused"), ("hg_stream",), ("w_partial",), ("w_update", "w_final")]
            per_iter, missing = 0.0, []
            for alt in subs:
                hit = [v["hbm_bytes_per_launch"] for k, v in kern.items() if any(a_ in k for a_ in alt)]
                if hit:
                    per_iter += hit[0]
                elif alt[0] not in ("w_partial",):        # (fused into the W-statistics kernel in later builds)
                    missing.append("/".join(alt))
            out["measured_hbm"] = {"bytes_per_em_iteration": per_iter, "GBps": per_iter / (iter_ms * 1e-3) / 1e9,
                                   "frac_of_8TBps": per_iter / (iter_ms * 1e-3) / PEAK_HBM, "kernels_missing_from_profile": missing,
                                   "note": "rocprofv3 PMC bytes (%s, not this run) / this run's kernel time per EM iteration" % traffic_src}
        else:
            out["measured_hbm"] = {"error": terr}
        # what this GPU delivers to a hand-written streaming read of the sample store's size (vaenmf_hbm_read_probe: 16 B per
        # lane, 8 wavefronts per SIMD, grid = resident set), outside the timed region
        pbuf = torch.empty(130 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()
        sink = torch.zeros(4, dtype=torch.int32, device=dev)
        gb = C.c_double()
        _lib.check(lib.vaenmf_hbm_read_probe(C.c_void_p(pbuf.data_ptr()), pbuf.numel() * 4, 10, C.c_void_p(sink.data_ptr()), C.byref(gb),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        out["hbm_read_probe"] = {"GBps": gb.value, "bytes": pbuf.numel() * 4, "frac_of_8TBps": gb.value * 1e9 / PEAK_HBM,
                                 "note": "hand-written HIP read kernel (aux.hip: 16 B per lane, 8 wavefronts per SIMD, 4 loads in flight per lane, grid = "
                                         "resident set) over a 520 MB buffer, 10 sweeps: a measured rate of one kernel on this GPU, not a bound"}
        del pbuf
        if configs:
            out["configs"] = configs
        if par is not None:
            out["parity_mode"] = par
        if cfg1_gpu is not None:
            out["config_1_gpu"] = cfg1_gpu
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(F, args.rank_k, args.niter, args.cpu_slice_utts)
        print(json.dumps(out))
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
