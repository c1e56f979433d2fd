#!/usr/bin/env python3
"""bench.py -- STFT-frames/s through the VAE-NMF reconstruct loop (BASELINE.json metric).

A "step" is one pass of the whole hot path over one batch of synthetic utterances per
GPU: waveforms resident in HBM -> STFT -> |X|^2 -> encoder -> 100 x (MH E-step, M-step)
-> Wiener chain + filter -> iSTFT -> SI-SDR sufficient statistics -> all-reduce of the
metric statistics over ranks (the job's only collective).  Workload at N=1 =
BASELINE.json configs[1]: 64 utterances x 4 s @16 kHz, 512-pt STFT (F=257, 501 frames
each), M1, NMF rank 8, 100 EM iterations, reference-faithful MH counts (60/30 per
E-step, 105/75 for the Wiener chain), decoder GEMMs on bf16 MFMA (fp32 accumulate; `--precision bf16x3`
selects the 3-term split mode used for the tight parity tests, also timed once and reported as
"parity_mode").  For N>1 every rank runs its own 64-utterance shard (weak scaling; utterances
are independent).

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "guided-vae-nmf_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

PEAK_BF16_DENSE = 2.5e15     # FLOP/s, MI355X_MICROARCH.md (dense bf16 MFMA)
PEAK_HBM = 8.0e12            # B/s spec (6.29e12 measured achievable)


def algorithmic(F, N_frames, niter, nsE, biE, nsW, biW):
    """minimal-pass bytes B_utt (fp32 sample tensor materialised once per
    iteration) and decoder flops per frame."""
    R, Rw = nsE, nsW
    bytes_per_frame = niter * 4 * F * (5 * R + 6) + 4 * F * (5 * Rw + 6) + 24 * F
    flop_row = 2 * (32 * 128 + 128 * 128 + 128 * F)
    return bytes_per_frame, flop_row


def cpu_baseline(F, n_frames, K, niter_full, sample_iters=12):
    """Oracle (numpy restatement of the reference path, checker only) timed on the host
    cores on a bounded sample: ONE utterance, `sample_iters` EM iterations + the Wiener
    chain, per-iteration time extrapolated to niter_full."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vaenmf_oracle as orc
    from vaenmf.synth import synth_utterance
    s, n, x, _ = synth_utterance(0)
    X = orc.stft(x, fs=16000, wlen_sec=32e-3 if F == 257 else 64e-3, hop_percent=0.25).T
    params = orc.xavier_normal_params([F, 32, [128, 128]], seed=0)
    m = orc.MCEMOracle("M1", sample_iters)
    m.init_parameters(X, params, K, 1e-8, orc.NumpyRNG(0))
    t0 = time.perf_counter()
    for _ in range(sample_iters):
        m.E_step(); m.M_step(); m.compute_expected_neg_log_like()
    t_it = (time.perf_counter() - t0) / sample_iters
    t0 = time.perf_counter()
    m.compute_WF(sample=True)
    t_wf = time.perf_counter() - t0
    t_utt = t_it * niter_full + t_wf
    cores = os.cpu_count() or 1
    return {"value": X.shape[0] / t_utt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "numpy oracle, 1 utterance (%d frames, F=%d, K=%d): %d EM iterations + Wiener chain timed "
                      "(%.3f s/iter, %.2f s WF), extrapolated to %d iterations; BLAS threads = host default"
                      % (X.shape[0], F, K, sample_iters, t_it, t_wf, niter_full)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--utts", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--niter", type=int, default=100)
    ap.add_argument("--nfft", type=int, default=512)
    ap.add_argument("--rank-k", type=int, default=8)
    ap.add_argument("--precision", default="bf16", choices=["bf16x3", "bf16"],
                    help="decoder MFMA mode: bf16 (BASELINE config 2) or bf16x3 (3-term split, ~fp32 accuracy)")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the extra bf16x3 measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-store", action="store_true", help="sample store also in bf16x3 mode (float rows)")
    ap.add_argument("--no-store", action="store_true",
                    help="M-step / Wiener filter decode the samples again instead of streaming the chain's stored variances")
    ap.add_argument("--model", default="M1", choices=["M1", "M2vad", "M2ibm"],
                    help="M1 (BASELINE config 2, default) or the guided M2 variants of config 3 (labels from a classifier)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from vaenmf import _lib
    from vaenmf.pipeline import Reconstructor, allreduce_stats
    from vaenmf.synth import synth_utterance, xavier_normal_params
    from vaenmf import metrics as vmet
    import ctypes as C

    fs, nfft = 16000, args.nfft
    F = nfft // 2 + 1
    wlen = nfft / fs
    U, T = args.utts, 64000
    # synthetic shard of this rank: utterance ids rank*U .. rank*U+U-1 (seeded, no dataset)
    ids = [rank * U + i for i in range(U)]
    base = {}
    wav_x, wav_s, wav_n, snr = [], [], [], []
    for uid in ids:
        s, n, x, sdb = synth_utterance(uid % 16)        # 16 distinct signals, cycled (host generation time)
        wav_x.append(x); wav_s.append(s); wav_n.append(n); snr.append(sdb)
    to_dev = lambda l: torch.from_numpy(np.concatenate(l).astype(np.float32)).to(dev)
    wav_x, wav_s, wav_n = to_dev(wav_x), to_dev(wav_s), to_dev(wav_n)
    counts = [T] * U
    Dy = {"M1": 0, "M2vad": 1, "M2ibm": F}[args.model]
    params = xavier_normal_params([F, 32, [128, 128]], seed=0, y_dim=Dy)
    clf = None
    if Dy:
        from vaenmf.synth import xavier_normal_classifier
        cp = xavier_normal_classifier([F, [128, 128], Dy], seed=1)
        clf = [(cp["hidden.0.weight"], cp["hidden.0.bias"]), (cp["hidden.1.weight"], cp["hidden.1.bias"]),
               (cp["output_layer.weight"], cp["output_layer.bias"])]
    rec = Reconstructor(params, F, args.rank_k, niter=args.niter, model="M1" if not Dy else "M2", reference_compat=True, fs=fs,
                        wlen_sec=wlen, precision=args.precision, device=dev, max_frames=U * 520, max_utts=U,
                        store=False if args.no_store else (True if args.force_store else None))
    nsE, biE, nsW, biW = rec.nsE, rec.biE, rec.nsW, rec.biW

    def step(i):
        s_hat, n_hat, cost = rec.enhance(wav_x, counts, seeds=[1000 * i + u for u in ids], init_seed=i, classifier=clf)
        G = vmet.gram3_batch(s_hat, wav_s, wav_n, counts)          # D2H of 6 doubles per utterance
        r = np.stack(vmet.ratios_from_gram(G), 1)
        st = allreduce_stats(vmet.sufficient_stats(r, snr), dev)   # RCCL all-reduce (<1 KB)
        return st, cost

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    n_launch = args.steps * (4 * args.niter + 8) + 16
    _lib.check(_lib.lib().vaenmf_profile_enable(rec.eng._plan, n_launch))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        st, cost = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    ms = (C.c_double * 5)()
    cn = (C.c_int64 * 5)()
    _lib.check(_lib.lib().vaenmf_profile_read(rec.eng._plan, ms, cn))
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # the same step in the bf16x3 (parity-grade) mode, one timed step, for the record
    par = None
    if args.precision == "bf16" and not args.no_parity_mode and world == 1 and not Dy:
        del rec
        torch.cuda.empty_cache()
        rec3 = Reconstructor(params, F, args.rank_k, niter=args.niter, model="M1", reference_compat=True, fs=fs,
                             wlen_sec=wlen, precision="bf16x3", device=dev, max_frames=U * 520, max_utts=U)
        rec3.enhance(wav_x, counts, seeds=[u for u in ids], init_seed=0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rec3.enhance(wav_x, counts, seeds=[7 + u for u in ids], init_seed=1)
        torch.cuda.synchronize()
        t3 = time.perf_counter() - t1
        par = {"dtype": "bf16x3", "value": U * rec3.frame_counts[0] / t3, "unit": "frames/s", "ms_per_step": t3 * 1e3}
        rec = rec3
    if rank == 0:
        frames_per_utt = rec.frame_counts[0]
        frames = U * frames_per_utt * world * args.steps
        value = frames / dt
        bpf, flop_row = algorithmic(F, frames_per_utt, args.niter, nsE, biE, nsW, biW)
        # dominant kernel = mh_chain: algorithmic decoder flops per launch / avg launch time
        chain_ms = ms[0] / max(cn[0], 1)
        rows_e = U * frames_per_utt * (nsE + biE)                  # one proposal decode per MH step
        rows_w = U * frames_per_utt * (nsW + biW)
        n_e = args.niter * args.steps
        n_w = args.steps
        flops_chain_avg = flop_row * (rows_e * n_e + rows_w * n_w) / max(n_e + n_w, 1)
        achieved = flops_chain_avg / (chain_ms * 1e-3) / 1e12 if chain_ms > 0 else 0.0
        # HBM bytes of the chain kernel per launch from the committed rocprofv3 PMC passes of this command
        # (profiles/round1_<precision>_traffic.json; collected with tools/profile.sh), else null
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "round1_%s_traffic.json" % args.precision)))
            traffic = [v["hbm_bytes_per_launch"] for k, v in tj["kernels"].items() if k.startswith("mh_chain_kernel")][0]
        except Exception:
            pass
        kernels = {k: {"ms_total": round(ms[i], 3), "launches": int(cn[i])}
                   for i, k in enumerate(["mh_chain", "m_wstats", "w_update", "m_hg", "wiener"])}
        out = {
            "metric": "STFT-frames/sec through VAE-NMF reconstruct loop; SI-SDR parity vs ref",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 (bf16 MFMA, 3-term hi/lo split, fp32 accumulate)" if args.precision == "bf16x3" else "bf16",
            "data": "synthetic",
            "config": {"workload": ("%d-utterance batch per GPU, %s reconstruct, %d-pt STFT (F=%d, %d frames/utt), "
                                    "NMF rank %d, %d EM iters, MH %d/%d per E-step + %d/%d Wiener chain")
                                   % (U, args.model, nfft, F, frames_per_utt, args.rank_k, args.niter, nsE + biE, nsE, nsW + biW, nsW),
                       "utterances_per_gpu": U, "parallelism": "utterance-shard x%d" % world},
            "roofline": {"bound": "mfma", "kernel": "mh_chain_kernel", "achieved": achieved, "peak": PEAK_BF16_DENSE / 1e12,
                         "unit": "TFLOP/s", "frac": achieved * 1e12 / PEAK_BF16_DENSE, "traffic": traffic,
                         "avg_launch_ms": chain_ms,
                         "note": "algorithmic decoder flops (1 proposal decode per MH step, %d flop/row); the bf16x3 mode "
                                 "issues 3 MFMAs per algorithmic product" % flop_row},
            "hbm_equiv": {"algorithmic_bytes_per_frame": bpf, "achieved_GBps": value * bpf / 1e9,
                          "frac_of_8TBps": value * bpf / PEAK_HBM,
                          "note": "SURVEY 8(d) B_utt credit (fp32 sample variances written once, read twice per iteration)"},
            "m_step_path": ("decode" if (args.no_store or args.precision != "bf16") else
                            "stored sample variances (bf16 rows written by the chain, streamed by the M-step and the Wiener filter)"),
            "kernels": kernels,
            "si_sdr_mean_db": float(st[0, 0, 1] / max(st[0, 0, 0], 1)),
            "final_cost_mean": float(cost[:, -1].mean().item()),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(F, frames_per_utt, args.rank_k, args.niter)
        if par is not None:
            out["parity_mode"] = par
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
