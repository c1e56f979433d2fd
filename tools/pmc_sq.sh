#!/bin/bash
# dev aid (GPU box): three SQ counter passes over a short bench run; prints per-kernel ratios
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcsq
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 0 --niter 6 --no-cpu-baseline --no-parity-mode --precision ${1:-bf16}"
timeout -k 10 240 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/a -- python $ARGS > $OUT/a.log 2>&1 &&
timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/b -- python $ARGS > $OUT/b.log 2>&1 &&
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/c -- python $ARGS > $OUT/c.log 2>&1 &&
python - <<'PY'
import csv, glob, collections
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
for p in "abc":
    fs = glob.glob("gpurun_out/pmcsq/%s/*/*_counter_collection.csv" % p)
    if not fs: print("no csv for pass", p); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if not any(s in k for s in ("mh_chain", "wchain", "decode_kernel", "stream_kernel", "fused_kernel", "rot_kernel", "w_update", "w_partial")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
        dur[(k, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in agg:
        n = len(nd[k]); t = sum(d for (kk, _), d in dur.items() if kk == k) / n
        print("%s  launches %d  avg %.1f us" % (k, n, t / 1e3))
        for c in sorted(agg[k]): print("    %-28s %16.0f per launch" % (c, agg[k][c] / n))
PY
