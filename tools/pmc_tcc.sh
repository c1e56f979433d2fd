#!/bin/bash
# dev aid (GPU box): L2 (TCC) hit / miss / memory-request counters of the streaming kernels, one bounded pass each
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmctcc
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 1 --warmup 0 --niter 6 --no-cpu-baseline --no-parity-mode --no-configs"
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/a -- python $ARGS > $OUT/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/b -- python $ARGS > $OUT/b.log 2>&1; echo "b rc=$?"
python - <<'PY'
import csv, glob, collections
short = lambda k: k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
for p in "ab":
    fs = glob.glob("gpurun_out/pmctcc/%s/*/*_counter_collection.csv" % p)
    if not fs: print("no csv for pass", p); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = short(r["Kernel_Name"])
        if not any(s in k for s in ("stream", "wchain", "fused", "rot_kernel")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
    for k in agg:
        print(k, "launches", len(nd[k]))
        for c in sorted(agg[k]): print("    %-32s %16.0f per launch" % (c, agg[k][c] / len(nd[k])))
PY
