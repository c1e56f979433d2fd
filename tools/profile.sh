#!/bin/bash
# dev aid (runs on the GPU box through gpurun): rocprofv3 kernel statistics and HBM-traffic / SQ counter passes of bench.py
# usage: tools/profile.sh <tag> [bench.py workload flags...]     e.g.  tools/profile.sh bf16
#                                                                      tools/profile.sh bf16_M2ibm_f257_k8 --model M2ibm
#                                                                      tools/profile.sh bf16_M1_f513_k32 --nfft 1024 --rank-k 32
# every rocprofv3 run is bounded (a hung pass must not eat the budget); the program itself follows "--" (no env / bash hop)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/prof_r3_$TAG
rm -rf $OUT; mkdir -p $OUT
B="--no-cpu-baseline --no-parity-mode --no-configs $*"
echo "python bench.py $B" > $OUT/command.txt
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run bench_trace --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 2 --warmup 1 $B
grep '^{' $OUT/bench_trace.log > $OUT/bench_trace.json
run bench_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 1 --warmup 0 --niter 10 $B
run bench_write --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 1 --warmup 0 --niter 10 $B
run bench_sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python bench.py --steps 1 --warmup 0 --niter 10 $B
run bench_sq2 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python bench.py --steps 1 --warmup 0 --niter 10 $B
run bench_grbm --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- python bench.py --steps 1 --warmup 0 --niter 10 $B
python tools/summarize_prof.py "$TAG" $OUT profiles/round3_${TAG}_summary.txt round3 > $OUT/summary_stdout.txt 2>&1; echo "summary rc=$?"
cp profiles/round3_${TAG}_summary.txt profiles/round3_${TAG}_traffic.json $OUT/ 2>/dev/null
ls $OUT
