#!/bin/bash
# dev aid (runs on the GPU box through gpurun): rocprofv3 kernel statistics and HBM-traffic counters of bench.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-mode --precision $1 > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 1 --warmup 0 --niter 10 --no-cpu-baseline --no-parity-mode --precision $1 > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 1 --warmup 0 --niter 10 --no-cpu-baseline --no-parity-mode --precision $1 > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python bench.py --steps 1 --warmup 0 --niter 10 --no-cpu-baseline --no-parity-mode --precision $1 > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- python bench.py --steps 1 --warmup 0 --niter 10 --no-cpu-baseline --no-parity-mode --precision $1 > $OUT/bench_grbm.log 2>&1
ls -R $OUT | head -40
