#!/bin/bash
# dev aid (GPU box): sample the GPU's clocks and power while bench.py runs
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clocks.txt &
SMI=$!
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-parity-mode --no-configs > gpurun_out/clocks_bench.log 2>&1
kill $SMI 2>/dev/null
tail -1 gpurun_out/clocks_bench.log | cut -c1-200
sort gpurun_out/clocks.txt | uniq -c | sort -rn | head -12
