#!/bin/bash
# dev aid (GPU box): sample the GPU's clocks and power while bench.py runs a long timed region
( for i in $(seq 1 200); do echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.15; done ) > gpurun_out/clocks.txt &
SMI=$!
python bench.py --steps ${1:-40} --warmup 5 --no-cpu-baseline --no-parity-mode --no-configs > gpurun_out/clocks_bench.log 2> gpurun_out/clocks_bench.err
kill $SMI 2>/dev/null
tail -1 gpurun_out/clocks_bench.log | cut -c1-220
grep -v amdgpu.ids gpurun_out/clocks_bench.err | head
awk '{print $2, $3, $4, $5, $6, $7}' gpurun_out/clocks.txt | uniq -c | head -60
