#!/bin/bash
# dev aid (GPU box): time bench.py under several environment settings in ONE call: tools/ab_env.sh "<bench args>" "VAR=1" "VAR=0" ...
args=$1; shift
for e in "$@"; do
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-mode $args > gpurun_out/abenv.log 2>&1 || { echo "$e FAILED"; tail -3 gpurun_out/abenv.log; continue; }
  tail -1 gpurun_out/abenv.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$e', round(d['ms_per_step'],2), 'ms/step', {n: round(v['ms_total']/v['launches'],4) for n,v in k.items()}, 'sisdr', round(d.get('si_sdr_mean_db',0),3))"
done
