#!/bin/bash
# dev aid (GPU box): time bench.py with several builds of the library in ONE call (box-to-box variance is +-3%)
# usage: tools/ab.sh "<bench args>" tagA tagB ...   (tag "main" = libvaenmf.so)
args=$1; shift
for t in "$@"; do
  lib=$GRAFT_REPO_ROOT/guided-vae-nmf_amd/vaenmf/libvaenmf_$t.so
  [ "$t" = main ] && lib=$GRAFT_REPO_ROOT/guided-vae-nmf_amd/vaenmf/libvaenmf.so
  VAENMF_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-mode $args > gpurun_out/ab_$t.log 2>&1 || { echo "$t FAILED"; tail -3 gpurun_out/ab_$t.log; continue; }
  tail -1 gpurun_out/ab_$t.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$t', round(d['ms_per_step'],2), 'ms/step', {n: round(v['ms_total']/v['launches'],4) for n,v in k.items()}, 'sisdr', round(d.get('si_sdr_mean_db',0),3))"
done
