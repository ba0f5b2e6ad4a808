#!/bin/bash
# same-box repeats of the headline leg under environment settings: bash scripts/ab_env.sh N "ENV=1" "ENV=0" ...
cd $GRAFT_REPO_ROOT
n=$1; shift
out=gpurun_out/ab_env.txt
: > $out
for i in $(seq $n); do
  for v in "$@"; do
    echo -n "$v : " >> $out
    env $v timeout 200 python3 bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j.get('self_check', {}).get('identical'))" >> $out
  done
done
sort $out
