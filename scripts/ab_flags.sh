#!/bin/bash
# same-box repeats of the headline leg under environment switches: bash scripts/ab_flags.sh N "ENV=1 ..." "..."
out=gpurun_out/ab_flags.txt
: > $out
n=$1; shift
for i in $(seq $n); do
  for v in "$@"; do
    echo -n "$v : " >> $out
    env $v timeout 200 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-side-legs --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['host_enqueue_ms_per_frame'])" >> $out
  done
done
sort $out
