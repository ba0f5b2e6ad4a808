#!/bin/bash
# what a launch is worth with the part full: the headline leg of a build with -DVP8HIP_EXPERIMENTS (its lines say INVALID) under
# VP8HIP_EXPERIMENT_SKIP values:  bash scripts/ab_flags_experiments.sh 2 "" "pack" "scan" "pack scan"
cd $GRAFT_REPO_ROOT
VP8HIP_EXTRA_FLAGS="-DVP8HIP_EXPERIMENTS" timeout 900 python -m vp8oclenc_amd.build > /dev/null 2>&1
export VP8_BENCH_EXPERIMENT_BUILD=prints-an-invalid-line
n=$1; shift
out=gpurun_out/ab_flags_experiments.txt
: > $out
for i in $(seq $n); do
  for v in "$@"; do
    echo -n "skip [$v] : " >> $out
    VP8HIP_EXPERIMENT_SKIP="$v" timeout 200 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-side-legs --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])" >> $out
  done
done
timeout 900 python -m vp8oclenc_amd.build > /dev/null 2>&1
sort $out
