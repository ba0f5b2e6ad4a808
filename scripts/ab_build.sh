#!/bin/bash
# same-box A/B of two BUILDS of the library: the headline leg with the tree as it is, then with VP8HIP_EXTRA_FLAGS="$1" compiled in
# (on the GPU box: hipcc is there), N repeats each:  bash scripts/ab_build.sh "-DVP8HIP_S2_PRE_ARRAY" 3
flags=$1; n=${2:-3}
out=gpurun_out/ab_build.txt
: > $out
run() {
  for i in $(seq $n); do
    echo -n "$1 : " >> $out
    timeout 200 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-side-legs --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])" >> $out
  done
}
run "tree as it is"
VP8HIP_EXTRA_FLAGS="$flags" timeout 900 python -m vp8oclenc_amd.build > /dev/null 2>&1 || echo "build failed" >> $out
run "with $flags"
timeout 900 python -m vp8oclenc_amd.build > /dev/null 2>&1
run "tree as it is (again)"
cat $out
