#!/bin/bash
# same-box A/B of round 3's switches: check_SSIM in the loop on / off, the batch's head-of-frame stream off / per batch / shared
out=gpurun_out/ab_r3.txt
: > $out
run() {
  echo "## $*" >> $out
  env "$@" python bench.py --no-side-legs --cpu-seconds 0 --steps 40 --warmup 10 2>>gpurun_out/ab_r3.err | python -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print(json.dumps({k:j[k] for k in ('value','ms_per_step')}), j['config'].get('check_ssim','')[:20], j['config'].get('frames_redone_as_key'))" >> $out
}
run VP8_BENCH_CHECK=0 VP8HIP_BATCH_PREP=0
run VP8_BENCH_CHECK=0 VP8HIP_BATCH_PREP=1
run VP8_BENCH_CHECK=0 VP8HIP_BATCH_PREP=2
run VP8_BENCH_CHECK=1 VP8HIP_BATCH_PREP=0
run VP8_BENCH_CHECK=1 VP8HIP_BATCH_PREP=1
run VP8_BENCH_CHECK=1 VP8HIP_BATCH_PREP=2
run VP8_BENCH_CHECK=1 VP8HIP_BATCH_PREP=1
run VP8_BENCH_CHECK=0 VP8HIP_BATCH_PREP=0
cat $out
