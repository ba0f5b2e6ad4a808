#!/bin/bash
# same-box A/B runs of the headline leg under environment switches (every run under its own timeout):
#   bash scripts/ab_round3.sh "VP8_BENCH_CHECK=0" "VP8_BENCH_CHECK=1" ...      [STEPS=20 WARMUP=5]
out=gpurun_out/ab_r3.txt
: > $out
run() {
  echo "## $*" >> $out
  env "$@" timeout 150 python bench.py --no-side-legs --cpu-seconds 0 --steps ${STEPS:-20} --warmup ${WARMUP:-5} 2>>gpurun_out/ab_r3.err | python -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print(json.dumps({k:j[k] for k in ('value','ms_per_step','host_enqueue_ms_per_frame')}), j['config'].get('check_ssim','')[:12], j['config'].get('frames_redone_as_key'))" >> $out
  echo "rc=$?" >> $out
}
for v in "$@"; do run $v; done
cat $out
