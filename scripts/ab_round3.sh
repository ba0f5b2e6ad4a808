#!/bin/bash
# same-box A/B of round 3's switches (every run under its own timeout)
out=gpurun_out/ab_r3.txt
: > $out
run() {
  echo "## $*" >> $out
  env "$@" timeout 150 python bench.py --no-side-legs --cpu-seconds 0 --steps 40 --warmup 10 2>>gpurun_out/ab_r3.err | python -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print(json.dumps({k:j[k] for k in ('value','ms_per_step')}), j['config'].get('check_ssim','')[:20], j['config'].get('frames_redone_as_key'))" >> $out
  echo "rc=$?" >> $out
}
for v in "$@"; do run $v; done
cat $out
