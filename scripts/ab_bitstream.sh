#!/bin/bash
# same-box repeats of the frames-out leg (with_bitstream) under environment switches: bash scripts/ab_bitstream.sh N "ENV=1 ..." "..."
out=gpurun_out/ab_bitstream.txt
: > $out
n=$1; shift
for i in $(seq $n); do
  for v in "$@"; do
    echo -n "$v : " >> $out
    env $v timeout 300 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --only-bitstream --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['with_bitstream']['value'], j['with_bitstream']['avg_frame_bytes'])" >> $out
  done
done
sort $out
