#!/bin/bash
# the frames-out leg at other chunk counts / batch sizes, same box: bash scripts/ab_bitstream_shape.sh N "G B [Q]" "G B [Q]" ...
out=gpurun_out/ab_bitstream_shape.txt
: > $out
n=$1; shift
for i in $(seq $n); do
  for v in "$@"; do
    read G B Q <<< "$v"
    echo -n "G=$G B=$B Q=${Q:-16} : " >> $out
    GPU_MAX_HW_QUEUES=${Q:-16} timeout 300 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --gops-per-gpu $G --batch $B --only-bitstream --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['with_bitstream']['value'])" >> $out
  done
done
sort $out
