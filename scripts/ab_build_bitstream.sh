#!/bin/bash
# same-box A/B of BUILDS of the library on the frames-out leg: bash scripts/ab_build_bitstream.sh N "<flags A>" "<flags B>" ...
n=$1; shift
out=gpurun_out/ab_build_bitstream.txt
: > $out
for flags in "$@"; do
  VP8HIP_EXTRA_FLAGS="$flags" timeout 900 python -m vp8oclenc_amd.build > /dev/null 2>&1 || echo "build failed: $flags" >> $out
  for i in $(seq $n); do
    echo -n "[$flags] : " >> $out
    timeout 300 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --only-bitstream --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['with_bitstream']['value'])" >> $out
  done
done
timeout 900 python -m vp8oclenc_amd.build > /dev/null 2>&1
cat $out
