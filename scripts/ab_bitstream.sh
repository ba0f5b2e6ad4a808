#!/bin/bash
# same-box repeats of the frames-out leg (with_bitstream) under environment switches: bash scripts/ab_bitstream.sh N "ENV=1 ..." "..."
# (the VP8HIP_EXPERIMENT_* switches exist only in a build with -DVP8HIP_EXPERIMENTS: EXPERIMENTS=1 builds one first and the tree's
# own again afterwards; bench.py prints a line for such a build only when told to, and the line says INVALID)
if [ -n "$EXPERIMENTS" ]; then VP8HIP_EXTRA_FLAGS="-DVP8HIP_EXPERIMENTS" timeout 900 python -m vp8oclenc_amd.build > /dev/null 2>&1; export VP8_BENCH_EXPERIMENT_BUILD=prints-an-invalid-line; fi
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
if [ -n "$EXPERIMENTS" ]; then timeout 900 python -m vp8oclenc_amd.build > /dev/null 2>&1; fi
sort $out
