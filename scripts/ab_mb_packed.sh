# the macroblock kernel with every lane on a block (k_mb_p) against the 32-lanes-per-macroblock form (k_mb_b), alternating on this box:
# the headline's chunks, then the same with the four-pass ladder (-SSIM-target 93)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for extra in "" "--ssim-target 0.93"; do
for m in 0 1 0 1; do
  VP8HIP_MB_PACKED=$m python3 bench.py --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 $extra 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('VP8HIP_MB_PACKED=$m $extra', d['value'], 'M MB/s; chunks against the oracle:', d['self_check']['against_the_oracle']['chunks_checked'], d['self_check']['against_the_oracle']['identical'])"
done
done
