# k_search2: groups of eight blocks a workgroup takes one after the other (VP8HIP_S2_ITER = 1, 2, 4), alternating on this box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1 2 4 1 2 4; do
  VP8HIP_S2_ITER=$v python3 bench.py --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('VP8HIP_S2_ITER=$v', d['value'], 'M MB/s; chunks against the oracle:', d['self_check']['against_the_oracle']['chunks_checked'], d['self_check']['against_the_oracle']['identical'])"
done
