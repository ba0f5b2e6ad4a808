# what a frame costs without one of its kernels (outputs are garbage: timing only)
cd $GRAFT_REPO_ROOT
for v in none mb lf s1 s2 "mb,lf" "s1,s2" "s1,s2,mb" none; do VP8HIP_EXPERIMENT_SKIP=$v python3 bench.py --steps 40 --warmup 10 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('skip $v', round(d['value']/1e6,2), d['config']['ms_per_frame'])"; done
