cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do python3 bench.py --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i', round(d['value']/1e6,2), d['config']['ms_per_frame'], d['roofline']['kernel'])"; done
