cd $GRAFT_REPO_ROOT
for cfg in "16 24" "24 24" "24 32" "32 32" "12 24" "16 24"; do set -- $cfg; GPU_MAX_HW_QUEUES=$2 python3 bench.py --steps 60 --warmup 10 --gops-per-gpu $1 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('gops $1 queues $2', round(d['value']/1e6,2), d['config']['ms_per_frame'])"; done
