cd $GRAFT_REPO_ROOT
for cfg in "20 5" "20 30" "20 100" "40 5" "80 5" "160 5" "20 5"; do set -- $cfg; python3 bench.py --steps $1 --warmup $2 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $1 warmup $2', d['value'], d['config']['ms_per_frame'], d['roofline']['avg_launch_ms'], d['host_enqueue_ms_per_frame'])"; done
