cd $GRAFT_REPO_ROOT
for cfg in "32 4" "40 4" "48 4" "64 4" "24 4" "32 4"; do set -- $cfg; python3 bench.py --steps 40 --warmup 10 --gops-per-gpu $1 --batch $2 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunks $1 batch $2', round(d['value']/1e6,2), d['config']['ms_per_frame'])"; done
