cd $GRAFT_REPO_ROOT
for p in 0 512 1024 1536 2048 3072 0; do VP8HIP_PERSIST=$p python3 bench.py --steps 40 --warmup 10 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('persist $p', round(d['value']/1e6,2), d['config']['ms_per_frame'])"; done
