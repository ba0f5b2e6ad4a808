# same-box A/B of an environment switch: bash scripts/ab.sh VAR valueA valueB [rounds]
cd $GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3; N=${4:-4}
for i in $(seq $N); do for v in "$A" "$B"; do env $VAR=$v python3 bench.py --steps 60 --warmup 10 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']/1e6,2), d['config']['ms_per_frame'])"; done; done
