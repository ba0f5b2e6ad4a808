# bench.py leaves through the ordinary teardown (every context destroyed, no os._exit): N runs of the driver's command with the side legs
# (each starts a child process that creates and destroys some 150 contexts) and M runs without; exit codes and values in gpurun_out/<tag>_exit_codes.txt
cd $GRAFT_REPO_ROOT
TAG=${1:-r04}; N=${2:-10}; M=${3:-40}
out=gpurun_out/${TAG}_exit_codes.txt
: > $out
for i in $(seq $N); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 > /tmp/b.json 2> /tmp/b.err; rc=$?
  echo "full $i rc=$rc $(python3 -c "import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['single_stream']['ms_per_frame'], d.get('few_stream_legs_error','legs ok'), d['self_check']['identical'], d['with_bitstream']['self_check']['identical'])" 2>&1 | tail -1)" >> $out
done
for i in $(seq $M); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 > /tmp/b.json 2> /tmp/b.err; rc=$?
  echo "headline $i rc=$rc $(python3 -c "import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['self_check']['identical'])" 2>&1 | tail -1)" >> $out
done
grep -c "rc=0" $out; grep -v "rc=0" $out | head; tail -3 $out
