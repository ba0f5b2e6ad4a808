# rocprofv3 kernel statistics of the native frame loop with complete frames out (scripts/native/frame_bench.cpp):
# one chunk (chain latency per kernel) and 16 chunks (what the default bench runs side by side)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
bash scripts/native/run_frame_bench.sh "1 60 8 threads 1" > /dev/null
for S in 1 16; do
rocprofv3 --kernel-trace --stats -d gpurun_out/r01k_native$S -o fr --output-format csv -- /tmp/frame_bench /tmp/frames.i420 1920 1088 $S 60 8 threads 1 > gpurun_out/r01k_native$S.txt 2>/dev/null
cat gpurun_out/r01k_native$S.txt
python3 - $S <<'PY'
import csv, sys
rows=list(csv.DictReader(open(f"gpurun_out/r01k_native{sys.argv[1]}/fr_kernel_stats.csv")))
for r in rows[:26]:
    print(f'{r["Name"][:52]:52s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us total {float(r["TotalDurationNs"])/1e6:8.1f} ms')
PY
done
