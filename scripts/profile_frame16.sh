cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r01h_frame16 -o fr --output-format csv -- python3 scripts/frame_bench.py --streams 16 --frames 30 > gpurun_out/r01h_frame16.txt 2>/dev/null
cat gpurun_out/r01h_frame16.txt | tail -2
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r01h_frame16/fr_kernel_stats.csv")))
for r in rows[:24]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us total {float(r["TotalDurationNs"])/1e6:9.1f} ms')
PY
