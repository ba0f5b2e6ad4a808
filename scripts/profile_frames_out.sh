# the frames-out leg (with_bitstream) on its own: the kernels of the entropy stage among the frame's, how many run at once
#     bash scripts/profile_frames_out.sh r06fo
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06fo}
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --only-bitstream --cpu-seconds 0 > gpurun_out/${TAG}_kt.json 2>/dev/null; echo "trace rc=$?"
cp gpurun_out/${TAG}_kt/kt_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
python3 scripts/concurrency_of_trace.py gpurun_out/${TAG}_kt/kt_kernel_trace.csv 160 > gpurun_out/${TAG}_concurrency.txt 2>/dev/null || true
python3 - <<PY > gpurun_out/${TAG}_last_second.txt
import csv, collections, glob
f = glob.glob("gpurun_out/${TAG}_kt/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
end = rows[-1][1]
# the frames-out leg is the last leg of the run: the last 0.25 s of the trace
win = [r for r in rows if r[0] > end - 250_000_000]
tot = collections.Counter(); n = collections.Counter()
import re
for s, e, k, q in win:
    m = re.search(r"(k_[a-z0-9_]+)", k); k = m.group(1) if m else k[:40]
    tot[k] += e - s; n[k] += 1
span = win[-1][1] - win[0][0]
print("window %.1f ms, %d launches" % (span / 1e6, len(win)))
for k, v in tot.most_common(40):
    print("%-34s %6d launches %10.1f us each %8.2f ms total  %5.1f %% of the sum" % (k, n[k], v / n[k] / 1e3, v / 1e6, 100 * v / sum(tot.values())))
PY
find gpurun_out/${TAG}_kt -name "*kernel_trace.csv" -size +30M -delete
python3 -c "import json;d=json.loads(open('gpurun_out/${TAG}_kt.json').read().strip().splitlines()[-1]);print(d['value'], d['with_bitstream']['value'])"
head -30 gpurun_out/${TAG}_kernel_stats.csv
cat gpurun_out/${TAG}_last_second.txt
head -12 gpurun_out/${TAG}_concurrency.txt
