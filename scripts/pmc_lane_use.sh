# how many of a wave's 64 lanes are active in the vector instructions each kernel issues (VALUUtilization = SQ_THREAD_CYCLES_VALU /
# (SQ_ACTIVE_INST_VALU x 64), a counter pass of its own): one video coded with its frames delivered, every kernel of the frame and of the entropy stage
#     bash scripts/pmc_lane_use.sh r06
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06}
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU -d gpurun_out/${TAG}_lanes -o lanes --output-format csv -- python3 bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --only-bitstream --cpu-seconds 0 > gpurun_out/${TAG}_lanes.json 2>/dev/null; echo "pmc rc=$?"
python3 - <<PY > gpurun_out/${TAG}_lane_use.txt
import csv, collections, glob, re
f = glob.glob("gpurun_out/${TAG}_lanes/*counter_collection.csv")[0]
t = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
    t[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_ACTIVE_INST_VALU": n[k] += 1
rows = [(v["SQ_ACTIVE_INST_VALU"], k, v) for k, v in t.items() if v["SQ_ACTIVE_INST_VALU"] > 0]
for a, k, v in sorted(rows, reverse=True):
    print("%-28s %6d launches   active lanes per vector instruction %5.1f %%   (instruction-cycles %.3g)" % (k, n[k], 100.0 * v["SQ_THREAD_CYCLES_VALU"] / (a * 64.0), a))
PY
find gpurun_out/${TAG}_lanes -name "*.csv" -size +20M -delete
cat gpurun_out/${TAG}_lane_use.txt
