# wave64 vector instructions per launch of every kernel of one video coded with its frames delivered (SQ_INSTS_VALU, a pass of its own)
#     bash scripts/pmc_frames_out.sh r06fo
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06fo}
rocprofv3 --pmc SQ_INSTS_VALU -d gpurun_out/${TAG}_pmc -o valu --output-format csv -- python3 bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --only-bitstream --cpu-seconds 0 > gpurun_out/${TAG}_pmc.json 2>/dev/null; echo "pmc rc=$?"
python3 - <<PY > gpurun_out/${TAG}_valu_by_kernel.txt
import csv, collections, glob, re
f = glob.glob("gpurun_out/${TAG}_pmc/*counter_collection.csv")[0]
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "SQ_INSTS_VALU": continue
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
    tot[k] += float(r["Counter_Value"]); n[k] += 1
for k, v in tot.most_common(40):
    print("%-28s %6d launches %12.0f wave instructions each %14.0f total" % (k, n[k], v / n[k], v))
PY
find gpurun_out/${TAG}_pmc -name "*.csv" -size +20M -delete
cat gpurun_out/${TAG}_valu_by_kernel.txt
