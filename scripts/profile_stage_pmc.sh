# the entropy stage's kernels by counters, one video with frames out (one frame per launch): instructions by class, wave cycles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r04p}
CMD="python3 scripts/trace_single_frames_out_native.py 40"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d gpurun_out/${TAG}_stage_a -o a --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d gpurun_out/${TAG}_stage_b -o b --output-format csv -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d gpurun_out/${TAG}_stage_c -o c --output-format csv -- $CMD > /dev/null 2>&1
TAG=$TAG python3 - <<'PY'
import csv, collections, glob, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04p"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/%s_stage_*/*counter_collection.csv" % "TAGX".replace("TAGX", __import__("os").environ.get("TAG", "r04p"))):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).split("::")[-1]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
print("%-24s" % "kernel" + "".join("%16s" % c[-15:] for c in names))
for k, v in sorted(acc.items()):
    print("%-24s" % k[:24] + "".join("%16.0f" % (sum(v[c]) / len(v[c]) if v[c] else 0) for c in names))
PY
