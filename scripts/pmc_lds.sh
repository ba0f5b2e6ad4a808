# is the LDS what a search kernel waits for?  Counter passes of their own over one video (k_search2, k_search1_pl, k_mb_p):
# LDS instructions' active cycles, bank-conflict cycles, LDS index unit active cycles, against the vector ALU's active cycles and the kernel's busy cycles
#     bash scripts/pmc_lds.sh r06
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06}
CMD="python3 bench.py --gops-per-gpu 1 --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0"
: > gpurun_out/${TAG}_lds_use.txt
for set in "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU"; do
  rm -rf gpurun_out/${TAG}_lds
  rocprofv3 --pmc $set --kernel-include-regex "k_search2|k_search1_pl|k_mb_p" -d gpurun_out/${TAG}_lds -o l --output-format csv -- $CMD > /dev/null 2>&1
  python3 - <<PY >> gpurun_out/${TAG}_lds_use.txt
import csv, collections, glob, re
fs = glob.glob("gpurun_out/${TAG}_lds/*counter_collection.csv")
if fs:
    t = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        t[(m.group(1), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(t.items()): print("%-16s %-26s %5d launches  %14.0f per launch" % (k[0], k[1], len(v), sum(v) / len(v)))
else:
    print("no counters for: $set")
PY
done
rm -rf gpurun_out/${TAG}_lds
cat gpurun_out/${TAG}_lds_use.txt
