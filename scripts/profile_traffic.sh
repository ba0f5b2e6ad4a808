# HBM traffic per launch of every kernel of the path: FETCH_SIZE and WRITE_SIZE, each in a pass of its own, on one chunk per launch
# (summarised into profiles/pmc_traffic.json by scripts/summarize_rocprof.py --fetch ... --write ...):  bash scripts/profile_traffic.sh r04m
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r04m}
CMD1="python3 bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --no-side-legs --cpu-seconds 0"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmc_fetch -o fetch --output-format csv -- $CMD1 > gpurun_out/${TAG}_pmc_fetch.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc_write -o write --output-format csv -- $CMD1 > gpurun_out/${TAG}_pmc_write.json 2>/dev/null
for d in fetch write; do find gpurun_out/${TAG}_pmc_$d -name "*.csv" -size +20M -delete; done
ls -la gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write
