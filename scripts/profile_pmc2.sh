# round 2 PMC passes (each counter set in its own run, kernel trace off): VALU instructions, FETCH_SIZE, WRITE_SIZE on a single GOP stream
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02f}
CMD="python3 bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --no-side-legs --cpu-seconds 0"
rocprofv3 --pmc SQ_INSTS_VALU -d gpurun_out/${TAG}_pmc_valu -o valu --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_valu.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmc_fetch -o fetch --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_fetch.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc_write -o write --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_write.json 2>/dev/null
ls -la gpurun_out/${TAG}_pmc_valu gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write
