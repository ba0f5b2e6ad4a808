# round 3: PMC passes (each counter in a run of its own, no kernel trace beside it) on one GOP chunk per launch, and the kernel trace +
# timeline of one video coded frame after frame; post-processing into profiles/ is done afterwards (make_pmc_valu.py, summarize_rocprof.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}
CMD="python3 bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --no-side-legs --cpu-seconds 0"
rocprofv3 --pmc SQ_INSTS_VALU -d gpurun_out/${TAG}_pmc_valu -o valu --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_valu.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmc_fetch -o fetch --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_fetch.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc_write -o write --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_write.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt1 -o kt1 --output-format csv -- python3 scripts/trace_single.py > gpurun_out/${TAG}_single.txt 2>/dev/null
python3 scripts/analyze_trace.py $(find gpurun_out/${TAG}_kt1 -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/${TAG}_single_stream_timeline.txt 2>&1
ls gpurun_out/${TAG}_pmc_valu gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_kt1 | head -30
cat gpurun_out/${TAG}_single.txt; head -40 gpurun_out/${TAG}_single_stream_timeline.txt
