# round 3: the driver's own bench command un-profiled, then the same command (side legs off) under rocprofv3 (kernel trace + stats),
# then the default command; summaries go to gpurun_out/<tag>_*, the ones to keep are copied into profiles/ by hand
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03a}
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 > gpurun_out/${TAG}_kt.json 2>/dev/null
cp gpurun_out/${TAG}_kt/kt_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
python3 scripts/concurrency_of_trace.py gpurun_out/${TAG}_kt/kt_kernel_trace.csv 160 > gpurun_out/${TAG}_concurrency.txt 2>/dev/null || true
if [ -z "$NO_DEFAULT" ]; then python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; fi
find gpurun_out/${TAG}_kt -name "*kernel_trace.csv" -size +30M -delete      # the raw trace is large; the stats and summaries stay
python3 scripts/show_bench.py gpurun_out/${TAG}_bench_driver.json
head -30 gpurun_out/${TAG}_kernel_stats.csv
cat gpurun_out/${TAG}_concurrency.txt | head -30
