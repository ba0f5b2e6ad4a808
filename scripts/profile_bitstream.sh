# the frames-out leg (with_bitstream) under rocprofv3 --kernel-trace --stats: which of the entropy stage's kernels the device time goes
# to, and how many kernels run at once during the leg (the trace holds the headline leg first: the second half of the window is the leg)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03e}
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_bs -o bs --output-format csv -- python3 bench.py --gpus 1 --steps 40 --warmup 5 --only-bitstream --cpu-seconds 0 > gpurun_out/${TAG}_bs.json 2>/dev/null
cp gpurun_out/${TAG}_bs/bs_kernel_stats.csv gpurun_out/${TAG}_bitstream_kernel_stats.csv
python3 scripts/concurrency_of_trace.py gpurun_out/${TAG}_bs/bs_kernel_trace.csv 60 > gpurun_out/${TAG}_bitstream_concurrency.txt 2>&1 || true
find gpurun_out/${TAG}_bs -name "*kernel_trace.csv" -size +30M -delete
cut -c1-170 gpurun_out/${TAG}_bitstream_kernel_stats.csv | head -24
cat gpurun_out/${TAG}_bitstream_concurrency.txt | head -40
