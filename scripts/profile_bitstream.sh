# the frames-out leg (with_bitstream) under rocprofv3 --kernel-trace --stats: which of the entropy stage's kernels the device time goes to
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03e}
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_bs -o bs --output-format csv -- python3 bench.py --gpus 1 --steps 40 --warmup 5 --only-bitstream --cpu-seconds 0 > gpurun_out/${TAG}_bs.json 2>/dev/null
cp gpurun_out/${TAG}_bs/bs_kernel_stats.csv gpurun_out/${TAG}_bitstream_kernel_stats.csv
find gpurun_out/${TAG}_bs -name "*kernel_trace.csv" -size +30M -delete
cut -c1-170 gpurun_out/${TAG}_bitstream_kernel_stats.csv | head -40
