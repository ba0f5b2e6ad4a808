cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r01h_frame -o fr --output-format csv -- python3 scripts/frame_bench.py --streams 1 --frames 40 > gpurun_out/r01h_frame.txt 2>/dev/null
head -30 gpurun_out/r01h_frame/fr_kernel_stats.csv | cut -c1-150
