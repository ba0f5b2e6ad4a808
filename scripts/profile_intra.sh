cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r01j_intra -o intra --output-format csv -- python3 scripts/intra_bench.py > gpurun_out/r01j_intra.txt 2>/dev/null
cat gpurun_out/r01j_intra.txt
grep -E "k_intra|k_ssim" gpurun_out/r01j_intra/intra_kernel_stats.csv | cut -c1-150
