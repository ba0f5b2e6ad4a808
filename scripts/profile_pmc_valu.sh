cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU -d gpurun_out/r01g_pmc_valu -o valu --output-format csv -- python3 bench.py --steps 32 --warmup 8 --gops-per-gpu 1 --cpu-seconds 0 > gpurun_out/r01g_pmc_valu.json 2>/dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU -d gpurun_out/r01g_pmc_waves -o waves --output-format csv -- python3 bench.py --steps 32 --warmup 8 --gops-per-gpu 1 --cpu-seconds 0 > gpurun_out/r01g_pmc_waves.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/r01g_intra -o intra --output-format csv -- python3 scripts/intra_bench.py > gpurun_out/r01g_intra.txt 2>/dev/null
ls gpurun_out/r01g_pmc_valu gpurun_out/r01g_pmc_waves gpurun_out/r01g_intra
