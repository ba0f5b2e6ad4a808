# How busy the VALUs are, by the hardware's own counters (a pass of its own, kernel trace off): SQ_ACTIVE_INST_VALU (cycles a
# SIMD's VALU is executing, summed over SIMDs), SQ_BUSY_CU_CYCLES and GRBM_GUI_ACTIVE per dispatch, on launches of six chunks
# as the headline run makes them.  Counter collection serialises the dispatches: each kernel is seen ALONE on the part.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02v}
CMD="python3 bench.py --gops-per-gpu 6 --batch 6 --steps 12 --warmup 4 --no-side-legs --cpu-seconds 0"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_pmc_busy -o busy --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_busy.json 2>gpurun_out/${TAG}_pmc_busy.err
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVES -d gpurun_out/${TAG}_pmc_busy2 -o busy2 --output-format csv -- $CMD > gpurun_out/${TAG}_pmc_busy2.json 2>gpurun_out/${TAG}_pmc_busy2.err
ls -la gpurun_out/${TAG}_pmc_busy gpurun_out/${TAG}_pmc_busy2
tail -3 gpurun_out/${TAG}_pmc_busy.err
