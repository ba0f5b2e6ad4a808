# round 4: the driver's own bench command un-profiled, the same command (side legs off) under rocprofv3 --kernel-trace --stats, the PMC
# passes (each counter set in a run of its own, no trace beside it: instructions, FETCH_SIZE, WRITE_SIZE on one chunk per launch; VALU-busy
# and MFMA counters on launches of six chunks), the one-video timeline, then the default command.  Summaries land in gpurun_out/<tag>_*;
# the ones to keep are copied into profiles/ afterwards (summarize_rocprof.py, make_pmc_valu.py).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r04c}
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err; echo "driver cmd rc=$?"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 > gpurun_out/${TAG}_kt.json 2>/dev/null; echo "trace rc=$?"
cp gpurun_out/${TAG}_kt/kt_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
python3 scripts/concurrency_of_trace.py gpurun_out/${TAG}_kt/kt_kernel_trace.csv 160 > gpurun_out/${TAG}_concurrency.txt 2>/dev/null || true
find gpurun_out/${TAG}_kt -name "*kernel_trace.csv" -size +30M -delete
CMD1="python3 bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --no-side-legs --cpu-seconds 0"
rocprofv3 --pmc SQ_INSTS_VALU -d gpurun_out/${TAG}_pmc_valu -o valu --output-format csv -- $CMD1 > gpurun_out/${TAG}_pmc_valu.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmc_fetch -o fetch --output-format csv -- $CMD1 > gpurun_out/${TAG}_pmc_fetch.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc_write -o write --output-format csv -- $CMD1 > gpurun_out/${TAG}_pmc_write.json 2>/dev/null
CMD6="python3 bench.py --gops-per-gpu 6 --batch 6 --steps 12 --warmup 4 --no-side-legs --cpu-seconds 0"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_pmc_busy -o busy --output-format csv -- $CMD6 > gpurun_out/${TAG}_pmc_busy.json 2>gpurun_out/${TAG}_pmc_busy.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/${TAG}_pmc_busy2 -o busy2 --output-format csv -- $CMD6 > gpurun_out/${TAG}_pmc_busy2.json 2>gpurun_out/${TAG}_pmc_busy2.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/${TAG}_pmc_mfma -o mfma --output-format csv -- $CMD6 > gpurun_out/${TAG}_pmc_mfma.json 2>gpurun_out/${TAG}_pmc_mfma.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt1 -o kt1 --output-format csv -- python3 scripts/trace_single.py > gpurun_out/${TAG}_single.txt 2>/dev/null
python3 scripts/analyze_trace.py $(find gpurun_out/${TAG}_kt1 -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/${TAG}_single_stream_timeline.txt 2>&1
cp gpurun_out/${TAG}_kt1/kt1_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_1gop.csv
find gpurun_out/${TAG}_kt1 -name "*kernel_trace.csv" -size +20M -delete
if [ -z "$NO_DEFAULT" ]; then python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "default rc=$?"; fi
for d in valu fetch write busy busy2 mfma; do find gpurun_out/${TAG}_pmc_$d -name "*.csv" -size +20M -delete; done
python3 scripts/show_bench.py gpurun_out/${TAG}_bench_driver.json
head -25 gpurun_out/${TAG}_kernel_stats.csv
cat gpurun_out/${TAG}_concurrency.txt | head -12
ls gpurun_out/${TAG}_pmc_*/ | head -40
tail -3 gpurun_out/${TAG}_pmc_mfma.err
