# round 6: the driver's own bench command un-profiled (every leg: drop_in, the chunk-parallel cpu_baseline, the self-checks against the
# new oracle tables), the same command (side legs off) under rocprofv3 --kernel-trace --stats, the one-video timeline, the FETCH_SIZE /
# WRITE_SIZE passes (each in a run of its own, no trace beside it), the A/B of the fused search launches in batches, the drop-in programs
# on their own with their host timelines, the full-length runs against the committed oracle digests.
# Summaries land in gpurun_out/<tag>_*; the ones to keep are copied into profiles/ afterwards.     bash scripts/profile_round6.sh r06a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06a}
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err; echo "driver cmd rc=$?"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 > gpurun_out/${TAG}_kt.json 2>/dev/null; echo "trace rc=$?"
cp gpurun_out/${TAG}_kt/kt_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
python3 scripts/concurrency_of_trace.py gpurun_out/${TAG}_kt/kt_kernel_trace.csv 160 > gpurun_out/${TAG}_concurrency.txt 2>/dev/null || true
find gpurun_out/${TAG}_kt -name "*kernel_trace.csv" -size +30M -delete
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt1 -o kt1 --output-format csv -- python3 scripts/trace_single.py > gpurun_out/${TAG}_single.txt 2>/dev/null
python3 scripts/analyze_trace.py $(find gpurun_out/${TAG}_kt1 -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/${TAG}_single_stream_timeline.txt 2>&1
cp gpurun_out/${TAG}_kt1/kt1_kernel_stats.csv gpurun_out/${TAG}_kernel_stats_1gop.csv
find gpurun_out/${TAG}_kt1 -name "*kernel_trace.csv" -size +20M -delete
CMD1="python3 bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --no-side-legs --cpu-seconds 0"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmc_fetch -o fetch --output-format csv -- $CMD1 > gpurun_out/${TAG}_pmc_fetch.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc_write -o write --output-format csv -- $CMD1 > gpurun_out/${TAG}_pmc_write.json 2>/dev/null
rocprofv3 --pmc SQ_INSTS_VALU -d gpurun_out/${TAG}_pmc_valu -o valu --output-format csv -- $CMD1 > gpurun_out/${TAG}_pmc_valu.json 2>/dev/null
for d in fetch write valu; do find gpurun_out/${TAG}_pmc_$d -name "*.csv" -size +20M -delete; done
# fewer launches per batch step?  the fused forms of the hierarchical search against a launch per level, alternating on this box
for m in 0 1 0 3 0 2 0; do
  VP8HIP_BATCH_S1_COARSE=$m python3 bench.py --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print('VP8HIP_BATCH_S1_COARSE=$m', d['value'], 'M MB/s; chunks against the oracle:', d['self_check']['against_the_oracle']['chunks_checked'], d['self_check']['against_the_oracle']['identical'])"
done > gpurun_out/${TAG}_fused_search_launches_ab.txt 2>&1
python3 scripts/drop_in_bench.py --repeats 4 --out gpurun_out/${TAG}_drop_in.json > /dev/null 2> gpurun_out/${TAG}_drop_in.err
python3 scripts/full_length_oracle.py --verify > gpurun_out/${TAG}_full_length_verify.txt 2>&1; echo "verify rc=$?"
if [ -z "$NO_DEFAULT" ]; then python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "default rc=$?"; fi
python3 scripts/show_bench.py gpurun_out/${TAG}_bench_driver.json
head -22 gpurun_out/${TAG}_kernel_stats.csv
head -12 gpurun_out/${TAG}_concurrency.txt
cat gpurun_out/${TAG}_fused_search_launches_ab.txt
tail -14 gpurun_out/${TAG}_full_length_verify.txt
