# round 2: the driver's own bench command under rocprofv3 (kernel trace + stats), plus the un-profiled line beside it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02c}
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o kt --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-side-legs --cpu-seconds 0 > gpurun_out/${TAG}_kt.json 2>/dev/null
python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -1 gpurun_out/${TAG}_bench_driver.json | cut -c1-300
tail -1 gpurun_out/${TAG}_bench_default.json | cut -c1-300
ls gpurun_out/${TAG}_kt
