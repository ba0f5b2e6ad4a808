cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 bench.py > gpurun_out/r01f_default.json 2> gpurun_out/r01f_default.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r01f_kt16 -o kt16 --output-format csv -- python3 bench.py --cpu-seconds 0 > gpurun_out/r01f_kt16.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d gpurun_out/r01f_kt1 -o kt1 --output-format csv -- python3 bench.py --steps 240 --warmup 16 --gops-per-gpu 1 --cpu-seconds 0 > gpurun_out/r01f_kt1.json 2>/dev/null
tail -1 gpurun_out/r01f_default.json | cut -c1-400
