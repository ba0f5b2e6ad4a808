# same-box A/B of the frames-out leg: entropy stage batched (vp8drv_batch_get_frame_begin) or member by member
cd $GRAFT_REPO_ROOT
N=${1:-3}
for i in $(seq $N); do for v in 1 0; do env VP8_BENCH_ENT_BATCH=$v python3 bench.py --steps 60 --warmup 10 --cpu-seconds 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ENT_BATCH=$v headline', round(d['value']/1e6,2), 'with_bitstream', round(d['with_bitstream']['value']/1e6,2), d['with_bitstream']['fps'])"; done; done
