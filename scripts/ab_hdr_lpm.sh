# same-box A/B of the header walks' lane mapping in batches (VP8HIP_HDR_BATCH_LPM lanes per macroblock): the frames-out leg
cd $GRAFT_REPO_ROOT
for v in 2 4 8 2 4 8 2 4 8 16; do
  echo "== VP8HIP_HDR_BATCH_LPM=$v"
  VP8HIP_HDR_BATCH_LPM=$v python3 bench.py --gpus 1 --steps 40 --warmup 5 --only-bitstream --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=p['with_bitstream']
print('value', p['value'], 'with_bitstream', b['value'], b['fps'], b['self_check']['identical'])"
done
