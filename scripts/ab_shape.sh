# same-box A/B of the headline's shape (chunks in flight x chunks per batched launch): value and the frames-out leg
cd $GRAFT_REPO_ROOT
for gb in "48 6" "56 7" "64 8" "48 6" "56 7" "64 8" "72 8"; do
  set -- $gb
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --gops-per-gpu $1 --batch $2 --only-bitstream --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=p['with_bitstream']
print('$1 x $2: value', round(p['value']/1e6,2), 'with_bitstream', round(b['value']/1e6,2), b['self_check']['identical'], p['self_check']['against_the_oracle']['chunks_checked'])"
done
