# builds scripts/native/frame_bench on the GPU box, writes the synthetic sequence as raw I420 and runs the configurations given
# as arguments ("<streams> <frames> <partitions> <mode> <bitstream>" each, quoted); default: the four corners at 16 chunks
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24}
W=${W:-1920}; H=${H:-1080}
/opt/rocm/bin/hipcc -O2 -std=c++17 -I include scripts/native/frame_bench.cpp -o /tmp/frame_bench -L vp8oclenc_amd -lvp8hip -Wl,-rpath,$PWD/vp8oclenc_amd -lpthread || exit 1
python3 - <<PY
import sys
sys.path.insert(0, ".")
import numpy as np
from vp8oclenc_amd.synth import SynthSequence
s = SynthSequence($W, $H, seed=1)
with open("/tmp/frames.i420", "wb") as f:
    for t in range(8):
        for p in s.frame(t): f.write(np.ascontiguousarray(p).tobytes())
print("coded size", s.W, s.H)
PY
HH=$(( (H + 15) / 16 * 16 ))
if [ $# -eq 0 ]; then set -- "16 60 8 threads 0" "16 60 8 threads 1" "16 60 8 pipeline 0" "16 60 8 pipeline 1"; fi
for cfg in "$@"; do set -- $cfg; /tmp/frame_bench /tmp/frames.i420 $W $HH $1 $2 $3 $4 $5 ${6:-0} ${7:-0}; done
