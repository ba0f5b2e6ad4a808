cd $GRAFT_REPO_ROOT
for ov in 0 1; do VP8_BENCH_OVERLAP=$ov python3 - <<PY
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import bench
from vp8oclenc_amd import api
ov = int(os.environ["VP8_BENCH_OVERLAP"])
leg = bench.Leg(torch, api, 1920, 1080, 1, "all", -1.0, 8, 0, 1, overlap_filter=ov)
for _ in range(30): leg.step()
el, enq, nrefs = leg.run(300)
print("overlap_filter", ov, "single stream: %.4f ms/frame, %.2f M MB/s, refs %.2f" % (el / 300 * 1e3, leg.mbs * 300 / el / 1e6, nrefs))
leg.close()
PY
done
