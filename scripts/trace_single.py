"""one GOP chunk, frame after frame, loop filter on its own stream: the workload of the single-stream kernel trace"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vp8oclenc_amd import api
ov = int(os.environ.get("VP8_BENCH_OVERLAP", "1"))
leg = bench.Leg(api, 1920, 1080, 1, "all", -1.0, 8, 0, 1, overlap_filter=ov)
for _ in range(20): leg.step()
el, enq, nrefs = leg.run(60)
print("overlap_filter", ov, "single stream: %.4f ms/frame" % (el / 60 * 1e3))
leg.close()
