"""one GOP frame after frame for several seconds: does the part hold the rate?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vp8oclenc_amd import api
leg = bench.Leg(api, 1920, 1080, 1, "all", -1.0, 8, 0, 1, overlap_filter=1)
for _ in range(20): leg.step()
for i in range(8):
    el, enq, nrefs = leg.run(2000)
    ms, n, ghz = leg.clock_read() if hasattr(leg, "clock_read") else (0, 0, 0)
    print("2000 frames: %.4f ms/frame; loop filter by its own clock %.4f ms (%d launches), clock %s" % (el / 2000 * 1e3, ms / max(n, 1), n, ghz), flush=True)
leg.close()
