"""32-chunk leg, then the single-stream leg in the same process (for a kernel trace of the latter)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vp8oclenc_amd import api
leg = bench.Leg(torch, api, 1920, 1080, 32, "all", -1.0, 8, 0, 1, batch=4)
for _ in range(5): leg.step()
el, enq, nrefs = leg.run(20)
leg.close()
r = bench.side_leg(torch, api, 1920, 1080, 1, "all", -1.0, 60, 20, 0, nd=8)
print("after load", r["ms_per_frame"], "lf", r["loop_filter_ms_by_its_own_clock"], flush=True)
