"""single-stream leg before and after a 32-chunk leg in the same process: how long does the part take to come back?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vp8oclenc_amd import api

def single(tag, n=200):
    r = bench.side_leg(api, 1920, 1080, 1, "all", -1.0, n, 20, 0, nd=8)
    print(tag, r["ms_per_frame"], "lf", r["loop_filter_ms_by_its_own_clock"], "clock", r["shader_clock_ghz"], flush=True)

single("fresh")
leg = bench.Leg(api, 1920, 1080, 32, "all", -1.0, 8, 0, 1, batch=4)
for _ in range(10): leg.step()
el, enq, nrefs = leg.run(int(os.environ.get("LOAD_STEPS", "120")))
print("32 chunks: %.2f M MB/s" % (leg.mbs * int(os.environ.get("LOAD_STEPS", "120")) * 32 / el / 1e6), flush=True)
leg.close()
t0 = time.time()
for i in range(12):
    single("%.1f s after the load" % (time.time() - t0), 1000)
    time.sleep(float(os.environ.get("NAP", "0")))
