#!/usr/bin/env python3
"""bench.py's headline flow (per-kernel events on chunk 0, the dominant kernel timed on every chunk, the frames-out leg, teardown)
over and over in ONE process: hunting the rare crash in the teardown.   python -X faulthandler scripts/stress_headline_flow.py [cycles] [noprof] [nobits]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vp8oclenc_amd import api
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 20
noprof, nobits = "noprof" in sys.argv, "nobits" in sys.argv
t0 = time.time()
for c in range(cycles):
    leg = bench.Leg(api, 1920, 1080, 48, "all", -1.0, 4, 0, seed=1, batch=6)
    if not noprof:
        leg.drv[0].hip.profile_enable(api.K_NAMES)
        for d in leg.drv:
            d.hip.profile_search2_clock(True)
    leg.clock_read()
    for _ in range(5):
        leg.step()
    api.device_synchronize()
    if not noprof:
        leg.drv[0].hip.profile_read()
        for d in leg.drv:
            d.hip.profile_search2_clock(False)
        leg.clock_read()
        leg.profile(["search2"])
    el, _, _ = leg.run(20)
    if not noprof:
        leg.profile_read()
        leg.profile([])
    bs = None if nobits else bench.bitstream_leg(leg, 20)["value"]
    leg.close()
    print(f"cycle {c}: {leg.mbs * 20 * 48 / el / 1e6:.1f} M, frames out {(bs or 0) / 1e6:.1f} M ({time.time() - t0:.0f} s)", flush=True)
print("done")
