#!/usr/bin/env python3
"""bench.py's side legs over and over in ONE process (the rare crash in the teardown of a leg's contexts was seen there).
    python -X faulthandler scripts/stress_side_leg.py [cycles] [which]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vp8oclenc_amd import api
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
which = sys.argv[2] if len(sys.argv) > 2 else "all"
legs = {"720p": ((1280, 720, 48, "last", -1.0, 20, 5), dict(batch=4)),
        "4k": ((3840, 2160, 16, "all", -1.0, 10, 3), dict(batch=2)),
        "ssim93": ((1920, 1080, 48, "all", 0.93, 20, 5), dict(batch=6)),
        "gop150": ((1920, 1080, 48, "all", -1.0, 20, 5), dict(batch=6, gop=150)),
        "conf": ((1920, 1080, 48, "all", -1.0, 20, 5), dict(batch=6, conformant=1))}
t0 = time.time()
for c in range(cycles):
    for name, (a, kw) in legs.items():
        if which != "all" and which != name:
            continue
        r = bench.side_leg(api, *a, 0, **kw)
        print(f"cycle {c} {name}: {r['value'] / 1e6:.1f} M  ({time.time() - t0:.0f} s)", flush=True)
print("done")
