#!/usr/bin/env python3
"""Create / run / destroy batches of contexts over and over (a bench leg in small): hunting a rare crash in the destroy path.
    python -X faulthandler scripts/stress_create_destroy.py [cycles] [W H] [gop] [chunks] [batch]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 100
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 192)
gop = int(sys.argv[4]) if len(sys.argv) > 4 else 4
G = int(sys.argv[5]) if len(sys.argv) > 5 else 48
B = int(sys.argv[6]) if len(sys.argv) > 6 else 6
seq = SynthSequence(W, H, seed=1)
dev = [tuple(api.to_device(p) for p in seq.frame(t)) for t in range(4)]
ptrs = [tuple(p.data_ptr() for p in f) for f in dev]
t0 = time.time()
for c in range(cycles):
    drv = [api.NativeDriver(seq.W, seq.H, gop_size=gop, device_params=1, check_ssim=1) for _ in range(G)]
    groups = [list(range(i, min(i + B, G))) for i in range(0, G, B)]
    batches = [api.NativeBatch([drv[k] for k in m]) for m in groups]
    api.NativeBatch.encode_frames_device_all(batches, 6, ptrs, [[(3 * k) % 4 for k in m] for m in groups])
    for d in drv:
        d.resolve()
    api.device_synchronize()
    for b in batches:
        b.close()
    for d in drv:
        d.close()
    if c % 20 == 19:
        print(f"cycle {c + 1}: {time.time() - t0:.1f} s", flush=True)
print("done", cycles)
