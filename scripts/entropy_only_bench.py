"""Entropy stage alone (vp8drv_get_frame repeated on one encoded frame), S streams on S host threads: the stage's
saturated device cost per frame, to set beside the inter path's 0.197 ms."""
import argparse, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, ".")
import torch
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, nargs="+", default=[1, 4, 8, 16])
ap.add_argument("--reps", type=int, default=60)
ap.add_argument("--partitions", type=int, default=8)
a = ap.parse_args()
seq = SynthSequence(1920, 1080, seed=1)
W, H = seq.W, seq.H
dev = [tuple(torch.from_numpy(p).cuda() for p in seq.frame(t)) for t in range(4)]
S = max(a.streams)
drvs = [api.NativeDriver(W, H, gop_size=1 << 30, num_partitions=a.partitions) for _ in range(S)]
ref = None
for k, d in enumerate(drvs):
    for t in range(3):
        y, u, v = dev[t]
        d.encode_frame_device(y.data_ptr(), u.data_ptr(), v.data_ptr())
    f = bytes(d.get_frame())
    ref = ref or f
    assert f == ref

def work(k, n):
    for _ in range(n):
        assert len(drvs[k].get_frame()) == len(ref)

for s in a.streams:
    th = [threading.Thread(target=work, args=(k, a.reps)) for k in range(s)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    print(f"{s:2d} streams: {el / (s * a.reps) * 1e3:7.3f} ms per frame (device-wide), {el / a.reps * 1e3:7.3f} ms per frame and stream")
for d in drvs: d.close()
