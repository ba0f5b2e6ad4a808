#!/usr/bin/env python3
"""Whole-encoder timing: native frame loop + vp8drv_get_frame (complete VP8 frames out), one thread per GOP stream.
    python scripts/frame_bench.py [--streams N] [--frames K] [--width W --height H] [--partitions P]"""
import argparse, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=16)
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--partitions", type=int, default=8)
ap.add_argument("--check-ssim", type=int, default=0)
ap.add_argument("--switch-interval", type=float, default=0.0, help="sys.setswitchinterval (0 = Python's default 5 ms)")
ap.add_argument("--only", choices=("both", "on", "off"), default="both", help="which of the two legs to time")
ap.add_argument("--pipeline", action="store_true", help="one host thread: encode + get_frame_begin on every stream, then get_frame_end on every stream")
a = ap.parse_args()
if a.switch_interval > 0:
    sys.setswitchinterval(a.switch_interval)
seq = SynthSequence(a.width, a.height, seed=1)
W, H = seq.W, seq.H
mbs = (W // 16) * (H // 16)
nd = 8
dev = [tuple(api.to_device(p) for p in seq.frame(t)) for t in range(nd)]
api.device_synchronize()
drvs = [api.NativeDriver(W, H, gop_size=1 << 30, num_partitions=a.partitions, check_ssim=a.check_ssim) for _ in range(a.streams)]
sizes = [0] * a.streams

def work(k, n, emit):
    d = drvs[k]
    for t in range(n):
        y, u, v = dev[(t + 3 * k) % nd]
        d.encode_frame_device(y.data_ptr(), u.data_ptr(), v.data_ptr())
        if emit:
            sizes[k] += len(d.get_frame())
    d.hip.synchronize()

def run_pipeline(n, emit):
    t0 = time.perf_counter()
    for t in range(n):
        for k, d in enumerate(drvs):
            y, u, v = dev[(t + 3 * k) % nd]
            d.encode_frame_device(y.data_ptr(), u.data_ptr(), v.data_ptr())
            if emit:
                d.get_frame_begin()
        if emit:
            for k, d in enumerate(drvs):
                sizes[k] += len(d.get_frame_end())
    for d in drvs: d.hip.synchronize()
    return time.perf_counter() - t0

def run(n, emit):
    if a.pipeline:
        return run_pipeline(n, emit)
    th = [threading.Thread(target=work, args=(k, n, emit)) for k in range(a.streams)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0

run(4, True)
for emit in {"both": (False, True), "on": (True,), "off": (False,)}[a.only]:
    for k in range(a.streams): sizes[k] = 0
    el = run(a.frames, emit)
    fps = a.streams * a.frames / el
    print(f"{W}x{H} {a.streams} streams x {a.frames} frames, bitstream {'on ' if emit else 'off'}: {fps:8.1f} fps, {fps * mbs / 1e6:6.2f} M MB/s, "
          f"{el / a.frames * 1e3 / 1:7.3f} ms per frame and stream" + (f", {sum(sizes) / (a.streams * a.frames) / 1024:.1f} KiB per frame" if emit else ""))
for d in drvs: d.close()
