#!/usr/bin/env python3
"""Long equality run of the ways one video can be coded: plain, loop filter overlapped (vp8hip_filter_overlap), and as a
member of a batch beside other videos -- the SHA-256 of all frames and of the last reconstruction must agree.
    python scripts/soak_modes.py [--width 1920 --height 1080 --frames 600 --gop 150 --partitions 4]"""
import argparse, hashlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=600); ap.add_argument("--gop", type=int, default=150); ap.add_argument("--partitions", type=int, default=4)
a = ap.parse_args()
s = SynthSequence(a.width, a.height, seed=5)
ND = 24
dev = [tuple(torch.from_numpy(p).cuda() for p in s.frame(t)) for t in range(ND)]
ptr = [tuple(p.data_ptr() for p in f) for f in dev]
cfg = dict(gop_size=a.gop, altref_range=5, num_partitions=a.partitions, device_params=1, check_ssim=0)


def single(overlap):
    d = api.NativeDriver(s.W, s.H, overlap_filter=overlap, **cfg)
    h = hashlib.sha256()
    t0 = time.perf_counter()
    for t in range(a.frames):
        d.encode_frame_device(*ptr[(t * 7) % ND])
        h.update(d.get_frame())
    el = time.perf_counter() - t0
    for p in d.hip.download_last():
        h.update(p.tobytes())
    d.close()
    return h.hexdigest(), el


def batched(n):
    drv = [api.NativeDriver(s.W, s.H, **cfg) for _ in range(n)]
    nb = api.NativeBatch(drv)
    h = hashlib.sha256()
    for t in range(a.frames):
        # member 0 codes the video; the others code the same frames in another order (their own GOP phases)
        nb.encode_frame_device([ptr[(t * 7) % ND]] + [ptr[(t * 5 + 3 * i) % ND] for i in range(1, n)])
        nb.get_frames_begin()
        frames = [d.get_frame_end() for d in drv]
        h.update(frames[0])
    for p in drv[0].hip.download_last():
        h.update(p.tobytes())
    nb.close()
    [d.close() for d in drv]
    return h.hexdigest()


ref, t_plain = single(0)
ov, t_ov = single(1)
b4 = batched(4)
print(f"{s.W}x{s.H} {a.frames} frames: plain {t_plain / a.frames * 1e3:.3f} ms/frame, overlapped {t_ov / a.frames * 1e3:.3f} ms/frame (frames out)")
assert ref == ov, "overlapped filter: different stream"
assert ref == b4, "as a member of a batch: different stream"
print("all identical:", ref[:16])
