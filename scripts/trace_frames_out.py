"""16 GOP chunks with finished frames delivered to the host (bench.py's with_bitstream leg alone): the workload of the
frames-out kernel trace"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vp8oclenc_amd import api
G = int(os.environ.get("VP8_CHUNKS", "48"))
leg = bench.Leg(torch, api, 1920, 1080, G, "all", -1.0, 8, 0, 1, batch=int(os.environ.get("VP8_BATCH", "6")))
r = bench.bitstream_leg(torch, leg, int(os.environ.get("VP8_FRAMES", "40")))
print(r["value"] / 1e6, "M MB/s", r["fps"], "frames/s", r["avg_frame_bytes"], "B/frame")
leg.close()
