"""Host cost of enqueuing one frame: vp8drv_encode_frame_device (inter path) and vp8drv_get_frame_begin (entropy stage),
GPU idle, one thread."""
import sys, time
sys.path.insert(0, ".")
import torch
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence
seq = SynthSequence(1920, 1080, seed=1); W, H = seq.W, seq.H
dev = [tuple(torch.from_numpy(p).cuda() for p in seq.frame(t)) for t in range(4)]
d = api.NativeDriver(W, H, gop_size=1 << 30, num_partitions=8)
for t in range(3):
    y, u, v = dev[t]; d.encode_frame_device(y.data_ptr(), u.data_ptr(), v.data_ptr()); d.get_frame()
T = {"encode_frame_device": 0.0, "get_frame_begin": 0.0, "get_frame_end": 0.0}
N = 50
for t in range(N):
    y, u, v = dev[t % 4]
    t0 = time.perf_counter(); d.encode_frame_device(y.data_ptr(), u.data_ptr(), v.data_ptr()); t1 = time.perf_counter()
    d.hip.synchronize()
    t2 = time.perf_counter(); d.get_frame_begin(); t3 = time.perf_counter()
    d.hip.synchronize()
    t4 = time.perf_counter(); d.get_frame_end(); t5 = time.perf_counter()
    T["encode_frame_device"] += t1 - t0; T["get_frame_begin"] += t3 - t2; T["get_frame_end"] += t5 - t4
for k, v in T.items(): print(f"{k:22s} {v / N * 1e6:8.1f} us host time per frame")
