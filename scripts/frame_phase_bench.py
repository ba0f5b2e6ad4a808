import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from vp8oclenc_amd import api, bitstream
from vp8oclenc_amd.synth import SynthSequence
seq = SynthSequence(1920, 1080, seed=1); W, H = seq.W, seq.H
dev = [tuple(torch.from_numpy(p).cuda() for p in seq.frame(t)) for t in range(4)]
d = api.NativeDriver(W, H, gop_size=1 << 30, num_partitions=8)
hip = d.hip
for t in range(3):
    y,u,v = dev[t]; d.encode_frame_device(y.data_ptr(), u.data_ptr(), v.data_ptr()); d.get_frame()
T = {}
def tick(name, f):
    t0 = time.perf_counter(); r = f(); T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
N = 20
for t in range(N):
    y,u,v = dev[t % 4]
    tick("encode_frame (enqueue)", lambda: d.encode_frame_device(y.data_ptr(), u.data_ptr(), v.data_ptr()))
    tick("synchronize (GPU frame)", hip.synchronize)
    nz = tick("filter_mask+nz", lambda: hip.prepare_filter_mask())
    probs, denom = tick("count_probs", lambda: hip.count_probs(8))
    p = bitstream.default_probs(probs, denom)
    parts = tick("encode_coefficients", lambda: hip.encode_coefficients(p, 8, W//16*(H//16)*100))
    res = tick("download_results", lambda: hip.download_results(recon=False))
    sd = tick("get_segments", hip.get_segments)
    hdr = tick("encode_header", lambda: bitstream.encode_header(W, H, (0,0,0), sd[0], res["MB_segment_id"], nz, p, denom, api.skip_prob(nz), ref_frame=res["MB_reference_frame"], parts=res["MB_parts"], vectors=res["MB_vectors"]))
    tick("get_frame (whole)", d.get_frame)
for k, v in T.items(): print(f"{k:28s} {v / N * 1e3:7.3f} ms")
print("header bytes", len(hdr[0]), "partition bytes", sum(len(x) for x in parts))
