"""one video, frame after frame, WITH frames out (the loop of bench.py's config5_literal / scripts/native/y4m_to_ivf.cpp: frame t + 1 under way
before frame t's bytes are taken): the workload of the frames-out single-stream kernel trace"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vp8oclenc_amd import api
leg = bench.Leg(api, 1920, 1080, 0, "all", -1.0, 8, 0, 1)
d = api.NativeDriver(leg.W, leg.H, gop_size=1 << 30, altref_range=bench.ALTREF_RANGE, qi_min=0, qi_max=48, ssim_target=-1.0, device_params=1,
                     check_ssim=1, overlap_filter=1, src_width=1920, src_height=1080)
d.hip.reserve_frame_path_dense()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
pending = None
nbytes = 0
for t in range(20 + N):
    if t == 20:
        api.device_synchronize(); t0 = time.perf_counter()
    d.encode_frame_device(*leg.ptrs[t % leg.nd])
    if pending is not None:
        nbytes += len(d.get_frame_end())
    d.get_frame_begin()
    pending = t
    d.resolve()
nbytes += len(d.get_frame_end())
d.hip.synchronize()
el = time.perf_counter() - t0
print("single stream with frames out: %.4f ms/frame, %d bytes/frame" % (el / N * 1e3, nbytes // (N + 20)))
d.close(); leg.close()
