"""one video with frames out through the NATIVE loop (vp8drv_encode_video_device): the kernel-trace workload beside trace_single_frames_out.py"""
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
d.encode_video_device(20, leg.ptrs, start=0)
api.device_synchronize(); t0 = time.perf_counter()
fr, _ = d.encode_video_device(N, leg.ptrs, start=20)
d.hip.synchronize()
el = time.perf_counter() - t0
print("single stream with frames out, native loop: %.4f ms/frame, %d bytes/frame" % (el / N * 1e3, sum(len(f) for f in fr) // N))
d.close(); leg.close()
