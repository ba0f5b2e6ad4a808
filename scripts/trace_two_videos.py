"""two videos side by side, each coded frame after frame by a host thread of its own (config3_literal's shape without its key frames):
the kernel-trace workload for what two chains cost each other"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from vp8oclenc_amd import api
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
leg = bench.Leg(api, 1920, 1080, 0, "all", -1.0, 8, 0, 1)
drv = [api.NativeDriver(leg.W, leg.H, gop_size=1 << 30, altref_range=bench.ALTREF_RANGE, qi_min=0, qi_max=48, ssim_target=-1.0, device_params=1,
                        check_ssim=1, overlap_filter=1, src_width=1920, src_height=1080) for _ in range(chunks)]
def work(k, n):
    d = drv[k]
    for t in range(n):
        d.encode_frame_device(*leg.ptrs[(3 * k + t) % leg.nd])
    d.resolve(); d.hip.synchronize()
for k in range(chunks): work(k, 12)
api.device_synchronize(); t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k, N)) for k in range(chunks)]
[t.start() for t in th]; [t.join() for t in th]
api.device_synchronize(); el = time.perf_counter() - t0
print(f"{chunks} videos side by side: {el / N * 1e3:.4f} ms per frame-step, {chunks * N / el:.0f} frames/s", flush=True)
for d in drv: d.close()
leg.close()
