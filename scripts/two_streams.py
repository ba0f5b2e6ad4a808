"""one video / two videos side by side (threads), frame after frame: ms per frame, with and without check_SSIM and frames out"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence
seq = SynthSequence(1920, 1080, seed=1)
dev = [tuple(api.to_device(p) for p in seq.frame(t)) for t in range(8)]
ptr = [tuple(p.data_ptr() for p in f) for f in dev]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
def run(chunks, check, bits, overlap=1, label=""):
    drv = [api.NativeDriver(seq.W, seq.H, gop_size=1 << 30, check_ssim=check, overlap_filter=overlap, device_params=1) for _ in range(chunks)]
    for d in drv:
        d.encode_frame_device(*ptr[0]); d.get_frame()
    def work(k):
        d = drv[k]
        for t in range(N):
            d.encode_frame_device(*ptr[(3 * k + t) % 8])
            if bits: d.get_frame()
        d.resolve(); d.hip.synchronize()
    api.device_synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(chunks)]
    [t.start() for t in th]; [t.join() for t in th]
    api.device_synchronize(); el = time.perf_counter() - t0
    print(f"{label or ''} chunks {chunks} check {check} frames_out {bits} overlap {overlap}: {el / N * 1e3:.4f} ms per frame-step, {chunks * N / el:.0f} fps total", flush=True)
    for d in drv: d.close()
for chunks in (1, 2):
    for check in (0, 1):
        for bits in (0, 1):
            run(chunks, check, bits)
run(1, 1, 0, overlap=0)
run(1, 0, 0, overlap=0)
