#!/usr/bin/env python3
"""Time the device intra path (key frame, check_SSIM fallback) with hipEvents; GPU box only.
    python scripts/intra_bench.py [W H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1088)
seq = SynthSequence(W, H, seed=1)
last, _ = api.quantizer_ladders(0, 48)
f0, f1 = seq.frame(0), seq.frame(1)
enc = api.Vp8Hip(W, H, 0.93)
enc.upload_current(*f0)
red, sh = api.loopfilter_strength(f0[0])
enc.set_segments(api.prepare_segments_data(True, last, 0, red, sh))
enc.profile_enable(["intra"])
for _ in range(3):
    enc.intra_transform()
enc.synchronize()
enc.profile_read()
N = 10
t = time.perf_counter()
for _ in range(N):
    enc.intra_transform()
enc.synchronize()
wall = (time.perf_counter() - t) / N
ms, n = enc.profile_read()["intra"]
mbs = (W // 16) * (H // 16)
print(f"key frame {W}x{H}: {ms / n:.3f} ms per frame (hipEvents), wall {wall * 1e3:.3f} ms, {mbs / (ms / n) / 1e3:.2f} M MB/s, "
      f"{(ms / n) * 1e3 / (W // 16 + 2 * (H // 16 - 1)):.2f} us per wavefront step")
# fallback: an inter frame with a high target so that a share of the macroblocks is tried
enc.prepare_filter_mask(False); enc.loop_filter()
for target_frac in (0.02, 0.10, 0.30):
    enc.upload_current(*f1)
    red, sh = api.loopfilter_strength(f1[0])
    enc.set_segments(api.prepare_segments_data(False, last, 0, red, sh))
    enc.inter_transform(1, 1, 0, 0)
    r = enc.download_results(recon=False)
    thr = float(np.quantile(r["MB_SSIM"], target_frac))
    e2 = api.Vp8Hip(W, H, thr)
    e2.upload_current(*f1)
    e2.set_segments(api.prepare_segments_data(False, last, 0, red, sh))
    rr = enc.download_results(recon=True)
    e2.upload_recon(rr["prefilter_Y"], rr["prefilter_U"], rr["prefilter_V"])
    e2.upload_mb_data(rr["MB_coeffs"], rr["MB_parts"], rr["MB_segment_id"])
    e2._debug_set_ssim(rr["MB_SSIM"])
    e2.profile_enable(["intra"])
    t = time.perf_counter()
    repl, new, mn = e2.check_ssim()
    wall = time.perf_counter() - t
    ms, n = e2.profile_read()["intra"]
    print(f"check_ssim target {thr:.4f} ({int((rr['MB_SSIM'] < thr).sum())} of {mbs} below): kernel {ms / n:.3f} ms, call {wall * 1e3:.3f} ms, replaced {repl}")
    e2.close()
    enc.prepare_filter_mask(False); enc.loop_filter()
enc.close()

# nothing below the target: what the call costs when it has nothing to do
e3 = api.Vp8Hip(W, H, -1.0)
e3.upload_current(*f1)
e3.set_segments(api.prepare_segments_data(False, last, 0, 4, 0))
e3.upload_last(*f0)
e3.inter_transform(1, 1, 0, 0)
e3.synchronize()
ts = []
for _ in range(20):
    t = time.perf_counter(); e3.check_ssim(); ts.append(time.perf_counter() - t)
print(f"check_ssim with no macroblock below the target: {1e3 * sorted(ts)[len(ts) // 2]:.3f} ms per call")
e3.close()
