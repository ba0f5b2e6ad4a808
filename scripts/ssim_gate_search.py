"""Is there a macroblock whose SSIM gate (`MB_SSIM > SSIM_target`, src/GPU_kernels.cl:1391; `MB_SSIM < SSIM_target`, src/vp8enc.cpp:244)
falls differently with the reference's mad() FUSED -- what AMD's OpenCL compiler makes of it on this very GPU -- than with it unfused,
which is what the product computes (bit-identical to the restatement and to the x86 build of the reference's kernels)?  The two
values differ by 1-2 ulp; the command line only reaches targets of the form n/100 (init.h:1512).  This runs the reference's kernels
on the MI355X (oracle/_ref, tests/oracle_lib.ref_cl_stages) and the product on the same frames and looks for a macroblock and a
reachable target between its two values.

    python scripts/ssim_gate_search.py [frames per size] -> gpurun_out/ssim_gate_search.json
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import ref_cl_stages
from pipeline import default_segments, run_inter_frame
from test_gpu_parity import _one_frame
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

st = ref_cl_stages()
assert st is not None, "oracle/_ref/libvp8ref_cl.so + code objects not present, or no OpenCL GPU device"
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
targets = (np.arange(1, 100, dtype=np.float32) / np.float32(100.0)).astype(np.float32)      # init.h:1512: ((float)buf)/100.0f
total = differ = 0
flips = []
maxulp = 0
for (W, H, seed, noise, qi) in [(640, 352, 3, 0, (0, 48)), (640, 352, 4, 12, (0, 48)), (1280, 720, 5, 0, (20, 80)), (1280, 720, 6, 20, (40, 110)), (1920, 1088, 7, 6, (0, 48))]:
    s = SynthSequence(W, H, seed=seed, noise=noise) if noise else SynthSequence(W, H, seed=seed)
    lastqi, _ = api.quantizer_ladders(*qi)
    for t in range(n_frames):
        cur, refs = s.frame(t + 3), [s.frame(t + 2), s.frame(t), s.frame(t + 1)]
        red, sh = api.loopfilter_strength(cur[0])
        sd = api.prepare_segments_data(False, lastqi, min(qi), red, sh)
        r = run_inter_frame(st, cur, refs, sd, 1, 1, -1.0)                      # the reference's kernels on this GPU: fused mad
        h, _ = _one_frame(s.W, s.H, [refs[0], refs[1], refs[2], cur], sd, (1, 1), -1.0)   # the product (and the restatement): unfused
        a, b = r["MB_SSIM"].astype(np.float32), h["MB_SSIM"].astype(np.float32)
        assert all(np.array_equal(r[k], h[k]) for k in ("MB_vectors", "MB_parts", "MB_reference_frame", "prefilter_Y")), "integer outputs must agree"
        total += a.size
        d = a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)
        differ += int((d != 0).sum())
        maxulp = max(maxulp, int(np.abs(d).max()))
        ga, gb = a[:, None] > targets[None, :], b[:, None] > targets[None, :]
        for mb, k in zip(*np.nonzero(ga != gb)):
            flips.append(dict(size=[s.W, s.H], seed=seed, frame=t, mb=int(mb), target=float(targets[k]), fused=float(a[mb]), unfused=float(b[mb]),
                              fused_bits=hex(int(a[mb:mb + 1].view(np.uint32)[0])), unfused_bits=hex(int(b[mb:mb + 1].view(np.uint32)[0]))))
    print(f"{W}x{H} seed {seed}: {total} macroblocks so far, {differ} with different bit patterns, {len(flips)} gate flips", flush=True)
out = dict(device=st.device_name, macroblocks=total, values_that_differ=differ, max_ulp=maxulp, reachable_targets="n/100, n = 1..99 (init.h:1512)",
           gate_flips=flips[:20], n_gate_flips=len(flips),
           product_follows="the unfused value: bit-identical to the restatement and to the x86 build of the reference's kernels")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ssim_gate_search.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "gate_flips"}))
for f in flips[:5]:
    print(f)
