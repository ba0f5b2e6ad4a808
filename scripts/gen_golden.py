#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own kernels (oracle/_ref/libvp8ref.so).

Runs only where /root/reference exists (this container): oracle/build_ref.sh compiles the reference's
src/GPU_kernels.cl and src/CPU_kernels.cl for x86 and this script drives them, stage by stage, in the
enqueue order of src/inter_part.h:96-384 + src/loop_filter.h (tests/pipeline.py) on seeded synthetic
frames.  What is committed is data only -- inputs (as generator parameters) and every stage output.
The fixtures pin the CPU restatement (tests/test_oracle.py, no GPU) and, through it and directly, the
HIP path (tests/test_gpu_parity.py::test_golden_vectors).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_lib import ref_stages  # noqa: E402
from pipeline import default_segments, run_inter_frame  # noqa: E402
from vp8oclenc_amd.synth import SynthSequence, noise_frames  # noqa: E402

CASES = [
    # name, W, H, seed, kind, ssim_target, use_golden, use_altref, synth kwargs
    ("g64x48_last_only", 64, 48, 1, "synth", -1.0, 0, 0, {}),
    ("g96x64_3refs", 96, 64, 2, "synth", -1.0, 1, 1, {}),
    ("g128x64_ssim93_saturated", 128, 64, 7, "synth", 0.93, 1, 1, dict(noise=20, saturate=True)),
    ("g64x64_noise_wrap", 64, 64, 9, "noise", -1.0, 1, 1, {}),
    ("g176x144_ssim97", 176, 144, 3, "synth", 0.97, 1, 0, dict(noise=10)),
    ("g128x96_mixed_partitions", 128, 96, 4, "synth", 0.9, 1, 1, dict(noise=1, n_rects=3)),
]


def case_frames(W, H, seed, kind, kw):
    if kind == "noise":
        nf = noise_frames(W, H, seed)
        return [nf[0], nf[1], nf[0], nf[1]]
    s = SynthSequence(W, H, seed=seed, **kw)
    return [s.frame(t) for t in range(4)]


def main():
    st = ref_stages()
    if st is None:
        raise SystemExit("oracle/_ref/libvp8ref.so is not available: run `make -C oracle ref` where /root/reference exists")
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, W, H, seed, kind, target, ug, ua, kw in CASES:
        f = case_frames(W, H, seed, kind, kw)
        cur, refs = f[3], [f[2], f[0], f[1]]
        sd = default_segments()
        r = run_inter_frame(st, cur, refs, sd, ug, ua, target)
        flat = {}
        for k, v in r.items():
            if isinstance(v, list):
                for i, a in enumerate(v):
                    flat[f"{k}_{i}"] = a
            elif k.startswith(("pred_", "resid_")):
                continue  # internal planes of the reference; implied by coefficients + reconstruction
            else:
                flat[k] = v
        # the inputs themselves, so that the fixtures do not depend on the generator staying bit-stable
        for nm, fr in (("cur", cur), ("ref0", refs[0]), ("ref1", refs[1]), ("ref2", refs[2])):
            for pn, pl in zip("YUV", fr):
                flat[f"in_{nm}_{pn}"] = pl
        meta = dict(W=W, H=H, seed=seed, kind=kind, ssim_target=target, use_golden=ug, use_altref=ua,
                    synth_kwargs=repr(kw))
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), segments=sd, meta=np.array(repr(meta)), **flat)
        print(name, {k: v.shape for k, v in list(flat.items())[:3]}, "...", len(flat), "arrays")


if __name__ == "__main__":
    main()
