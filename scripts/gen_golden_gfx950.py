#!/usr/bin/env python3
"""Golden vectors from the REFERENCE's own kernels executed ON THE MI355X (tests/golden/gfx950/*.npz).

oracle/build_ref.sh (run in the build container, where /root/reference exists) compiles the reference's
src/GPU_kernels.cl and src/CPU_kernels.cl with AMD's OpenCL C compiler for gfx950 -- vendor built-in library,
images and samplers; no stand-in of ours -- into oracle/_ref/*.co; oracle/ref_cl_driver.c runs them through the
vendor's OpenCL runtime with the reference host's NDRanges.  This script drives that on the GPU box, stage by
stage in the enqueue order of src/inter_part.h:96-384 + src/loop_filter.h (tests/pipeline.py), on seeded
synthetic frames, and writes inputs + every stage output.  It also runs the CPU restatement
(oracle/vp8_oracle.c) and, where present, the x86 build of the same kernels on the same inputs and records
what differs (report.json) -- that is the pin of the restatement and of oracle/ref_shim.cl.

    gpurun -- 'python3 scripts/gen_golden_gfx950.py --out gpurun_out/golden_gfx950'
    cp gpurun_out/golden_gfx950/*.npz tests/golden/gfx950/ ; cp .../report*.json tests/golden/gfx950/
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from entropy_cases import from_inter_path, run_stage, synthetic  # noqa: E402
from oracle_lib import Oracle, ref_cl_stages, ref_stages  # noqa: E402
from pipeline import default_segments, run_inter_frame  # noqa: E402
from vp8oclenc_amd.synth import SynthSequence, noise_frames  # noqa: E402

CASES = [
    # name, W, H, seed, kind, ssim_target, use_golden, use_altref, synth kwargs
    ("c64x48_last_only", 64, 48, 1, "synth", -1.0, 0, 0, {}),
    ("c96x64_3refs", 96, 64, 2, "synth", -1.0, 1, 1, {}),
    ("c128x64_ssim93_saturated", 128, 64, 7, "synth", 0.93, 1, 1, dict(noise=20, saturate=True)),
    ("c64x64_noise_wrap", 64, 64, 9, "noise", -1.0, 1, 1, {}),
    ("c176x144_ssim97", 176, 144, 3, "synth", 0.97, 1, 0, dict(noise=10)),
    ("c128x96_mixed_partitions", 128, 96, 4, "synth", 0.9, 1, 1, dict(noise=1, n_rects=3)),
    ("c48x32_empty_levels", 48, 32, 12, "synth", -1.0, 1, 1, {}),
    ("c352x288_cif_3refs_ssim95", 352, 288, 3, "synth", 0.95, 1, 1, {}),   # BASELINE configs[0] geometry
]


# The metric's geometries (BASELINE configs[1]-[3]): too large to commit as arrays, so the fixture is the CRC-32 of every stage output of
# the reference's kernels (+ MB_SSIM itself, the one float output, for the 1e-4 comparison).  1920x1080 is the only configuration with padded
# rows (wrk 1088) and a half block row at pyramid level 4 (src/inter_part.h:110, src/init.h:383-386).  tests/large_cases.py regenerates the inputs.
from large_cases import LARGE_CASES, crc_of_outputs, large_case_frames  # noqa: E402


def case_frames(W, H, seed, kind, kw):
    if kind == "noise":
        nf = noise_frames(W, H, seed)
        return [nf[0], nf[1], nf[0], nf[1]]
    s = SynthSequence(W, H, seed=seed, **kw)
    return [s.frame(t) for t in range(4)]


def compare(a: dict, b: dict) -> dict:
    """what differs between two stage-output dicts: {key: count or max abs float difference}"""
    bad = {}
    for k in a:
        va, vb = a[k], b[k]
        if isinstance(va, list):
            for i, (x, y) in enumerate(zip(va, vb)):
                if not np.array_equal(x, y):
                    bad[f"{k}_{i}"] = int((x != y).sum())
        elif va.dtype == np.float32:
            if not np.array_equal(va.view(np.uint32), vb.view(np.uint32)):
                bad[k] = {"max_abs_diff": float(np.abs(va.astype(np.float64) - vb.astype(np.float64)).max()),
                          "values_with_other_bits": int((va.view(np.uint32) != vb.view(np.uint32)).sum())}
        elif not np.array_equal(va, vb):
            bad[k] = int((va != vb).sum())
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "golden_gfx950"))
    ap.add_argument("--only-large", action="store_true", help="only the L* fixtures of the metric's geometries (report_large.json)")
    args = ap.parse_args()
    cl = ref_cl_stages()
    if cl is None:
        raise SystemExit("no OpenCL GPU device, or oracle/_ref/{libvp8ref_cl.so,ref_*_gfx950.co} missing (make -C oracle ref)")
    ora = Oracle.stages()
    x86 = ref_stages()
    os.makedirs(args.out, exist_ok=True)
    report = {"device": cl.device_name, "CL_DEVICE_IMAGE_SUPPORT": cl.image_support, "opencl_runtime": "AMD (libamdocl64), clCreateProgramWithBinary on the code objects of oracle/build_ref.sh",
              "cases": {}, "entropy": {}}
    for name, W, H, seed, kind, target, ug, ua, kw in ([] if args.only_large else CASES):
        t0 = time.time()
        f = case_frames(W, H, seed, kind, kw)
        cur, refs = f[3], [f[2], f[0], f[1]]
        sd = default_segments()
        r = run_inter_frame(cl, cur, refs, sd, ug, ua, target)
        o = run_inter_frame(ora, cur, refs, sd, ug, ua, target)
        rep = {"restatement_vs_gfx950": compare(o, r), "seconds": None}
        if x86 is not None:
            rep["x86_shim_build_vs_gfx950"] = compare(run_inter_frame(x86, cur, refs, sd, ug, ua, target), r)
        flat = {}
        for k, v in r.items():
            if isinstance(v, list):
                for i, a in enumerate(v):
                    flat[f"{k}_{i}"] = a
            elif k.startswith(("pred_", "resid_")):
                continue
            else:
                flat[k] = v
        for nm, fr in (("cur", cur), ("ref0", refs[0]), ("ref1", refs[1]), ("ref2", refs[2])):
            for pn, pl in zip("YUV", fr):
                flat[f"in_{nm}_{pn}"] = pl
        meta = dict(W=W, H=H, seed=seed, kind=kind, ssim_target=target, use_golden=ug, use_altref=ua, synth_kwargs=repr(kw),
                    device=cl.device_name)
        np.savez_compressed(os.path.join(args.out, name + ".npz"), segments=sd, meta=np.array(json.dumps(meta)), **flat)
        rep["seconds"] = round(time.time() - t0, 1)
        report["cases"][name] = rep
        print(name, rep, flush=True)

    large = {"device": cl.device_name, "CL_DEVICE_IMAGE_SUPPORT": cl.image_support, "cases": {}}
    for name, W, H, seed, target, ug, ua, kw in LARGE_CASES:
        t0 = time.time()
        cur, refs = large_case_frames(W, H, seed, kw)
        sd = default_segments()
        r = run_inter_frame(cl, cur, refs, sd, ug, ua, target)
        t_cl = time.time() - t0
        o = run_inter_frame(ora, cur, refs, sd, ug, ua, target)
        rep = {"restatement_vs_gfx950": compare(o, r), "seconds_reference_kernels": round(t_cl, 1)}
        if x86 is not None and W * H <= 1920 * 1088:
            rep["x86_shim_build_vs_gfx950"] = compare(run_inter_frame(x86, cur, refs, sd, ug, ua, target), r)
        crcs = crc_of_outputs(r)
        ins = {f"in_{nm}_{pn}": pl for nm, fr in (("cur", cur), ("ref0", refs[0]), ("ref1", refs[1]), ("ref2", refs[2])) for pn, pl in zip("YUV", fr)}
        meta = dict(W=W, H=H, wrk_W=int(cur[0].shape[1]), wrk_H=int(cur[0].shape[0]), seed=seed, ssim_target=target, use_golden=ug, use_altref=ua,
                    synth_kwargs=repr(kw), device=cl.device_name, crc32=crcs, crc32_inputs=crc_of_outputs(ins),
                    segments_histogram=np.bincount(r["MB_segment_id"], minlength=4).tolist(),
                    reference_histogram=np.bincount(r["MB_reference_frame"], minlength=4).tolist())
        np.savez_compressed(os.path.join(args.out, name + ".npz"), segments=sd, meta=np.array(json.dumps(meta)), MB_SSIM=r["MB_SSIM"])
        rep["seconds"] = round(time.time() - t0, 1)
        large["cases"][name] = rep
        print(name, rep, flush=True)
    with open(os.path.join(args.out, "report_large.json"), "w") as fjs:
        json.dump(large, fjs, indent=1)
    if args.only_large:
        return

    # coefficient entropy stage (src/CPU_kernels.cl:347-778) on the same device
    ent = [("e_synthetic_6x4_p1", synthetic(6, 4, 11), 6, 4, 1),
           ("e_synthetic_9x7_dense_p4", synthetic(9, 7, 12, density=0.5, big=0.1), 9, 7, 4),
           ("e_synthetic_10x9_skips_p8", synthetic(10, 9, 13, skip=0.5), 10, 9, 8),
           ("e_interframe_176x144_p2", from_inter_path(176, 144, 5), 11, 9, 2)]
    for name, (c, p, n), mbw, mbh, P in ent:
        r = run_stage(cl, c, p, n, mbw, mbh, P)
        o = run_stage(ora, c, p, n, mbw, mbh, P)
        bad = [k for k in ("counts", "denom", "probs", "sizes", "third_context") if not np.array_equal(r[k], o[k])]
        bad += [f"partition_{i}" for i in range(P) if not np.array_equal(r["partitions"][i], o["partitions"][i])]
        d = dict(mbw=mbw, mbh=mbh, P=P, coeffs=c, parts=p, nz=n, counts=r["counts"], denom=r["denom"], probs=r["probs"],
                 sizes=r["sizes"], third_context=r["third_context"])
        for i in range(P):
            d[f"partition_{i}"] = r["partitions"][i]
        np.savez_compressed(os.path.join(args.out, name + ".npz"), **d)
        report["entropy"][name] = {"restatement_vs_gfx950": bad, "sizes": r["sizes"].tolist()}
        print(name, report["entropy"][name], flush=True)
    with open(os.path.join(args.out, "report.json"), "w") as fjs:
        json.dump(report, fjs, indent=1)


if __name__ == "__main__":
    main()
