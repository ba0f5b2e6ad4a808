"""The metric's geometries as reference-kernel fixtures: BASELINE configs[1]-[3] (1280x720 LAST only, 1920x1080 with three references --
the only geometry with padded rows, wrk 1088, and a half block row at pyramid level 4: src/inter_part.h:110, src/init.h:383-386 -- and
3840x2160), one of them per size with -SSIM-target 0.93 (the four-pass segment ladder, src/inter_part.h:268-365).

The arrays are too large to commit, so tests/golden/gfx950/L*.npz hold the CRC-32 of every stage output of the REFERENCE'S OWN kernels
run on the MI355X (scripts/gen_golden_gfx950.py) plus MB_SSIM itself (the one float output: compared at 1e-4).  The inputs are
regenerated from the seed (their CRC-32 is in the fixture too: a different numpy that renders another picture is reported as such, not as
a parity failure).  Test infrastructure.
"""
from __future__ import annotations

import glob
import json
import os
import zlib

import numpy as np

from vp8oclenc_amd.synth import SynthSequence

# name, W, H, seed, ssim_target, use_golden, use_altref, synth kwargs
LARGE_CASES = [
    ("L1280x720_last_only", 1280, 720, 26, -1.0, 0, 0, {}),
    ("L1920x1080_3refs", 1920, 1080, 27, -1.0, 1, 1, {}),
    ("L1920x1080_3refs_ssim93", 1920, 1080, 28, 0.93, 1, 1, dict(noise=12)),
    ("L3840x2160_3refs", 3840, 2160, 29, -1.0, 1, 1, {}),
    ("L3840x2160_3refs_ssim93", 3840, 2160, 30, 0.93, 1, 1, dict(noise=12)),
]
LARGE_BY_NAME = {c[0]: c for c in LARGE_CASES}
FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gfx950", "L*.npz")))
SSIM_TOL = 1e-4


def large_case_frames(W, H, seed, kw):
    """(current, [LAST, GOLDEN, ALTREF]) at the wrk size; LAST is the frame closest in time"""
    s = SynthSequence(W, H, seed=seed, **kw)
    f = [s.frame(t) for t in range(4)]
    return f[3], [f[2], f[0], f[1]]


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def crc_of_outputs(out: dict) -> dict:
    """CRC-32 of every integer output (lists -- pyramids -- per level); predictor / residual planes and floats are left out"""
    d = {}
    for k, v in out.items():
        if k.startswith(("pred_", "resid_")):
            continue
        if isinstance(v, list):
            for i, a in enumerate(v):
                d[f"{k}_{i}"] = crc(a)
        elif v.dtype != np.float32:
            d[k] = crc(v)
    if "MB_coeffs" in out and "MB_parts" in out:
        # block 24 (the second-order block) exists only for 16x16 macroblocks; what the slot holds otherwise is left over from the passes
        # (src/GPU_kernels.cl:1545-1608) and the HIP path does not write it
        c = out["MB_coeffs"].copy()
        c[out["MB_parts"] != 0, 24] = 0
        d["MB_coeffs_block24_where_it_exists"] = crc(c)
    return d


def load_fixture(path):
    z = np.load(path)
    return json.loads(str(z["meta"])), z["segments"], z["MB_SSIM"]


def diff_against_fixture(out: dict, meta: dict, ssim: np.ndarray, rename=None, skip=()) -> list:
    """what in `out` (a stage-output dict, keys as tests/pipeline.py names them or mapped by `rename`) differs from the fixture"""
    mine = crc_of_outputs(out)
    bad, seen = [], 0
    for k, c in mine.items():
        fk = (rename or {}).get(k, k)
        if fk in skip or fk not in meta["crc32"]:
            continue
        seen += 1
        if meta["crc32"][fk] != c:
            bad.append(fk)
    if "MB_SSIM" in out:
        d = float(np.abs(out["MB_SSIM"].astype(np.float64) - ssim.astype(np.float64)).max())
        if not d <= SSIM_TOL:
            bad.append(("MB_SSIM", d))
    assert seen >= 10, f"only {seen} outputs were compared"
    return bad
