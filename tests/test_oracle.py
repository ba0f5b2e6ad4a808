"""The CPU restatement (oracle/vp8_oracle.c) against
  (1) the committed golden vectors, which were produced by the reference's own kernels -- tests/golden/gfx950/c*.npz by
      the kernels compiled with AMD's OpenCL compiler and RUN ON THE MI355X with the vendor's built-in library
      (scripts/gen_golden_gfx950.py, oracle/ref_cl_driver.c), tests/golden/*.npz by the same kernels compiled for x86
      (scripts/gen_golden.py) -- runs everywhere, no GPU, no /root/reference;
  (2) the reference's own kernels executed live, stage by stage (only where oracle/_ref was built).
Bit-exact for every integer output; MB_SSIM (float in the reference) within 1e-4.
"""
import glob
import os

import numpy as np
import pytest

from large_cases import FIXTURES as LARGE_FIXTURES, LARGE_BY_NAME, crc_of_outputs, diff_against_fixture, large_case_frames, load_fixture
from oracle_lib import Oracle
from pipeline import default_segments, load_meta, run_inter_frame
from vp8oclenc_amd.synth import SynthSequence, noise_frames

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
GOLDEN_GFX950 = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "gfx950", "c*.npz")))
SSIM_TOL = 1e-4


def load_case(path):
    z = np.load(path)
    meta = load_meta(z)
    cur = tuple(np.ascontiguousarray(z[f"in_cur_{p}"]) for p in "YUV")
    refs = [tuple(np.ascontiguousarray(z[f"in_ref{r}_{p}"]) for p in "YUV") for r in range(3)]
    return z, meta, cur, refs


def diff_against_golden(out: dict, z) -> list:
    bad = []
    for k, v in out.items():
        if isinstance(v, list):
            for i, a in enumerate(v):
                if f"{k}_{i}" in z and not np.array_equal(a, z[f"{k}_{i}"]):
                    bad.append((f"{k}_{i}", int((a != z[f'{k}_{i}']).sum())))
        elif k in z.files:
            if v.dtype == np.float32:
                d = float(np.abs(v.astype(np.float64) - z[k].astype(np.float64)).max())
                if not d <= SSIM_TOL:
                    bad.append((k, d))
            elif not np.array_equal(v, z[k]):
                bad.append((k, int((v != z[k]).sum())))
    return bad


def test_golden_fixtures_present():
    assert len(GOLDEN) >= 5
    assert len(GOLDEN_GFX950) >= 8


@pytest.mark.parametrize("path", GOLDEN_GFX950, ids=[os.path.basename(p)[:-4] for p in GOLDEN_GFX950])
def test_restatement_matches_reference_kernels_run_on_gfx950(path, oracle_stages):
    """The pin: outputs of the reference's kernels executed on an MI355X (vendor compiler, vendor built-ins)."""
    z, meta, cur, refs = load_case(path)
    assert "gfx950" in meta["device"]
    out = run_inter_frame(oracle_stages, cur, refs, z["segments"], meta["use_golden"], meta["use_altref"],
                          meta["ssim_target"])
    bad = diff_against_golden(out, z)
    assert not bad, f"{os.path.basename(path)}: restatement differs from the reference kernels run on gfx950: {bad}"


def test_gfx950_report_says_what_the_fixtures_pin():
    """report.json of the generating run: every integer output identical between the gfx950 run, the x86 build of the same
    kernels (oracle/ref_shim.cl built-ins) and the restatement; MB_SSIM (float, fused mad on the GPU) within 1e-6."""
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "gfx950", "report.json")) as f:
        rep = json.load(f)
    assert "gfx950" in rep["device"] and rep["CL_DEVICE_IMAGE_SUPPORT"] == 0
    assert len(rep["cases"]) >= 8
    for name, c in rep["cases"].items():
        for who in ("restatement_vs_gfx950", "x86_shim_build_vs_gfx950"):
            d = dict(c[who])
            ssim = d.pop("MB_SSIM", None)
            assert not d, (name, who, d)
            assert ssim is None or ssim["max_abs_diff"] < 1e-6, (name, who, ssim)
    for name, e in rep["entropy"].items():
        assert e["restatement_vs_gfx950"] == [], name


@pytest.mark.parametrize("path", LARGE_FIXTURES, ids=[os.path.basename(p)[:-4] for p in LARGE_FIXTURES])
def test_restatement_matches_reference_kernels_run_on_gfx950_at_the_metrics_geometry(path, oracle_stages):
    """BASELINE configs[1]-[3] (1280x720, 1920x1080 -> wrk 1088, 3840x2160; one per size with -SSIM-target 0.93): CRC-32 of every stage
    output of the reference's own kernels run on an MI355X (scripts/gen_golden_gfx950.py --only-large), MB_SSIM at 1e-4."""
    meta, seg, ssim = load_fixture(path)
    assert "gfx950" in meta["device"]
    name, W, H, seed, target, ug, ua, kw = LARGE_BY_NAME[os.path.basename(path)[:-4]]
    assert (W, H, seed, target, ug, ua, repr(kw)) == (meta["W"], meta["H"], meta["seed"], meta["ssim_target"], meta["use_golden"], meta["use_altref"], meta["synth_kwargs"])
    cur, refs = large_case_frames(W, H, seed, kw)
    ins = {f"in_{nm}_{pn}": pl for nm, fr in (("cur", cur), ("ref0", refs[0]), ("ref1", refs[1]), ("ref2", refs[2])) for pn, pl in zip("YUV", fr)}
    assert crc_of_outputs(ins) == meta["crc32_inputs"], "this numpy renders other synthetic frames than the box the fixture was made on"
    out = run_inter_frame(oracle_stages, cur, refs, seg, ug, ua, target)
    bad = diff_against_fixture(out, meta, ssim)
    assert not bad, f"{name}: restatement differs from the reference kernels run on gfx950: {bad}"


def test_large_fixtures_present_and_their_report_is_clean():
    import json
    assert len(LARGE_FIXTURES) == len(LARGE_BY_NAME) >= 5
    with open(os.path.join(os.path.dirname(__file__), "golden", "gfx950", "report_large.json")) as f:
        rep = json.load(f)
    assert "gfx950" in rep["device"] and set(rep["cases"]) == set(LARGE_BY_NAME)
    for name, c in rep["cases"].items():
        for who in ("restatement_vs_gfx950", "x86_shim_build_vs_gfx950"):
            d = dict(c.get(who, {}))
            ssim = d.pop("MB_SSIM", None)
            assert not d, (name, who, d)
            assert ssim is None or ssim["max_abs_diff"] < 1e-6, (name, who, ssim)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_restatement_matches_golden_vectors(path, oracle_stages):
    z, meta, cur, refs = load_case(path)
    out = run_inter_frame(oracle_stages, cur, refs, z["segments"], meta["use_golden"], meta["use_altref"],
                          meta["ssim_target"])
    bad = diff_against_golden(out, z)
    assert not bad, f"{os.path.basename(path)}: restatement differs from the reference kernels' outputs: {bad}"


@pytest.mark.parametrize("path", GOLDEN[:3], ids=[os.path.basename(p)[:-4] for p in GOLDEN[:3]])
def test_frame_driver_matches_golden_vectors(path):
    """vp8o_inter_transform/vp8o_loop_filter (the whole-frame driver the GPU tests compare with) vs golden."""
    z, meta, cur, refs = load_case(path)
    W, H = meta["W"], meta["H"]
    ora = Oracle(W, H, meta["ssim_target"])
    ora.set_segments(z["segments"])
    # make refs[1] GOLDEN and refs[2] ALTREF through the reference's rotation rules (inter_part.h:35-50)
    ora.upload_last(*refs[1]); ora.upload_current(*cur); ora.inter_transform(1, 0, 0, 0); ora.loop_filter()
    ora.upload_last(*refs[2]); ora.upload_current(*cur); ora.inter_transform(0, 1, 0, 0); ora.loop_filter()
    ora.upload_last(*refs[0]); ora.upload_current(*cur)
    ora.inter_transform(0, 0, meta["use_golden"], meta["use_altref"])
    res = ora.download_results()
    ora.loop_filter()
    res.update(ora.filter_outputs())
    # block 24 of non-16x16 macroblocks is never written by the reference (stale by design): compare the rest
    keys = ["MB_parts", "MB_reference_frame", "MB_vectors", "MB_segment_id", "MB_SSIM", "prefilter_Y", "prefilter_U",
            "prefilter_V", "MB_non_zero_coeffs", "mb_mask", "recon_Y", "recon_U", "recon_V"]
    bad = diff_against_golden({k: res[k] for k in keys}, z)
    c, g = res["MB_coeffs"].copy(), z["MB_coeffs"].copy()
    c[res["MB_parts"] != 0, 24] = 0
    g[z["MB_parts"] != 0, 24] = 0
    if not np.array_equal(c, g):
        bad.append(("MB_coeffs", int((c != g).sum())))
    assert not bad, bad
    ora.close()


CASES = [
    (64, 48, 1, "synth", -1.0, 1, 1, {}),
    (256, 128, 5, "synth", 0.93, 1, 1, {}),
    (256, 128, 7, "synth", 0.97, 1, 1, dict(noise=20, saturate=True)),
    (128, 64, 9, "noise", -1.0, 1, 1, {}),
    (352, 288, 3, "synth", 0.95, 1, 0, {}),      # BASELINE configs[0] geometry
    (64, 64, 11, "synth", -1.0, 0, 0, {}),
    (48, 32, 12, "synth", -1.0, 1, 1, {}),       # coarsest pyramid levels have zero 8x8 blocks
]


@pytest.mark.parametrize("W,H,seed,kind,target,ug,ua,kw", CASES)
def test_restatement_matches_reference_kernels_live(W, H, seed, kind, target, ug, ua, kw, oracle_stages,
                                                    reference_stages):
    if kind == "noise":
        nf = noise_frames(W, H, seed)
        f = [nf[0], nf[1], nf[0], nf[1]]
    else:
        s = SynthSequence(W, H, seed=seed, **kw)
        f = [s.frame(t) for t in range(4)]
    cur, refs = f[3], [f[2], f[0], f[1]]
    sd = default_segments()
    a = run_inter_frame(oracle_stages, cur, refs, sd, ug, ua, target)
    b = run_inter_frame(reference_stages, cur, refs, sd, ug, ua, target)
    bad = []
    for k in a:
        va, vb = a[k], b[k]
        if isinstance(va, list):
            bad += [(k, i) for i, (x, y) in enumerate(zip(va, vb)) if not np.array_equal(x, y)]
        elif va.dtype == np.float32:
            if float(np.abs(va - vb).max()) > SSIM_TOL:
                bad.append((k, float(np.abs(va - vb).max())))
        elif not np.array_equal(va, vb):
            bad.append((k, int((va != vb).sum())))
    assert not bad, bad


def test_quirks_are_exercised_by_the_fixtures():
    """The saturated fixture must actually reach the clamps the restatement special-cases."""
    path = [p for p in GOLDEN if "saturated" in p][0]
    z, meta, cur, refs = load_case(path)
    assert (z["recon_Y"] == 255).any() and (z["recon_Y"] == 0).any()
    assert len(np.unique(z["MB_segment_id"])) >= 2          # several segment passes (SSIM gate)
    mixed = np.load([p for p in GOLDEN if "mixed" in p][0])
    assert (mixed["MB_parts"] == 0).any() and (mixed["MB_parts"] == 1).any()  # both WHT and non-WHT macroblocks
    assert len(np.unique(mixed["MB_reference_frame"])) == 3                   # all three references chosen


def test_weight_metric_known_answers():
    """weight_opt on hand-checked blocks (GPU_kernels.cl:85-190)."""
    lib = Oracle.lib()

    def weight_py(d):
        d = [int(v) for v in d]
        R = [0] * 16
        for c in range(4):
            r0, r1, r2, r3 = d[c], d[4 + c], d[8 + c], d[12 + c]
            a, dd, cc = (r0 + r3) * 8, (r0 - r3) * 8, (r1 - r2) * 8   # the reference's b1 is overwritten (:100-108)
            R[c], R[8 + c] = a + cc, a - cc
            R[4 + c] = (r2 * 2217 + dd * 5352 + 14500) >> 12           # raw r2, not c1 (:124)
            R[12 + c] = (dd * 2217 - r2 * 5352 + 7500) >> 12
        tot = 0
        for i in range(4):
            e0, e1, e2, e3 = R[4 * i:4 * i + 4]
            a1, d1, b1, c1 = e0 + e3, e0 - e3, e1 + e2, e1 - e2
            o = [(a1 + b1 + 7) >> 4, ((c1 * 2217 + d1 * 5352 + 12000) >> 16) + (d1 != 0), (a1 - b1 + 7) >> 4,
                 (d1 * 2217 - c1 * 5352 + 51000) >> 16]
            tot += (abs(o[0]) // 4 if i == 0 else abs(o[0])) + abs(o[1]) + abs(o[2]) + abs(o[3])
        return tot

    # the rounding constants make even an all-zero difference cost something: (14500>>12)=3 and (7500>>12)=1
    # per column feed the row pass
    assert lib.vp8o_weight(np.zeros(16, np.int32)) == weight_py(np.zeros(16)) > 0
    rng = np.random.default_rng(0)
    for amp in (1, 8, 64, 255):
        for _ in range(50):
            d = rng.integers(-amp, amp + 1, size=16).astype(np.int32)
            assert lib.vp8o_weight(d) == weight_py(d)
    assert lib.vp8o_weight(np.full(16, 255, np.int32)) == weight_py(np.full(16, 255))
    assert lib.vp8o_weight(np.full(16, -255, np.int32)) == weight_py(np.full(16, -255))
