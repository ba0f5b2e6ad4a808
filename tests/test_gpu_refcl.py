"""The reference's OWN kernels executed on this MI355X (AMD's OpenCL compiler + device library + OpenCL runtime:
oracle/build_ref.sh -> oracle/_ref/*.co, oracle/ref_cl_driver.c) against the CPU restatement and against the HIP
path, live, on fresh seeded inputs.  Needs the prebuilt oracle/_ref (it travels to the GPU box; built only where
/root/reference exists) -- skipped otherwise; the committed fixtures tests/golden/gfx950/ (made by the same
machinery, scripts/gen_golden_gfx950.py) cover that case in test_oracle.py / test_gpu_parity.py.

Integers bit-exact.  MB_SSIM is float in the reference and its mad() is a fused multiply-add on this GPU (1-2 ulp
from the unfused x86 / restatement value): tolerance 1e-4 as north_star states, observed < 3e-7.
"""
import numpy as np
import pytest

from entropy_cases import run_stage, synthetic
from large_cases import FIXTURES, LARGE_CASES, diff_against_fixture, large_case_frames, load_fixture
from oracle_lib import Oracle, ref_cl_stages
from pipeline import default_segments, run_inter_frame
from vp8oclenc_amd.synth import SynthSequence, noise_frames

pytestmark = pytest.mark.gpu
SSIM_TOL = 1e-4


@pytest.fixture(scope="module")
def cl_stages():
    st = ref_cl_stages()
    if st is None:
        pytest.skip("oracle/_ref/libvp8ref_cl.so + code objects not present, or no OpenCL GPU device")
    return st


CASES = [
    (64, 48, 21, "synth", -1.0, 1, 1, {}),
    (160, 96, 22, "synth", 0.93, 1, 1, dict(noise=12)),
    (128, 64, 23, "noise", -1.0, 1, 1, {}),                      # ushort cost wrap-around
    (256, 128, 24, "synth", 0.97, 1, 0, dict(noise=20, saturate=True)),
    (48, 32, 25, "synth", -1.0, 0, 0, {}),                       # coarsest levels have no 8x8 block
]


def _frames(W, H, seed, kind, kw):
    if kind == "noise":
        nf = noise_frames(W, H, seed)
        return [nf[0], nf[1], nf[0], nf[1]]
    s = SynthSequence(W, H, seed=seed, **kw)
    return [s.frame(t) for t in range(4)]


def _diff(a, b):
    bad = []
    for k in a:
        va, vb = a[k], b[k]
        if isinstance(va, list):
            bad += [(k, i) for i, (x, y) in enumerate(zip(va, vb)) if not np.array_equal(x, y)]
        elif va.dtype == np.float32:
            d = float(np.abs(va.astype(np.float64) - vb.astype(np.float64)).max())
            if not d <= SSIM_TOL:
                bad.append((k, d))
        elif not np.array_equal(va, vb):
            bad.append((k, int((va != vb).sum())))
    return bad


def test_device_is_gfx950_without_image_hardware(cl_stages):
    assert "gfx950" in cl_stages.device_name
    # the reason the two image-sampling kernels come from the build with oracle/ref_image_as_buffer.cl in front
    assert cl_stages.image_support == 0


@pytest.mark.parametrize("W,H,seed,kind,target,ug,ua,kw", CASES)
def test_restatement_matches_reference_kernels_on_this_gpu(W, H, seed, kind, target, ug, ua, kw, cl_stages, oracle_stages):
    f = _frames(W, H, seed, kind, kw)
    cur, refs = f[3], [f[2], f[0], f[1]]
    sd = default_segments()
    a = run_inter_frame(oracle_stages, cur, refs, sd, ug, ua, target)
    b = run_inter_frame(cl_stages, cur, refs, sd, ug, ua, target)
    bad = _diff(a, b)
    assert not bad, f"restatement differs from the reference kernels run on {cl_stages.device_name}: {bad}"


def _hip_against(r, cur, refs, sd, ug, ua, target):
    """what of the same frame through the C ABI of libvp8hip.so differs from the stage outputs `r`"""
    from test_gpu_parity import _one_frame
    h, _ = _one_frame(cur[0].shape[1], cur[0].shape[0], [refs[0], refs[1], refs[2], cur], sd, (ug, ua), target)
    bad = []
    for k in ("MB_parts", "MB_reference_frame", "MB_vectors", "MB_segment_id", "prefilter_Y", "prefilter_U", "prefilter_V",
              "MB_non_zero_coeffs", "mb_mask", "recon_Y", "recon_U", "recon_V"):
        if not np.array_equal(h[k], r[k]):
            bad.append((k, int((h[k] != r[k]).sum())))
    d = float(np.abs(h["MB_SSIM"].astype(np.float64) - r["MB_SSIM"].astype(np.float64)).max())
    if not d <= SSIM_TOL:
        bad.append(("MB_SSIM", d))
    c, g = h["MB_coeffs"].copy(), r["MB_coeffs"].copy()
    c[h["MB_parts"] != 0, 24] = 0       # block 24 exists only for 16x16 macroblocks
    g[r["MB_parts"] != 0, 24] = 0
    if not np.array_equal(c, g):
        bad.append(("MB_coeffs", int((c != g).sum())))
    for ref in range(3):
        if ref == 0 or (ug, ua)[ref - 1]:
            for hk, rk in ((f"net1_r{ref}", f"net1_r{ref}"), (f"bdiff_r{ref}", f"bdiff_r{ref}"), (f"net2_r{ref}", f"net_r{ref}_l0")):
                if not np.array_equal(h[hk], r[rk]):
                    bad.append((hk, int((h[hk] != r[rk]).sum())))
    for l in range(5):
        for hk, rk in ((f"cur_pyr{l}", "cur_pyr"), (f"last_pyr{l}", "ref0_pyr")):
            if not np.array_equal(h[hk], r[rk][l]):
                bad.append((hk, int((h[hk] != r[rk][l]).sum())))
    return bad


@pytest.mark.parametrize("W,H,seed,kind,target,ug,ua,kw", CASES[:4])
def test_hip_path_matches_reference_kernels_on_this_gpu(W, H, seed, kind, target, ug, ua, kw, cl_stages):
    """The same frame through the C ABI of libvp8hip.so and through the reference's kernels, both on this GPU."""
    f = _frames(W, H, seed, kind, kw)
    cur, refs = f[3], [f[2], f[0], f[1]]
    sd = default_segments()
    r = run_inter_frame(cl_stages, cur, refs, sd, ug, ua, target)
    bad = _hip_against(r, cur, refs, sd, ug, ua, target)
    assert not bad, f"HIP differs from the reference kernels run on {cl_stages.device_name}: {bad}"


@pytest.mark.parametrize("case", LARGE_CASES, ids=[c[0] for c in LARGE_CASES])
def test_reference_kernels_restatement_and_hip_agree_at_the_metrics_geometry(case, cl_stages, oracle_stages):
    """BASELINE configs[1]-[3]: 1280x720, 1920x1080 (wrk 1088: padded rows, half a block row at level 4) and 3840x2160, three references,
    one per size with the four-pass SSIM ladder -- the reference's own kernels on this GPU, the CPU restatement and libvp8hip.so on the
    same frame; the committed CRC fixture of the case (tests/golden/gfx950/L*.npz) must be what the reference's kernels give here."""
    name, W, H, seed, target, ug, ua, kw = case
    cur, refs = large_case_frames(W, H, seed, kw)
    sd = default_segments()
    r = run_inter_frame(cl_stages, cur, refs, sd, ug, ua, target)
    bad = _diff(run_inter_frame(oracle_stages, cur, refs, sd, ug, ua, target), r)
    assert not bad, f"{name}: restatement differs from the reference kernels run on {cl_stages.device_name}: {bad}"
    bad = _hip_against(r, cur, refs, sd, ug, ua, target)
    assert not bad, f"{name}: HIP differs from the reference kernels run on {cl_stages.device_name}: {bad}"
    fx = [p for p in FIXTURES if p.endswith(name + ".npz")]
    assert fx, f"{name}: no committed fixture (scripts/gen_golden_gfx950.py --only-large)"
    meta, seg, ssim = load_fixture(fx[0])
    assert np.array_equal(seg, sd)
    bad = diff_against_fixture(r, meta, ssim)
    assert not bad, f"{name}: the reference kernels give other outputs here than the committed fixture holds: {bad}"


@pytest.mark.parametrize("mbw,mbh,seed,P,kw", [(8, 5, 31, 2, {}), (11, 9, 32, 8, dict(density=0.5, big=0.1))])
def test_entropy_restatement_matches_reference_kernels_on_this_gpu(mbw, mbh, seed, P, kw, cl_stages):
    c, p, n = synthetic(mbw, mbh, seed, **kw)
    a = run_stage(Oracle.stages(), c, p, n, mbw, mbh, P)
    b = run_stage(cl_stages, c, p, n, mbw, mbh, P)
    for k in ("counts", "denom", "probs", "sizes"):
        assert np.array_equal(a[k], b[k]), k
    m = np.repeat(n != 0, 25)
    assert np.array_equal(a["third_context"][m], b["third_context"][m])
    for i in range(P):
        assert np.array_equal(a["partitions"][i], b["partitions"][i]), f"partition {i}"
