"""Coefficient entropy stage on the GPU (vp8hip_count_probs, ...) against the CPU oracle
(oracle/vp8_entropy_oracle.c, itself pinned to the reference's kernels in test_entropy_oracle.py) and against
the committed golden vectors of the reference's kernels.  Integer / byte work: bit-exact."""
import glob
import os

import numpy as np
import pytest

from entropy_cases import nz_counts, run_stage, synthetic
from oracle_lib import Oracle
from pipeline import default_segments
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "entropy", "*.npz"))) + \
    sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "gfx950", "e_*.npz")))   # the same kernels run on an MI355X


def device_with(coeffs, parts, mbw, mbh):
    hip = api.Vp8Hip(mbw * 16, mbh * 16)
    hip.set_segments(default_segments())
    hip.upload_mb_data(coeffs, parts, np.zeros(mbw * mbh, np.int32))
    nz = hip.prepare_filter_mask()
    return hip, nz


def check_counts(hip, exp, nz, P, tag, parts=None):
    probs, denom0 = hip.count_probs(P)
    assert np.array_equal(probs, exp["probs"]), f"{tag}: probabilities differ at {np.nonzero(probs != exp['probs'])[0][:8]}"
    assert np.array_equal(denom0, exp["denom"][:1056]), f"{tag}: partition-0 denominators"
    third = hip.debug(api.DBG_THIRD_CONTEXT).reshape(-1)
    m = np.repeat(nz != 0, 25)      # only coded blocks have an entry: the rest keeps whatever was there
    if parts is not None:
        m &= (np.tile(np.arange(25), len(nz)) < 24) | np.repeat(parts == 0, 25)
    assert np.array_equal(third[m], exp["third_context"][m]), f"{tag}: third_context"
    if "partitions" in exp:          # the boolean coder: byte-exact partitions
        got = hip.encode_coefficients(probs, P)
        for p in range(P):
            e = np.asarray(exp["partitions"][p])
            assert len(got[p]) == len(e), f"{tag}: partition {p} is {len(got[p])} bytes, expected {len(e)}"
            bad = np.nonzero(got[p] != e)[0]
            assert bad.size == 0, f"{tag}: partition {p} differs at bytes {bad[:8]} of {len(e)}"


CASES = [  # mbw, mbh, seed, partitions, kwargs
    (4, 3, 1, 1, {}),
    (8, 5, 2, 2, {}),
    (11, 9, 3, 4, dict(density=0.5, big=0.1)),
    (22, 18, 4, 8, {}),
    (6, 7, 5, 8, dict(skip=0.7)),
    (5, 4, 6, 2, dict(p16=0.0)),
    (5, 4, 7, 2, dict(p16=1.0, density=0.9, big=0.3)),
    (120, 68, 8, 8, {}),                          # 1080p geometry
    (480, 270, 9, 8, dict(skip=0.5)),             # 7680x4320: 3.2 million block slots (the scan of their bool counts in two levels)
    (480, 270, 10, 1, dict(skip=0.8, density=0.2)),
]


@pytest.mark.parametrize("mbw,mbh,seed,P,kw", CASES)
def test_count_probs_matches_oracle(mbw, mbh, seed, P, kw):
    coeffs, parts, nz = synthetic(mbw, mbh, seed, **kw)
    hip, dnz = device_with(coeffs, parts, mbw, mbh)
    assert np.array_equal(dnz, nz)
    exp = run_stage(Oracle.stages(), coeffs, parts, nz, mbw, mbh, P)
    check_counts(hip, exp, nz, P, f"{mbw}x{mbh} seed {seed} P{P}")
    # a second call on the same context must not accumulate
    check_counts(hip, exp, nz, P, "second call")
    hip.close()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_count_probs_matches_reference_golden_vectors(path):
    z = np.load(path)
    mbw, mbh, P = (int(z[k]) for k in ("mbw", "mbh", "P"))
    coeffs, parts, nz = (np.ascontiguousarray(z[k]) for k in ("coeffs", "parts", "nz"))
    hip, dnz = device_with(coeffs, parts, mbw, mbh)
    assert np.array_equal(dnz, nz)
    exp = dict(probs=z["probs"], denom=z["denom"], third_context=z["third_context"],
               partitions=[z[f"partition_{p}"] for p in range(P)])
    check_counts(hip, exp, nz, P, os.path.basename(path))
    hip.close()


def test_count_probs_after_inter_transform():
    """The stage consumes what the inter path left on the device: no host round trip of the coefficients."""
    W, H = 352, 288
    s = SynthSequence(W, H, seed=3)
    hip = api.Vp8Hip(s.W, s.H)
    hip.set_segments(default_segments())
    hip.upload_last(*s.frame(0))
    hip.upload_current(*s.frame(1))
    hip.inter_transform(0, 0, 0, 0)
    r = hip.download_results(recon=False)
    coeffs, parts = np.ascontiguousarray(r["MB_coeffs"]), np.ascontiguousarray(r["MB_parts"])
    nz = nz_counts(coeffs, parts)
    assert np.array_equal(hip.debug(api.DBG_MB_NZ), nz)
    for P in (1, 4):
        exp = run_stage(Oracle.stages(), coeffs, parts, nz, s.W // 16, s.H // 16, P)
        check_counts(hip, exp, nz, P, f"after inter_transform, P{P}")
    hip.close()


def test_encode_needs_count_first_grows_its_scratch_and_reports_overflow():
    mbw, mbh = 6, 4
    rng = np.random.default_rng(3)
    coeffs = (rng.integers(67, 2049, size=(mbw * mbh, 25, 16)) * rng.choice([-1, 1], size=(mbw * mbh, 25, 16))).astype(np.int16)
    parts = np.zeros(mbw * mbh, np.int32)
    hip, nz = device_with(coeffs, parts, mbw, mbh)
    probs = np.full(1056, 128, np.uint32)
    with pytest.raises(api.Vp8HipError, match="out of order"):
        hip.encode_coefficients(probs, 2)                   # block contexts not computed yet
    # 19 bools per coefficient, 304 per block: far beyond what the device scratch is sized for at first (64 per block);
    # the stage enlarges it and codes the frame again -- byte-exact like any other frame
    big = mbw * mbh * 25 * 400
    dense = run_stage(Oracle.stages(), coeffs, parts, nz, mbw, mbh, 2, step=big)
    probs, denom0 = hip.count_probs(2)
    assert np.array_equal(probs, dense["probs"])
    got = hip.encode_coefficients(probs, 2, partition_step=big)
    for p in range(2):
        assert np.array_equal(got[p], dense["partitions"][p]), f"dense frame, partition {p}"
    assert sum(len(x) for x in got) > 20000      # ~180 k bools, three times the initial scratch
    with pytest.raises(api.Vp8HipError, match="out of order"):
        hip.encode_coefficients(probs, 4)                   # other partition count than the statistics
    # a frame that fits is fine on the same context afterwards
    coeffs2, parts2, nz2 = synthetic(mbw, mbh, 9)
    hip.upload_mb_data(coeffs2, parts2, np.zeros(mbw * mbh, np.int32))
    assert np.array_equal(hip.prepare_filter_mask(), nz2)
    exp = run_stage(Oracle.stages(), coeffs2, parts2, nz2, mbw, mbh, 2)
    check_counts(hip, exp, nz2, 2, "after overflow", parts=parts2)
    with pytest.raises(api.Vp8HipError, match="do not fit"):
        hip.encode_coefficients(exp["probs"], 2, partition_step=16)   # caller's buffer too small
    hip.close()


def test_entropy_stage_after_key_frame():
    """vp8hip_intra_transform -> filter mask -> count_probs / encode_coefficients: intra macroblocks (MB_parts = 2)
    through the device entropy stage, against the oracle chain."""
    from intra_cases import key_case
    from oracle_lib import oracle_intra
    W, H = 352, 288
    cur, sd = key_case(W, H, 19, 10)
    hip = api.Vp8Hip(W, H)
    hip.upload_current(*cur)
    hip.set_segments(sd)
    hip.intra_transform()
    nz = hip.prepare_filter_mask()
    k = oracle_intra().intra_transform(cur, sd)
    coeffs, parts = np.ascontiguousarray(k["MB_coeffs"]), np.ascontiguousarray(k["MB_parts"])
    assert np.array_equal(nz, nz_counts(coeffs, parts))
    for P in (1, 4):
        exp = run_stage(Oracle.stages(), coeffs, parts, nz, W // 16, H // 16, P)
        check_counts(hip, exp, nz, P, f"key frame P{P}", parts=parts)
    hip.close()
