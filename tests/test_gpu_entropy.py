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

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "entropy", "*.npz")))


def device_with(coeffs, parts, mbw, mbh):
    hip = api.Vp8Hip(mbw * 16, mbh * 16)
    hip.set_segments(default_segments())
    hip.upload_mb_data(coeffs, parts, np.zeros(mbw * mbh, np.int32))
    nz = hip.prepare_filter_mask()
    return hip, nz


def check_counts(hip, exp, nz, P, tag):
    probs, denom0 = hip.count_probs(P)
    assert np.array_equal(probs, exp["probs"]), f"{tag}: probabilities differ at {np.nonzero(probs != exp['probs'])[0][:8]}"
    assert np.array_equal(denom0, exp["denom"][:1056]), f"{tag}: partition-0 denominators"
    third = hip.debug(api.DBG_THIRD_CONTEXT).reshape(-1)
    m = np.repeat(nz != 0, 25)
    assert np.array_equal(third[m], exp["third_context"][m]), f"{tag}: third_context"


CASES = [  # mbw, mbh, seed, partitions, kwargs
    (4, 3, 1, 1, {}),
    (8, 5, 2, 2, {}),
    (11, 9, 3, 4, dict(density=0.5, big=0.1)),
    (22, 18, 4, 8, {}),
    (6, 7, 5, 8, dict(skip=0.7)),
    (5, 4, 6, 2, dict(p16=0.0)),
    (5, 4, 7, 2, dict(p16=1.0, density=0.9, big=0.3)),
    (120, 68, 8, 8, {}),                          # 1080p geometry
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
    exp = dict(probs=z["probs"], denom=z["denom"], third_context=z["third_context"])
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
