"""Coefficient entropy stage (count_probs / num_div_denom / encode_coefficients, src/CPU_kernels.cl:347-778):
the CPU restatement (oracle/vp8_entropy_oracle.c) against
  (1) committed golden vectors produced by the reference's own kernels (scripts/gen_golden_entropy.py);
  (2) the reference's own kernels executed live (only where oracle/_ref was built).
Everything here is integer / byte work: bit-exact."""
import glob
import os

import numpy as np
import pytest

from entropy_cases import from_inter_path, nz_counts, run_stage, synthetic
from oracle_lib import Oracle, ref_stages

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "entropy", "*.npz"))) + \
    sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "gfx950", "e_*.npz")))   # the same kernels run on an MI355X


def same(a: dict, b: dict, P: int, tag: str, coded_only=None):
    for k in ("counts", "denom", "probs", "sizes"):
        assert np.array_equal(a[k], b[k]), f"{tag}: {k} differs at {np.nonzero(np.asarray(a[k]) != np.asarray(b[k]))[0][:8]}"
    ta, tb = a["third_context"], b["third_context"]
    if coded_only is not None:          # entries of skipped macroblocks are never written by either side
        m = np.repeat(coded_only, 25)
        ta, tb = ta[m], tb[m]
    assert np.array_equal(ta, tb), f"{tag}: third_context"
    for p in range(P):
        assert np.array_equal(a["partitions"][p], b["partitions"][p]), f"{tag}: partition {p} bytes"


def test_entropy_golden_fixtures_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_restatement_matches_entropy_golden_vectors(path):
    z = np.load(path)
    mbw, mbh, P = (int(z[k]) for k in ("mbw", "mbh", "P"))
    coeffs, parts, nz = (np.ascontiguousarray(z[k]) for k in ("coeffs", "parts", "nz"))
    got = run_stage(Oracle.stages(), coeffs, parts, nz, mbw, mbh, P)
    exp = dict(counts=z["counts"], denom=z["denom"], probs=z["probs"], sizes=z["sizes"], third_context=z["third_context"],
               partitions=[z[f"partition_{p}"] for p in range(P)])
    same(got, exp, P, os.path.basename(path), coded_only=nz != 0)


CASES = [  # mbw, mbh, seed, partitions, kwargs
    (4, 3, 1, 1, {}),
    (8, 5, 2, 2, {}),
    (11, 9, 3, 4, dict(density=0.5, big=0.1)),
    (22, 18, 4, 8, {}),
    (6, 7, 5, 8, dict(skip=0.7)),                 # fewer live rows than partitions in places
    (5, 4, 6, 2, dict(p16=0.0)),                  # no Y2 anywhere
    (5, 4, 7, 2, dict(p16=1.0, density=0.9, big=0.3)),   # dense, every category
]


@pytest.mark.parametrize("mbw,mbh,seed,P,kw", CASES)
def test_restatement_matches_reference_kernels_live(mbw, mbh, seed, P, kw):
    ref = ref_stages()
    if ref is None:
        pytest.skip("oracle/_ref not built (no reference checkout here)")
    coeffs, parts, nz = synthetic(mbw, mbh, seed, **kw)
    a = run_stage(Oracle.stages(), coeffs, parts, nz, mbw, mbh, P)
    b = run_stage(ref, coeffs, parts, nz, mbw, mbh, P)
    same(a, b, P, f"{mbw}x{mbh} seed {seed} P{P}", coded_only=nz != 0)
    assert sum(int(s) for s in a["sizes"]) > 4 * P or kw.get("skip", 0) > 0.5


def test_restatement_matches_reference_kernels_on_real_frame():
    ref = ref_stages()
    if ref is None:
        pytest.skip("oracle/_ref not built (no reference checkout here)")
    W, H = 352, 288
    coeffs, parts, nz = from_inter_path(W, H, 3)
    assert np.array_equal(nz, nz_counts(coeffs, parts))
    for P in (1, 4):
        a = run_stage(Oracle.stages(), coeffs, parts, nz, W // 16, H // 16, P)
        b = run_stage(ref, coeffs, parts, nz, W // 16, H // 16, P)
        same(a, b, P, f"real frame P{P}", coded_only=nz != 0)


def test_restatement_matches_reference_kernels_on_key_frame():
    """MB_parts = are4x4 (intra macroblocks): no Y2 block, luma blocks coded from coefficient 0."""
    from entropy_cases import from_key_frame
    ref = ref_stages()
    if ref is None:
        pytest.skip("oracle/_ref not built (no reference checkout here)")
    W, H = 176, 144
    coeffs, parts, nz = from_key_frame(W, H, 9)
    assert (parts == 2).all() and (nz > 0).any()
    for P in (1, 2):
        a = run_stage(Oracle.stages(), coeffs, parts, nz, W // 16, H // 16, P)
        b = run_stage(ref, coeffs, parts, nz, W // 16, H // 16, P)
        same(a, b, P, f"key frame P{P}", coded_only=nz != 0)
