"""Host intra path -- key frames (intra_transform, src/intra_part.h:1089-1109) and check_SSIM's intra fallback
(src/vp8enc.cpp:231-263, src/intra_part.h:855-1087): the CPU restatement (oracle/vp8_intra_oracle.c) against
  (1) committed golden vectors produced by the reference's own code (scripts/gen_golden_intra.py);
  (2) the reference's own code executed live (only where oracle/_ref was built).
Integer / byte work is bit-exact; the SSIM of an attempt decides a comparison, so it is compared by bit pattern too."""
import glob
import os

import numpy as np
import pytest

from intra_cases import CHECK_KEYS, INTRA_KEYS, compare_check, compare_key, fallback_case, key_case
from oracle_lib import oracle_intra, ref_intra

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "intra", "*.npz")))
KEY_GOLDEN = [p for p in GOLDEN if os.path.basename(p).startswith("key_")]
CHECK_GOLDEN = [p for p in GOLDEN if os.path.basename(p).startswith("check_")]


def ids(paths):
    return [os.path.basename(p)[:-4] for p in paths]


def load_check(z):
    cur = tuple(np.ascontiguousarray(z["cur_" + p]) for p in "YUV")
    inter = {k[3:]: np.ascontiguousarray(z[k]) for k in z.files if k.startswith("in_")}
    exp = {k: z["out_" + k] for k in CHECK_KEYS}
    exp.update(replaced=int(z["replaced"]), new_SSIM=np.float32(z["new_SSIM"]), filter_updated=int(z["filter_updated"]))
    return cur, np.ascontiguousarray(z["sd"]), float(z["target"]), inter, exp


def test_intra_golden_fixtures_present():
    assert len(KEY_GOLDEN) >= 3 and len(CHECK_GOLDEN) >= 3


@pytest.mark.parametrize("path", KEY_GOLDEN, ids=ids(KEY_GOLDEN))
def test_restatement_matches_key_frame_golden_vectors(path):
    z = np.load(path)
    cur = tuple(np.ascontiguousarray(z["cur_" + p]) for p in "YUV")
    got = oracle_intra().intra_transform(cur, z["sd"])
    compare_key(got, {k: z["out_" + k] for k in INTRA_KEYS}, os.path.basename(path))
    assert (got["MB_parts"] == 2).all() and (got["MB_segment_id"] == 0).all()


@pytest.mark.parametrize("path", CHECK_GOLDEN, ids=ids(CHECK_GOLDEN))
def test_restatement_matches_check_ssim_golden_vectors(path):
    cur, sd, target, inter, exp = load_check(np.load(path))
    got = oracle_intra().check_ssim(cur, sd, target, inter)
    compare_check(got, exp, os.path.basename(path))
    untouched = got["is_inter"] == 1
    assert np.array_equal(got["MB_coeffs"][untouched], inter["MB_coeffs"][untouched])
    assert (got["MB_parts"][~untouched] == 2).all()


needs_ref = pytest.mark.skipif(ref_intra() is None, reason="oracle/_ref/libvp8refhost.so not built (no /root/reference here)")


@needs_ref
def test_sub_block_mode_decision_against_reference_code():
    """pick_luma_predictor (src/intra_part.h:252-515): mode, predictor and residual for random and smooth neighbourhoods."""
    o, r = oracle_intra(), ref_intra()
    rng = np.random.default_rng(0)
    seen = set()
    for t in range(4000):
        base = int(rng.integers(0, 256))
        if t % 4 == 0:
            orig, top, left, tl = rng.integers(0, 256, 16), rng.integers(0, 256, 8), rng.integers(0, 256, 4), int(rng.integers(0, 256))
        elif t % 4 == 1:     # smooth: many modes tie, first minimum must win
            orig, top, left, tl = (base + rng.integers(-2, 3, 16)).clip(0, 255), (base + rng.integers(-2, 3, 8)).clip(0, 255), \
                (base + rng.integers(-2, 3, 4)).clip(0, 255), base
        elif t % 4 == 2:     # a directional ramp
            g = np.add.outer(np.arange(4) * int(rng.integers(-20, 21)), np.arange(4) * int(rng.integers(-20, 21))) + base
            orig, top, left, tl = g.clip(0, 255).ravel(), (base + np.arange(8) * int(rng.integers(-20, 21))).clip(0, 255), \
                (base + np.arange(4) * int(rng.integers(-20, 21))).clip(0, 255), base
        else:                # frame-edge constants
            orig, top, left, tl = rng.integers(0, 256, 16), np.full(8, 127), np.full(4, 129), 127
        a = o.pick_luma_predictor(orig, top, left, tl)
        b = r.pick_luma_predictor(orig, top, left, tl)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), t
        seen.add(a[0])
    assert seen == set(range(10))


@needs_ref
def test_macroblock_ssim_against_reference_code():
    """count_SSIM_16x16 (src/intra_part.h:744-853), including the accumulators it carries from plane to plane."""
    o, r = oracle_intra(), ref_intra()
    rng = np.random.default_rng(1)
    for t in range(300):
        a = [rng.integers(0, 256, (16, 16)).astype(np.uint8), rng.integers(0, 256, (8, 8)).astype(np.uint8), rng.integers(0, 256, (8, 8)).astype(np.uint8)]
        amp = int(rng.integers(1, 40))
        b = [(p.astype(int) + rng.integers(-amp, amp + 1, p.shape) + (int(rng.integers(-12, 13)) if t % 3 == 0 else 0)).clip(0, 255).astype(np.uint8) for p in a]
        x, y = o.count_ssim_16x16(a, b), r.count_ssim_16x16(a, b)
        assert x.view(np.uint32) == y.view(np.uint32), (t, x, y)


@needs_ref
@pytest.mark.parametrize("W,H,seed,qi,kind", [(16, 16, 1, 0, "synth"), (32, 16, 2, 5, "noise"), (16, 48, 3, 20, "synth"), (176, 144, 4, 0, "synth"),
                                               (64, 64, 5, 60, "noise"), (80, 48, 6, 127, "synth"), (48, 48, 7, 0, "flat")])
def test_key_frame_against_reference_code(W, H, seed, qi, kind):
    cur, sd = key_case(W, H, seed, qi, kind)
    compare_key(oracle_intra().intra_transform(cur, sd), ref_intra().intra_transform(cur, sd), f"{W}x{H} {kind} q{qi}")


@needs_ref
@pytest.mark.parametrize("W,H,seed,target,cut,qi", [(64, 48, 11, 0.97, True, (40, 100)), (96, 64, 12, 0.93, True, (70, 127)), (176, 144, 13, 0.995, False, (0, 48)),
                                                     (48, 48, 14, 0.99, True, (10, 60)), (64, 32, 15, -1.0, True, (0, 48)), (128, 64, 16, 2.0, True, (30, 90))])
def test_check_ssim_against_reference_code(W, H, seed, target, cut, qi):
    cur, sd, inter = fallback_case(W, H, seed, target if 0 < target < 1 else 0.9, scene_cut=cut, qi=qi)
    a = oracle_intra().check_ssim(cur, sd, target, inter)
    b = ref_intra().check_ssim(cur, sd, target, inter)
    compare_check(a, b, f"{W}x{H} t{target}")
    if target < 0:
        assert a["replaced"] == 0 and np.array_equal(a["recon_Y"], inter["recon_Y"])
