"""Inputs for the host intra path (key frames and check_SSIM's intra fallback): shared by the oracle tests, the
golden-vector generator (scripts/gen_golden_intra.py) and the GPU parity tests."""
from __future__ import annotations

import numpy as np

from oracle_lib import Oracle
from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

INTRA_KEYS = ("recon_Y", "recon_U", "recon_V", "MB_coeffs", "MB_parts", "MB_segment_id", "modes")
CHECK_KEYS = ("recon_Y", "recon_U", "recon_V", "MB_coeffs", "MB_parts", "MB_segment_id", "MB_SSIM", "is_inter", "modes")


def noise_frame(W, H, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (H, W)).astype(np.uint8), rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8),
            rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8))


def flat_frame(W, H, y=128, u=90, v=200):
    return (np.full((H, W), y, np.uint8), np.full((H // 2, W // 2), u, np.uint8), np.full((H // 2, W // 2), v, np.uint8))


def key_case(W, H, seed, qi_min=0, kind="synth"):
    """(cur planes, key-frame segment data) -- prepare_segments_data with current_is_key_frame, src/vp8enc.cpp:133-166."""
    cur = SynthSequence(W, H, seed=seed).frame(1) if kind == "synth" else noise_frame(W, H, seed) if kind == "noise" else flat_frame(W, H)
    last, _ = api.quantizer_ladders(qi_min, 48)
    red, sh = api.loopfilter_strength(cur[0])
    return cur, api.prepare_segments_data(True, last, qi_min, red, sh)


def fallback_case(W, H, seed, target, scene_cut=False, qi=(0, 48), gap=2):
    """An inter frame through the CPU oracle, stopped before the loop filter: (cur, sd, inter results).
    scene_cut: LAST comes from unrelated content, so most macroblocks fall below the target and are tried as intra."""
    seq = SynthSequence(W, H, seed=seed)
    cur = seq.frame(gap)
    prev = SynthSequence(W, H, seed=seed + 100).frame(0) if scene_cut else seq.frame(0)
    last, _ = api.quantizer_ladders(*qi)
    red, sh = api.loopfilter_strength(cur[0])
    sd = api.prepare_segments_data(False, last, qi[0], red, sh)
    orc = Oracle(W, H, target)
    orc.upload_last(*prev)
    orc.set_segments(sd)
    orc.upload_current(*cur)
    orc.inter_transform(0, 0, 0, 0)
    res = orc.download_results(recon=True)
    orc.close()
    inter = {"recon_Y": res["prefilter_Y"], "recon_U": res["prefilter_U"], "recon_V": res["prefilter_V"],
             "MB_coeffs": res["MB_coeffs"], "MB_parts": res["MB_parts"], "MB_segment_id": res["MB_segment_id"],
             "MB_SSIM": res["MB_SSIM"]}
    return cur, sd, inter


def compare_check(a: dict, b: dict, tag: str):
    """Everything check_SSIM leaves behind, bit for bit; block 24 of a replaced macroblock is undefined in the reference
    (uninitialised stack copy, src/intra_part.h:858,1066)."""
    assert a["replaced"] == b["replaced"], (tag, a["replaced"], b["replaced"])
    assert np.float32(a["new_SSIM"]) == np.float32(b["new_SSIM"]), (tag, a["new_SSIM"], b["new_SSIM"])
    assert a["filter_updated"] == b["filter_updated"], tag
    for k in CHECK_KEYS:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if k == "MB_coeffs":
            repl = np.asarray(a["is_inter"]) == 0
            assert np.array_equal(x[:, :24], y[:, :24]), f"{tag}: {k}"
            assert np.array_equal(x[~repl, 24], y[~repl, 24]), f"{tag}: Y2 block of untouched macroblocks"
        elif k == "MB_SSIM":
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f"{tag}: {k} (bit pattern)"
        else:
            assert np.array_equal(x, y), f"{tag}: {k} differs in {int((x != y).sum())} places"


def compare_key(a: dict, b: dict, tag: str):
    for k in INTRA_KEYS:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if k == "MB_coeffs":
            x, y = x[:, :24], y[:, :24]
        assert np.array_equal(x, y), f"{tag}: {k} differs in {int((x != y).sum())} places"
