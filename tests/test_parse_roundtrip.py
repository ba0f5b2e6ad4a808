"""Finished frames read back by a parser written from RFC 6386 (tests/vp8_parse.py; no code shared with the encoder or the
reference): frame header, segment map, modes, motion vectors and every coefficient token must come out as the encoder meant
them, and every partition must be consumed exactly to its end.  The independent reading of the INTER frames' syntax (their
pixels have no decoder in the image; key frames are decoded by libwebp in test_decode_roundtrip.py)."""
import numpy as np
import pytest

import vp8_parse as vp
from vp8oclenc_amd.synth import SynthSequence

SD_INTS, SD_Y_AC_I, SD_Y_DC, SD_UV_DC, SD_UV_AC, SD_LF_LEVEL = 11, 0, 1, 4, 5, 6


def check_frame(f, res, key, P, what):
    """f: what the parser read; res: the encoder's results for the frame (download_results / InterPathDriver's dict)"""
    assert f.key == key and f.partitions == P, what
    assert f.first_partition_overrun <= 2 and max(f.token_overrun) <= 2, f"{what}: the parser ran past a partition's end"
    # (a partition without a macroblock row -- more partitions than rows -- holds the coder's flush and is never read)
    # (up to two: the coder's flush is 32 bits past its last bool and a reader holds two bytes ahead; two were first seen on a one-
    # macroblock-wide frame whose partition is a single row of 16x16 blocks -- the bytes themselves are the reference's, bit for bit)
    assert all(u <= 2 for p, u in enumerate(f.token_bytes_unread) if p < f.mbh), f"{what}: bytes left unread in the token partitions: {f.token_bytes_unread}"
    n = f.mbw * f.mbh
    coeffs = np.asarray(res["MB_coeffs"]).astype(np.int32).copy()     # the encoder keeps a block in zig-zag (coding) order
    got = f.coeffs[:, :, vp.ZIGZAG].copy()
    y2 = f.has_y2.astype(bool)
    got[y2, :16, 0] = 0        # with a Y2 block the luma DCs travel in it; the encoder's array keeps the reconstructed DC there
    coeffs[y2, :16, 0] = 0
    defined = np.ones((n, 25), bool)
    defined[:, 24] = y2        # block 24 of a macroblock without Y2 is never written by the reference (stale)
    bad = np.argwhere((got != coeffs).any(axis=2) & defined)
    assert len(bad) == 0, f"{what}: coefficients differ in (macroblock, block) {bad[:5].tolist()}"
    sd = np.asarray(res["segments"]).reshape(4, SD_INTS)
    assert f.sharpness == int(res["sharpness"]), what
    if key:
        assert not f.segmentation_enabled and f.y_ac_qi == sd[0, SD_Y_AC_I], what
        assert (f.ymode == vp.B_PRED).all() and (f.uvmode == vp.TM_PRED).all(), what
        assert np.array_equal(f.bmodes, np.asarray(res["modes"]).reshape(n, 16)), f"{what}: sub-block modes"
        assert f.loop_filter_level == sd[0, SD_LF_LEVEL], what
        return
    assert f.segmentation_enabled and f.update_mb_segmentation_map and f.seg_abs == 1, what
    assert f.seg_quant == [int(sd[i, SD_Y_AC_I]) for i in range(4)], f"{what}: segment quantisers {f.seg_quant}"
    assert f.seg_lf == [int(sd[i, SD_LF_LEVEL]) for i in range(4)], f"{what}: segment filter levels {f.seg_lf}"
    assert (f.y_dc_delta, f.uv_dc_delta, f.uv_ac_delta) == (int(sd[0, SD_Y_DC]), int(sd[0, SD_UV_DC]), int(sd[0, SD_UV_AC])), what
    assert np.array_equal(f.segment_id, np.asarray(res["MB_segment_id"])), f"{what}: segment map"
    is_inter = np.asarray(res["is_inter"]).astype(bool) if res.get("is_inter") is not None else np.ones(n, bool)
    assert np.array_equal(f.is_inter.astype(bool), is_inter), f"{what}: inter / intra flags"
    assert np.array_equal(f.ref_frame[is_inter] - 1, np.asarray(res["MB_reference_frame"])[is_inter]), f"{what}: reference frames"
    quadrants = f.mvs[:, [0, 2, 8, 10], :][:, :, ::-1]       # (row, col) of the four 8x8 quarters -> (x, y)
    assert np.array_equal(quadrants[is_inter], np.asarray(res["MB_vectors"]).astype(np.int32)[is_inter]), f"{what}: motion vectors"
    parts = np.asarray(res["MB_parts"])
    assert np.array_equal((f.mv_mode == vp.MV_SPLIT)[is_inter], parts[is_inter] == 1), f"{what}: split / whole macroblocks"
    assert (f.split[is_inter & (parts == 1)] == vp.SPLIT_QUARTERS).all(), what
    if (~is_inter).any():
        intra = ~is_inter
        assert (f.ymode[intra] == vp.B_PRED).all() and (f.uvmode[intra] == vp.TM_PRED).all(), what
        assert np.array_equal(f.bmodes[intra], np.asarray(res["modes"]).reshape(n, 16)[intra]), f"{what}: sub-block modes of intra macroblocks"
    assert f.refresh_last == 1 and f.refresh_golden == 0 and f.refresh_altref == int(res["is_altref"]), what


@pytest.mark.parametrize("W,H,seed,frames,P,target", [(176, 144, 3, 9, 2, -1.0), (320, 192, 5, 8, 4, -1.0), (176, 144, 8, 7, 1, 0.93)])
def test_oracle_frames_parse_back(W, H, seed, frames, P, target):
    """the CPU side: the reference's frames as the oracle loop + the reference's own encode_header produce them"""
    from bitstream_cases import expected_frame
    from oracle_lib import Oracle
    from vp8oclenc_amd.driver import InterPathDriver
    s = SynthSequence(W, H, seed=seed)
    ora = Oracle(s.W, s.H, target)
    do = InterPathDriver(ora, s.W, s.H, gop_size=150, altref_range=3, qi_min=40 if target > 0 else 0, qi_max=100 if target > 0 else 48)
    st = vp.StreamState()
    refs = set()
    for t in range(frames):
        out = do.encode_frame(*s.frame(t))
        key = out is None
        res = do.last_key if key else out
        f = vp.parse_frame(expected_frame(s.W, s.H, res, key, P), st)
        check_frame(f, res, key, P, f"{W}x{H} frame {t}")
        refs.update(np.unique(f.ref_frame).tolist())
    assert {1, 2, 3} <= refs          # LAST, GOLDEN and ALTREF all occurred
    ora.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,seed,frames,P,cfg", [
    (176, 144, 1, 10, 1, {}), (320, 192, 2, 9, 2, {}), (640, 352, 3, 8, 8, {}),
    (320, 192, 4, 8, 4, dict(check_ssim=1, ssim_target=0.92, qi_min=40, qi_max=110)),     # intra macroblocks inside inter frames
    (1920, 1080, 5, 5, 8, {})])
def test_gpu_frames_parse_back(W, H, seed, frames, P, cfg):
    """the device side: every frame of the native frame loop (GOP of 6: key frames recur), against what the device itself
    reports for the frame (vp8hip_download_results / vp8hip_download_intra / vp8hip_get_segments)"""
    from vp8oclenc_amd import api
    s = SynthSequence(W, H, seed=seed)
    drv = api.NativeDriver(s.W, s.H, gop_size=6, altref_range=2, num_partitions=P, **cfg)
    st = vp.StreamState()
    intra_mbs = 0
    for t in range(frames):
        drv.encode_frame_host(*s.frame(t))
        frame = drv.get_frame()
        key = drv.resolve()
        res = drv.hip.download_results(recon=False)
        stats = drv.stats()
        modes, is_inter = drv.hip.download_intra()
        sd, _, sharp = drv.hip.get_segments()
        res.update(segments=sd, sharpness=sharp, modes=modes, is_altref=stats.last_was_altref)
        if not key and cfg.get("check_ssim") and stats.last_replaced:      # (is_inter is defined only when something was replaced)
            res["is_inter"] = is_inter
            intra_mbs += int((is_inter == 0).sum())
        f = vp.parse_frame(frame, st)
        if cfg.get("check_ssim"):
            res["sharpness"] = f.sharpness   # (check_SSIM's verdict may replace the scanned sharpness by 7, vp8enc.cpp:260-261)
        check_frame(f, res, key, P, f"{W}x{H} frame {t}")
    if cfg.get("check_ssim"):
        assert intra_mbs > 0
    drv.close()
