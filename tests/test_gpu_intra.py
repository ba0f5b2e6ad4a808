"""GPU parity of the host intra path moved to the device (kernels_intra.hip) through the C ABI:
vp8hip_intra_transform (key frames) and vp8hip_check_ssim (intra fallback of inter frames) against the CPU oracle
(oracle/vp8_intra_oracle.c) and the committed golden vectors of the reference's own code.  Bit-exact, including
the SSIM of every attempt (it decides comparisons)."""
import glob
import os

import numpy as np
import pytest

from intra_cases import CHECK_KEYS, INTRA_KEYS, compare_check, compare_key, fallback_case, key_case
from oracle_lib import oracle_intra
from vp8oclenc_amd import api

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "intra", "*.npz")))
KEY_GOLDEN = [p for p in GOLDEN if os.path.basename(p).startswith("key_")]
CHECK_GOLDEN = [p for p in GOLDEN if os.path.basename(p).startswith("check_")]


def ids(paths):
    return [os.path.basename(p)[:-4] for p in paths]


def hip_key_frame(cur, sd):
    H, W = cur[0].shape
    enc = api.Vp8Hip(W, H)
    try:
        enc.upload_current(*cur)
        enc.set_segments(sd)
        enc.intra_transform()
        r = enc.download_results(recon=True)
        modes, _ = enc.download_intra()
    finally:
        enc.close()
    return {"recon_Y": r["prefilter_Y"], "recon_U": r["prefilter_U"], "recon_V": r["prefilter_V"], "MB_coeffs": r["MB_coeffs"],
            "MB_parts": r["MB_parts"], "MB_segment_id": r["MB_segment_id"], "modes": modes}


def hip_check_ssim(cur, sd, target, inter, modes_of_kept=False):
    """Put the inter results on the device exactly as vp8hip_inter_transform would have left them, then check_ssim."""
    H, W = cur[0].shape
    enc = api.Vp8Hip(W, H, target)
    try:
        if modes_of_kept:
            enc.conformant_stream(True)
        enc.upload_current(*cur)
        enc.set_segments(sd)
        enc.upload_recon(inter["recon_Y"], inter["recon_U"], inter["recon_V"])
        enc.upload_mb_data(inter["MB_coeffs"], inter["MB_parts"], inter["MB_segment_id"])
        enc._debug_set_ssim(inter["MB_SSIM"])
        repl, new, mn = enc.check_ssim()
        r = enc.download_results(recon=True)
        modes, is_inter = enc.download_intra()
    finally:
        enc.close()
    return {"recon_Y": r["prefilter_Y"], "recon_U": r["prefilter_U"], "recon_V": r["prefilter_V"], "MB_coeffs": r["MB_coeffs"],
            "MB_parts": r["MB_parts"], "MB_segment_id": r["MB_segment_id"], "MB_SSIM": r["MB_SSIM"], "modes": modes, "is_inter": is_inter,
            "replaced": repl, "new_SSIM": new, "min_SSIM": mn, "filter_updated": int(mn > np.float32(0.95))}


@pytest.mark.parametrize("path", KEY_GOLDEN, ids=ids(KEY_GOLDEN))
def test_key_frame_matches_reference_golden_vectors(path):
    z = np.load(path)
    cur = tuple(np.ascontiguousarray(z["cur_" + p]) for p in "YUV")
    compare_key(hip_key_frame(cur, z["sd"]), {k: z["out_" + k] for k in INTRA_KEYS}, os.path.basename(path))


@pytest.mark.parametrize("W,H,seed,qi,kind", [(16, 16, 1, 0, "synth"), (32, 16, 2, 5, "noise"), (16, 48, 3, 20, "synth"), (176, 144, 4, 0, "synth"),
                                               (64, 64, 5, 60, "noise"), (80, 48, 6, 127, "synth"), (48, 48, 7, 0, "flat"), (352, 288, 8, 10, "synth"),
                                               (1040, 32, 9, 30, "noise")])
def test_key_frame_matches_oracle(W, H, seed, qi, kind):
    cur, sd = key_case(W, H, seed, qi, kind)
    compare_key(hip_key_frame(cur, sd), oracle_intra().intra_transform(cur, sd), f"{W}x{H} {kind} q{qi}")


def test_key_frame_1080p_matches_oracle():
    cur, sd = key_case(1920, 1088, 21, 0, "synth")
    compare_key(hip_key_frame(cur, sd), oracle_intra().intra_transform(cur, sd), "1080p")


@pytest.mark.parametrize("path", CHECK_GOLDEN, ids=ids(CHECK_GOLDEN))
def test_check_ssim_matches_reference_golden_vectors(path):
    z = np.load(path)
    cur = tuple(np.ascontiguousarray(z["cur_" + p]) for p in "YUV")
    inter = {k[3:]: np.ascontiguousarray(z[k]) for k in z.files if k.startswith("in_")}
    exp = {k: z["out_" + k] for k in CHECK_KEYS}
    exp.update(replaced=int(z["replaced"]), new_SSIM=np.float32(z["new_SSIM"]), filter_updated=int(z["filter_updated"]))
    compare_check(hip_check_ssim(cur, z["sd"], float(z["target"]), inter), exp, os.path.basename(path))


@pytest.mark.parametrize("W,H,seed,target,cut,qi", [(64, 48, 11, 0.97, True, (40, 100)), (96, 64, 12, 0.93, True, (70, 127)), (176, 144, 13, 0.995, False, (0, 48)),
                                                     (48, 48, 14, 0.99, True, (10, 60)), (64, 32, 15, -1.0, True, (0, 48)), (128, 64, 16, 2.0, True, (30, 90)),
                                                     (352, 288, 17, 0.97, True, (40, 100)), (1280, 720, 18, 0.98, True, (30, 90))])
def test_check_ssim_matches_oracle(W, H, seed, target, cut, qi):
    H = H // 16 * 16
    cur, sd, inter = fallback_case(W, H, seed, target if 0 < target < 1 else 0.9, scene_cut=cut, qi=qi)
    a = hip_check_ssim(cur, sd, target, inter)
    b = oracle_intra().check_ssim(cur, sd, target, inter)
    compare_check(a, b, f"{W}x{H} t{target}")
    assert np.float32(a["min_SSIM"]) == np.float32(b["min_SSIM"])


@pytest.mark.parametrize("legacy", [0, 1])
def test_check_ssim_modes_of_kept_attempt(legacy, monkeypatch):
    """vp8hip_conformant_stream (NOT the reference: the decodable variant, include/vp8hip.h), its check_SSIM half, against the oracle's
    switch of the same meaning; nothing but the modes of macroblocks with a kept attempt followed by a failed one may change"""
    from oracle_lib import Oracle
    import subprocess, sys, textwrap
    if legacy:     # the one-wavefront kernel behind VP8HIP_INTRA_CHECK_1WAVE (read once per process): its own interpreter
        code = textwrap.dedent("""
            import sys; sys.path[:0] = [%r, %r]
            import pytest; sys.exit(pytest.main(["-q", "-x", "-m", "gpu", %r + "::test_check_ssim_modes_of_kept_attempt[0]"]))
        """) % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__), __file__)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VP8HIP_INTRA_CHECK_1WAVE="1"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        return
    changed = 0
    for (W, H, seed, target, qi) in ((96, 64, 12, 0.93, (70, 127)), (352, 288, 17, 0.97, (40, 100)), (176, 144, 8, 0.93, (40, 100))):
        cur, sd, inter = fallback_case(W, H, seed, target, scene_cut=True, qi=qi)
        ref = oracle_intra().check_ssim(cur, sd, target, inter)
        Oracle.lib().vp8o_set_conformant_stream(1)
        try:
            b = oracle_intra().check_ssim(cur, sd, target, inter)
        finally:
            Oracle.lib().vp8o_set_conformant_stream(0)
        a = hip_check_ssim(cur, sd, target, inter, modes_of_kept=True)
        compare_check(a, b, f"{W}x{H} kept")
        for k in CHECK_KEYS:
            if k != "modes":
                assert np.array_equal(np.asarray(a[k]), np.asarray(ref[k])), k      # everything but the modes is the reference's
        kept = np.asarray(a["is_inter"]) == 0
        changed += int((np.asarray(a["modes"]).reshape(-1, 16)[kept] != np.asarray(ref["modes"]).reshape(-1, 16)[kept]).any(axis=1).sum())
    assert changed > 0      # the cases do contain macroblocks where the reference's modes are not those of its coefficients


@pytest.mark.parametrize("device_params", [0, 1])
def test_native_frame_loop_with_intra_fallback_and_scene_cut(device_params):
    """The native frame loop (vp8_driver.cpp) with check_SSIM on, coarse quantizers and a scene cut in the middle:
    macroblocks get replaced by intra ones, the cut frame is recoded as a key frame (vp8enc.cpp:443-453) -- against
    the same loop in Python driving the CPU oracle, frame by frame."""
    from oracle_lib import Oracle
    from vp8oclenc_amd.driver import InterPathDriver
    from vp8oclenc_amd.synth import SynthSequence
    W, H, target, qmin, qmax = 320, 192, 0.90, 50, 110
    a_seq, b_seq = SynthSequence(W, H, seed=41), SynthSequence(W, H, seed=97)
    frames = [a_seq.frame(t) for t in range(4)] + [b_seq.frame(t) for t in range(4)]   # the cut hits an ordinary P frame
    drv = api.NativeDriver(W, H, gop_size=150, altref_range=5, qi_min=qmin, qi_max=qmax, ssim_target=target,
                           device_params=device_params, check_ssim=1)
    ora = Oracle(W, H, target)
    do = InterPathDriver(ora, W, H, gop_size=150, altref_range=5, qi_min=qmin, qi_max=qmax, ssim_target=target)
    replaced_total = 0
    for t, (y, u, v) in enumerate(frames):
        drv.encode_frame_host(y, u, v)
        was_key = drv.resolve()     # check_SSIM's verdict: with device parameters it is not waited for inside the call
        b = do.encode_frame(y, u, v)
        assert was_key == (b is None), f"frame {t}: key decision"
        st = drv.stats()
        assert st.key_frames == do.key_frames and st.redone_as_key == do.redone_as_key, t
        ly, lu, lv = drv.hip.download_last()
        oy, ou, ov = ora.download_last()
        assert np.array_equal(ly, oy) and np.array_equal(lu, ou) and np.array_equal(lv, ov), f"filtered recon, frame {t}"
        got = drv.hip.download_results(recon=False)
        modes, is_inter = drv.hip.download_intra()
        if b is None:
            exp = do.last_key
            assert np.array_equal(got["MB_coeffs"][:, :24], exp["MB_coeffs"][:, :24]) and np.array_equal(modes, exp["modes"]), t
            assert (got["MB_parts"] == 2).all()
            continue
        assert st.last_replaced == b["replaced"] and np.float32(st.last_new_ssim) == np.float32(b["new_SSIM"]), t
        replaced_total += b["replaced"]
        repl = b["is_inter"] == 0
        if b["replaced"] or not device_params:    # (defined only when something was replaced: vp8hip_check_ssim_async)
            assert np.array_equal(is_inter, b["is_inter"]) and np.array_equal(modes, b["modes"]), t
        assert np.array_equal(got["MB_coeffs"][:, :24], b["MB_coeffs"][:, :24]), t
        y2 = b["MB_parts"] == 0        # only 16x16 macroblocks have a Y2 block; elsewhere block 24 is stale in every implementation
        assert np.array_equal(got["MB_coeffs"][y2, 24], b["MB_coeffs"][y2, 24]), t
        for k in ("MB_parts", "MB_segment_id", "MB_vectors", "MB_reference_frame"):
            assert np.array_equal(got[k], b[k]), (t, k)
        assert np.array_equal(got["MB_SSIM"].view(np.uint32), b["MB_SSIM"].view(np.uint32)), t
    assert replaced_total > 0, "the case must exercise the fallback"
    assert do.redone_as_key >= 1, "the scene cut must be recoded as a key frame"
    drv.close()
    ora.close()


def test_key_frame_then_inter_frames_follow_the_reference_flow():
    """intra_transform -> filter mask -> loop filter -> LAST = GOLDEN = ALTREF, then an inter frame on top: the
    device chain against the same chain on the CPU oracle."""
    from oracle_lib import Oracle
    from vp8oclenc_amd.synth import SynthSequence
    W, H = 176, 144
    seq = SynthSequence(W, H, seed=31)
    key, nxt = seq.frame(0), seq.frame(1)
    last, _ = api.quantizer_ladders(0, 48)
    red, sh = api.loopfilter_strength(key[0])
    sd_key = api.prepare_segments_data(True, last, 0, red, sh)
    red, sh = api.loopfilter_strength(nxt[0])
    sd_int = api.prepare_segments_data(False, last, 0, red, sh)

    enc = api.Vp8Hip(W, H)
    enc.upload_current(*key)
    enc.set_segments(sd_key)
    enc.intra_transform()
    enc.prepare_filter_mask(False)
    enc.loop_filter()
    got_key = enc.download_last()
    enc.upload_current(*nxt)
    enc.set_segments(sd_int)
    enc.inter_transform(1, 1, 0, 0)
    got = enc.download_results(recon=True)
    enc.close()

    k = oracle_intra().intra_transform(key, sd_key)
    orc = Oracle(W, H)
    orc.set_segments(sd_key)
    orc.upload_recon(k["recon_Y"], k["recon_U"], k["recon_V"])
    orc.upload_mb_data(k["MB_coeffs"], k["MB_parts"], k["MB_segment_id"])
    orc.prepare_filter_mask(False)
    orc.loop_filter()
    exp_key = orc.download_last()
    for a, b, n in zip(got_key, exp_key, "YUV"):
        assert np.array_equal(a, b), f"loop-filtered key frame, plane {n}"
    orc.set_segments(sd_int)
    orc.upload_current(*nxt)
    orc.inter_transform(1, 1, 0, 0)
    exp = orc.download_results(recon=True)
    orc.close()
    for kk in ("MB_parts", "MB_vectors", "MB_coeffs", "MB_segment_id", "prefilter_Y", "prefilter_U", "prefilter_V"):
        assert np.array_equal(got[kk], exp[kk]), kk


def test_intra_wavefront_waits_are_bounded():
    """Row 0 of the key-frame wavefront never publishes (test hook): every other row runs into its bounded wait, the
    launch ends by itself, the next synchronising call reports VP8HIP_ERR_TIMEOUT once, and the context then codes the
    same key frame correctly."""
    import time
    W, H = 128, 96
    cur, sd = key_case(W, H, 3, 10)
    enc = api.Vp8Hip(W, H)
    enc.upload_current(*cur)
    enc.set_segments(sd)
    assert enc.lib.vp8hip_debug_lf_stall(enc.h, 1) == 0
    t0 = time.perf_counter()
    enc.intra_transform()
    rc = enc.lib.vp8hip_synchronize(enc.h)
    dt = time.perf_counter() - t0
    assert rc == -6, rc
    assert dt < 60, f"time-out took {dt:.1f} s"
    assert enc.lib.vp8hip_synchronize(enc.h) == 0
    assert enc.lib.vp8hip_debug_lf_stall(enc.h, 0) == 0
    enc.intra_transform()
    r = enc.download_results(recon=True)
    exp = oracle_intra().intra_transform(cur, sd)
    assert np.array_equal(r["prefilter_Y"], exp["recon_Y"]) and np.array_equal(r["MB_coeffs"][:, :24], exp["MB_coeffs"][:, :24])
    enc.close()


def test_native_frame_loop_detects_scene_cuts_itself():
    """vp8drv_config.scene_detect: scene_change() (vp8enc.cpp:265-311, 408-416) inside the native loop -- chroma differences on
    the device, decision with its hold-over on the host.  Against the same decision made outside (numpy on the source planes +
    vp8host_scene_change) and handed to a second driver as force_key: the same key frames, the same bytes.  The sequence has
    two cuts three frames apart (the second one falls into the "no two forced key frames within 4 frames" rule and comes out of
    the hold-over later) and a GOP boundary."""
    from vp8oclenc_amd.synth import SynthSequence
    W, H = 320, 192
    a, b, c = (SynthSequence(W, H, seed=s) for s in (3, 50, 77))
    def tint(planes, du, dv):
        y, u, v = planes
        return y, np.clip(u.astype(int) + du, 0, 255).astype(np.uint8), np.clip(v.astype(int) + dv, 0, 255).astype(np.uint8)
    frames = [a.frame(t) for t in range(5)] + [tint(b.frame(t), 40, -30) for t in range(3)] + [tint(c.frame(t), -35, 25) for t in range(9)]
    cfg = dict(gop_size=12, altref_range=3, num_partitions=2)
    auto = api.NativeDriver(W, H, scene_detect=1, **cfg)
    manual = api.NativeDriver(W, H, **cfg)
    st = api.SceneState(0, 0)
    gop = api.Gop(12, 3)
    prev = None
    forced, keys = [], []
    for t, (y, u, v) in enumerate(frames):
        g = gop.next()
        force = False
        if not g.current_is_key:
            ud = int(np.abs(prev[1].astype(np.int64) - u).sum() // u.size)
            vd = int(np.abs(prev[2].astype(np.int64) - v).sum() // v.size)
            force = api.scene_change(st, ud, vd, t)
        if g.current_is_key or force:
            st.last_key_detect = t                      # intra_transform, intra_part.h:1093
            gop.key_coded()
        gop.frame_done()
        prev = (y, u, v)
        k1 = auto.encode_frame_host(y, u, v)
        k2 = manual.encode_frame_host(y, u, v, force_key=force)
        assert k1 == k2 == bool(g.current_is_key or force), t
        assert auto.get_frame() == manual.get_frame(), t
        forced.append(force)
        keys.append(k1)
    assert forced[5] and sum(forced) == 2 and forced.index(True, 6) >= 9, forced       # the second cut waits for the hold-over
    assert auto.stats().scene_changes == 2 and manual.stats().scene_changes == 0
    assert keys[0] and sum(keys) >= 3
    with pytest.raises(api.Vp8HipError):               # the batched loop never blocks: it takes force_key only
        api.NativeBatch([auto, api.NativeDriver(W, H, scene_detect=1, **cfg)])
    auto.close(); manual.close()
