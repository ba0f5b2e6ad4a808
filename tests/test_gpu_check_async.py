"""check_SSIM without the host in the middle (vp8hip_check_ssim_async, vp8drv_config.check_ssim with device parameters): the
fallback, the statistics and the filter update `if (min1 > 0.95) prepare_segments_data(1, 7)` (src/vp8enc.cpp:231-263) run on the
device in front of the loop filter, the "redo as key frame" decision (:443-453) is taken one call later -- and every byte must
be what the reference's sequence of calls gives: the step-by-step entry points with the host deciding, and the same loop on the
CPU oracle."""
import numpy as np
import pytest

from bitstream_cases import expected_frame
from oracle_lib import Oracle
from vp8oclenc_amd import api
from vp8oclenc_amd.driver import InterPathDriver
from vp8oclenc_amd.synth import SynthSequence

pytestmark = pytest.mark.gpu


def _frame(hip, frames, t, refqi, qi_min, key):
    hip.upload_current(*frames[t])
    hip.auto_segments(key, refqi, qi_min)


@pytest.mark.parametrize("W,H,qi,target,want", [(320, 192, (0, 6), -1.0, "update"), (320, 192, (40, 110), 0.92, "replace"),
                                                 (640, 352, (0, 4), 0.999, "both"), (1920, 1088, (0, 8), -1.0, "update")])
def test_async_check_equals_the_host_driven_sequence(W, H, qi, target, want):
    s = SynthSequence(W, H, seed=7)
    frames = [s.frame(t) for t in range(4)]
    lastqi, altrefqi = api.quantizer_ladders(*qi)
    qi_min = min(qi)
    a, b = api.Vp8Hip(s.W, s.H, target), api.Vp8Hip(s.W, s.H, target)
    seen = set()
    for hip in (a, b):      # frame 0: key frame
        _frame(hip, frames, 0, altrefqi, qi_min, True)
        hip.intra_transform()
        hip.prepare_filter_mask(False)
        hip.loop_filter()
    for t in range(1, 4):
        first = t == 1
        # a: check_SSIM as the reference's host does it
        _frame(a, frames, t, lastqi, qi_min, False)
        a.inter_transform(first, first, 0 if first else 1, 0)
        repl, new, mn = a.check_ssim()
        sharp_a = -1
        if mn > np.float32(0.95):
            sd, red, sharp = a.get_segments()
            a.set_segments(api.prepare_segments_data(False, lastqi, qi_min, red, sharp, True, 7))
            sharp_a = 7
            seen.add("update")
        if repl:
            seen.add("replace")
            a.prepare_filter_mask(False)
        ra, ia = a.download_results(), a.download_intra()
        nza, maska = a.debug(api.DBG_MB_NZ), a.debug(api.DBG_MB_MASK)
        a.loop_filter()
        # b: nobody waits
        _frame(b, frames, t, lastqi, qi_min, False)
        b.inter_transform(first, first, 0 if first else 1, 0)
        b.check_ssim_async(lastqi, qi_min)
        rb, ib = b.download_results(), b.download_intra()
        nzb, maskb = b.debug(api.DBG_MB_NZ), b.debug(api.DBG_MB_MASK)
        b.loop_filter()
        replb, newb, mnb, upd = b.check_ssim_result()
        assert (repl, new.view(np.uint32), mn.view(np.uint32)) == (replb, newb.view(np.uint32), mnb.view(np.uint32)), t
        assert upd == (sharp_a == 7)
        for k in ra:
            x, y = ra[k], rb[k]
            if k == "MB_SSIM":
                x, y = x.view(np.uint32), y.view(np.uint32)
            if k == "MB_coeffs":
                x, y = x[:, :24], y[:, :24]
            assert np.array_equal(x, y), (t, k)
        if repl:      # e_data.mode / is_inter_mb: written (and read by the header coder) only when the fallback had work
            assert np.array_equal(ia[0], ib[0]) and np.array_equal(ia[1], ib[1]), t
        assert np.array_equal(nza, nzb) and np.array_equal(maska, maskb), t
        assert np.array_equal(a.get_segments()[0], b.get_segments()[0]), t
        for p_, q_ in zip(a.download_last(), b.download_last()):
            assert np.array_equal(p_, q_), t
        # the frame header carries the sharpness in force: explicit on a, the device's own word on b
        class P(api.C.Structure):
            _fields_ = [(n, api.C.c_int32) for n in ("is_key", "is_golden", "is_altref", "loop_filter_type", "loop_filter_sharpness",
                                                      "partitions_log2", "width", "height", "use_intra_info")]
        outs = []
        for hip, sh in ((a, sharp_a), (b, -1)):
            hip.lib.vp8hip_encode_frame.argtypes = [api.C.c_void_p, api.C.c_int, api.C.c_void_p, api.C.c_void_p, api.C.c_size_t, api.C.POINTER(api.C.c_size_t)]
            buf, n = np.zeros(hip.mbs * 900 + 65536, np.uint8), api.C.c_size_t(0)
            p = P(0, 0, 0, 0, sh, 0, 0, 0, 1 if repl else 0)
            assert hip.lib.vp8hip_encode_frame(hip.h, 2, api.C.byref(p), buf.ctypes.data, len(buf), api.C.byref(n)) == 0
            outs.append(buf[:n.value].tobytes())
        assert outs[0] == outs[1], t
    a.close()
    b.close()
    if want in ("update", "both"):
        assert "update" in seen
    if want in ("replace", "both"):
        assert "replace" in seen


@pytest.mark.parametrize("W,H,qi,target,gop", [(320, 192, (0, 6), -1.0, 150), (320, 192, (40, 110), 0.92, 6), (320, 192, (50, 110), 0.90, 150)])
def test_native_loop_with_async_check_emits_the_oracle_loops_bytes(W, H, qi, target, gop):
    """frames on which the worst macroblock is above 0.95 (low quantizers: the filter update), frames with replaced macroblocks, and
    frames sent back to be key frames -- the native loop's bytes against the reference's loop on the CPU oracle"""
    a, b = SynthSequence(W, H, seed=41), SynthSequence(W, H, seed=97)
    frames = [a.frame(t) for t in range(4)] + [b.frame(t) for t in range(4)]      # the cut hits an ordinary P frame
    drv = api.NativeDriver(W, H, num_partitions=2, check_ssim=1, device_params=1, gop_size=gop, qi_min=qi[0], qi_max=qi[1], ssim_target=target)
    ora = Oracle(W, H, target)
    do = InterPathDriver(ora, W, H, gop_size=gop, qi_min=qi[0], qi_max=qi[1], ssim_target=target)
    updates = 0
    for t, (y, u, v) in enumerate(frames):
        d = [api.to_device(p) for p in (y, u, v)]
        api.device_synchronize()
        drv.encode_frame_device(*(x.data_ptr() for x in d))
        got = drv.get_frame() if t % 3 else None     # without a bitstream request the verdict is taken by the next call
        out = do.encode_frame(y, u, v)
        if out is not None and out.get("min_SSIM", 0) > np.float32(0.95):
            updates += 1
        if got is not None:
            assert drv.resolve() == (out is None), t
            assert got == expected_frame(W, H, do.last_key if out is None else out, out is None, 2), t
        for p_, q_ in zip(drv.hip.download_last() if got is not None else (), ora.download_last()):
            assert np.array_equal(p_, q_), t
    drv.resolve()
    st = drv.stats()
    assert (st.inter_frames, st.key_frames, st.redone_as_key) == (do.inter_frames, do.key_frames, do.redone_as_key)
    for p_, q_ in zip(drv.hip.download_last(), ora.download_last()):
        assert np.array_equal(p_, q_)
    if qi[1] <= 8:
        assert updates >= 2
    if qi[0] == 50:
        assert do.redone_as_key >= 1
    drv.close()
    ora.close()


def test_a_frame_that_overflows_the_callers_buffer_is_delivered_again():
    """vp8drv_get_frame into a buffer that is too small says VP8HIP_ERR_OVERFLOW and keeps the coded frame: the same call with a
    larger buffer delivers it, and the next frame codes as if nothing had happened"""
    W, H = 320, 192
    s = SynthSequence(W, H, seed=5)
    a, b = api.NativeDriver(W, H, check_ssim=1), api.NativeDriver(W, H, check_ssim=1)
    for t in range(3):
        y, u, v = s.frame(t)
        a.encode_frame_host(y, u, v)
        b.encode_frame_host(y, u, v)
        b._frame_buf = np.zeros(64, np.uint8)         # far too small: get_frame doubles it until the frame fits
        fa, fb = a.get_frame(), b.get_frame()
        assert fa == fb and len(fb) > 64, t
    # a caller that gives the frame up instead goes on with the next one
    y, u, v = s.frame(3)
    b.encode_frame_host(y, u, v)
    n = api.C.c_size_t(0)
    small = np.zeros(16, np.uint8)
    b.lib.vp8drv_get_frame.argtypes = [api.C.c_void_p, api.C.c_void_p, api.C.c_size_t, api.C.POINTER(api.C.c_size_t)]
    assert b.lib.vp8drv_get_frame(b.h, small.ctypes.data, 16, api.C.byref(n)) == api.ERR_OVERFLOW
    a.encode_frame_host(y, u, v)
    a.get_frame()
    y, u, v = s.frame(4)
    a.encode_frame_host(y, u, v)
    b.encode_frame_host(y, u, v)
    assert a.get_frame() == b.get_frame()
    a.close()
    b.close()
