"""copy_with_padding (src/encIO.h:141-196), the input side of the path: the restatement against the reference's own function
(oracle/_ref), and vp8hip_set_source_size -- the same on the device, inside the launch that brings a frame into the context's
surfaces -- against the restatement, through every way a frame enters (host planes, device planes, a batch, the native loop)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle_lib import REF_HOST_SO, Oracle


def oracle_pad(planes, W, H):
    y, u, v = (np.ascontiguousarray(p) for p in planes)
    sh, sw = y.shape
    out = [np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)]
    lib = Oracle.lib()
    lib.vp8o_copy_with_padding.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int]
    lib.vp8o_copy_with_padding.restype = None
    lib.vp8o_copy_with_padding(y.ctypes.data, u.ctypes.data, v.ctypes.data, sw, sh, *[o.ctypes.data for o in out], W, H)
    return out


def source(sw, sh, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (sh, sw)).astype(np.uint8), rng.integers(0, 256, (sh // 2, sw // 2)).astype(np.uint8),
            rng.integers(0, 256, (sh // 2, sw // 2)).astype(np.uint8))


SIZES = [(1920, 1080, 1920, 1088), (1280, 720, 1280, 720), (500, 300, 512, 304), (1366, 768, 1376, 768), (176, 130, 176, 144), (30, 18, 32, 32)]


@pytest.mark.parametrize("sw,sh,W,H", SIZES)
def test_restatement_is_edge_replication(sw, sh, W, H):
    src = source(sw, sh, 1)
    for got, s in zip(oracle_pad(src, W, H), src):
        exp = np.pad(s, ((0, got.shape[0] - s.shape[0]), (0, got.shape[1] - s.shape[1])), mode="edge")
        assert np.array_equal(got, exp)


@pytest.mark.skipif(not os.path.exists(REF_HOST_SO), reason="oracle/_ref/libvp8refhost.so not built (no /root/reference here)")
@pytest.mark.parametrize("sw,sh,W,H", SIZES)
def test_restatement_against_the_reference_function(sw, sh, W, H):
    """the reference's own copy_with_padding: identical for Y and U always, and for V whenever the width needs no padding
    (every BASELINE config: 1080p pads rows only).  With a width that needs padding its V lines read U and write into U's next
    row (encIO.h:180-183): U still comes out right (the next iteration rewrites those samples; one row past the plane's end is
    written when no bottom padding follows), V's right padding is never written -- shown here, so that the deviation of the
    restatement (and of the device path) is a known one: they give V the padding the reference means"""
    ref = C.CDLL(REF_HOST_SO)
    ref.ref_copy_with_padding.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_int]
    ref.ref_copy_with_padding.restype = None
    y, u, v = source(sw, sh, 2)
    FILL = 0xA5
    # one slack row after U's source and destination planes: the reference reads and writes one row past them
    ub = np.concatenate([u, np.full((1, sw // 2), 7, np.uint8)])
    out = [np.full((H, W), FILL, np.uint8), np.full((H // 2 + 1, W // 2), FILL, np.uint8), np.full((H // 2, W // 2), FILL, np.uint8)]
    ref.ref_copy_with_padding(y.ctypes.data, ub.ctypes.data, v.ctypes.data, sw, sh, *[o.ctypes.data for o in out], W, H)
    exp = oracle_pad((y, u, v), W, H)
    assert np.array_equal(out[0], exp[0]) and np.array_equal(out[1][:H // 2], exp[1])
    if sw == W:
        assert np.array_equal(out[2], exp[2])
    else:
        cw = sw // 2
        assert np.array_equal(out[2][:, :cw], exp[2][:, :cw])              # inside the source: the same
        assert (out[2][:sh // 2, cw:] == FILL).all()                       # V's right padding: never written
        if sh == H:
            assert (out[1][H // 2, cw:] == 7).all()                        # and U's row past the end: written


@pytest.mark.gpu
@pytest.mark.parametrize("sw,sh,W,H", SIZES)
def test_device_padding_matches_the_restatement(sw, sh, W, H):
    from vp8oclenc_amd import api
    hip = api.Vp8Hip(W, H)
    hip.set_source_size(sw, sh)
    for k, how in enumerate(("host", "device", "device-unaligned")):
        src = source(sw, sh, 10 + k)
        exp = oracle_pad(src, W, H)
        if how == "host":
            hip.upload_current(*src)
        else:
            off = 3 if how.endswith("unaligned") else 0     # source planes at an odd address: the byte-wise path
            dev = [api.DeviceBuffer(p.size + 8) for p in src]
            for d, p in zip(dev, src):
                d.upload(p.reshape(-1), offset=off)
            hip.set_current_device(*[d.data_ptr() + off for d in dev])
        got = [hip.debug(api.DBG_PYRAMID, 3, 0), hip.debug(api.DBG_CURRENT_CHROMA, 0), hip.debug(api.DBG_CURRENT_CHROMA, 1)]
        for name, g, e in zip("YUV", got, exp):
            assert np.array_equal(g, e), (how, name)
    hip.set_source_size(0, 0)                 # back to planes of the coded size
    full = source(W, H, 20)
    hip.upload_current(*full)
    assert np.array_equal(hip.debug(api.DBG_PYRAMID, 3, 0), full[0])
    for bad in ((W + 2, H), (W - 16, H), (W, H - 16), (W - 1, H)):
        if bad[0] > 0 and bad[1] > 0:
            with pytest.raises(api.Vp8HipError):
                hip.set_source_size(*bad)
    hip.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sw,sh,W,H", [(1920, 1080, 1920, 1088), (500, 300, 512, 304)])
def test_native_loop_and_batch_with_a_source_size(sw, sh, W, H):
    """vp8drv_config.src_width/height: frames of the source size in == frames padded by the restatement in, byte for byte
    (host planes, device planes, and two chunks in a batch); key frames carry the source size as display size"""
    import vp8_parse as vp
    from vp8oclenc_amd import api
    rng = np.random.default_rng(3)
    base = [rng.integers(0, 256, (sh + 16, sw + 16)).astype(np.uint8) for _ in range(2)]

    def frame(i, t):      # panned noise-free content: smooth ramps modulated by a coarse pattern
        yy, xx = np.mgrid[0:sh, 0:sw]
        y = ((xx * 2 + yy * 3 + 7 * t + 40 * i) % 256 ^ (base[i][t:t + sh, t:t + sw] >> 5)).astype(np.uint8)
        return y, np.ascontiguousarray(y[::2, ::2] // 2 + 60), np.ascontiguousarray(200 - y[1::2, 1::2] // 2)

    cfg = dict(gop_size=4, altref_range=2, num_partitions=2)
    padded = api.NativeDriver(W, H, display_width=sw, display_height=sh, **cfg)
    host = api.NativeDriver(W, H, src_width=sw, src_height=sh, **cfg)
    dev = api.NativeDriver(W, H, src_width=sw, src_height=sh, **cfg)
    members = [api.NativeDriver(W, H, src_width=sw, src_height=sh, **cfg) for _ in range(2)]
    singles1 = api.NativeDriver(W, H, display_width=sw, display_height=sh, **cfg)
    batch = api.NativeBatch(members)
    for t in range(6):
        src = [frame(i, t) for i in range(2)]
        exp = [oracle_pad(s, W, H) for s in src]
        k = padded.encode_frame_host(*exp[0])
        want = padded.get_frame()
        assert host.encode_frame_host(*src[0]) == k and host.get_frame() == want, t
        d = [[api.to_device(p) for p in s] for s in src]
        ptr = [tuple(p.data_ptr() for p in f) for f in d]
        assert dev.encode_frame_device(*ptr[0]) == k and dev.get_frame() == want, t
        batch.encode_frame_device(ptr)
        batch.get_frames_begin()
        assert members[0].get_frame_end() == want, t
        singles1.encode_frame_host(*exp[1])
        assert members[1].get_frame_end() == singles1.get_frame(), t
        if k:
            f = vp.parse_frame(want, vp.StreamState())
            assert (f.width, f.height) == (sw, sh) and (f.mbw, f.mbh) == (W // 16, H // 16)
        api.device_synchronize()
    batch.close()
    with pytest.raises(api.Vp8HipError):
        api.NativeDriver(W, H, src_width=sw, src_height=sh, device_params=0)
    for x in (padded, host, dev, singles1, *members):
        x.close()
