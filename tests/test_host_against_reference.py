"""The host producers of the path's parameters -- vp8_host.cpp's mirrors of get_loopfilter_strength, prepare_segments_data,
scene_change, ParseArgs' defaults and quantizer ladders -- against the REFERENCE'S OWN functions, compiled as they are into
oracle/_ref/libvp8refhost.so (oracle/ref_host_driver.cpp): random planes, quantisers and frame types.  The `-m gpu` half holds
the device versions (vp8hip_loopfilter_strength, vp8hip_auto_segments, vp8hip_chroma_change) against the same functions."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle_lib import REF_HOST_SO
from vp8oclenc_amd import api

pytestmark = pytest.mark.skipif(not os.path.exists(REF_HOST_SO), reason="oracle/_ref/libvp8refhost.so not built (no /root/reference here)")
i32p = C.POINTER(C.c_int32)


def ref():
    lib = C.CDLL(REF_HOST_SO)
    lib.ref_loopfilter_strength.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, i32p]
    lib.ref_prepare_segments_data.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, C.c_int, C.c_int, C.c_int, i32p, i32p]
    lib.ref_scene_change.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, i32p]
    lib.ref_parse_args.argtypes = [C.c_int, C.POINTER(C.c_char_p), i32p, C.POINTER(C.c_float)]
    return lib


def planes(rng, W, H, kind):
    if kind == "noise":
        return rng.integers(0, 256, (H, W)).astype(np.uint8)
    if kind == "flat":
        return np.full((H, W), int(rng.integers(0, 256)), np.uint8)
    y = np.add.outer(np.arange(H) * int(rng.integers(1, 5)), np.arange(W) * int(rng.integers(1, 5))) // int(rng.integers(1, 6)) % 256
    return np.clip(y + rng.integers(-6, 7, (H, W)), 0, 255).astype(np.uint8)


def ref_segments(lib, y, is_key, is_altref, lastqi, altrefqi, qi_min, update=0, shrpnss=0):
    sd = (C.c_int32 * 44)()
    sh = C.c_int32()
    a, b = (C.c_int32 * 4)(*lastqi), (C.c_int32 * 4)(*altrefqi)
    lib.ref_prepare_segments_data(y.ctypes.data, y.shape[1], y.shape[0], int(is_key), int(is_altref), a, b, qi_min, update, shrpnss, sd, C.byref(sh))
    return np.array(list(sd), np.int32).reshape(4, 11), sh.value


def test_loopfilter_strength_and_segment_data():
    lib = ref()
    rng = np.random.default_rng(3)
    for case in range(40):
        W, H = 16 * int(rng.integers(1, 30)), 16 * int(rng.integers(1, 20))
        y = planes(rng, W, H, ["noise", "flat", "ramp", "ramp"][case % 4])
        r, s = C.c_int32(), C.c_int32()
        lib.ref_loopfilter_strength(y.ctypes.data, W, H, C.byref(r), C.byref(s))
        assert api.loopfilter_strength(y) == (r.value, s.value), case
        qmin = int(rng.integers(0, 100)); qmax = int(min(127, qmin + rng.integers(0, 60)))
        lastqi, altrefqi = api.quantizer_ladders(qmin, qmax)
        for is_key, is_altref, update in ((1, 1, 0), (0, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1)):
            exp, exp_sharp = ref_segments(lib, y, is_key, is_altref, lastqi, altrefqi, qmin, update, 7)
            got = api.prepare_segments_data(bool(is_key), altrefqi if is_altref else lastqi, qmin, r.value, s.value, bool(update), 7)
            assert np.array_equal(got, exp), (case, is_key, is_altref, update)
            assert exp_sharp == (7 if update else s.value)


def test_defaults_and_quantizer_ladders():
    lib = ref()
    rng = np.random.default_rng(4)
    out, st = (C.c_int32 * 13)(), C.c_float()

    def parse(*args):
        argv = [b"vp8enc", b"-i", b"in.y4m", b"-o", b"out.ivf", *[str(a).encode() for a in args]]
        arr = (C.c_char_p * len(argv))(*argv)
        assert lib.ref_parse_args(len(argv), arr, out, C.byref(st)) == 0
        return list(out), st.value

    o, target = parse()                      # no options: the defaults vp8drv_default_config states
    cfg = api.DrvConfig()
    api.load_library().vp8drv_default_config(C.byref(cfg))
    assert (o[0], o[1], o[2], o[3], o[4], target) == (cfg.qi_min, cfg.qi_max, cfg.gop_size, cfg.altref_range, cfg.num_partitions, cfg.ssim_target)
    assert (o[5:9], o[9:13]) == tuple(api.quantizer_ladders(cfg.qi_min, cfg.qi_max))
    for _ in range(60):
        a, b = int(rng.integers(0, 128)), int(rng.integers(0, 128))
        o, _t = parse("-qmin", a, "-qmax", b)
        assert (o[0], o[1]) == (min(a, b), max(a, b))                # "wrong quantizer min-max range -> swap"
        assert (o[5:9], o[9:13]) == tuple(api.quantizer_ladders(a, b)), (a, b)
    o, target = parse("-SSIM-target", 93, "-g", 30, "-altref-range", 3, "-partitions", 4)
    assert (o[2], o[3], o[4]) == (30, 3, 4) and abs(target - 0.93) < 1e-7


def test_scene_change_on_planes():
    """the reference's scene_change() on chroma planes (its hold-over is a function-static: ONE sequence in this process)
    against numpy differences + vp8host_scene_change; a key frame sets last_key_detect (intra_part.h:1093)"""
    lib = ref()
    rng = np.random.default_rng(5)
    n = 48 * 32
    st = api.SceneState(0, 0)
    last_key = C.c_int32(0)
    prev = None
    verdicts = []
    for t in range(120):
        base = int(rng.integers(0, 200)) if rng.random() < 0.25 else (0 if prev is None else int(prev[0][0]) - int(prev[0][0]) % 8)
        u = np.clip(base + rng.integers(0, 12, n), 0, 255).astype(np.uint8)
        v = np.clip(base // 2 + rng.integers(0, 12, n), 0, 255).astype(np.uint8)
        if prev is not None:
            r = lib.ref_scene_change(u.ctypes.data, v.ctypes.data, prev[0].ctypes.data, prev[1].ctypes.data, n, t, C.byref(last_key))
            ud = int(np.abs(prev[0].astype(np.int64) - u).sum() // n)
            vd = int(np.abs(prev[1].astype(np.int64) - v).sum() // n)
            got = api.scene_change(st, ud, vd, t)
            assert int(got) == r and st.last_key_detect == last_key.value, t
            if r:
                last_key.value = st.last_key_detect = t
            verdicts.append(r)
        prev = (u, v)
    assert 3 < sum(verdicts) < 60, verdicts


@pytest.mark.gpu
def test_device_parameter_producers_against_the_reference_functions():
    lib = ref()
    rng = np.random.default_rng(6)
    for (W, H) in ((64, 48), (352, 288), (1920, 1088)):
        hip = api.Vp8Hip(W, H)
        prev = None
        for k in range(3):
            y = planes(rng, W, H, ["noise", "ramp", "ramp"][k])
            u = rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8) if k != 2 else np.clip(prev[0].astype(int) + 5, 0, 255).astype(np.uint8)
            v = rng.integers(90, 160, (H // 2, W // 2)).astype(np.uint8)
            hip.upload_current(y, u, v)
            r, s = C.c_int32(), C.c_int32()
            lib.ref_loopfilter_strength(y.ctypes.data, W, H, C.byref(r), C.byref(s))
            assert hip.loopfilter_strength() == (r.value, s.value), (W, H, k)
            lastqi, altrefqi = api.quantizer_ladders(10, 90)
            for is_key, is_altref in ((1, 1), (0, 0), (0, 1)):
                hip.auto_segments(bool(is_key), altrefqi if is_altref else lastqi, 10)
                sd, red, sh = hip.get_segments()
                exp, _ = ref_segments(lib, y, is_key, is_altref, lastqi, altrefqi, 10)
                assert np.array_equal(np.asarray(sd).reshape(4, 11), exp), (W, H, k, is_key, is_altref)
            if prev is not None:
                n = u.size
                assert hip.chroma_change() == (int(np.abs(prev[0].astype(np.int64) - u).sum() // n), int(np.abs(prev[1].astype(np.int64) - v).sum() // n))
            prev = (u, v)
        hip.close()
