"""End to end on the GPU: the native frame loop (vp8_driver.cpp) with vp8drv_get_frame -- every stage on the device
except the first-partition coder -- must emit the bytes the reference emits: the same loop on the CPU oracle, the
entropy oracle for the coefficient partitions and the reference's own encode_header (where oracle/_ref is built)
for the first partition."""
import numpy as np
import pytest

from bitstream_cases import expected_frame
from oracle_lib import Oracle
from vp8oclenc_amd import api, bitstream
from vp8oclenc_amd.driver import InterPathDriver
from vp8oclenc_amd.synth import SynthSequence

pytestmark = pytest.mark.gpu


def run_sequence(W, H, frames, P=1, **cfg):
    drv = api.NativeDriver(W, H, num_partitions=P, check_ssim=1, **cfg)
    ora = Oracle(W, H, cfg.get("ssim_target", -1.0))
    do = InterPathDriver(ora, W, H, gop_size=cfg.get("gop_size", 150), altref_range=cfg.get("altref_range", 5),
                         qi_min=cfg.get("qi_min", 0), qi_max=cfg.get("qi_max", 48), ssim_target=cfg.get("ssim_target", -1.0))
    stream = []
    for t, (y, u, v) in enumerate(frames):
        was_key = drv.encode_frame_host(y, u, v)
        got = drv.get_frame()
        out = do.encode_frame(y, u, v)
        assert was_key == (out is None), t
        exp = expected_frame(W, H, do.last_key if out is None else out, out is None, P)
        assert len(got) == len(exp), f"frame {t}: {len(got)} bytes, expected {len(exp)}"
        if got != exp:
            a, b = np.frombuffer(got, np.uint8), np.frombuffer(exp, np.uint8)
            raise AssertionError(f"frame {t} differs at bytes {np.nonzero(a != b)[0][:8]} of {len(a)}")
        stream.append(got)
    stats = drv.stats()
    drv.close()
    ora.close()
    return stream, stats, do


@pytest.mark.parametrize("P", [1, 4])
def test_bitstream_of_a_gop_with_golden_and_altref(P):
    W, H = 320, 192
    s = SynthSequence(W, H, seed=51)
    stream, st, _ = run_sequence(W, H, [s.frame(t) for t in range(13)], P=P, gop_size=12, altref_range=5)
    assert st.key_frames == 2 and st.inter_frames == 11          # a second key frame at the GOP boundary
    assert stream[0][3:6] == b"\x9d\x01\x2a" and stream[12][3:6] == b"\x9d\x01\x2a"
    assert all(f[0] & 1 for f in stream[1:12])                    # inter frames
    # the container around it
    ivf = bitstream.ivf_file_header(W, H, 30, 1, len(stream)) + b"".join(bitstream.ivf_frame_header(len(f), i) + f for i, f in enumerate(stream))
    assert ivf[:4] == b"DKIF" and len(ivf) == 32 + sum(12 + len(f) for f in stream)


def test_bitstream_with_intra_fallback_and_scene_cut():
    W, H = 320, 192
    a, b = SynthSequence(W, H, seed=41), SynthSequence(W, H, seed=97)
    frames = [a.frame(t) for t in range(4)] + [b.frame(t) for t in range(3)]
    _, st, do = run_sequence(W, H, frames, P=2, qi_min=50, qi_max=110, ssim_target=0.90)
    assert st.redone_as_key >= 1 and do.redone_as_key == st.redone_as_key


def test_bitstream_from_device_resident_frames_and_device_parameters():
    """vp8drv_encode_frame_device: planes already in HBM, segment data computed on the device."""
    import torch
    W, H = 176, 144
    s = SynthSequence(W, H, seed=61)
    drv = api.NativeDriver(W, H, check_ssim=1, device_params=1)
    ora = Oracle(W, H)
    do = InterPathDriver(ora, W, H)
    for t in range(4):
        y, u, v = s.frame(t)
        d = [torch.from_numpy(p).cuda() for p in (y, u, v)]
        torch.cuda.synchronize()
        was_key = drv.encode_frame_device(*(x.data_ptr() for x in d))
        got = drv.get_frame()
        out = do.encode_frame(y, u, v)
        assert got == expected_frame(W, H, do.last_key if out is None else out, out is None, 1), t
    drv.close()
    ora.close()
