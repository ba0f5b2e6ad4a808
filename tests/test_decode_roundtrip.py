"""Key frames through an INDEPENDENT decoder (libwebp, tests/webp_decode.py): the emitted frame, wrapped as a WebP file, must
decode to exactly the reconstruction the encoder keeps as its next reference -- bitstream syntax, token coding, dequantisation,
inverse transforms, intra prediction and the normal loop filter all in one comparison that involves none of this repository's
code (nor the reference's) on the decoding side.  SURVEY 8f.3: "a decodable stream ... end-to-end validation with an external
VP8 decoder"."""
import numpy as np
import pytest

import webp_decode
from vp8oclenc_amd.synth import SynthSequence

needs_libwebp = pytest.mark.skipif(webp_decode.libwebp() is None, reason="no libwebp in this image")


def _same(planes, recon, what):
    for name, a, b in zip("YUV", planes, recon):
        assert a.shape == b.shape, (what, name, a.shape, b.shape)
        d = np.abs(a.astype(np.int32) - b.astype(np.int32))
        assert d.max() == 0, f"{what}: plane {name} differs in {int((d > 0).sum())} samples, by up to {int(d.max())}"


@needs_libwebp
@pytest.mark.parametrize("W,H,seed,P", [(176, 144, 3, 1), (320, 192, 5, 4), (640, 352, 7, 8)])
def test_oracle_key_frames_decode_to_the_oracle_reconstruction(W, H, seed, P):
    """the CPU side of the same statement: the reference's key-frame path as the oracle restates it (its first partition by
    the reference's own encode_header where oracle/_ref is built) emits what libwebp decodes to its reconstruction"""
    from bitstream_cases import expected_frame
    from oracle_lib import Oracle
    from vp8oclenc_amd.driver import InterPathDriver
    s = SynthSequence(W, H, seed=seed)
    ora = Oracle(s.W, s.H, -1.0)
    do = InterPathDriver(ora, s.W, s.H, gop_size=150)
    assert do.encode_frame(*s.frame(0)) is None          # a key frame
    frame = expected_frame(s.W, s.H, do.last_key, True, P)
    _same(webp_decode.decode_key_frame(frame), ora.download_last(), f"{W}x{H}")
    ora.close()


@pytest.mark.gpu
@needs_libwebp
@pytest.mark.parametrize("W,H,seed,P,qi", [(176, 144, 1, 1, (0, 48)), (320, 192, 2, 2, (20, 100)), (640, 352, 3, 4, (60, 127)),
                                           (1280, 720, 4, 8, (0, 48)), (1920, 1080, 5, 8, (0, 48)), (3840, 2160, 6, 8, (10, 60))])
def test_gpu_key_frames_decode_to_the_device_reconstruction(W, H, seed, P, qi):
    """every key frame the native frame loop emits (GOP of 3: key frames at 0, 3, 6 with inter frames in between, so the later
    ones follow a reference rotation) decodes, by libwebp, to the filtered reconstruction the device keeps as LAST"""
    from vp8oclenc_amd import api
    s = SynthSequence(W, H, seed=seed)
    drv = api.NativeDriver(s.W, s.H, gop_size=3, num_partitions=P, qi_min=qi[0], qi_max=qi[1])
    keys = 0
    for t in range(7 if W <= 1280 else 4):
        was_key = drv.encode_frame_host(*s.frame(t))
        frame = drv.get_frame()
        if was_key:
            keys += 1
            _same(webp_decode.decode_key_frame(frame), drv.hip.download_last(), f"{W}x{H} frame {t}")
    assert keys >= 2
    drv.close()


@pytest.mark.gpu
@needs_libwebp
def test_gpu_key_frame_with_display_size_decodes_cropped():
    """display size below the coded size (a 180x100 picture coded as 192x112): the header carries the display size and the
    decoder crops to it"""
    from vp8oclenc_amd import api
    s = SynthSequence(192, 112, seed=9)
    drv = api.NativeDriver(s.W, s.H, num_partitions=2, display_width=180, display_height=100)
    assert drv.encode_frame_host(*s.frame(0))
    Y, U, V = webp_decode.decode_key_frame(drv.get_frame())
    ry, ru, rv = drv.hip.download_last()
    assert Y.shape == (100, 180)
    _same((Y, U, V), (ry[:100, :180], ru[:50, :90], rv[:50, :90]), "cropped")
    drv.close()
