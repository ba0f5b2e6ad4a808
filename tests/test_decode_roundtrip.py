"""Key frames through an INDEPENDENT decoder (libwebp, tests/webp_decode.py): the emitted frame, wrapped as a WebP file, must
decode to exactly the reconstruction the encoder keeps as its next reference -- bitstream syntax, token coding, dequantisation,
inverse transforms, intra prediction and the normal loop filter all in one comparison that involves none of this repository's
code (nor the reference's) on the decoding side.  SURVEY 8f.3: "a decodable stream ... end-to-end validation with an external
VP8 decoder".

Inter frames: the image holds no VP8 video decoder, so the second half of this file uses one written for these tests from
RFC 6386 alone (tests/vp8_parse.py + tests/vp8_decode.py: dequantisation, inverse transforms, intra and six-tap inter
prediction, loop filter, golden / altref buffer rules).  It is first held against libwebp on the key frames (exact), then
decodes whole sequences from the emitted bytes only and must arrive, frame after frame, at exactly the reconstruction the
encoder keeps -- which closes the loop the byte-level pins leave open: that a decoder following the format sees the pictures
the encoder believes it coded."""
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
        drv.encode_frame_host(*s.frame(t))
        frame = drv.get_frame()
        was_key = drv.resolve()
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


# ---- whole sequences, inter frames included, through the decoder written from the RFC ------------------------------------------

def _decode_sequence(frames_and_recons, what):
    """frames_and_recons: iterable of (frame bytes, is_key, encoder's reconstruction).  Returns statistics of what the stream
    exercised."""
    import vp8_decode
    import vp8_parse as vp
    dec = vp8_decode.Decoder()
    seen = dict(refs=set(), split=0, fractional=0, intra_in_inter=0, inter_frames=0, skipped=0)
    for t, (frame, key, recon) in enumerate(frames_and_recons):
        f, planes = dec.decode(frame)
        assert f.key == key
        if key and webp_decode.libwebp() is not None:
            _same(planes, webp_decode.decode_key_frame(frame), f"{what} frame {t}: the RFC decoder against libwebp")
        _same(planes, recon, f"{what} frame {t} ({'key' if key else 'inter'})")
        if not key:
            inter = f.is_inter.astype(bool)
            seen["inter_frames"] += 1
            seen["refs"].update(np.unique(f.ref_frame[inter]).tolist())
            seen["split"] += int((f.mv_mode[inter] == vp.MV_SPLIT).sum())
            seen["fractional"] += int(((f.mvs[inter] & 3) != 0).any(axis=(1, 2)).sum())
            seen["intra_in_inter"] += int((~inter).sum())
            seen["skipped"] += int(f.skip.sum())
    return seen


def _oracle_frames(W, H, seed, frames, P, target, seq=None, qi=None):
    from bitstream_cases import expected_frame
    from oracle_lib import Oracle
    from vp8oclenc_amd.driver import InterPathDriver
    s = seq or SynthSequence(W, H, seed=seed)
    ora = Oracle(s.W, s.H, target)
    qi = qi or ((40, 100) if target > 0 else (0, 48))
    do = InterPathDriver(ora, s.W, s.H, gop_size=150, altref_range=3, qi_min=qi[0], qi_max=qi[1])
    for t in range(frames):
        out = do.encode_frame(*s.frame(t))
        key = out is None
        yield expected_frame(s.W, s.H, do.last_key if key else out, key, P), key, ora.download_last()
    ora.close()


@pytest.mark.parametrize("W,H,seed,frames,P", [(176, 144, 3, 9, 2), (320, 192, 5, 6, 4)])
def test_oracle_sequences_decode_to_the_oracle_reconstruction(W, H, seed, frames, P):
    """the CPU side: what the reference's frame loop (oracle) reconstructs is what a decoder gets out of its bytes"""
    seen = _decode_sequence(_oracle_frames(W, H, seed, frames, P, -1.0), f"{W}x{H}")
    assert {1, 2, 3} <= seen["refs"] and seen["split"] > 0 and seen["fractional"] > 0, seen


def test_reference_fallback_streams_do_not_decode_to_their_reconstruction_and_why():
    """A defect of the REFERENCE that this decoder brought to light, kept here as a test because the product reproduces the
    reference byte for byte by default: check_SSIM's intra fallback writes a macroblock's sub-block modes on every attempt
    (intra_part.h:964) but its coefficients only when the attempt is kept, so some replaced macroblocks go out with modes
    that do not belong to their coefficients.  A decoder then reconstructs something else than the encoder, and the error
    lives on through inter prediction until the next key frame.  With the modes of the KEPT attempt (the oracle's switch;
    vp8hip_conformant_stream / vp8drv_config.conformant_stream in the product) the same sequence decodes exactly."""
    import vp8_decode
    from oracle_lib import Oracle
    W, H, seed, frames, P, target = 176, 144, 8, 6, 1, 0.93
    dec = vp8_decode.Decoder()
    first_bad = None
    for t, (frame, key, recon) in enumerate(_oracle_frames(W, H, seed, frames, P, target)):
        f, planes = dec.decode(frame)
        d = np.abs(planes[0].astype(np.int32) - recon[0].astype(np.int32))
        if d.max() > 0 and first_bad is None:
            first_bad = t
            assert not key
            intra = {divmod(int(mb), f.mbw) for mb in np.where(f.is_inter == 0)[0]}
            for y, x in np.argwhere(d > 0):     # the damage starts inside replaced macroblocks (the loop filter carries it a few samples out)
                assert any(abs(y // 16 - my) <= 1 and abs(x // 16 - mx) <= 1 for my, mx in intra), (t, y, x)
        if key and t > 0 and first_bad is not None:
            assert d.max() == 0       # a key frame ends the drift
    assert first_bad is not None, "the sequence was chosen because it shows the defect"
    Oracle.lib().vp8o_set_conformant_stream(1)
    try:
        seen = _decode_sequence(_oracle_frames(W, H, seed, frames, P, target), "kept-attempt modes")
    finally:
        Oracle.lib().vp8o_set_conformant_stream(0)
    assert seen["intra_in_inter"] > 0


@pytest.mark.parametrize("qi", [(0, 48), (100, 127)])
def test_reference_predictor_wraps_where_a_decoder_saturates(qi):
    """The second defect of the REFERENCE the decoder brought to light (again reproduced by default, because the product is
    the reference byte for byte): `construct` narrows the last three of the nine first-pass lines of a 4x4 predictor with a
    plain (uchar) cast (GPU_kernels.cl:702-758) where the format saturates (RFC 6386 section 18.3).  On content that makes the
    filter overshoot there -- hard edges -- the encoder's reconstruction and a decoder's differ.  Three statements: the plain
    decoder does NOT arrive at the encoder's reconstruction; a decoder that wraps the same three lines does, exactly (so that
    is the whole difference); and with the format's predictor in the encoder (the oracle's switch; vp8hip_conformant_stream in
    the product) the plain decoder does."""
    import vp8_decode
    from hard_edges import HardEdgeSequence
    from oracle_lib import Oracle
    W, H, frames = 176, 144, 6
    plain, wrapping = vp8_decode.Decoder(), vp8_decode.Decoder(reference_wrap=True)
    differing = 0
    for t, (frame, key, recon) in enumerate(_oracle_frames(W, H, 0, frames, 1, -1.0, seq=HardEdgeSequence(W, H), qi=qi)):
        _, a = plain.decode(frame)
        _, b = wrapping.decode(frame)
        differing += int(sum((p != r).sum() for p, r in zip(a, recon)))
        _same(b, recon, f"frame {t}, decoder with the reference's wrap")
    assert differing > 0, "the content was chosen because it shows the defect"
    Oracle.lib().vp8o_set_conformant_stream(1)
    try:
        seen = _decode_sequence(_oracle_frames(W, H, 0, frames, 1, -1.0, seq=HardEdgeSequence(W, H), qi=qi), "format's predictor")
    finally:
        Oracle.lib().vp8o_set_conformant_stream(0)
    assert seen["fractional"] > 0


class _PannedNoise:
    """white noise panned by whole pixels plus fresh noise on top: costs wrap, every coefficient is alive"""

    def __init__(self, W, H, seed, amp=8):
        self.W, self.H, self.amp = W, H, amp
        self.rng = np.random.default_rng(seed)
        self.base = self.rng.integers(0, 256, (H + 64, W + 64)).astype(np.int32)

    def frame(self, t):
        y = self.base[t:t + self.H, 2 * t:2 * t + self.W] + self.rng.integers(-self.amp, self.amp + 1, (self.H, self.W))
        y = np.clip(y, 0, 255).astype(np.uint8)
        return y, np.ascontiguousarray(y[::2, ::2]), np.ascontiguousarray(255 - y[1::2, 1::2])


@pytest.mark.parametrize("qi", [(0, 48), (60, 127), (120, 127)])
@pytest.mark.parametrize("content", ["noise", "hard", "synth+fallback"])
def test_conformant_streams_decode_exactly_whatever_the_content(content, qi):
    """with the two repairs there is nothing left between the encoder and a decoder: noise, hard edges panned by 1.25 / 0.25
    pixels, and the SSIM fallback, each at fine, coarse and the coarsest quantisers (loop filter levels up to 63)"""
    from hard_edges import HardEdgeSequence
    from oracle_lib import Oracle
    W, H = 176, 144
    seq = {"noise": _PannedNoise(W, H, 1), "hard": HardEdgeSequence(W, H, seed=2, step=(1.25, 0.25)), "synth+fallback": SynthSequence(W, H, seed=3)}[content]
    Oracle.lib().vp8o_set_conformant_stream(1)
    try:
        seen = _decode_sequence(_oracle_frames(W, H, 0, 5, 1, 0.9 if "fallback" in content else -1.0, seq=seq, qi=qi), f"{content} {qi}")
    finally:
        Oracle.lib().vp8o_set_conformant_stream(0)
    assert seen["inter_frames"] == 4 or "fallback" in content      # check_SSIM may recode frames as key frames (all of them at the coarsest)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,frames,P,cfg", [
    (176, 144, 8, 1, dict(gop_size=150, altref_range=3, check_ssim=1)),
    (640, 352, 5, 4, dict(gop_size=4, altref_range=2, check_ssim=1, qi_min=100, qi_max=127)),
    (320, 192, 8, 2, dict(gop_size=6, altref_range=2, check_ssim=1, ssim_target=0.92, qi_min=40, qi_max=110)),
])
def test_gpu_conformant_stream_on_hard_edges(W, H, frames, P, cfg):
    """vp8hip_conformant_stream (vp8drv_config.conformant_stream; NOT the reference, opt-in): byte for byte the oracle loop with
    its switch of the same meaning, and every frame decodes to exactly the device's reconstruction -- on content where the
    reference's own stream does not (asserted too, so that the switch is known to bite)."""
    import vp8_decode
    from bitstream_cases import expected_frame
    from hard_edges import HardEdgeSequence
    from oracle_lib import Oracle
    from vp8oclenc_amd import api
    from vp8oclenc_amd.driver import InterPathDriver
    s = HardEdgeSequence(W, H)
    ref_drv = api.NativeDriver(s.W, s.H, num_partitions=P, **cfg)
    drv = api.NativeDriver(s.W, s.H, num_partitions=P, conformant_stream=1, **cfg)
    target = cfg.get("ssim_target", -1.0)
    ora = Oracle(s.W, s.H, target)
    do = InterPathDriver(ora, s.W, s.H, gop_size=cfg["gop_size"], altref_range=cfg["altref_range"], qi_min=cfg.get("qi_min", 0),
                         qi_max=cfg.get("qi_max", 48), ssim_target=target)
    dec, ref_dec = vp8_decode.Decoder(), vp8_decode.Decoder()
    reference_drift = changed = 0
    Oracle.lib().vp8o_set_conformant_stream(1)
    try:
        for t in range(frames):
            y, u, v = s.frame(t)
            drv.encode_frame_host(y, u, v)
            frame = drv.get_frame()
            key = drv.resolve()
            out = do.encode_frame(y, u, v)
            assert bool(key) == (out is None), t
            assert frame == expected_frame(s.W, s.H, do.last_key if out is None else out, out is None, P), f"frame {t}: not the oracle's bytes"
            _, planes = dec.decode(frame)
            _same(planes, drv.hip.download_last(), f"{W}x{H} frame {t}")
            ref_drv.encode_frame_host(y, u, v)
            ref_frame = ref_drv.get_frame()
            changed += ref_frame != frame
            _, planes = ref_dec.decode(ref_frame)
            reference_drift += int(sum((p != r).sum() for p, r in zip(planes, ref_drv.hip.download_last())))
    finally:
        Oracle.lib().vp8o_set_conformant_stream(0)
    assert reference_drift > 0 and changed > 0
    for d in (drv, ref_drv):
        d.close()
    ora.close()


@pytest.mark.gpu
def test_gpu_conformant_stream_in_a_batch():
    """the batched launch takes the same switch (all members must agree): frames of a batch of two == the single contexts'"""
    from hard_edges import HardEdgeSequence
    from vp8oclenc_amd import api
    seqs = [HardEdgeSequence(320, 192, seed=s) for s in (1, 2)]
    cfg = dict(gop_size=5, altref_range=2, num_partitions=2, device_params=1, check_ssim=0, conformant_stream=1)
    singles = [api.NativeDriver(320, 192, **cfg) for _ in seqs]
    plain = [api.NativeDriver(320, 192, **dict(cfg, conformant_stream=0)) for _ in seqs]
    members = [api.NativeDriver(320, 192, **cfg) for _ in seqs]
    batch = api.NativeBatch(members)
    changed = 0
    for t in range(7):
        dev = [tuple(api.to_device(p) for p in s.frame(t)) for s in seqs]
        ptr = [tuple(p.data_ptr() for p in f) for f in dev]
        batch.encode_frame_device(ptr)
        batch.get_frames_begin()
        for i, d in enumerate(singles):
            d.encode_frame_device(*ptr[i])
            plain[i].encode_frame_device(*ptr[i])
            frame = members[i].get_frame_end()
            assert d.get_frame() == frame, (t, i)
            changed += plain[i].get_frame() != frame
        api.device_synchronize()
    assert changed > 0
    batch.close()
    with pytest.raises(api.Vp8HipError):      # one launch, one predictor
        api.NativeBatch([members[0], plain[0]])


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,seed,frames,P,cfg", [
    (176, 144, 1, 12, 1, dict(gop_size=5, altref_range=2)),                                                  # three GOPs
    (640, 352, 3, 7, 4, dict(gop_size=150, altref_range=3, qi_min=20, qi_max=100)),
    (320, 192, 4, 8, 4, dict(gop_size=6, altref_range=2, check_ssim=1, ssim_target=0.92, qi_min=40, qi_max=110, conformant_stream=1)),   # intra macroblocks inside inter frames
    (1280, 720, 5, 4, 8, dict(gop_size=150, altref_range=2)),
    (1920, 1080, 6, 3, 8, dict(gop_size=150, altref_range=2)),
])
def test_gpu_sequences_decode_to_the_device_reconstruction(W, H, seed, frames, P, cfg):
    """the frames the native loop emits, decoded from their bytes alone, against the filtered reconstruction the device keeps
    after each of them (vp8hip_download_last)"""
    from vp8oclenc_amd import api
    s = SynthSequence(W, H, seed=seed)
    drv = api.NativeDriver(s.W, s.H, num_partitions=P, **cfg)

    def run():
        for t in range(frames):
            drv.encode_frame_host(*s.frame(t))
            frame = drv.get_frame()
            yield frame, bool(drv.resolve()), drv.hip.download_last()

    seen = _decode_sequence(run(), f"{W}x{H}")
    assert seen["inter_frames"] >= 2 and seen["fractional"] > 0 and {1, 3} <= seen["refs"], seen
    if "ssim_target" in cfg:
        assert seen["intra_in_inter"] > 0, seen
    drv.close()


# ---- the decoder on streams it has never seen the encoder of ---------------------------------------------------------------------

def _foreign_key_frame(rgb, quality, method):
    """a VP8 key frame made by libwebp's ENCODER (through Pillow): the 'VP8 ' chunk of a lossy .webp"""
    import io
    import struct
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(rgb, "RGB").save(buf, format="WEBP", quality=quality, method=method)
    webp = buf.getvalue()
    assert webp[:4] == b"RIFF" and webp[8:12] == b"WEBP"
    off = 12
    while off < len(webp):
        tag, size = webp[off:off + 4], struct.unpack("<I", webp[off + 4:off + 8])[0]
        if tag == b"VP8 ":
            return webp[off + 8:off + 8 + size]
        off += 8 + size + (size & 1)
    raise ValueError("no VP8 chunk")


def _pillow_writes_webp():
    try:
        from PIL import features
        return bool(features.check("webp"))
    except Exception:
        return False


@needs_libwebp
@pytest.mark.skipif(not _pillow_writes_webp(), reason="no Pillow with WebP in this image")
def test_the_rfc_decoder_decodes_foreign_streams_as_libwebp_does():
    """tests/vp8_decode.py is a VP8 decoder, not an echo of this encoder: key frames made by libwebp's encoder -- segments with
    their own quantisers and filter levels, all four 16x16 luma modes beside B_PRED, every chroma mode, sizes that are not
    multiples of 16, qualities 5 to 99 -- decode to exactly what libwebp's decoder makes of them"""
    import vp8_decode
    import vp8_parse as vp
    rng = np.random.default_rng(1)
    ymodes, uvmodes, segs, levels = np.zeros(5, np.int64), np.zeros(4, np.int64), 0, set()
    for case in range(24):
        W, H = int(rng.integers(2, 30)) * 8 + int(rng.integers(0, 8)), int(rng.integers(2, 24)) * 8 + int(rng.integers(0, 8))
        y = SynthSequence(320, 256, seed=case).frame(case)[0]
        rgb = np.stack([y[:H, :W], np.roll(y, 3, 1)[:H, :W], 255 - y[:H, :W]], axis=2)
        if case % 3 == 0:
            rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        elif case % 3 == 1:
            rgb = (rgb // 64 * 64).astype(np.uint8)            # flat areas: the 16x16 modes
        frame = _foreign_key_frame(np.ascontiguousarray(rgb), int(rng.integers(5, 100)), int(rng.integers(0, 7)))
        f, planes = vp8_decode.Decoder().decode(frame)
        ref = webp_decode.decode_key_frame(frame)
        for name, a, b in zip("YUV", planes, ref):
            assert np.array_equal(a[:b.shape[0], :b.shape[1]], b), (case, W, H, name)
        ymodes += np.bincount(f.ymode, minlength=5)
        uvmodes += np.bincount(f.uvmode, minlength=4)
        segs += int(f.segmentation_enabled)
        levels.add(int(f.loop_filter_level))
    assert (ymodes > 0).all() and (uvmodes > 0).all() and segs > 0 and len(levels) > 5, (ymodes, uvmodes, segs, levels)
