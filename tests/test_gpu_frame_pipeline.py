"""One video with the next frame under way before this one's bytes are taken (vp8drv_get_frame_begin, encode of frame t + 1,
vp8drv_get_frame_end; include/vp8hip.h, vp8hip_encode_frame_begin): the entropy stage of frame t runs on the context's third
stream beside its loop filter and beside frame t + 1's input side, and the bytes must be the ones the one-shot order gives --
which the other tests hold against the reference's."""
import numpy as np
import pytest

from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

pytestmark = pytest.mark.gpu


def _frames(W, H, n, cut):
    a, b = SynthSequence(W, H, seed=11), SynthSequence(W, H, seed=12)
    return a.W, a.H, [(b if (cut and t >= cut) else a).frame(t) for t in range(n)]


@pytest.mark.parametrize("W,H,n,cfg,cut", [
    (320, 240, 10, dict(gop_size=4, num_partitions=2), 0),                                   # key frames in the flow
    (640, 352, 8, dict(gop_size=150, num_partitions=8, qi_min=30, qi_max=100, ssim_target=0.93), 0),   # macroblocks replaced by the fallback
    (320, 192, 9, dict(gop_size=150, num_partitions=1, ssim_target=0.6), 5),                  # a cut: frames sent back to be key frames
    (1920, 1088, 6, dict(gop_size=150, num_partitions=4, qi_min=0, qi_max=8), 0),             # the filter update (min SSIM > 0.95)
])
def test_next_frame_started_before_the_bytes_are_taken(W, H, n, cfg, cut):
    W, H, frames = _frames(W, H, n, cut)
    base = dict(device_params=1, check_ssim=1, overlap_filter=1, altref_range=3)
    base.update(cfg)
    one = api.NativeDriver(W, H, **base)
    two = api.NativeDriver(W, H, **base)
    two.hip.reserve_frame_path_dense()
    want, keys_one = [], []
    for f in frames:
        one.encode_frame_host(*f)
        want.append(one.get_frame())
        keys_one.append(one.resolve())
    got, keys_two, pending = [], [], False
    for f in frames:
        two.encode_frame_host(*f)          # frame t under way ...
        if pending:
            got.append(two.get_frame_end())   # ... then frame t - 1's bytes
        two.get_frame_begin()
        keys_two.append(two.resolve())
        pending = True
    got.append(two.get_frame_end())
    assert keys_one == keys_two
    for t, (a, b) in enumerate(zip(want, got)):
        assert a == b, f"frame {t}: {len(a)} vs {len(b)} bytes"
    for p, q in zip(one.hip.download_last(), two.hip.download_last()):
        assert np.array_equal(p, q)
    so, st = one.stats(), two.stats()
    assert (so.key_frames, so.redone_as_key) == (st.key_frames, st.redone_as_key)
    one.close()
    two.close()


def test_a_recode_whose_input_is_gone_is_refused_not_garbled():
    """Without the dense reservation a frame denser than the coder's scratch is coded again -- impossible once the next frame has
    overwritten its results: VP8HIP_ERR_STATE, never wrong bytes."""
    s = SynthSequence(320, 192, seed=3)
    rng = np.random.default_rng(5)
    noise = [(rng.integers(0, 256, (s.H, s.W)).astype(np.uint8), rng.integers(0, 256, (s.H // 2, s.W // 2)).astype(np.uint8),
              rng.integers(0, 256, (s.H // 2, s.W // 2)).astype(np.uint8)) for _ in range(3)]
    d = api.NativeDriver(s.W, s.H, device_params=1, check_ssim=1, overlap_filter=1, gop_size=150, qi_min=0, qi_max=0, num_partitions=1)
    d.encode_frame_host(*noise[0])     # key frame of noise at quantiser 0: far more than 64 bools per block
    d.get_frame_begin()
    d.resolve()
    d.encode_frame_host(*noise[1])
    with pytest.raises(api.Vp8HipError, match=r"\(-4\)"):
        d.get_frame_end()
    d.close()


def test_native_video_loop_delivers_the_frames_of_the_call_by_call_loop():
    """vp8drv_encode_video_device (one video, frames out, the whole loop natively) against encode / get_frame call by call: the same
    bytes frame by frame -- with the filter-overlap mode (the stage on its third stream, the next frame started before a frame's bytes
    are taken) and without, key frames inside the run, check_SSIM replacing macroblocks and sending frames back"""
    from vp8oclenc_amd.synth import SynthSequence
    W, H, ND, N = 320, 192, 7, 23
    seq = SynthSequence(W, H, seed=33)
    dev = [tuple(api.to_device(p) for p in seq.frame(t)) for t in range(ND)]
    ptrs = [tuple(p.data_ptr() for p in f) for f in dev]
    for cfg in (dict(overlap_filter=1, gop_size=150), dict(overlap_filter=0, gop_size=6, num_partitions=4),
                dict(overlap_filter=1, gop_size=9, ssim_target=0.92, qi_min=40, qi_max=110)):
        kw = dict(device_params=1, check_ssim=1, altref_range=3)
        kw.update(cfg)
        a = api.NativeDriver(seq.W, seq.H, **kw)
        want = []
        for t in range(N):
            a.encode_frame_device(*ptrs[(2 + t) % ND])
            want.append(a.get_frame())
        a.resolve()
        b = api.NativeDriver(seq.W, seq.H, **kw)
        b.hip.reserve_frame_path_dense()
        got, keys = b.encode_video_device(N, ptrs, start=2)
        assert got == want, [i for i, (x, y) in enumerate(zip(got, want)) if x != y][:4]
        sa, sb = a.stats(), b.stats()
        assert (sa.key_frames, sa.inter_frames, sa.redone_as_key) == (sb.key_frames, sb.inter_frames, sb.redone_as_key) and keys == sa.key_frames
        for p_, q_ in zip(a.hip.download_last(), b.hip.download_last()):
            assert np.array_equal(p_, q_)
        # ... and with out = NULL (no frames out, no interpreter in the loop): the same video, views into a caller's buffer the same bytes
        c = api.NativeDriver(seq.W, seq.H, **kw)
        c.encode_video_device_no_frames(N, ptrs, start=2)
        sc = c.stats()
        assert (sa.key_frames, sa.inter_frames, sa.redone_as_key) == (sc.key_frames, sc.inter_frames, sc.redone_as_key)
        for p_, q_ in zip(a.hip.download_last(), c.hip.download_last()):
            assert np.array_equal(p_, q_)
        c.hip.reserve_frame_path_dense()
        buf = c.video_out_buffer(3)
        more, _ = c.encode_video_device(3, ptrs, start=2 + N, out=buf, views=True)
        for t in range(3):
            a.encode_frame_device(*ptrs[(2 + N + t) % ND])
            assert bytes(more[t]) == a.get_frame()
        a.close(); b.close(); c.close()
