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
    cfg = {k: v for k, v in cfg.items() if k != "host_bitstream"}
    ora = Oracle(W, H, cfg.get("ssim_target", -1.0))
    do = InterPathDriver(ora, W, H, gop_size=cfg.get("gop_size", 150), altref_range=cfg.get("altref_range", 5),
                         qi_min=cfg.get("qi_min", 0), qi_max=cfg.get("qi_max", 48), ssim_target=cfg.get("ssim_target", -1.0))
    stream = []
    for t, (y, u, v) in enumerate(frames):
        drv.encode_frame_host(y, u, v)     # "inter frame" is provisional until check_SSIM's verdict is in (get_frame waits for it)
        got = drv.get_frame()
        was_key = drv.resolve()
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


@pytest.mark.parametrize("P,host_bitstream", [(1, 0), (4, 0), (8, 0), (2, 1)])
def test_bitstream_of_a_gop_with_golden_and_altref(P, host_bitstream):
    W, H = 320, 192
    s = SynthSequence(W, H, seed=51)
    stream, st, _ = run_sequence(W, H, [s.frame(t) for t in range(13)], P=P, gop_size=12, altref_range=5, host_bitstream=host_bitstream)
    assert st.key_frames == 2 and st.inter_frames == 11          # a second key frame at the GOP boundary
    assert stream[0][3:6] == b"\x9d\x01\x2a" and stream[12][3:6] == b"\x9d\x01\x2a"
    assert all(f[0] & 1 for f in stream[1:12])                    # inter frames
    # the container around it
    ivf = bitstream.ivf_file_header(W, H, 30, 1, len(stream)) + b"".join(bitstream.ivf_frame_header(len(f), i) + f for i, f in enumerate(stream))
    assert ivf[:4] == b"DKIF" and len(ivf) == 32 + sum(12 + len(f) for f in stream)


@pytest.mark.parametrize("device_params,host_bitstream", [(0, 0), (1, 0), (0, 1)])
def test_negative_sharpness_of_the_overflowed_strength_accumulator(device_params, host_bitstream):
    """get_loopfilter_strength's second `int` accumulator (vp8enc.cpp:112-126) overflows on a large frame of noise and leaves a
    NEGATIVE sharpness; the reference writes its low three bits into every frame header and derives the interior limits from it.
    (Found by the fuzz run once it drew 1080p: the header took "negative" for "still on the device".)"""
    W, H = 1024, 512
    rng = np.random.default_rng(9559)
    frames = [(rng.integers(0, 256, (H, W)).astype(np.uint8), rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8),
               rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8)) for _ in range(3)]
    red, sharp = api.loopfilter_strength(frames[0][0])
    assert sharp < 0, sharp
    run_sequence(W, H, frames, P=2, qi_min=27, qi_max=50, device_params=device_params, host_bitstream=host_bitstream)


@pytest.mark.parametrize("host_bitstream", [0, 1])
def test_bitstream_with_intra_fallback_and_scene_cut(host_bitstream):
    W, H = 320, 192
    a, b = SynthSequence(W, H, seed=41), SynthSequence(W, H, seed=97)
    frames = [a.frame(t) for t in range(4)] + [b.frame(t) for t in range(3)]
    _, st, do = run_sequence(W, H, frames, P=2, qi_min=50, qi_max=110, ssim_target=0.90, host_bitstream=host_bitstream)
    assert st.redone_as_key >= 1 and do.redone_as_key == st.redone_as_key


def test_bitstream_from_device_resident_frames_and_device_parameters():
    """vp8drv_encode_frame_device: planes already in HBM, segment data computed on the device."""
    W, H = 176, 144
    s = SynthSequence(W, H, seed=61)
    drv = api.NativeDriver(W, H, check_ssim=1, device_params=1)
    ora = Oracle(W, H)
    do = InterPathDriver(ora, W, H)
    for t in range(4):
        y, u, v = s.frame(t)
        d = [api.to_device(p) for p in (y, u, v)]
        api.device_synchronize()
        drv.encode_frame_device(*(x.data_ptr() for x in d))
        got = drv.get_frame()
        out = do.encode_frame(y, u, v)
        assert got == expected_frame(W, H, do.last_key if out is None else out, out is None, 1), t
    drv.close()
    ora.close()


# ---- the first partition coded on the device (vp8hip_encode_header, kernels_hdr.hip) ------------------------------------
import ctypes as C
import glob
import os

from bitstream_cases import default_sd, random_inter_case, ref_encode_header, ref_header_lib

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "bitstream", "*.npz")))


def device_header(W, H, flags, sd, c, sharpness=0, partitions_log2=0, dst=None):
    """Drive the device coder with explicit inputs (hidden test hook vp8hip_debug_upload_header_inputs)."""
    hip = api.Vp8Hip(W, H)
    lib = hip.lib
    lib.vp8hip_debug_upload_header_inputs.argtypes = [C.c_void_p] * 11
    keep = []

    def ptr(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data

    key = flags[0] == 1
    rc = lib.vp8hip_debug_upload_header_inputs(hip.h, ptr(c["seg"], np.int32), ptr(c["nz"], np.int32), ptr(None if key else c["ref_frame"], np.int32),
                                               ptr(None if key else c["parts"], np.int32), ptr(None if key else c["vectors"], np.int16),
                                               ptr(c.get("is_inter"), np.int32), ptr(c.get("modes"), np.int32), ptr(c["probs"], np.uint32),
                                               ptr(c["denom"], np.uint32), ptr(np.asarray(sd).reshape(-1), np.int32))
    assert rc == 0
    use_intra = (not key) and c.get("is_inter") is not None
    out = hip.encode_header(flags[0], flags[1], flags[2], sharpness=sharpness, partitions_log2=partitions_log2, use_intra_info=use_intra,
                            width=(dst or (0, 0))[0], height=(dst or (0, 0))[1])
    hip.close()
    return out


@pytest.mark.parametrize("mbw,mbh,seed,kw", [(4, 3, 1, {}), (11, 9, 2, {}), (22, 18, 3, dict(long_mv=0.5)), (7, 5, 4, dict(split=1.0)),
                                             (7, 5, 5, dict(split=0.0, zero=0.6)), (9, 6, 6, dict(intra=0.2)), (9, 6, 7, dict(intra=0.02)),
                                             (5, 4, 8, dict(intra=1.0)), (30, 17, 9, dict(copy_neighbour=0.8)), (1, 1, 10, {}), (1, 6, 11, {}), (6, 1, 12, {}),
                                             (120, 68, 13, {}), (240, 135, 14, dict(long_mv=0.3))])
def test_device_header_matches_host_coder_on_stress_inputs(mbw, mbh, seed, kw):
    c = random_inter_case(mbw, mbh, seed, **kw)
    # the device derives skip_prob and `replaced` itself: they must come out as the host computes them
    for flags, sharp, plog in (((0, 0, 0), 0, 0), ((0, 0, 1), 7, 3)):
        host, _ = bitstream.encode_header(mbw * 16, mbh * 16, flags, default_sd(), c["seg"], c["nz"], c["probs"], c["denom"], c["skip_prob"],
                                          ref_frame=c["ref_frame"], parts=c["parts"], vectors=c["vectors"], is_inter=c["is_inter"], modes=c["modes"],
                                          replaced=c["replaced"], sharpness=sharp, partitions_log2=plog)
        dev = device_header(mbw * 16, mbh * 16, flags, default_sd(), c, sharpness=sharp, partitions_log2=plog)
        assert len(dev) == len(host), (flags, len(dev), len(host))
        assert np.array_equal(dev, host), (flags, np.nonzero(dev != host)[0][:8], len(host))


@pytest.mark.parametrize("mbw,mbh,seed", [(1, 1, 1), (4, 3, 2), (11, 9, 3), (120, 68, 4)])
def test_device_key_frame_header_matches_host_coder(mbw, mbh, seed):
    c = random_inter_case(mbw, mbh, seed)
    c.update(is_inter=None, replaced=0)
    host, _ = bitstream.encode_header(mbw * 16, mbh * 16, (1, 1, 1), default_sd(True), c["seg"], c["nz"], c["probs"], c["denom"], c["skip_prob"],
                                      modes=c["modes"], sharpness=3, dst=(mbw * 16 - 5, mbh * 16 - 2))
    dev = device_header(mbw * 16, mbh * 16, (1, 1, 1), default_sd(True), c, sharpness=3, dst=(mbw * 16 - 5, mbh * 16 - 2))
    assert np.array_equal(dev, host), np.nonzero(dev[:min(len(dev), len(host))] != host[:min(len(dev), len(host))])[0][:8]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_device_header_matches_reference_golden_vectors(path):
    z = np.load(path)
    c = {k: (np.ascontiguousarray(z[k]) if k in z.files else None) for k in ("seg", "nz", "ref_frame", "parts", "vectors", "is_inter", "modes", "probs", "denom")}
    flags = tuple(int(x) for x in z["flags"])
    dev = device_header(int(z["W"]), int(z["H"]), flags, z["sd"], c, sharpness=int(z["sharpness"]), partitions_log2=int(z["partitions_log2"]))
    assert np.array_equal(dev, z["header"]), np.nonzero(dev[:min(len(dev), len(z["header"]))] != z["header"][:min(len(dev), len(z["header"]))])[0][:8]


def test_sharded_encoder_writes_the_reference_ivf(tmp_path):
    """gop_shard (closed-GOP chunks, one encoder per chunk) with the native encoder on the GPU against the same chunks
    on the CPU oracle: the two .ivf files are byte-identical."""
    from test_gop_shard import FRAMES, GOP, OracleEncoder
    from vp8oclenc_amd import gop_shard
    W, H = 64, 48
    seq = SynthSequence(W, H, seed=5)
    chunks = gop_shard.gop_chunks(FRAMES, GOP)
    dev = gop_shard.encode_chunks_frames(lambda: gop_shard.NativeEncoder(seq.W, seq.H, num_partitions=2, check_ssim=1), seq, chunks)
    cpu = gop_shard.encode_chunks_frames(lambda: OracleEncoder(seq.W, seq.H), seq, chunks)
    a, b = str(tmp_path / "dev.ivf"), str(tmp_path / "cpu.ivf")
    gop_shard.write_ivf(a, gop_shard.gather_frames(dev, FRAMES), seq.W, seq.H)
    gop_shard.write_ivf(b, gop_shard.gather_frames(cpu, FRAMES), seq.W, seq.H)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_get_frame_in_two_halves_across_chunks_equals_get_frame():
    """vp8drv_get_frame_begin/_end: one host thread keeps several GOP chunks in flight; same bytes as the blocking call,
    and the call order is enforced."""
    W, H, P = 320, 192, 4
    seqs = [SynthSequence(W, H, seed=s) for s in (61, 62, 63)]
    blocking = []
    for s in seqs:
        d = api.NativeDriver(W, H, num_partitions=P, check_ssim=1, gop_size=4)
        frames = []
        for t in range(6):
            d.encode_frame_host(*s.frame(t))
            frames.append(d.get_frame())
        blocking.append(frames)
        d.close()
    drvs = [api.NativeDriver(W, H, num_partitions=P, check_ssim=1, gop_size=4) for _ in seqs]
    with pytest.raises(api.Vp8HipError):
        drvs[0].get_frame_begin()                     # nothing coded yet
    for t in range(6):
        for d, s in zip(drvs, seqs):
            d.encode_frame_host(*s.frame(t))
            d.get_frame_begin()
        with pytest.raises(api.Vp8HipError):
            drvs[0].get_frame_begin()                 # one frame pending already
        for k, d in enumerate(drvs):
            assert d.get_frame_end() == blocking[k][t], (k, t)
        with pytest.raises(api.Vp8HipError):
            drvs[0].get_frame_end()                   # nothing pending
    host = api.NativeDriver(W, H, num_partitions=P, host_bitstream=1)
    host.encode_frame_host(*seqs[0].frame(0))
    with pytest.raises(api.Vp8HipError):
        host.get_frame_begin()                        # the host coder has no asynchronous half
    host.close()
    for d in drvs:
        d.close()


def test_loop_filter_on_its_own_stream_changes_nothing():
    """overlap_filter: the filter runs beside the frame's entropy stage; frames and reconstructions stay the same, with and
    without check_SSIM's fallback, across key frames."""
    W, H = 320, 192
    a, b = SynthSequence(W, H, seed=71), SynthSequence(W, H, seed=72)
    frames = [a.frame(t) for t in range(5)] + [b.frame(t) for t in range(4)]
    outs = []
    for overlap in (0, 1):
        d = api.NativeDriver(W, H, num_partitions=2, check_ssim=1, gop_size=6, qi_min=40, qi_max=100, ssim_target=0.9, overlap_filter=overlap)
        got = []
        for y, u, v in frames:
            d.encode_frame_host(y, u, v)
            f = d.get_frame()
            rec = d.hip.download_last()
            got.append((f, [p.copy() for p in rec]))
        outs.append(got)
        d.close()
    for t, ((f0, r0), (f1, r1)) in enumerate(zip(*outs)):
        assert f0 == f1, t
        for p0, p1 in zip(r0, r1):
            assert np.array_equal(p0, p1), t


@pytest.mark.parametrize("frames_out", [True, False])
def test_overlap_three_references_device_frames(frames_out):
    """overlap_filter on the path the single_stream leg of bench.py times: frames already in device memory, parameters scanned
    on the device, LAST + GOLDEN + ALTREF.  The context trades streams at every filter (the next frame's pack, parameter scan
    and GOLDEN / ALTREF searches run beside it, the LAST search behind it on the filter's own stream): same frames, same
    reconstructions, with the entropy stage beside the filter or without one."""
    W, H = 640, 352
    s = SynthSequence(W, H, seed=73)
    dev = [tuple(api.to_device(p) for p in s.frame(t)) for t in range(16)]
    outs = []
    for overlap in (0, 1):
        d = api.NativeDriver(s.W, s.H, num_partitions=4, check_ssim=0, gop_size=150, altref_range=3, device_params=1, overlap_filter=overlap)
        got, refs = [], set()
        for t, f in enumerate(dev):
            d.encode_frame_device(*(p.data_ptr() for p in f))
            st = d.stats()
            refs.add((st.last_use_golden, st.last_use_altref))
            if frames_out:
                got.append(d.get_frame())
            if t % 5 == 4 or not frames_out:
                got.append([p.copy() for p in d.hip.download_last()])
        assert (1, 1) in refs
        outs.append(got)
        d.close()
    for t, (g0, g1) in enumerate(zip(*outs)):
        if isinstance(g0, bytes):
            assert g0 == g1, t
        else:
            for p0, p1 in zip(g0, g1):
                assert np.array_equal(p0, p1), t


def test_frames_beyond_a_million_blocks():
    """4096x2736 = 1.09 million 4x4 block slots: the frame path's prefix sums have no size limit (the step-by-step
    vp8hip_encode_coefficients had a 1 Mi one until its top-level scan learnt to walk: tests/test_gpu_entropy.py at 480x270 macroblocks); key frame + inter frame, 8 partitions, byte-exact.  (The format
    itself stops a little further on: the frame tag holds the first partition's size in 19 bits, see the next test.)"""
    W, H = 4096, 2736
    s = SynthSequence(W, H, seed=5)
    stream, st, _ = run_sequence(W, H, [s.frame(t) for t in range(2)], P=8)
    assert st.key_frames == 1 and st.inter_frames == 1 and len(stream[0]) > 1 << 19


def test_first_partition_beyond_the_frame_tag_is_an_error():
    """A 7680x4320 key frame codes 129 600 x 16 sub-block modes: its first partition passes 512 KiB, which the frame tag's
    19-bit size field cannot say (RFC 6386 9.1).  The reference writes the truncated size and the frame cannot be decoded;
    here the call fails with VP8HIP_ERR_FORMAT instead of returning such a frame (both coders and the inter frames
    of that size: tests/test_gpu_soak.py::test_8k_frames_do_not_fit_the_format_and_say_so)."""
    W, H = 7680, 4320
    s = SynthSequence(W, H, seed=5)
    d = api.NativeDriver(W, H, num_partitions=8)
    assert d.encode_frame_host(*s.frame(0))
    try:
        frame = d.get_frame()
    except api.Vp8HipError as e:
        assert "(-8)" in str(e) and "19 bits" in str(e)
    else:   # content whose key frame happens to fit: then the tag must tell the truth
        first = (frame[0] | frame[1] << 8 | frame[2] << 16) >> 5
        assert first < (1 << 19) and len(frame) > first
    d.close()


def test_entropy_stage_variants_emit_the_same_bytes():
    """The A/B switches of the frame path (step-by-step bool-string kernels, device buffer + copy instead of writing into
    pinned host memory) are read once per process: run each in its own interpreter and compare the stream's hash."""
    import subprocess
    import sys
    prog = (
        "import hashlib, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from vp8oclenc_amd import api\n"
        "from vp8oclenc_amd.synth import SynthSequence\n"
        "s = SynthSequence(320, 192, seed=81); d = api.NativeDriver(320, 192, num_partitions=4, check_ssim=1, gop_size=4)\n"
        "h = hashlib.sha256()\n"
        "for t in range(7):\n"
        "    d.encode_frame_host(*s.frame(t)); h.update(d.get_frame())\n"
        "print(h.hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for extra in ({}, {"VP8HIP_ENT_STEPWISE": "1"}, {"VP8HIP_FRAME_ZEROCOPY": "0"}, {"VP8HIP_ENT_STEPWISE": "1", "VP8HIP_FRAME_ZEROCOPY": "0"}):
        out = subprocess.run([sys.executable, "-c", prog], cwd=root, env={**os.environ, **extra}, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append(out.stdout.strip().splitlines()[-1])
    assert len(set(digests)) == 1, digests
