"""First partition (frame header, per-macroblock modes and motion vectors), frame assembly and IVF
(vp8oclenc_amd/csrc/vp8_bitstream.cpp, include/vp8hip_bitstream.h) against the reference's own encode_header
compiled from /root/reference (oracle/_ref/libvp8refhost.so) and against committed golden vectors made by it.
Host code on both sides; byte-exact."""
import glob
import os

import numpy as np
import pytest

from bitstream_cases import default_sd, random_inter_case, ref_encode_header, ref_header_lib
from vp8oclenc_amd import api, bitstream

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "bitstream", "*.npz")))
needs_ref = pytest.mark.skipif(ref_header_lib() is None, reason="oracle/_ref/libvp8refhost.so not built (no /root/reference here)")


def both(W, H, flags, sd, c, **kw):
    args = dict(ref_frame=c["ref_frame"], parts=c["parts"], vectors=c["vectors"], is_inter=c["is_inter"], modes=c["modes"],
                replaced=c["replaced"])
    args.update(kw)
    mine, mvp = bitstream.encode_header(W, H, flags, sd, c["seg"], c["nz"], c["probs"], c["denom"], c["skip_prob"], **args)
    ref = ref_encode_header(W, H, flags, sd, c["seg"], c["nz"], c["probs"], c["denom"], c["skip_prob"], **args)
    return mine, ref, mvp


@needs_ref
@pytest.mark.parametrize("mbw,mbh,seed,kw", [(4, 3, 1, {}), (11, 9, 2, {}), (22, 18, 3, dict(long_mv=0.5)), (7, 5, 4, dict(split=1.0)),
                                             (7, 5, 5, dict(split=0.0, zero=0.6)), (9, 6, 6, dict(intra=0.2)), (9, 6, 7, dict(intra=0.02)),
                                             (5, 4, 8, dict(intra=1.0)), (30, 17, 9, dict(copy_neighbour=0.8)), (1, 1, 10, {}), (1, 6, 11, {}), (6, 1, 12, {})])
def test_inter_frame_header_matches_reference_code(mbw, mbh, seed, kw):
    c = random_inter_case(mbw, mbh, seed, **kw)
    for flags, sharp, plog in (((0, 0, 0), 0, 0), ((0, 1, 0), 3, 1), ((0, 0, 1), 7, 2), ((0, 1, 1), 5, 3)):
        mine, ref, mvp = both(mbw * 16, mbh * 16, flags, default_sd(), c, sharpness=sharp, partitions_log2=plog)
        assert len(mine) == len(ref), (flags, len(mine), len(ref))
        assert np.array_equal(mine, ref), (flags, np.nonzero(mine != ref)[0][:8])
        assert (mvp >= 2).all() and (mvp <= 254).all() and (mvp % 2 == 0).all()


@needs_ref
@pytest.mark.parametrize("mbw,mbh,seed", [(1, 1, 1), (4, 3, 2), (11, 9, 3), (40, 23, 4)])
def test_key_frame_header_matches_reference_code(mbw, mbh, seed):
    c = random_inter_case(mbw, mbh, seed)
    c.update(is_inter=None, replaced=0)
    for dst in (None, (mbw * 16 - 3, mbh * 16 - 7)):
        mine, ref, _ = both(mbw * 16, mbh * 16, (1, 1, 1), default_sd(True), c, dst=dst, ref_frame=None, parts=None, vectors=None)
        assert np.array_equal(mine, ref), np.nonzero(mine[:min(len(mine), len(ref))] != ref[:min(len(mine), len(ref))])[0][:8]
        assert mine[3:6].tobytes() == b"\x9d\x01\x2a"


@needs_ref
def test_headers_of_a_real_sequence_match_reference_code():
    """Key frame + inter frames (golden / altref in use, check_SSIM replacing macroblocks) from the CPU oracle loop."""
    from entropy_cases import nz_counts, run_stage
    from oracle_lib import Oracle
    from vp8oclenc_amd.driver import InterPathDriver
    from vp8oclenc_amd.synth import SynthSequence
    W, H, target = 176, 144, 0.93
    a, b = SynthSequence(W, H, seed=5), SynthSequence(W, H, seed=77)
    frames = [a.frame(t) for t in range(7)] + [b.frame(0)]
    ora = Oracle(W, H, target)
    drv = InterPathDriver(ora, W, H, qi_min=40, qi_max=100, ssim_target=target)
    seen_replaced = 0
    for t, (y, u, v) in enumerate(frames):
        out = drv.encode_frame(y, u, v)
        key = out is None
        r = drv.last_key if key else out
        g = drv.gop.s
        flags = (int(key), int(g.prev_is_golden) if False else int(key), int(key))   # replaced below for inter frames
        coeffs, parts = r["MB_coeffs"], r["MB_parts"]
        nz = nz_counts(coeffs, parts)
        st = run_stage(Oracle.stages(), np.ascontiguousarray(coeffs), np.ascontiguousarray(parts), nz, W // 16, H // 16, 1)
        probs = bitstream.default_probs(st["probs"], st["denom"][:1056])
        sd = np.asarray(r["segments"]).reshape(4, 11)
        c = dict(seg=r["MB_segment_id"], nz=nz, probs=probs, denom=st["denom"][:1056], skip_prob=api.skip_prob(nz),
                 ref_frame=None if key else r["MB_reference_frame"], parts=None if key else parts,
                 vectors=None if key else r["MB_vectors"], is_inter=None if key else r["is_inter"], modes=r["modes"],
                 replaced=0 if key else int(r["replaced"]))
        if not key:
            flags = (0, 0, int(out.get("was_altref", 0)))
            seen_replaced += c["replaced"]
        mine, ref, _ = both(W, H, flags, sd, c, sharpness=int(sd[0, 9] < sd[0, 6]))
        assert np.array_equal(mine, ref), (t, len(mine), len(ref))
    ora.close()
    assert seen_replaced > 0


def test_default_probs_gather_and_ivf():
    probs, denom = np.arange(1056, dtype=np.uint32) % 200 + 1, (np.arange(1056) % 3).astype(np.uint32)
    p = bitstream.default_probs(probs, denom)
    assert np.array_equal(p[denom >= 2], probs[denom >= 2]) and (p[denom < 2] >= 1).all() and (p[denom < 2] <= 255).all()
    assert p[0] == 128 and p[33] == 253 and p[34] == 136       # RFC 6386 13.5: default_coeff_probs[0][0][0][0], [0][1][0][0..1]
    hdr = np.arange(7, dtype=np.uint8)
    parts = [np.full(5, 1, np.uint8), np.full(300, 2, np.uint8), np.full(70000, 3, np.uint8), np.full(2, 4, np.uint8)]
    fr = bitstream.gather_frame(hdr, parts)
    assert len(fr) == 7 + 9 + 5 + 300 + 70000 + 2
    assert fr[7:16].tolist() == [5, 0, 0, 300 & 255, 300 >> 8, 0, 70000 & 255, (70000 >> 8) & 255, 70000 >> 16]
    assert fr[16:21].tolist() == [1] * 5 and fr[-2:].tolist() == [4, 4]
    h = bitstream.ivf_file_header(1920, 1080, 30, 1, 300)
    assert h[:4] == b"DKIF" and h[8:12] == b"VP80" and h[4:8] == bytes([0, 0, 32, 0])
    assert int.from_bytes(h[12:14], "little") == 1920 and int.from_bytes(h[14:16], "little") == 1080
    assert int.from_bytes(h[16:20], "little") == 30 and int.from_bytes(h[20:24], "little") == 1 and int.from_bytes(h[24:28], "little") == 300
    fh = bitstream.ivf_frame_header(123456, 7)
    assert int.from_bytes(fh[:4], "little") == 123456 and int.from_bytes(fh[4:], "little") == 7


def test_bitstream_golden_fixtures_present():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_header_matches_reference_golden_vectors(path):
    z = np.load(path)
    opt = lambda k: np.ascontiguousarray(z[k]) if k in z.files else None
    mine, _ = bitstream.encode_header(int(z["W"]), int(z["H"]), tuple(int(x) for x in z["flags"]), z["sd"], z["seg"], z["nz"], z["probs"],
                                      z["denom"], int(z["skip_prob"]), ref_frame=opt("ref_frame"), parts=opt("parts"), vectors=opt("vectors"),
                                      is_inter=opt("is_inter"), modes=opt("modes"), replaced=int(z["replaced"]), sharpness=int(z["sharpness"]),
                                      partitions_log2=int(z["partitions_log2"]))
    assert np.array_equal(mine, z["header"]), np.nonzero(mine[:min(len(mine), len(z["header"]))] != z["header"][:min(len(mine), len(z["header"]))])[0][:8]


@needs_ref
def test_ivf_file_against_the_reference_writers(tmp_path):
    """write_output_header / write_output_file (encIO.h:32-139), the reference's own functions, against gop_shard.write_ivf
    (vp8bs_ivf_file_header / vp8bs_ivf_frame_header): the same file, byte for byte"""
    import ctypes as C
    from oracle_lib import REF_HOST_SO
    from vp8oclenc_amd import gop_shard
    rng = np.random.default_rng(8)
    frames = [rng.integers(0, 256, int(n)).astype(np.uint8).tobytes() for n in (10, 70000, 3, 513, 1 << 17)]
    ref = C.CDLL(REF_HOST_SO)
    ref.ref_write_ivf.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]
    arr = (C.c_char_p * len(frames))(*frames)
    sizes = (C.c_int32 * len(frames))(*[len(f) for f in frames])
    a, b = str(tmp_path / "ref.ivf"), str(tmp_path / "own.ivf")
    assert ref.ref_write_ivf(a.encode(), 1918, 1078, 25, len(frames), arr, sizes) == 0
    gop_shard.write_ivf(b, frames, 1918, 1078, framerate=25)
    assert open(a, "rb").read() == open(b, "rb").read()
