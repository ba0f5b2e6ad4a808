"""YUV4MPEG2, the reference's input format: the native restatement of its header parser (vp8host_y4m_parse_header) against the
reference's own OpenYUV420FileAndParseHeader (oracle/_ref) on headers as ffmpeg and mjpegtools write them, on tag orders the
reference happens to accept and on everything it refuses; and the frame reader on a file written here."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle_lib import REF_HOST_SO
from vp8oclenc_amd import y4m

FRAME = b"FRAME\n" + bytes(range(48))
HEADERS = [
    b"YUV4MPEG2 W352 H288 F30:1 Ip A128:117 C420jpeg XYSCSS=420JPEG\n",
    b"YUV4MPEG2 W1920 H1080 F30000:1001 Ip A1:1 C420mpeg2\n",
    b"YUV4MPEG2 H144 W176 F25:1 Ip\n",                      # any order
    b"YUV4MPEG2 W16 H16 F15:2 \n",                           # 7.5 rounds to 8
    b"YUV4MPEG2 W640 H360 F24:1 Ip A1:1\nFRAME\n",          # (the header ends at the first FRAME line)
    b"YUV4MPEG2 C420 W320 H240 F60:1 Ip\n",                 # tags without W/H/F letters before the three
    b"YUV4MPEG2 W352 H288 Ip F30:1 \n",
    b"YUV4MPEG2 W352 H288 W17 F30:1 \n",                    # three tags are read whatever they are: W accumulates, F is never seen
    b"YUV4MPEG  W352 H288 F30:1 \n",                         # refused: magic word
    b"YUV4MPEG2 W352 H288 F30:1 Ip\nFRAME Ip\n",            # refused: FRAME line with parameters
    b"YUV4MPEG2 W352 H288 F30:1 Ip\nFRAMX\nFRAME\n",        # a false start is skipped
    b"YUV4MPEG2 W352",                                       # refused: ends inside a tag
    b"YUV4MPEG2 W0 H0 F30:1 \n",                             # refused: no size
]


def in_format(w, h):
    """the one place where the restatement parts from the reference's parser: a size that does not fit the 14 bits VP8 has for it
    (RFC 6386 9.1) is refused; the reference goes on with it (W352 ... W17 accumulates to 35217)"""
    return 1 <= w <= 16383 and 1 <= h <= 16383


@pytest.mark.parametrize("i", range(len(HEADERS)))
def test_header_restatement(i):
    data = HEADERS[i] + (b"" if HEADERS[i].endswith(b"FRAME\n") or i in (11,) else FRAME)
    try:
        got = y4m.parse_header(data)
    except ValueError:
        got = None
    expect = {0: (352, 288, 30), 1: (1920, 1080, 30), 2: (176, 144, 25), 3: (16, 16, 8), 4: (640, 360, 24), 5: (320, 240, 60),
              6: (352, 288, 30), 7: None, 8: None, 9: None, 10: (352, 288, 30), 11: None, 12: None}
    if i in expect:
        assert (got[:3] if got else None) == expect[i]
        if got:
            assert data[got[3] - 6:got[3]] == b"FRAME\n"
    if not os.path.exists(REF_HOST_SO):
        return
    ref = C.CDLL(REF_HOST_SO)          # the reference's own function on the same bytes
    ref.ref_parse_y4m_header.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "in.y4m")
        open(p, "wb").write(data)
        w, h, f, off = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        rc = ref.ref_parse_y4m_header(p.encode(), os.path.join(d, "out.ivf").encode(), C.byref(w), C.byref(h), C.byref(f), C.byref(off))
    if rc != 0 or not in_format(w.value, h.value):
        assert got is None, (i, got)
    else:
        assert got == (w.value, h.value, f.value, off.value), (i, got, (w.value, h.value, f.value, off.value))


def test_frames_come_back(tmp_path):
    rng = np.random.default_rng(4)
    frames = [(rng.integers(0, 256, (36, 50)).astype(np.uint8), rng.integers(0, 256, (18, 25)).astype(np.uint8), rng.integers(0, 256, (18, 25)).astype(np.uint8))
              for _ in range(5)]
    p = str(tmp_path / "a.y4m")
    y4m.write_y4m(p, frames, framerate=24)
    f = y4m.Y4mFile(p)
    assert (f.W, f.H, f.framerate, f.n) == (50, 36, 24, 5)
    for t, fr in enumerate(frames):
        for a, b in zip(f.frame(t), fr):
            assert np.array_equal(a, b)
    raw = bytearray(open(p, "rb").read())
    raw[f.first + f.fsz + 4] = ord("X")           # FRAMX before frame 1: "broken stream" (encIO.h:245-248)
    open(p, "wb").write(bytes(raw))
    g = y4m.Y4mFile(p)
    g.frame(0)
    with pytest.raises(ValueError, match="broken stream"):
        g.frame(1)


@pytest.mark.skipif(not os.path.exists(REF_HOST_SO), reason="oracle/_ref/libvp8refhost.so not built (no /root/reference here)")
def test_random_headers_against_the_reference_function(tmp_path):
    """400 random headers -- tags in any order, repeated, with letters inside numbers, FRAME look-alikes, truncations -- through the
    reference's own parser and the restatement: the same verdict and the same numbers (kept below the reference's 128-byte header
    array and away from a zero denominator, which it divides by)"""
    ref = C.CDLL(REF_HOST_SO)
    ref.ref_parse_y4m_header.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    rng = np.random.default_rng(11)
    tags = [lambda: b"W%d" % rng.integers(0, 5000), lambda: b"H%d" % rng.integers(0, 5000), lambda: b"F%d:%d" % (rng.integers(0, 99999), rng.integers(1, 2000)),
            lambda: b"Ip", lambda: b"A%d:%d" % (rng.integers(1, 200), rng.integers(1, 200)), lambda: b"C420jpeg", lambda: b"XYSCSS=420", lambda: b"W1x7",
            lambda: b"H", lambda: b"It", lambda: b"C444"]
    p, scratch = str(tmp_path / "h.y4m"), str(tmp_path / "o.ivf")
    agree_ok = agree_bad = 0
    for _ in range(400):
        head = b"YUV4MPEG2 " if rng.random() < 0.95 else b"YUV4MPEG1 "
        picks = [int(i) for i in rng.integers(0, len(tags), int(rng.integers(1, 7)))]
        if rng.random() < 0.6:       # the three tags the reference needs, in a random order, before whatever else
            picks = [int(i) for i in rng.permutation(3)] + [i for i in picks if i > 2]
        body = b" ".join(tags[i]() for i in picks)
        tail = [b" \nFRAME\n", b"\nFRAME\n", b" \nFRAMX\nFRAME\n", b" \nFRAME Ip\n", b" \nFRA", b" \nFFRAME\n"][int(rng.integers(0, 6))]
        data = (head + body + tail)[:110] + bytes(range(32))
        open(p, "wb").write(data)
        w, h, f, off = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        rc = ref.ref_parse_y4m_header(p.encode(), scratch.encode(), C.byref(w), C.byref(h), C.byref(f), C.byref(off))
        try:
            got = y4m.parse_header(data)
        except ValueError:
            got = None
        if rc != 0 or not in_format(w.value, h.value):
            assert got is None, (data, got)
            agree_bad += 1
        else:
            assert got == (w.value, h.value, f.value, off.value), (data, got, (w.value, h.value, f.value, off.value))
            agree_ok += 1
    assert agree_ok > 100 and agree_bad > 50, (agree_ok, agree_bad)
