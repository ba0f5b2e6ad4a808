"""YUV4MPEG2, the reference's input format (OpenYUV420FileAndParseHeader, init.h:1610-1737; get_yuv420_frame,
encIO.h:203-254), read the way the reference reads it: the header by the native restatement (vp8host_y4m_parse_header),
frames as tight I420 of the header's size, each followed by the next frame's 6-byte FRAME line whose bytes 0 and 4 the
reference checks."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api


def parse_header(data: bytes):
    """(width, height, framerate, offset of the first frame's samples); raises on what the reference refuses"""
    lib = api.load_library()
    lib.vp8host_y4m_parse_header.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]
    w, h, f, off = C.c_int32(), C.c_int32(), C.c_int32(), C.c_size_t()
    if lib.vp8host_y4m_parse_header(data, len(data), C.byref(w), C.byref(h), C.byref(f), C.byref(off)) != 0:
        raise ValueError("not a YUV4MPEG2 stream the reference accepts (magic word, W / H / F tags ended by spaces, a plain FRAME line)")
    return w.value, h.value, f.value, off.value


class Y4mFile:
    """frames of a .y4m file: .W, .H (the SOURCE size: hand them to an encoder created for the padded size with
    src_width / src_height), .framerate, .n, .frame(t) -> (y, u, v)"""

    def __init__(self, path: str):
        self.m = np.memmap(path, np.uint8, "r")
        self.W, self.H, self.framerate, self.first = parse_header(bytes(self.m[:4096]))
        if self.W % 2 or self.H % 2:
            raise ValueError("odd frame sizes are not I420 the reference can read")
        self.fsz = self.W * self.H * 3 // 2
        self.n = (len(self.m) - self.first + 6) // (self.fsz + 6)
        lib = api.load_library()
        lib.vp8host_y4m_frame_marker_ok.argtypes = [C.c_char_p]
        self._ok = lib.vp8host_y4m_frame_marker_ok

    def frame(self, t: int):
        if not 0 <= t < self.n:
            raise IndexError(f"frame {t} of {self.n}")
        a = self.first + t * (self.fsz + 6)
        if t > 0 and not self._ok(bytes(self.m[a - 6:a])):
            raise ValueError(f"broken stream before frame {t}")          # encIO.h:245-248
        b = self.m[a:a + self.fsz]
        W, H = self.W, self.H
        return (np.ascontiguousarray(b[:W * H].reshape(H, W)), np.ascontiguousarray(b[W * H:W * H * 5 // 4].reshape(H // 2, W // 2)),
                np.ascontiguousarray(b[W * H * 5 // 4:].reshape(H // 2, W // 2)))


def write_y4m(path: str, frames, framerate: int = 30):
    """a .y4m as ffmpeg writes it (for tests and tools)"""
    frames = list(frames)
    H, W = frames[0][0].shape
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F{framerate}:1 Ip A1:1 C420jpeg XYSCSS=420JPEG\n".encode())
        for y, u, v in frames:
            f.write(b"FRAME\n")
            for p in (y, u, v):
                f.write(np.ascontiguousarray(p).tobytes())
