"""An INDEPENDENT VP8 decoder for key frames, as test infrastructure: libwebp (WebP lossy = one VP8 key frame in a RIFF
container), loaded from where the image has it -- the system's libwebp.so or the copy Pillow bundles.  WebPDecodeYUV hands
back the decoder's own Y, U, V planes, loop-filtered, before any colour conversion: what an encoder's reconstruction of the
same frame must equal sample for sample if the stream says what the encoder meant.  (Inter frames have no decoder in the
image: their bytes are pinned to the reference's own encode_header and coefficient coder instead.)"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
import struct

import numpy as np

_lib = None


def libwebp():
    global _lib
    if _lib is not None:
        return _lib or None
    cands = []
    name = ctypes.util.find_library("webp")
    if name:
        cands.append(name)
    cands += sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libwebp.so*"))
    try:
        import PIL
        cands += sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), "pillow.libs", "libwebp-*.so*")))
    except Exception:
        pass
    for c in cands:
        try:
            lib = C.CDLL(c)
            lib.WebPDecodeYUV.restype = C.POINTER(C.c_uint8)
            lib.WebPDecodeYUV.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint8)),
                                          C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
            lib.WebPFree.argtypes = [C.c_void_p]
            lib.WebPFree.restype = None
            _lib = lib
            return lib
        except (OSError, AttributeError):
            continue
    _lib = False
    return None


def as_webp(vp8_key_frame: bytes) -> bytes:
    """RIFF / WEBP / 'VP8 ' around the frame exactly as the encoder emitted it (frame tag, start code, dimensions, partitions)"""
    data = vp8_key_frame + (b"\0" if len(vp8_key_frame) & 1 else b"")
    chunk = b"VP8 " + struct.pack("<I", len(vp8_key_frame)) + data
    return b"RIFF" + struct.pack("<I", 4 + len(chunk)) + b"WEBP" + chunk


def decode_key_frame(vp8_key_frame: bytes):
    """(Y, U, V) uint8 planes as libwebp decodes the frame, or raises"""
    lib = libwebp()
    if lib is None:
        raise RuntimeError("no libwebp in this image")
    blob = as_webp(vp8_key_frame)
    w, h, ys, uvs = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    u, v = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint8)()
    y = lib.WebPDecodeYUV(blob, len(blob), C.byref(w), C.byref(h), C.byref(u), C.byref(v), C.byref(ys), C.byref(uvs))
    if not y:
        raise ValueError("libwebp rejects the frame")
    try:
        W, H = w.value, h.value
        cw, ch = (W + 1) // 2, (H + 1) // 2
        Y = np.ctypeslib.as_array(y, shape=(H * ys.value,))[: H * ys.value].reshape(H, ys.value)[:, :W].copy()
        U = np.ctypeslib.as_array(u, shape=(ch * uvs.value,)).reshape(ch, uvs.value)[:, :cw].copy()
        V = np.ctypeslib.as_array(v, shape=(ch * uvs.value,)).reshape(ch, uvs.value)[:, :cw].copy()
    finally:
        lib.WebPFree(y)
    return Y, U, V
