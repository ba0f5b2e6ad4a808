"""ctypes binding of include/vp8hip_bitstream.h: first partition (frame header, modes, motion vectors), frame
assembly and IVF -- the host half of the reference's entropy stage (src/entropy_host.cpp, src/encIO.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api


class Frame(C.Structure):
    """vp8bs_frame"""
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "mb_width", "mb_height", "is_key", "is_golden", "is_altref",
                                          "loop_filter_type", "loop_filter_sharpness", "partitions_log2", "skip_prob", "replaced")] + \
               [(n, C.c_void_p) for n in ("segments", "MB_segment_id", "MB_non_zero_coeffs", "MB_reference_frame", "MB_parts",
                                          "MB_vectors", "is_inter_mb", "modes", "new_probs", "new_probs_denom")]


def _lib():
    lib = api.load_library()
    if not getattr(lib, "_vp8bs_bound", False):
        lib.vp8bs_default_probs.argtypes = [C.c_void_p, C.c_void_p]
        lib.vp8bs_default_probs.restype = None
        lib.vp8bs_encode_header.argtypes = [C.POINTER(Frame), C.c_void_p, C.c_size_t, C.c_void_p]
        lib.vp8bs_encode_header.restype = C.c_size_t
        lib.vp8bs_gather_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.vp8bs_gather_frame.restype = C.c_size_t
        lib.vp8bs_ivf_file_header.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.vp8bs_ivf_file_header.restype = C.c_size_t
        lib.vp8bs_ivf_frame_header.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        lib.vp8bs_ivf_frame_header.restype = C.c_size_t
        lib._vp8bs_bound = True
    return lib


def default_probs(probs: np.ndarray, denom: np.ndarray) -> np.ndarray:
    """Contexts never seen take the default probability (vp8enc.cpp:69-76); returns the patched copy."""
    p = np.ascontiguousarray(probs, np.uint32).copy()
    d = np.ascontiguousarray(denom, np.uint32)
    _lib().vp8bs_default_probs(p.ctypes.data, d.ctypes.data)
    return p


def encode_header(width, height, flags, segments, seg, nz, probs, denom, skip_prob, ref_frame=None, parts=None, vectors=None,
                  is_inter=None, modes=None, replaced=0, loop_filter_type=0, sharpness=0, partitions_log2=0, dst=None):
    """encode_header (entropy_host.cpp:709-1256): returns (first partition bytes incl. the frame tag, mv probs[2][19])."""
    keep = []

    def ptr(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data

    mbs = (width // 16) * (height // 16)
    f = Frame(width=(dst or (width, height))[0], height=(dst or (width, height))[1], mb_width=width // 16, mb_height=height // 16,
              is_key=int(flags[0]), is_golden=int(flags[1]), is_altref=int(flags[2]), loop_filter_type=loop_filter_type,
              loop_filter_sharpness=sharpness, partitions_log2=partitions_log2, skip_prob=int(skip_prob), replaced=int(replaced),
              segments=ptr(np.asarray(segments).reshape(-1), np.int32), MB_segment_id=ptr(seg, np.int32),
              MB_non_zero_coeffs=ptr(nz, np.int32), MB_reference_frame=ptr(ref_frame, np.int32), MB_parts=ptr(parts, np.int32),
              MB_vectors=ptr(vectors, np.int16), is_inter_mb=ptr(is_inter, np.int32), modes=ptr(modes, np.int32),
              new_probs=ptr(probs, np.uint32), new_probs_denom=ptr(denom, np.uint32))
    cap = 4096 + mbs * 96
    out = np.zeros(cap, np.uint8)
    mvp = np.zeros((2, 19), np.uint8)
    n = _lib().vp8bs_encode_header(C.byref(f), out.ctypes.data, cap, mvp.ctypes.data)
    if n == C.c_size_t(-1).value:
        raise api.Vp8HipError("vp8bs_encode_header: first partition of 512 KB or more: the VP8 frame tag has 19 bits for its size")
    if n == 0:
        raise api.Vp8HipError("vp8bs_encode_header failed (bad arguments or the partition does not fit)")
    return out[:n].copy(), mvp


def gather_frame(header: np.ndarray, partitions) -> np.ndarray:
    """gather_frame (encIO.h:1-30): header + partition sizes + coefficient partitions."""
    P = len(partitions)
    step = max(len(p) for p in partitions) + 1
    buf = np.zeros(P * step, np.uint8)
    sizes = np.zeros(P, np.int32)
    for i, p in enumerate(partitions):
        buf[i * step: i * step + len(p)] = p
        sizes[i] = len(p)
    cap = len(header) + 3 * P + int(sizes.sum()) + 16
    frame = np.zeros(cap, np.uint8)
    frame[:len(header)] = header
    n = _lib().vp8bs_gather_frame(frame.ctypes.data, len(header), cap, P, buf.ctypes.data, step, sizes.ctypes.data)
    if n == 0:
        raise api.Vp8HipError("vp8bs_gather_frame failed")
    return frame[:n].copy()


def ivf_file_header(width, height, framerate, timescale, frame_count) -> bytes:
    out = np.zeros(32, np.uint8)
    _lib().vp8bs_ivf_file_header(out.ctypes.data, width, height, framerate, timescale, frame_count)
    return out.tobytes()


def ivf_frame_header(frame_size, timestamp) -> bytes:
    out = np.zeros(12, np.uint8)
    _lib().vp8bs_ivf_frame_header(out.ctypes.data, frame_size, timestamp)
    return out.tobytes()
