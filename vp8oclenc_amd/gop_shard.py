"""Closed-GOP sharding of a sequence over ranks (one process per GPU) -- SURVEY.md section 8e.

A key frame resets every reference (intra_part.h:1091-1098, inter_part.h:35-50), so GOPs are independent
units: rank r encodes chunks r, r+world, ... with the ordinary single-GPU pipeline and the outputs are
concatenated in frame order.  No collective sits on the data path; the process group -- vp8oclenc_amd.api.Group, i.e. vp8hip_group_*:
RCCL inside libvp8hip.so -- carries only the barrier, the max-over-ranks time and the gathering of results.  (The multi-process CPU
tests hand in an adapter of the same shape over gloo, tests/torch_transport.py: there is no second transport in this package.)
"""
from __future__ import annotations

import zlib

import numpy as np


def gop_chunks(total_frames: int, gop_size: int) -> list[tuple[int, int]]:
    """(first frame, length) of every closed GOP when a key frame is forced every gop_size frames."""
    return [(s, min(gop_size, total_frames - s)) for s in range(0, total_frames, gop_size)]


def chunks_of_rank(total_frames: int, gop_size: int, rank: int, world: int) -> list[tuple[int, int]]:
    return gop_chunks(total_frames, gop_size)[rank::world]


def frame_digest(out: dict | None, last_planes) -> int:
    """CRC of everything the entropy coder and the next frame consume from one frame of the path."""
    crc = 0
    if out is not None:
        for k in ("MB_parts", "MB_reference_frame", "MB_vectors", "MB_segment_id"):
            crc = zlib.crc32(np.ascontiguousarray(out[k]).tobytes(), crc)
        c = out["MB_coeffs"].copy()
        c[out["MB_parts"] != 0, 24] = 0      # block 24 only exists for 16x16 macroblocks
        crc = zlib.crc32(c.tobytes(), crc)
    for p in last_planes:
        crc = zlib.crc32(np.ascontiguousarray(p).tobytes(), crc)
    return crc


def encode_chunks(make_backend, sequence, chunks, width, height, **driver_kw) -> dict[int, int]:
    """Run the inter-path driver over the given GOP chunks; returns {frame number: digest}."""
    from .driver import InterPathDriver
    digests = {}
    for start, length in chunks:
        be = make_backend()
        drv = InterPathDriver(be, width, height, gop_size=1 << 30, **driver_kw)
        for t in range(start, start + length):
            y, u, v = sequence.frame(t)
            out = drv.encode_frame(y, u, v)
            digests[t] = frame_digest(out, be.download_last() if out is not None else (y, u, v))
        be.close()
    return digests


def gather_digests(local: dict[int, int], total_frames: int, group=None) -> np.ndarray:
    """All ranks' digests in frame order (-1 = not mine): one all_gather of a dense int64 vector over `group` (anything with
    all_gather_object: ref_shard.ThreadGroup members, the tests' gloo adapter), or this process's alone"""
    vec = np.full(total_frames, -1, np.int64)
    for t, d in local.items():
        vec[t] = d
    if group is None or group.get_world_size() == 1:
        return vec
    allv = np.stack([np.asarray(v, np.int64) for v in group.all_gather_object(vec)])
    assert ((allv >= 0).sum(axis=0) == 1).all(), "every frame must be encoded by exactly one rank"
    return allv.max(axis=0)


# ---- the same with the bitstream: every rank codes its GOP chunks to complete VP8 frames, rank order is restored ----
class NativeEncoder:
    """One GOP chunk on one GPU through the native frame loop: encode(y, u, v) -> the frame's bytes."""

    def __init__(self, width: int, height: int, device: int = 0, **cfg):
        from . import api
        cfg.setdefault("gop_size", 1 << 30)          # the chunk starts with its key frame; no other forced one inside
        cfg.setdefault("overlap_filter", 1)          # one chunk at a time per GPU here: loop filter beside the entropy stage
        self.drv = api.NativeDriver(width, height, device=device, **cfg)

    def encode(self, y, u, v) -> bytes:
        self.drv.encode_frame_host(y, u, v)
        return self.drv.get_frame()

    def close(self):
        self.drv.close()


def encode_chunks_frames(make_encoder, sequence, chunks) -> dict[int, bytes]:
    """{frame number: VP8 frame} of the given GOP chunks; make_encoder() -> object with encode(y, u, v) and close()."""
    frames = {}
    for start, length in chunks:
        enc = make_encoder()
        for t in range(start, start + length):
            frames[t] = enc.encode(*sequence.frame(t))
        enc.close()
    return frames


def gather_frames(local: dict[int, bytes], total_frames: int, group=None, dst: int = 0):
    """The sequence's frames in frame order on rank `dst` (None on the other ranks).  group: vp8oclenc_amd.api.Group (or anything with its
    rank / world / gather_bytes); None = one process, which owns every frame."""
    if group is None:
        assert len(local) == total_frames
        return [local[t] for t in range(total_frames)]
    return gather_frames_group(local, total_frames, group, dst)


def gather_frames_group(local: dict[int, bytes], total_frames: int, group, dst: int = 0):
    """gather_frames over vp8oclenc_amd.api.Group (vp8hip_group_gather_bytes: ncclSend / ncclRecv inside the library, the same code
    path at every world size, one rank included).  Every rank sends ONE self-describing blob -- [count][(frame number, length) x
    count] as little-endian uint64, then the frames' bytes in that order -- and `dst` takes the blobs apart."""
    mine = sorted(local)
    head = np.empty(1 + 2 * len(mine), np.uint64)
    head[0] = len(mine)
    head[1::2] = mine
    head[2::2] = [len(local[t]) for t in mine]
    blob = head.tobytes() + b"".join(local[t] for t in mine)
    data, counts = group.gather_bytes(blob, root=dst)
    if group.rank != dst:
        return None
    out: list = [None] * total_frames
    at = 0
    for r in range(group.world):
        piece = data[at:at + int(counts[r])]
        at += int(counts[r])
        n = int(piece[:8].view(np.uint64)[0])
        index = piece[8:8 + 16 * n].view(np.uint64).reshape(n, 2)
        off = 8 + 16 * n
        for t, ln in index:
            assert out[int(t)] is None, "every frame must be encoded by exactly one rank"
            out[int(t)] = piece[off:off + int(ln)].tobytes()
            off += int(ln)
        assert off == len(piece)
    assert all(b is not None for b in out), "every frame must be encoded by exactly one rank"
    return out


def write_ivf(path: str, frames: list[bytes], width: int, height: int, framerate: int = 30, timescale: int = 1) -> int:
    """The reference's output file (encIO.h:32-139): 32-byte header, then a 12-byte header + the bytes of every frame."""
    from . import bitstream
    with open(path, "wb") as f:
        # (the count the reference's program writes: one more than the frames in the file -- REFERENCE_DEFECTS.md #8)
        f.write(bitstream.ivf_file_header(width, height, framerate, timescale, len(frames) + 1))
        for t, b in enumerate(frames):
            f.write(bitstream.ivf_frame_header(len(b), t))
            f.write(b)
    return 32 + sum(12 + len(b) for b in frames)
