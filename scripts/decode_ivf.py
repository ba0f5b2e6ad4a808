#!/usr/bin/env python3
"""Decode an .ivf of VP8 frames with the decoder the tests use (tests/vp8_decode.py, written from RFC 6386; numpy, about
20 s per 1080p frame) -- the image has no other VP8 video decoder -- and report per-frame PSNR against a source.

    python scripts/decode_ivf.py in.ivf [--out decoded.yuv] [--yuv source.yuv | --synth-seed 1] [--frames N]
"""
import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def read_ivf(path):
    """-> (width, height, framerate, timescale, [frame bytes]) -- the layout of encIO.h:32-72 / vp8bs_ivf_*"""
    with open(path, "rb") as f:
        head = f.read(32)
        assert head[:4] == b"DKIF" and head[8:12] == b"VP80", "not an IVF file of VP8 frames"
        hlen, = struct.unpack_from("<H", head, 6)
        W, H, rate, scale, count = struct.unpack_from("<HHIII", head, 12)
        f.seek(hlen)
        frames = []
        while True:
            fh = f.read(12)
            if len(fh) < 12:
                break
            size, _pts = struct.unpack("<IQ", fh)
            frames.append(f.read(size))
    return W, H, rate, scale, frames


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def main():
    import vp8_decode
    ap = argparse.ArgumentParser()
    ap.add_argument("ivf"); ap.add_argument("--out"); ap.add_argument("--yuv"); ap.add_argument("--synth-seed", type=int)
    ap.add_argument("--frames", type=int, default=1 << 30)
    a = ap.parse_args()
    W, H, rate, scale, frames = read_ivf(a.ivf)
    src = None
    if a.yuv:
        from encode_ivf import YuvFile
        src = YuvFile(a.yuv, W, H)
    elif a.synth_seed is not None:
        from vp8oclenc_amd.synth import SynthSequence
        src = SynthSequence(W, H, seed=a.synth_seed)
    dec = vp8_decode.Decoder()
    out = open(a.out, "wb") if a.out else None
    worst = 99.0
    for t, fr in enumerate(frames[:a.frames]):
        f, (Y, U, V) = dec.decode(fr)
        Y, U, V = Y[:H, :W], U[:(H + 1) // 2, :(W + 1) // 2], V[:(H + 1) // 2, :(W + 1) // 2]
        line = f"frame {t:4d} {'key  ' if f.key else 'inter'} {len(fr):8d} bytes"
        if src is not None:
            y, u, v = src.frame(t)
            p = [psnr(Y, y[:H, :W]), psnr(U, u), psnr(V, v)]
            worst = min(worst, p[0])
            line += f"  PSNR Y {p[0]:.2f} U {p[1]:.2f} V {p[2]:.2f} dB"
        print(line)
        if out:
            out.write(Y.tobytes()); out.write(U.tobytes()); out.write(V.tobytes())
    if src is not None:
        print(f"lowest luma PSNR {worst:.2f} dB over {min(len(frames), a.frames)} frames")


if __name__ == "__main__":
    main()
