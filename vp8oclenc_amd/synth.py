"""Seeded synthetic YUV420 sequences ("foreman-like") for parity tests and bench.py.

The reference ships no test content, so every test and benchmark input
comes from here: a low-frequency luminance field with a mid-frequency texture that drifts by a
fractional number of pixels per frame, textured rectangles moving at +-(1..6) px/frame including
fractional motion, uniform noise of small amplitude, smooth chroma.  Dimensions are padded to
multiples of 16 by edge replication the way src/encIO.h:141-202 pads ("wrk" size).
"""
from __future__ import annotations

import numpy as np


def wrk_size(width: int, height: int) -> tuple[int, int]:
    """Padded working size, src/init.h:381-389."""
    return (width + 15) // 16 * 16, (height + 15) // 16 * 16


def _bilinear_shift(tex: np.ndarray, oy: float, ox: float, h: int, w: int) -> np.ndarray:
    iy, ix = int(np.floor(oy)), int(np.floor(ox))
    fy, fx = oy - iy, ox - ix
    a = tex[iy:iy + h, ix:ix + w]
    b = tex[iy:iy + h, ix + 1:ix + w + 1]
    c = tex[iy + 1:iy + h + 1, ix:ix + w]
    d = tex[iy + 1:iy + h + 1, ix + 1:ix + w + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


class SynthSequence:
    """Deterministic frame generator; frame(t) -> (Y, U, V) uint8 arrays of the wrk size."""

    def __init__(self, width: int, height: int, seed: int = 1, noise: int = 4, n_rects: int = 6,
                 saturate: bool = False):
        self.width, self.height = width, height
        self.W, self.H = wrk_size(width, height)
        self.noise = noise
        self.seed = seed
        rng = np.random.default_rng(seed)
        m = 96  # margin for drift
        hh, ww = height + 2 * m, width + 2 * m
        yy, xx = np.mgrid[0:hh, 0:ww].astype(np.float32)
        tex = 128 + 60 * np.sin(xx / 97.0 + 0.3) * np.cos(yy / 71.0) + 25 * np.sin(xx / 13.0 + yy / 17.0)
        tex += 10 * np.sin(xx / 3.1) * np.sin(yy / 2.7)
        tex += rng.uniform(-6, 6, size=tex.shape).astype(np.float32)
        if saturate:  # push parts of the picture to 0 / 255 to exercise every clamp
            tex = (tex - 128) * 2.6 + 128
        self.tex = tex.astype(np.float32)
        self.m = m
        self.bg_v = rng.uniform(-1.5, 1.5, size=2)
        self.rects = []
        for _ in range(n_rects):
            rh = int(rng.integers(max(8, height // 8), max(9, height // 3)))
            rw = int(rng.integers(max(8, width // 8), max(9, width // 3)))
            pos = np.array([rng.uniform(0, height - rh), rng.uniform(0, width - rw)])
            vel = rng.choice([-1, 1], size=2) * rng.uniform(1, 6, size=2)
            if rng.random() < 0.5:
                vel = np.round(vel * 4) / 4  # quarter-pel motion
            ph = rng.uniform(0, 6.28)
            ry, rx = np.mgrid[0:rh + 2, 0:rw + 2].astype(np.float32)
            rt = 128 + 70 * np.sin(rx / 5.0 + ph) * np.cos(ry / 7.0 + ph) + rng.uniform(-12, 12, size=ry.shape)
            if saturate:
                rt = (rt - 128) * 2.2 + 128
            cu, cv = rng.uniform(-40, 40, size=2)
            self.rects.append((rh, rw, pos, vel, rt.astype(np.float32), cu, cv))

    def frame(self, t: int):
        h, w, m = self.height, self.width, self.m
        oy = m + np.clip(self.bg_v[0] * t, -m + 2, m - 2)
        ox = m + np.clip(self.bg_v[1] * t, -m + 2, m - 2)
        y = _bilinear_shift(self.tex, float(oy), float(ox), h, w).copy()
        yy, xx = np.mgrid[0:h // 2 + h % 2, 0:w // 2 + w % 2].astype(np.float32)
        u = 128 + 30 * np.sin(xx / 41.0 + 0.02 * t) + 10 * np.cos(yy / 29.0)
        v = 128 + 30 * np.cos(xx / 37.0) + 10 * np.sin(yy / 23.0 - 0.03 * t)
        for rh, rw, pos, vel, rt, cu, cv in self.rects:
            p = pos + vel * t
            # bounce inside the frame
            for k, lim in ((0, h - rh), (1, w - rw)):
                period = 2 * max(lim, 1)
                q = p[k] % period
                p[k] = q if q <= lim else period - q
            iy, ix = int(np.floor(p[0])), int(np.floor(p[1]))
            fy, fx = p[0] - iy, p[1] - ix
            patch = _bilinear_shift(rt, 1 - fy if fy > 0 else 0.0, 1 - fx if fx > 0 else 0.0, rh, rw)
            y[iy:iy + rh, ix:ix + rw] = patch
            u[iy // 2:(iy + rh) // 2, ix // 2:(ix + rw) // 2] = 128 + cu
            v[iy // 2:(iy + rh) // 2, ix // 2:(ix + rw) // 2] = 128 + cv
        rng = np.random.default_rng((self.seed << 20) + t)
        if self.noise:
            y += rng.integers(-self.noise, self.noise + 1, size=y.shape)
        Y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
        U = np.clip(np.rint(u), 0, 255).astype(np.uint8)
        V = np.clip(np.rint(v), 0, 255).astype(np.uint8)
        return self._pad(Y, self.H, self.W), self._pad(U, self.H // 2, self.W // 2), self._pad(V, self.H // 2, self.W // 2)

    @staticmethod
    def _pad(a: np.ndarray, H: int, W: int) -> np.ndarray:
        a = a[:H, :W]
        return np.ascontiguousarray(np.pad(a, ((0, H - a.shape[0]), (0, W - a.shape[1])), mode="edge"))


def noise_frames(width: int, height: int, seed: int, amp: int = 120):
    """Two nearly unrelated frames: drives the ushort cost of the 1-step search toward wrap-around."""
    W, H = wrk_size(width, height)
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(2):
        Y = np.clip(128 + rng.integers(-amp, amp + 1, size=(H, W)), 0, 255).astype(np.uint8)
        U = np.clip(128 + rng.integers(-amp, amp + 1, size=(H // 2, W // 2)), 0, 255).astype(np.uint8)
        V = np.clip(128 + rng.integers(-amp, amp + 1, size=(H // 2, W // 2)), 0, 255).astype(np.uint8)
        out.append((Y, U, V))
    return out


def bench_frames(width: int, height: int, seed: int, nd: int = 8):
    """The `nd` distinct synthetic frames bench.py's legs cycle through (frame t of a chunk that starts at phase p is frames[(p + t) % nd]),
    as (W, H, source, padded): `source` = what is handed to the encoder (the source size when it is below the coded size: 1920x1080 in a
    1920x1088 context -- copy_with_padding, encIO.h:141-196, then runs on the device), `padded` = the same frames padded on the host the
    way copy_with_padding pads (what the CPU oracle and the CPU baseline code).  One definition, so that bench.py, the full-length oracle
    runs (scripts/full_length_oracle.py) and the committed digests agree by construction."""
    seq = SynthSequence(width, height, seed=seed)
    W, H = seq.W, seq.H
    frames = [seq.frame(t) for t in range(nd)]
    if (width, height) != (W, H) and width % 2 == 0 and height % 2 == 0 and W - width < 16 and H - height < 16:
        source = [(np.ascontiguousarray(y[:height, :width]), np.ascontiguousarray(u[:height // 2, :width // 2]), np.ascontiguousarray(v[:height // 2, :width // 2]))
                  for y, u, v in frames]
        pad = lambda p, h, w: np.pad(p, ((0, h - p.shape[0]), (0, w - p.shape[1])), mode="edge")
        padded = [(pad(y, H, W), pad(u, H // 2, W // 2), pad(v, H // 2, W // 2)) for y, u, v in source]
        return W, H, source, padded
    return W, H, frames, frames
