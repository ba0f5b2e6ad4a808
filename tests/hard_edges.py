"""Content that makes the six-tap filters overshoot: a field of black and white 4x4 tiles (chroma following luma) panned by a
fractional number of pixels per frame, so inter frames pick fractional vectors across hard edges.  On such content the
reference's predictor (`construct`, last three first-pass lines wrapped instead of saturated) and a decoder's part ways."""
import numpy as np


class HardEdgeSequence:
    def __init__(self, width: int, height: int, seed: int = 5, step=(0.75, 0.5)):
        self.W, self.H = width // 16 * 16, height // 16 * 16
        rng = np.random.default_rng(seed)
        tiles = (rng.integers(0, 2, (self.H // 4 + 32, self.W // 4 + 32)) * 255).astype(np.float64)
        self.big = np.kron(tiles, np.ones((4, 4)))
        self.step = step

    def frame(self, t: int):
        dx, dy = self.step[0] * t, self.step[1] * t
        ix, iy = int(dx), int(dy)
        fx, fy = dx - ix, dy - iy
        a = self.big[iy:iy + self.H + 1, ix:ix + self.W + 1]
        y = (1 - fy) * ((1 - fx) * a[:-1, :-1] + fx * a[:-1, 1:]) + fy * ((1 - fx) * a[1:, :-1] + fx * a[1:, 1:])
        y = np.clip(np.rint(y), 0, 255).astype(np.uint8)
        u = (y[::2, ::2] // 2 + 64).astype(np.uint8)
        v = (255 - y[1::2, 1::2] // 2 - 64).astype(np.uint8)
        return y, np.ascontiguousarray(u), np.ascontiguousarray(v)
