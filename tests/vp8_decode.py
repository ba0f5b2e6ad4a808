"""The rest of a VP8 decoder on top of tests/vp8_parse.py, written from RFC 6386 (test infrastructure, numpy, slow):
dequantisation (9.6, 14.1), inverse WHT / DCT (14.3, 14.4), intra prediction (12), inter prediction with the six-tap
filters (18), the normal loop filter (15), and the golden / altref buffer rules (9.7).  It shares no code with the encoder
or the reference.  tests/test_decode_roundtrip.py first holds it against libwebp on key frames (exact), then uses it to
decode whole sequences -- inter frames included -- and compares with the encoder's own reconstruction."""
from __future__ import annotations

import os
import re

import numpy as np

import vp8_parse as vp

ROOT = vp.ROOT


def _q_table(name):
    text = open(os.path.join(ROOT, "vp8oclenc_amd", "csrc", "vp8hip_dev.h")).read()
    m = re.search(r"%s\[128\]\s*=\s*\{(.*?)\};" % name, text, re.S)
    t = [int(x) for x in re.findall(r"\d+", m.group(1))]
    assert len(t) == 128
    return t


DC_Q, AC_Q = _q_table("k_dc_q"), _q_table("k_ac_q")      # 14.1 (numbers of the format)
SIXTAP = np.array([[0, 0, 128, 0, 0, 0], [0, -6, 123, 12, -1, 0], [2, -11, 108, 36, -8, 1], [0, -9, 93, 50, -6, 0],
                   [3, -16, 77, 77, -16, 3], [0, -6, 50, 93, -9, 0], [1, -8, 36, 108, -11, 2], [0, -1, 12, 123, -6, 0]], np.int32)   # 18.3
BORDER = 32    # replicated edge kept around every reference plane (vectors of this encoder stay inside the frame; 18.1 allows more)


def clamp255(a):
    return np.clip(a, 0, 255)


class Decoder:
    def __init__(self, reference_wrap: bool = False):
        """reference_wrap: NOT the format -- predict the way the reference encoder's `construct` does, per 4x4 block
        with the last 3 of the 9 first-pass lines WRAPPED to 8 bits instead of clamped (GPU_kernels.cl:702-758).  Only there to
        show that this is the whole difference between the encoder's reconstruction and a decoder's on content that overshoots."""
        self.reference_wrap = reference_wrap
        self.st = vp.StreamState()
        self.ref = {}          # 1 last, 2 golden, 3 altref -> (Y, U, V) uint8 planes of the CODED size (multiples of 16)

    # ---- 9.6 / 14.1 -------------------------------------------------------------------------------------------------------
    def _quant(self, f, seg):
        if f.segmentation_enabled:
            q = f.seg_quant[seg] if f.seg_abs else f.y_ac_qi + f.seg_quant[seg]
        else:
            q = f.y_ac_qi
        c = lambda v: min(max(v, 0), 127)
        y2ac = AC_Q[c(q + f.y2_ac_delta)] * 155 // 100
        return dict(y1=(DC_Q[c(q + f.y_dc_delta)], AC_Q[c(q)]), y2=(DC_Q[c(q + f.y2_dc_delta)] * 2, max(y2ac, 8)),
                    uv=(min(DC_Q[c(q + f.uv_dc_delta)], 132), AC_Q[c(q + f.uv_ac_delta)]))

    # ---- 14.3, 14.4 -------------------------------------------------------------------------------------------------------
    @staticmethod
    def _iwht(b):
        a = b.reshape(4, 4).astype(np.int64)
        a1, b1, c1, d1 = a[0] + a[3], a[1] + a[2], a[1] - a[2], a[0] - a[3]
        t = np.stack([a1 + b1, c1 + d1, a1 - b1, d1 - c1])
        a1, b1, c1, d1 = t[:, 0] + t[:, 3], t[:, 1] + t[:, 2], t[:, 1] - t[:, 2], t[:, 0] - t[:, 3]
        o = np.stack([a1 + b1, c1 + d1, a1 - b1, d1 - c1], axis=1)
        return ((o + 3) >> 3).reshape(16)

    @staticmethod
    def _idct(b):
        ip = np.moveaxis(b.reshape(b.shape[:-1] + (4, 4)).astype(np.int64), -2, 0)     # [row, ..., col]
        c, s = 20091, 35468

        def pass1(x0, x1, x2, x3):
            a1, b1 = x0 + x2, x0 - x2
            c1 = ((x1 * s) >> 16) - (x3 + ((x3 * c) >> 16))
            d1 = (x1 + ((x1 * c) >> 16)) + ((x3 * s) >> 16)
            return a1 + d1, b1 + c1, b1 - c1, a1 - d1

        t = np.stack(pass1(ip[0], ip[1], ip[2], ip[3]))                 # down the columns: [row, ..., col]
        o = np.stack(pass1(t[..., 0], t[..., 1], t[..., 2], t[..., 3]), axis=-1)   # along the rows
        return np.moveaxis((o + 4) >> 3, 0, -2)

    # ---- 12: intra prediction ---------------------------------------------------------------------------------------------
    @staticmethod
    def _edges(P, x0, y0, n):
        """above row (n + 4 samples: the 4 above-right ones for luma sub-blocks), left column, corner, with the frame's virtual
        borders: 127 above, 129 to the left (12.2)"""
        H, W = P.shape
        above = np.full(n + 4, 127, np.int32)
        left = np.full(n, 129, np.int32)
        corner = 127 if y0 == 0 else (129 if x0 == 0 else int(P[y0 - 1, x0 - 1]))
        if y0 > 0:
            w = min(n + 4, W - x0)
            above[:w] = P[y0 - 1, x0:x0 + w]
            if w < n + 4:
                above[w:] = above[w - 1] if w > n else P[y0 - 1, W - 1]
        if x0 > 0:
            left[:] = P[y0:y0 + n, x0 - 1]
        return above, left, corner

    def _pred_mb(self, P, x0, y0, n, mode):
        above, left, corner = self._edges(P, x0, y0, n)
        a, l = above[:n], left
        if mode == vp.DC_PRED:
            have_a, have_l = y0 > 0, x0 > 0
            if have_a and have_l:
                dc = (int(a.sum()) + int(l.sum()) + n) // (2 * n)
            elif have_a:
                dc = (int(a.sum()) + n // 2) // n
            elif have_l:
                dc = (int(l.sum()) + n // 2) // n
            else:
                dc = 128
            return np.full((n, n), dc, np.int32)
        if mode == vp.V_PRED:
            return np.tile(a, (n, 1))
        if mode == vp.H_PRED:
            return np.tile(l[:, None], (1, n))
        return clamp255(l[:, None] + a[None, :] - corner)    # TM_PRED

    @staticmethod
    def _pred_b(mode, A, L, P):
        """12.3: A = above[0..7] (with above-right), L = left[0..3], P = corner; the edge array E of the RFC: L3 L2 L1 L0 P A0..A7"""
        A, L = [int(v) for v in A], [int(v) for v in L]
        E = [L[3], L[2], L[1], L[0], P] + A
        o = np.zeros((4, 4), np.int32)
        avg3 = lambda x, y, z: (x + 2 * y + z + 2) >> 2
        avg2 = lambda x, y: (x + y + 1) >> 1
        if mode == vp.B_DC:
            o[:] = (sum(A[:4]) + sum(L) + 4) >> 3
        elif mode == vp.B_TM:
            o[:] = clamp255(np.array(L)[:, None] + np.array(A[:4])[None, :] - P)
        elif mode == vp.B_VE:
            o[:] = [avg3(P, A[0], A[1]), avg3(A[0], A[1], A[2]), avg3(A[1], A[2], A[3]), avg3(A[2], A[3], A[4])]
        elif mode == vp.B_HE:
            col = [avg3(P, L[0], L[1]), avg3(L[0], L[1], L[2]), avg3(L[1], L[2], L[3]), avg3(L[2], L[3], L[3])]
            o[:] = np.array(col)[:, None]
        elif mode == vp.B_LD:
            v = [avg3(A[i], A[i + 1], A[i + 2]) for i in range(6)] + [avg3(A[6], A[7], A[7])]
            for r in range(4):
                for c in range(4):
                    o[r, c] = v[r + c]
        elif mode == vp.B_RD:
            v = [avg3(E[i], E[i + 1], E[i + 2]) for i in range(7)]      # v[3] sits on the main diagonal
            for r in range(4):
                for c in range(4):
                    o[r, c] = v[3 - r + c]
        elif mode == vp.B_VR:
            o[3, 0] = avg3(E[1], E[2], E[3])
            o[2, 0] = avg3(E[2], E[3], E[4])
            o[3, 1] = o[1, 0] = avg3(E[3], E[4], E[5])
            o[2, 1] = o[0, 0] = avg2(E[4], E[5])
            o[3, 2] = o[1, 1] = avg3(E[4], E[5], E[6])
            o[2, 2] = o[0, 1] = avg2(E[5], E[6])
            o[3, 3] = o[1, 2] = avg3(E[5], E[6], E[7])
            o[2, 3] = o[0, 2] = avg2(E[6], E[7])
            o[1, 3] = avg3(E[6], E[7], E[8])
            o[0, 3] = avg2(E[7], E[8])
        elif mode == vp.B_VL:
            o[0, 0] = avg2(A[0], A[1])
            o[1, 0] = avg3(A[0], A[1], A[2])
            o[2, 0] = o[0, 1] = avg2(A[1], A[2])
            o[1, 1] = o[3, 0] = avg3(A[1], A[2], A[3])
            o[2, 1] = o[0, 2] = avg2(A[2], A[3])
            o[3, 1] = o[1, 2] = avg3(A[2], A[3], A[4])
            o[2, 2] = o[0, 3] = avg2(A[3], A[4])
            o[3, 2] = o[1, 3] = avg3(A[3], A[4], A[5])
            o[2, 3] = avg3(A[4], A[5], A[6])
            o[3, 3] = avg3(A[5], A[6], A[7])
        elif mode == vp.B_HD:
            o[3, 0] = avg2(E[0], E[1])
            o[3, 1] = avg3(E[0], E[1], E[2])
            o[2, 0] = o[3, 2] = avg2(E[1], E[2])
            o[2, 1] = o[3, 3] = avg3(E[1], E[2], E[3])
            o[2, 2] = o[1, 0] = avg2(E[2], E[3])
            o[2, 3] = o[1, 1] = avg3(E[2], E[3], E[4])
            o[1, 2] = o[0, 0] = avg2(E[3], E[4])
            o[1, 3] = o[0, 1] = avg3(E[3], E[4], E[5])
            o[0, 2] = avg3(E[4], E[5], E[6])
            o[0, 3] = avg3(E[5], E[6], E[7])
        else:   # B_HU
            o[0, 0] = avg2(L[0], L[1])
            o[0, 1] = avg3(L[0], L[1], L[2])
            o[0, 2] = o[1, 0] = avg2(L[1], L[2])
            o[0, 3] = o[1, 1] = avg3(L[1], L[2], L[3])
            o[1, 2] = o[2, 0] = avg2(L[2], L[3])
            o[1, 3] = o[2, 1] = avg3(L[2], L[3], L[3])
            o[2, 2] = o[2, 3] = o[3, 0] = o[3, 1] = o[3, 2] = o[3, 3] = L[3]
        return o

    # ---- 18: inter prediction ---------------------------------------------------------------------------------------------
    @staticmethod
    def _padded(P):
        return np.pad(P, BORDER, mode="edge").astype(np.int32)

    @staticmethod
    def _sixtap(R, x, y, w, h, mv_r, mv_c, wrap_last3=False):
        """block (x, y, w x h) of the padded reference R displaced by (mv_r, mv_c) in EIGHTHS of a sample"""
        ix, iy, fx, fy = x + (mv_c >> 3) + BORDER, y + (mv_r >> 3) + BORDER, mv_c & 7, mv_r & 7
        if fx == 0 and fy == 0:
            return R[iy:iy + h, ix:ix + w].copy()
        src = R[iy - 2:iy + h + 3, ix - 2:ix + w + 3]
        fh, fv = SIXTAP[fx], SIXTAP[fy]
        tmp = sum(fh[k] * src[:, k:k + w] for k in range(6))
        if wrap_last3:
            assert h == 4
            q = tmp + 64
            tmp = q >> 7
            tmp[6:] = (np.sign(q[6:]) * (np.abs(q[6:]) // 128)) & 255     # the reference divides (truncation), then casts to 8 bits
        else:
            tmp = (tmp + 64) >> 7
        tmp = clamp255(tmp)                                   # every line of the first pass is clamped
        out = sum(fv[k] * tmp[k:k + h, :] for k in range(6))
        return clamp255((out + 64) >> 7)

    # ---- 15: loop filter (normal) -----------------------------------------------------------------------------------------
    @staticmethod
    def _lf_edge(P, idx, hev_thr, ilim, elim, mb_edge):
        """filter across one edge.  idx: tuple of 8 index arrays (p3 p2 p1 p0 q0 q1 q2 q3), each selecting the samples along
        the edge"""
        p3, p2, p1, p0, q0, q1, q2, q3 = [P[i].astype(np.int32) for i in idx]
        mask = ((np.abs(p0 - q0) * 2 + (np.abs(p1 - q1) >> 1)) <= elim) & (np.abs(p3 - p2) <= ilim) & (np.abs(p2 - p1) <= ilim) & \
               (np.abs(p1 - p0) <= ilim) & (np.abs(q1 - q0) <= ilim) & (np.abs(q2 - q1) <= ilim) & (np.abs(q3 - q2) <= ilim)
        hev = (np.abs(p1 - p0) > hev_thr) | (np.abs(q1 - q0) > hev_thr)
        s = lambda v: v - 128                    # to signed
        c = lambda v: np.clip(v, -128, 127)
        sp2, sp1, sp0, sq0, sq1, sq2 = s(p2), s(p1), s(p0), s(q0), s(q1), s(q2)
        if mb_edge:
            w = c(c(sp1 - sq1) + 3 * (sq0 - sp0))
            w = np.where(mask, w, 0)
            # high edge variance: the common filter on the inner two samples only
            a = np.where(hev, w, 0)
            f1 = c(a + 4) >> 3
            f2 = c(a + 3) >> 3
            nq0, np0 = c(sq0 - f1), c(sp0 + f2)
            w2 = np.where(hev, 0, w)
            a = c((27 * w2 + 63) >> 7)
            nq0, np0 = c(nq0 - a), c(np0 + a)
            a = c((18 * w2 + 63) >> 7)
            nq1, np1 = c(sq1 - a), c(sp1 + a)
            a = c((9 * w2 + 63) >> 7)
            nq2, np2 = c(sq2 - a), c(sp2 + a)
            for i, v in ((idx[1], np2), (idx[2], np1), (idx[3], np0), (idx[4], nq0), (idx[5], nq1), (idx[6], nq2)):
                P[i] = (v + 128).astype(np.uint8)
        else:
            a = c(sp1 - sq1)
            a = np.where(hev, a, 0)
            a = c(a + 3 * (sq0 - sp0))
            a = np.where(mask, a, 0)
            f1 = c(a + 4) >> 3
            f2 = c(a + 3) >> 3
            nq0, np0 = c(sq0 - f1), c(sp0 + f2)
            a = (f1 + 1) >> 1
            a = np.where(hev, 0, a)
            nq1, np1 = c(sq1 - a), c(sp1 + a)
            for i, v in ((idx[2], np1), (idx[3], np0), (idx[4], nq0), (idx[5], nq1)):
                P[i] = (v + 128).astype(np.uint8)

    def _loop_filter(self, f, planes, skip_inner):
        """macroblocks in raster order, each: left edge, inner vertical edges, top edge, inner horizontal edges (15.2).  A
        macroblock only touches what its left, above and above-right neighbours have finished with, so all macroblocks with the
        same mx + 2 * my are independent and are filtered in one vectorised step -- the result is that of the raster order"""
        n_mb = f.mbw * f.mbh
        lvl = np.zeros(n_mb, np.int32)
        for mb in range(n_mb):
            if f.segmentation_enabled:
                seg = int(f.segment_id[mb])
                l = f.seg_lf[seg] if f.seg_abs else f.loop_filter_level + f.seg_lf[seg]
            else:
                l = f.loop_filter_level
            lvl[mb] = min(max(l, 0), 63)
        il = lvl.copy()
        if f.sharpness:
            il >>= 2 if f.sharpness > 4 else 1
            il = np.minimum(il, 9 - f.sharpness)
        il = np.maximum(il, 1)
        if f.key:
            hev = np.where(lvl >= 40, 2, np.where(lvl >= 15, 1, 0))
        else:
            hev = np.where(lvl >= 40, 3, np.where(lvl >= 20, 2, np.where(lvl >= 15, 1, 0)))
        mbl, sbl = (lvl + 2) * 2 + il, lvl * 2 + il
        for d in range(f.mbw + 2 * f.mbh):
            wave = [(d - 2 * my, my) for my in range(f.mbh) if 0 <= d - 2 * my < f.mbw]
            wave = [(mx, my) for mx, my in wave if lvl[my * f.mbw + mx] > 0]
            if not wave:
                continue
            for P, n in zip(planes, (16, 8, 8)):
                def run(sel, pos, vertical, mb_edge):
                    """one edge (at offset pos inside the macroblock) of every selected macroblock of the wave"""
                    if not sel:
                        return
                    ids = np.array([my * f.mbw + mx for mx, my in sel])
                    along = np.concatenate([np.arange(n) + (my if vertical else mx) * n for mx, my in sel])
                    across = np.repeat(np.array([(mx if vertical else my) * n + pos for mx, my in sel]), n)
                    rep = lambda a: np.repeat(a[ids], n)
                    idx = tuple((along, across + k) if vertical else (across + k, along) for k in range(-4, 4))
                    self._lf_edge(P, idx, rep(hev), rep(il), rep(mbl if mb_edge else sbl), mb_edge)
                inner = [(mx, my) for mx, my in wave if not skip_inner[my * f.mbw + mx]]
                run([(mx, my) for mx, my in wave if mx > 0], 0, True, True)
                for x in range(4, n, 4):
                    run(inner, x, True, False)
                run([(mx, my) for mx, my in wave if my > 0], 0, False, True)
                for y in range(4, n, 4):
                    run(inner, y, False, False)

    # ---- one frame --------------------------------------------------------------------------------------------------------
    def decode(self, data: bytes):
        """returns (parsed Frame, (Y, U, V) of the coded size after the loop filter)"""
        f = vp.parse_frame(data, self.st)
        W, H = f.mbw * 16, f.mbh * 16
        Y, U, V = np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)
        padded = {r: tuple(self._padded(p) for p in planes) for r, planes in self.ref.items()} if not f.key else {}
        skip_inner = np.zeros(f.mbw * f.mbh, bool)
        for my in range(f.mbh):
            for mx in range(f.mbw):
                mb = my * f.mbw + mx
                q = self._quant(f, int(f.segment_id[mb]) if f.segmentation_enabled else 0)
                co = f.coeffs[mb].astype(np.int64)
                dq = np.zeros((24, 16), np.int64)
                dq[:16] = co[:16] * q["y1"][1]
                dq[:16, 0] = co[:16, 0] * q["y1"][0]
                dq[16:] = co[16:24] * q["uv"][1]
                dq[16:, 0] = co[16:24, 0] * q["uv"][0]
                if f.has_y2[mb]:
                    y2 = co[24] * q["y2"][1]
                    y2[0] = co[24, 0] * q["y2"][0]
                    dq[:16, 0] = self._iwht(y2)
                res = self._idct(dq)
                nonzero = bool(np.any(co != 0))
                is4x4 = (not f.is_inter[mb] and f.ymode[mb] == vp.B_PRED) or (f.is_inter[mb] and f.mv_mode[mb] == vp.MV_SPLIT)
                skip_inner[mb] = (not nonzero) and not is4x4
                x0, y0 = mx * 16, my * 16
                if f.is_inter[mb]:
                    RY, RU, RV = padded[int(f.ref_frame[mb])]
                    if f.mv_mode[mb] != vp.MV_SPLIT and not self.reference_wrap:      # one vector: the same samples as sixteen 4x4 predictions, in one call
                        r, c = int(f.mvs[mb, 0, 0]), int(f.mvs[mb, 0, 1])
                        blk = np.block([[res[4 * i + j] for j in range(4)] for i in range(4)])
                        Y[y0:y0 + 16, x0:x0 + 16] = clamp255(self._sixtap(RY, x0, y0, 16, 16, 2 * r, 2 * c) + blk)
                        for P, R, off in ((U, RU, 16), (V, RV, 20)):
                            blk = np.block([[res[off + 2 * i + j] for j in range(2)] for i in range(2)])
                            P[y0 // 2:y0 // 2 + 8, x0 // 2:x0 // 2 + 8] = clamp255(self._sixtap(R, x0 // 2, y0 // 2, 8, 8, r, c) + blk)
                        continue
                    for b in range(16):
                        by, bx = b >> 2, b & 3
                        r, c = int(f.mvs[mb, b, 0]) * 2, int(f.mvs[mb, b, 1]) * 2     # quarter pixels -> eighths (18.3)
                        p = self._sixtap(RY, x0 + 4 * bx, y0 + 4 * by, 4, 4, r, c, self.reference_wrap)
                        Y[y0 + 4 * by:y0 + 4 * by + 4, x0 + 4 * bx:x0 + 4 * bx + 4] = clamp255(p + res[b])
                    for b in range(4):       # chroma: one vector per 2x2 luma sub-blocks, full precision (18.3 / 5)
                        by, bx = b >> 1, b & 1
                        grp = [(2 * by + j) * 4 + 2 * bx + i for j in range(2) for i in range(2)]
                        sr = int(sum(int(f.mvs[mb, g, 0]) for g in grp)) * 2
                        sc = int(sum(int(f.mvs[mb, g, 1]) for g in grp)) * 2
                        r = _div_trunc(sr + (4 if sr >= 0 else -4), 8)
                        c = _div_trunc(sc + (4 if sc >= 0 else -4), 8)
                        for P, R, off in ((U, RU, 16), (V, RV, 20)):
                            p = self._sixtap(R, x0 // 2 + 4 * bx, y0 // 2 + 4 * by, 4, 4, r, c, self.reference_wrap)
                            P[y0 // 2 + 4 * by:y0 // 2 + 4 * by + 4, x0 // 2 + 4 * bx:x0 // 2 + 4 * bx + 4] = clamp255(p + res[off + b])
                else:
                    if f.ymode[mb] == vp.B_PRED:
                        above_mb, _, _ = self._edges(Y, x0, y0, 16)      # above row of the macroblock with its above-right 4
                        for b in range(16):
                            by, bx = b >> 2, b & 3
                            xx, yy = x0 + 4 * bx, y0 + 4 * by
                            A, L, C = self._edges(Y, xx, yy, 4)
                            if bx == 3:
                                A[4:8] = above_mb[16:20]               # the right column takes the macroblock's above-right (12.3)
                            pred = self._pred_b(int(f.bmodes[mb, b]), A, L, C)
                            Y[yy:yy + 4, xx:xx + 4] = clamp255(pred + res[b])
                    else:
                        pred = self._pred_mb(Y, x0, y0, 16, int(f.ymode[mb]))
                        blk = np.block([[res[4 * r + c] for c in range(4)] for r in range(4)])
                        Y[y0:y0 + 16, x0:x0 + 16] = clamp255(pred + blk)
                    for P, off in ((U, 16), (V, 20)):
                        pred = self._pred_mb(P, x0 // 2, y0 // 2, 8, int(f.uvmode[mb]))
                        blk = np.block([[res[off + 2 * r + c] for c in range(2)] for r in range(2)])
                        P[y0 // 2:y0 // 2 + 8, x0 // 2:x0 // 2 + 8] = clamp255(pred + blk)
        self.prefilter = (Y.copy(), U.copy(), V.copy())      # for diagnosis: the reconstruction before the loop filter
        self._loop_filter(f, (Y, U, V), skip_inner)
        # 9.7: buffer updates
        new = (Y, U, V)
        if f.key:
            self.ref = {1: new, 2: new, 3: new}
        else:
            old = dict(self.ref)
            if f.copy_to_golden:
                self.ref[2] = old[1] if f.copy_to_golden == 1 else old[3]
            if f.copy_to_altref:
                self.ref[3] = old[1] if f.copy_to_altref == 1 else old[2]
            if f.refresh_golden:
                self.ref[2] = new
            if f.refresh_altref:
                self.ref[3] = new
            if f.refresh_last:
                self.ref[1] = new
        return f, new


def _div_trunc(a, b):
    return a // b if a >= 0 else -((-a) // b)
