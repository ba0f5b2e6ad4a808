"""A VP8 bitstream PARSER written from RFC 6386 (test infrastructure): frame header, per-macroblock modes and motion
vectors, coefficient tokens -- everything a decoder reads before it starts predicting.  It shares no code with the encoder
(csrc/vp8_bitstream.cpp, vp8_mbhdr.h, the device coders) or with the reference; the tests hand it finished frames and
compare what it reads -- segment ids, reference frames, vectors, intra modes, every coefficient, the segment quantisers and
filter levels -- with what the encoder meant to say, and require that every partition is consumed to its end.
Key frames are also checked pixel for pixel by libwebp (tests/webp_decode.py); for inter frames the image has no decoder,
so this is the independent reading of their syntax.

Section numbers are RFC 6386's."""
from __future__ import annotations

import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_table(name, shape):
    """a constant table of the format out of csrc/vp8_rfc6386_tables.inc (numbers only; generated from the RFC's tables)"""
    text = open(os.path.join(ROOT, "vp8oclenc_amd", "csrc", "vp8_rfc6386_tables.inc")).read()
    m = re.search(r"%s(?:\[\d+\])+\s*=\s*\{(.*?)\};" % re.escape(name), text, re.S)
    return np.array([int(x) for x in re.findall(r"-?\d+", m.group(1))], np.int32).reshape(shape)


KF_BMODE_PROBS = _c_table("k_kf_bmode_probs", (10, 10, 9))          # 11.5
COEFF_UPDATE_PROBS = _c_table("k_coeff_update_probs", (4, 8, 3, 11))  # 13.4
DEFAULT_COEFF_PROBS = _c_table("k_default_coeff_probs", (4, 8, 3, 11))  # 13.5
MV_UPDATE_PROBS = _c_table("k_mv_update_probs", (2, 19))            # 17.2

# trees (8.1): positive = next index, non-positive = -(leaf value)
DC_PRED, V_PRED, H_PRED, TM_PRED, B_PRED = 0, 1, 2, 3, 4
B_DC, B_TM, B_VE, B_HE, B_LD, B_RD, B_VR, B_VL, B_HD, B_HU = range(10)
KF_YMODE_TREE = [-B_PRED, 2, 4, 6, -DC_PRED, -V_PRED, -H_PRED, -TM_PRED]          # 11.2
YMODE_TREE = [-DC_PRED, 2, 4, 6, -V_PRED, -H_PRED, -TM_PRED, -B_PRED]             # 16.1
UV_MODE_TREE = [-DC_PRED, 2, -V_PRED, 4, -H_PRED, -TM_PRED]
BMODE_TREE = [-B_DC, 2, -B_TM, 4, -B_VE, 6, 8, 12, -B_HE, 10, -B_RD, -B_VR, -B_LD, 14, -B_VL, 16, -B_HD, -B_HU]
SEGMENT_TREE = [2, 4, -0, -1, -2, -3]                                              # 9.3
MV_ZERO, MV_NEAREST, MV_NEAR, MV_NEW, MV_SPLIT = range(5)
MV_REF_TREE = [-MV_ZERO, 2, -MV_NEAREST, 4, -MV_NEAR, 6, -MV_NEW, -MV_SPLIT]      # 16.3
SPLIT_16x8, SPLIT_8x16, SPLIT_QUARTERS, SPLIT_4x4 = range(4)
SPLIT_MV_TREE = [-SPLIT_4x4, 2, -SPLIT_QUARTERS, 4, -SPLIT_16x8, -SPLIT_8x16]   # 16.3: sixteenths "0", quarters "10", top/bottom "110", left/right "111"
LEFT4x4, ABOVE4x4, ZERO4x4, NEW4x4 = range(4)
SUB_MV_REF_TREE = [-LEFT4x4, 2, -ABOVE4x4, 4, -ZERO4x4, -NEW4x4]
SMALL_MV_TREE = [2, 8, 4, 6, -0, -1, -2, -3, 10, 12, -4, -5, -6, -7]              # 17.1
KF_YMODE_PROB = [145, 156, 163, 128]
YMODE_PROB = [112, 86, 140, 37]
KF_UV_MODE_PROB = [142, 114, 183]
UV_MODE_PROB = [162, 101, 204]
BMODE_PROB = [120, 90, 79, 133, 87, 85, 80, 111, 151]
SPLIT_MV_PROB = [110, 111, 150]
SUB_MV_REF_PROB = [[147, 136, 18], [106, 145, 1], [179, 121, 1], [223, 1, 34], [208, 1, 1]]
MODE_CONTEXTS = [[7, 1, 1, 143], [14, 18, 14, 107], [135, 64, 57, 68], [60, 56, 128, 65], [159, 134, 128, 34], [234, 188, 128, 28]]
DEFAULT_MV_PROBS = [[162, 128, 225, 146, 172, 147, 214, 39, 156, 128, 129, 132, 75, 145, 178, 206, 239, 254, 254],
                    [164, 128, 204, 170, 119, 235, 140, 230, 228, 128, 130, 130, 74, 148, 180, 203, 236, 254, 254]]
MV_IS_SHORT, MV_SIGN, MV_SHORT, MV_LONG, MV_LONG_BITS = 0, 1, 2, 9, 10
SPLIT_PARTITION = {SPLIT_16x8: [0] * 8 + [1] * 8, SPLIT_8x16: [0, 0, 1, 1] * 4,
                   SPLIT_QUARTERS: [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3], SPLIT_4x4: list(range(16))}
# tokens (13.2)
DCT_0, DCT_1, DCT_2, DCT_3, DCT_4, CAT1, CAT2, CAT3, CAT4, CAT5, CAT6, DCT_EOB = range(12)
COEFF_TREE = [-DCT_EOB, 2, -DCT_0, 4, -DCT_1, 6, 8, 12, -DCT_2, 10, -DCT_3, -DCT_4, 14, 16, -CAT1, -CAT2, 18, 20, -CAT3, -CAT4, -CAT5, -CAT6]
COEFF_BANDS = [0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7]
ZIGZAG = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
CAT_PROBS = {CAT1: [159], CAT2: [165, 145], CAT3: [173, 148, 140], CAT4: [176, 155, 140, 135], CAT5: [180, 157, 141, 134, 130],
             CAT6: [254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129]}
CAT_BASE = {CAT1: 5, CAT2: 7, CAT3: 11, CAT4: 19, CAT5: 35, CAT6: 67}


class BoolDecoder:
    """7.3"""

    def __init__(self, data: bytes):
        self.data = data
        self.pos = 2
        self.value = (data[0] << 8 | data[1]) if len(data) >= 2 else (data[0] << 8 if data else 0)
        self.range = 255
        self.bit_count = 0

    def read(self, prob: int) -> int:
        split = 1 + (((self.range - 1) * prob) >> 8)
        big = split << 8
        if self.value >= big:
            ret = 1
            self.range -= split
            self.value -= big
        else:
            ret = 0
            self.range = split
        while self.range < 128:
            self.value <<= 1
            self.range <<= 1
            self.bit_count += 1
            if self.bit_count == 8:
                self.bit_count = 0
                if self.pos < len(self.data):
                    self.value |= self.data[self.pos]
                self.pos += 1
        return ret

    def flag(self) -> int:
        return self.read(128)

    def literal(self, n: int) -> int:
        v = 0
        for _ in range(n):
            v = (v << 1) | self.read(128)
        return v

    def signed(self, n: int) -> int:
        v = self.literal(n)
        return -v if self.flag() else v

    def tree(self, t, probs, start=0) -> int:
        i = start
        while True:
            i = t[i + self.read(probs[i >> 1])]
            if i <= 0:
                return -i

    def overrun(self) -> int:
        """bytes read past the end of the data (a conforming stream never needs more than two of padding)"""
        return max(0, self.pos - len(self.data))


class Frame:
    pass


class StreamState:
    """what persists from frame to frame (9.7, 9.10, 9.11, 13.4, 17.2): probabilities, segment parameters, sign biases"""

    def __init__(self):
        self.coeff_probs = DEFAULT_COEFF_PROBS.copy()
        self.mv_probs = [list(r) for r in DEFAULT_MV_PROBS]
        self.ymode_prob = list(YMODE_PROB)
        self.uv_mode_prob = list(UV_MODE_PROB)
        self.seg_quant = [0] * 4
        self.seg_lf = [0] * 4
        self.seg_abs = 0
        self.seg_tree_probs = [255] * 3
        self.segmentation_enabled = 0
        self.ref_lf_delta = [0] * 4
        self.mode_lf_delta = [0] * 4
        self.sign_bias = [0, 0, 0, 0]   # by reference frame: intra, last, golden, altref


def parse_frame(data: bytes, st: StreamState) -> Frame:
    """one frame (9.1 .. 9.11, 19.2, 19.3, 13): returns a Frame with everything read"""
    f = Frame()
    tag = data[0] | data[1] << 8 | data[2] << 16
    f.key = not (tag & 1)
    f.version = (tag >> 1) & 7
    f.show = (tag >> 4) & 1
    f.first_part_size = tag >> 5
    off = 3
    if f.key:
        assert data[3:6] == b"\x9d\x01\x2a", "start code"
        f.width = (data[6] | data[7] << 8) & 0x3fff
        f.height = (data[8] | data[9] << 8) & 0x3fff
        f.hscale, f.vscale = data[7] >> 6, data[9] >> 6
        off = 10
        st.__init__()
        st.width, st.height = f.width, f.height
    else:
        f.width, f.height = st.width, st.height
    mbw, mbh = (f.width + 15) // 16, (f.height + 15) // 16
    f.mbw, f.mbh = mbw, mbh
    d = BoolDecoder(data[off:off + f.first_part_size])
    if f.key:
        f.color_space, f.clamping_type = d.flag(), d.flag()
    st.segmentation_enabled = d.flag()
    f.update_mb_segmentation_map = 0
    if st.segmentation_enabled:   # 9.3
        f.update_mb_segmentation_map = d.flag()
        if d.flag():              # update_segment_feature_data
            st.seg_abs = d.flag()
            st.seg_quant = [d.signed(7) if d.flag() else 0 for _ in range(4)]
            st.seg_lf = [d.signed(6) if d.flag() else 0 for _ in range(4)]
        if f.update_mb_segmentation_map:
            st.seg_tree_probs = [d.literal(8) if d.flag() else 255 for _ in range(3)]
    f.filter_type = d.flag()      # 9.6
    f.loop_filter_level = d.literal(6)
    f.sharpness = d.literal(3)
    f.lf_delta_enabled = d.flag()
    if f.lf_delta_enabled and d.flag():
        for i in range(4):
            if d.flag():
                st.ref_lf_delta[i] = d.signed(6)
        for i in range(4):
            if d.flag():
                st.mode_lf_delta[i] = d.signed(6)
    f.partitions = 1 << d.literal(2)   # 9.5
    f.y_ac_qi = d.literal(7)           # 9.6
    f.y_dc_delta, f.y2_dc_delta, f.y2_ac_delta, f.uv_dc_delta, f.uv_ac_delta = [d.signed(4) if d.flag() else 0 for _ in range(5)]
    if f.key:
        f.refresh_golden = f.refresh_altref = 1
        f.copy_to_golden = f.copy_to_altref = 0
        f.refresh_entropy = d.flag()
        f.refresh_last = 1
    else:                              # 9.7
        f.refresh_golden = d.flag()
        f.refresh_altref = d.flag()
        f.copy_to_golden = 0 if f.refresh_golden else d.literal(2)
        f.copy_to_altref = 0 if f.refresh_altref else d.literal(2)
        st.sign_bias[2] = d.flag()
        st.sign_bias[3] = d.flag()
        f.refresh_entropy = d.flag()
        f.refresh_last = d.flag()
    saved = None if f.refresh_entropy else (st.coeff_probs.copy(), [list(r) for r in st.mv_probs], list(st.ymode_prob), list(st.uv_mode_prob))
    for i in range(4):                 # 13.4
        for j in range(8):
            for k in range(3):
                for t in range(11):
                    if d.read(int(COEFF_UPDATE_PROBS[i, j, k, t])):
                        st.coeff_probs[i, j, k, t] = d.literal(8)
    f.mb_no_coeff_skip = d.flag()      # 9.11
    f.prob_skip_false = d.literal(8) if f.mb_no_coeff_skip else 0
    if not f.key:
        f.prob_intra = d.literal(8)
        f.prob_last = d.literal(8)
        f.prob_gf = d.literal(8)
        if d.flag():
            st.ymode_prob = [d.literal(8) for _ in range(4)]
        if d.flag():
            st.uv_mode_prob = [d.literal(8) for _ in range(3)]
        for c in range(2):             # 17.2
            for i in range(19):
                if d.read(int(MV_UPDATE_PROBS[c, i])):
                    x = d.literal(7)
                    st.mv_probs[c][i] = x << 1 if x else 1
    f.seg_quant, f.seg_lf, f.seg_abs = list(st.seg_quant), list(st.seg_lf), st.seg_abs
    f.segmentation_enabled = st.segmentation_enabled
    # ---- token partitions (9.5): 3-byte sizes of all but the last, then the partitions -----------------------------------
    p0 = off + f.first_part_size
    sizes = [data[p0 + 3 * i] | data[p0 + 3 * i + 1] << 8 | data[p0 + 3 * i + 2] << 16 for i in range(f.partitions - 1)]
    start = p0 + 3 * (f.partitions - 1)
    bounds = []
    for s in sizes:
        bounds.append((start, start + s))
        start += s
    bounds.append((start, len(data)))
    f.partition_sizes = [b - a for a, b in bounds]
    assert all(b >= a for a, b in bounds) and bounds[-2][1] <= len(data) if len(bounds) > 1 else True, "partition table runs past the frame"
    tok = [BoolDecoder(data[a:b]) for a, b in bounds]
    # ---- per-macroblock data (19.3) -------------------------------------------------------------------------------------
    n = mbw * mbh
    f.segment_id = np.zeros(n, np.int32)
    f.skip = np.zeros(n, np.int32)
    f.is_inter = np.zeros(n, np.int32)
    f.ref_frame = np.zeros(n, np.int32)          # 0 intra, 1 last, 2 golden, 3 altref
    f.ymode = np.zeros(n, np.int32)
    f.uvmode = np.zeros(n, np.int32)
    f.bmodes = np.zeros((n, 16), np.int32)
    f.mv_mode = np.full(n, -1, np.int32)
    f.split = np.full(n, -1, np.int32)
    f.mvs = np.zeros((n, 16, 2), np.int32)       # per 4x4 sub-block (row, col), quarter pixels
    f.coeffs = np.zeros((n, 25, 16), np.int32)   # raster order inside a block; block 24 = Y2
    f.has_y2 = np.zeros(n, np.int32)
    above_nz = np.zeros((mbw, 9), np.int32)      # 4 Y, 2 U, 2 V, 1 Y2 (13.3)
    for my in range(mbh):
        left_nz = np.zeros(9, np.int32)
        t = tok[my % f.partitions]
        for mx in range(mbw):
            mb = my * mbw + mx
            if f.update_mb_segmentation_map:
                f.segment_id[mb] = d.tree(SEGMENT_TREE, st.seg_tree_probs)
            f.skip[mb] = d.read(f.prob_skip_false) if f.mb_no_coeff_skip else 0
            if f.key:
                _intra_modes_key(d, f, mb, mx, my)
            elif d.read(f.prob_intra):
                f.is_inter[mb] = 1
                f.ref_frame[mb] = 1 + ((1 + d.read(f.prob_gf)) if d.read(f.prob_last) else 0)
                _inter_modes(d, f, st, mb, mx, my)
            else:
                f.ymode[mb] = d.tree(YMODE_TREE, st.ymode_prob)
                if f.ymode[mb] == B_PRED:
                    for b in range(16):
                        f.bmodes[mb, b] = d.tree(BMODE_TREE, BMODE_PROB)
                f.uvmode[mb] = d.tree(UV_MODE_TREE, st.uv_mode_prob)
            f.has_y2[mb] = int(not (f.ymode[mb] == B_PRED and not f.is_inter[mb]) and not (f.is_inter[mb] and f.mv_mode[mb] == MV_SPLIT))
            if f.skip[mb]:
                # 13: a skipped macroblock with a Y2 leaves the Y2 context alone only if it HAS no Y2; contexts of the rest clear
                if f.has_y2[mb]:
                    left_nz[:] = 0
                    above_nz[mx, :] = 0
                else:
                    left_nz[:8] = 0
                    above_nz[mx, :8] = 0
                continue
            _tokens(t, f, st, mb, mx, left_nz, above_nz)
    f.first_partition_overrun = d.overrun()
    f.token_overrun = [t.overrun() for t in tok]
    f.token_bytes_unread = [max(0, len(t.data) - t.pos) for t in tok]
    if saved is not None:
        st.coeff_probs, st.mv_probs, st.ymode_prob, st.uv_mode_prob = saved
    return f


def _intra_modes_key(d, f, mb, mx, my):
    f.ymode[mb] = d.tree(KF_YMODE_TREE, KF_YMODE_PROB)
    if f.ymode[mb] == B_PRED:
        for b in range(16):
            by, bx = b >> 2, b & 3
            above = _bmode_at(f, mx, my, bx, by - 1)
            left = _bmode_at(f, mx, my, bx - 1, by)
            f.bmodes[mb, b] = d.tree(BMODE_TREE, [int(x) for x in KF_BMODE_PROBS[above, left]])
    else:
        f.bmodes[mb, :] = {DC_PRED: B_DC, V_PRED: B_VE, H_PRED: B_HE, TM_PRED: B_TM}[int(f.ymode[mb])]   # 11.3: implied contexts
    f.uvmode[mb] = d.tree(UV_MODE_TREE, KF_UV_MODE_PROB)


def _bmode_at(f, mx, my, bx, by):
    if bx < 0:
        mx, bx = mx - 1, 3
    if by < 0:
        my, by = my - 1, 3
    if mx < 0 or my < 0:
        return B_DC
    return int(f.bmodes[my * f.mbw + mx, by * 4 + bx])


def _read_mv_component(d, p):
    if d.read(p[MV_IS_SHORT]):
        x = 0
        for i in range(3):
            x += d.read(p[MV_LONG + i]) << i
        for i in range(MV_LONG_BITS - 1, 3, -1):
            x += d.read(p[MV_LONG + i]) << i
        if not (x & 0xFFF0) or d.read(p[MV_LONG + 3]):
            x += 8
    else:
        x = d.tree(SMALL_MV_TREE, p[MV_SHORT:MV_SHORT + 7])
    return -x if (x and d.read(p[MV_SIGN])) else x


def _read_mv(d, st):
    r = _read_mv_component(d, st.mv_probs[0])
    c = _read_mv_component(d, st.mv_probs[1])
    return np.array([r, c], np.int32)


def _near_mvs(f, st, mb, mx, my):
    """18.3: census of the above, left and above-left macroblocks (weights 2, 2, 1): up to three distinct non-zero vectors
    with their counts, zero vectors counted in slot 0; returns ([best, nearest, near, -], the four mode-context counts)"""
    mv = [np.zeros(2, np.int32) for _ in range(4)]
    cnt = [0, 0, 0, 0]
    k = 0
    this_ref = int(f.ref_frame[mb])
    order = ((mx, my - 1, 2), (mx - 1, my, 2), (mx - 1, my - 1, 1))
    for n, (nx, ny, w) in enumerate(order):
        if nx < 0 or ny < 0:
            continue
        m = ny * f.mbw + nx
        if not f.is_inter[m]:
            continue
        v = f.mvs[m, 15].copy()
        if st.sign_bias[int(f.ref_frame[m])] != st.sign_bias[this_ref]:
            v = -v
        if v.any():
            if n == 0:
                k += 1
                mv[k] = v
                cnt[k] += w
            elif np.array_equal(v, mv[k]) and k > 0:
                cnt[k] += w
            else:
                k += 1
                mv[k] = v
                cnt[k] += w
        else:
            cnt[0] += w
    # CNT_SPLITMV slot: neighbours that are split (weights as above)
    if cnt[3] and np.array_equal(mv[k], mv[1]):
        cnt[1] += 1
    split = 0
    for n, (nx, ny, w) in enumerate(order):
        if nx < 0 or ny < 0:
            continue
        m = ny * f.mbw + nx
        if f.is_inter[m] and f.mv_mode[m] == MV_SPLIT:
            split += w
    cnt[3] = split
    if cnt[2] > cnt[1]:
        cnt[1], cnt[2] = cnt[2], cnt[1]
        mv[1], mv[2] = mv[2], mv[1]
    if cnt[1] >= cnt[0]:
        mv[0] = mv[1]
    return mv, cnt   # mv[0] best, mv[1] nearest, mv[2] near


def _clamp(v, mx, my, f):
    """18.3: nearest / near / best may move a macroblock at most one macroblock beyond the frame (quarter pixels)"""
    lo_r, hi_r = -((my + 1) * 16) << 2, ((f.mbh - my) * 16) << 2
    lo_c, hi_c = -((mx + 1) * 16) << 2, ((f.mbw - mx) * 16) << 2
    return np.array([min(max(int(v[0]), lo_r), hi_r), min(max(int(v[1]), lo_c), hi_c)], np.int32)


def _inter_modes(d, f, st, mb, mx, my):
    mv, cnt = _near_mvs(f, st, mb, mx, my)
    probs = [MODE_CONTEXTS[cnt[i]][i] for i in range(4)]
    mode = d.tree(MV_REF_TREE, probs)
    f.mv_mode[mb] = mode
    best = _clamp(mv[0], mx, my, f)
    if mode == MV_ZERO:
        f.mvs[mb, :] = 0
    elif mode == MV_NEAREST:
        f.mvs[mb, :] = _clamp(mv[1], mx, my, f)
    elif mode == MV_NEAR:
        f.mvs[mb, :] = _clamp(mv[2], mx, my, f)
    elif mode == MV_NEW:
        f.mvs[mb, :] = _read_mv(d, st) + best
    else:
        sp = d.tree(SPLIT_MV_TREE, SPLIT_MV_PROB)
        f.split[mb] = sp
        part = SPLIT_PARTITION[sp]
        done = {}
        for b in range(16):
            p = part[b]
            if p in done:
                f.mvs[mb, b] = done[p]
                continue
            by, bx = b >> 2, b & 3
            left = _sub_mv(f, mx, my, bx - 1, by, mb)
            above = _sub_mv(f, mx, my, bx, by - 1, mb)
            lez, aez, lea = not left.any(), not above.any(), np.array_equal(left, above)
            ctx = (4 if lez else 3) if lea else (2 if aez else (1 if lez else 0))
            sm = d.tree(SUB_MV_REF_TREE, SUB_MV_REF_PROB[ctx])
            if sm == LEFT4x4:
                v = left
            elif sm == ABOVE4x4:
                v = above
            elif sm == ZERO4x4:
                v = np.zeros(2, np.int32)
            else:
                v = _read_mv(d, st) + mv[0]      # the UNclamped best vector (libvpx decode_split_mv; RFC 5 dixie likewise)
            done[p] = v
            f.mvs[mb, b] = v


def _sub_mv(f, mx, my, bx, by, mb):
    if bx >= 0 and by >= 0:
        return f.mvs[mb, by * 4 + bx].copy()
    if bx < 0:
        mx, bx = mx - 1, 3
    if by < 0:
        my, by = my - 1, 3
    if mx < 0 or my < 0:
        return np.zeros(2, np.int32)
    m = my * f.mbw + mx
    if not f.is_inter[m]:
        return np.zeros(2, np.int32)
    return f.mvs[m, by * 4 + bx].copy()


def _tokens(t, f, st, mb, mx, left_nz, above_nz):
    """13: the 25 (or 24) blocks of a macroblock; contexts = whether the neighbouring block of the same kind had a non-zero"""
    if f.has_y2[mb]:
        blk, nz = _block(t, st, 1, int(left_nz[8] + above_nz[mx, 8]), 0)
        f.coeffs[mb, 24] = blk
        left_nz[8] = above_nz[mx, 8] = nz
        ytype, first = 0, 1
    else:
        ytype, first = 3, 0
    for b in range(16):
        by, bx = b >> 2, b & 3
        blk, nz = _block(t, st, ytype, int(left_nz[by] + above_nz[mx, bx]), first)
        f.coeffs[mb, b] = blk
        left_nz[by] = above_nz[mx, bx] = nz
    for pl in range(2):
        for b in range(4):
            by, bx = b >> 1, b & 1
            li, ai = 4 + 2 * pl + by, 4 + 2 * pl + bx
            blk, nz = _block(t, st, 2, int(left_nz[li] + above_nz[mx, ai]), 0)
            f.coeffs[mb, 16 + 4 * pl + b] = blk
            left_nz[li] = above_nz[mx, ai] = nz


def _block(t, st, plane_type, ctx, first):
    """one block's tokens; returns (coefficients in raster order, 1 if at least one token other than EOB was read)"""
    out = np.zeros(16, np.int32)
    i = first
    prev_zero = False
    c = ctx
    probs = st.coeff_probs[plane_type]
    while i < 16:
        p = probs[COEFF_BANDS[i], c]
        tokn = t.tree(COEFF_TREE, p, 2 if prev_zero else 0)   # no EOB right after a zero (13.2)
        if tokn == DCT_EOB:
            break
        if tokn == DCT_0:
            v = 0
        elif tokn <= DCT_4:
            v = tokn
        else:
            extra = 0
            for pr in CAT_PROBS[tokn]:
                extra = (extra << 1) | t.read(pr)
            v = CAT_BASE[tokn] + extra
        if v and t.flag():
            v = -v
        out[ZIGZAG[i]] = v
        prev_zero = v == 0
        c = 0 if v == 0 else (1 if abs(v) == 1 else 2)
        i += 1
    return out, int(i > first)
