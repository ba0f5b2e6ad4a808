"""CPU backend of vp8oclenc_amd.ref_shard.RefShardDriver for the gloo tests: the restatement's per-stage functions
(oracle/vp8_oracle.c through oracle_lib.Stages) sequenced like include/vp8hip.h's vp8hip_inter_search /
vp8hip_inter_finish, state in numpy, exchanges as numpy parcels over object collectives (ref_shard.ThreadGroup; tests/torch_transport.py
over gloo).  Test infrastructure."""
import numpy as np

from oracle_lib import Oracle, oracle_intra
from pipeline import pyramid


class OracleRefBackend:
    def __init__(self, W, H):
        self.W, self.H = W, H
        self.mbs = (W // 16) * (H // 16)
        self.b8 = self.mbs * 4
        self.st = Oracle.stages()
        self.refs = [None, None, None]          # LAST, GOLDEN, ALTREF: (Y, U, V)
        self.net = [np.zeros((self.b8, 2), np.int16) for _ in range(3)]
        self.bd = [np.full(self.b8, 0x7FFFFFFF, np.int32) for _ in range(3)]

    def upload_current(self, y, u, v):
        self.cur = tuple(np.ascontiguousarray(p) for p in (y, u, v))

    def set_segments(self, sd):
        self.sd = np.ascontiguousarray(sd, np.int32).reshape(-1).copy()

    def intra_transform(self):
        k = oracle_intra().intra_transform(self.cur, self.sd)
        self.MB, self.parts, self.seg = k["MB_coeffs"], k["MB_parts"], k["MB_segment_id"]
        self.recon = [k["recon_Y"], k["recon_U"], k["recon_V"]]

    def inter_search(self, prev_is_golden, prev_is_altref, use_golden, use_altref, mask):
        if prev_is_golden:
            self.refs[1] = self.refs[0]
        if prev_is_altref:
            self.refs[2] = self.refs[0]
        W, H, st = self.W, self.H, self.st
        use = [1, int(use_golden), int(use_altref)]
        cur_pyr = pyramid(st, self.cur[0])
        for r in range(3):
            if not (use[r] and (mask >> r) & 1):
                continue
            net = [np.zeros((self.b8, 2), np.int16), np.zeros((self.b8, 2), np.int16)]
            bd = np.full(self.b8, 0x7FFFFFFF, np.int32)
            ref_pyr = pyramid(st, self.refs[r][0])
            src = 0
            for l in range(4, -1, -1):
                st.luma_search_1step(cur_pyr[l], ref_pyr[l], net[src], net[src ^ 1], (W // 16) * 2, W >> l, H >> l, 1 << l)
                src ^= 1
            st.luma_search_2step(self.cur[0], np.ascontiguousarray(self.refs[r][0]), net[1], net[0], bd, W, H)
            self.net[r], self.bd[r] = net[0], bd

    def export_search(self, ref):
        return np.stack([self.net[ref].view(np.int32).reshape(-1), self.bd[ref]]).copy()

    def import_search(self, ref, t):
        a = np.asarray(t)
        self.net[ref] = np.ascontiguousarray(a[0]).view(np.int16).reshape(-1, 2).copy()
        self.bd[ref] = np.ascontiguousarray(a[1]).copy()

    def inter_finish(self, use_golden, use_altref):
        W, H, st, mbs = self.W, self.H, self.st, self.mbs
        use = [1, int(use_golden), int(use_altref)]
        net = [self.net[r] if use[r] else np.zeros((self.b8, 2), np.int16) for r in range(3)]
        bd = [self.bd[r] if use[r] else np.full(self.b8, 0x7FFFFFFF, np.int32) for r in range(3)]
        MB_ref, MB_vec = np.zeros(mbs, np.int32), np.zeros((mbs, 4, 2), np.int16)
        parts, ssim = np.zeros(mbs, np.int32), np.zeros(mbs, np.float32)
        st.select_reference(net[0], net[1], net[2], bd[0], bd[1], bd[2], MB_ref, MB_vec, W, H, use[1], use[2])
        st.pack_8x8_into_16x16(MB_vec, parts, ssim, mbs)
        planes = [(self.cur[0], W, H), (self.cur[1], W // 2, H // 2), (self.cur[2], W // 2, H // 2)]
        pred = [np.zeros_like(p[0]) for p in planes]
        resid = [np.zeros(p[0].shape, np.int16) for p in planes]
        recon = [np.zeros_like(p[0]) for p in planes]
        for r in range(3):
            if use[r]:
                for p, (pl, w, h) in enumerate(planes):
                    st.prepare_predictors_and_residual(np.ascontiguousarray(pl), np.ascontiguousarray(self.refs[r][p]), pred[p], resid[p],
                                                       MB_ref, MB_vec, w, h, p, r)
        MB = np.zeros((mbs, 25, 16), np.int16)
        seg = np.zeros(mbs, np.int32)
        metric = [np.zeros(mbs, np.float32) for _ in range(3)]
        for s in range(3, -1, -1):
            for p, (pl, w, h) in enumerate(planes):
                st.dct4x4(resid[p], MB, seg, parts, ssim, w, h, self.sd, s, -1.0, p)
            st.wht4x4_iwht4x4(MB, seg, parts, self.sd, s, mbs)
            for p, (pl, w, h) in enumerate(planes):
                st.idct4x4(recon[p], pred[p], MB, seg, parts, w, h, self.sd, s, p)
            for p, (pl, w, h) in enumerate(planes):
                st.count_SSIM(np.ascontiguousarray(pl), recon[p], seg, metric[p], w, h, s, 16 if p == 0 else 8)
            st.gather_SSIM(metric[0], metric[1], metric[2], ssim, mbs)
        self.MB, self.parts, self.seg, self.recon = MB, parts, seg, recon
        self.out = dict(MB_parts=parts, MB_reference_frame=MB_ref, MB_vectors=MB_vec, MB_coeffs=MB, MB_segment_id=seg, MB_SSIM=ssim,
                        prefilter_Y=recon[0].copy(), prefilter_U=recon[1].copy(), prefilter_V=recon[2].copy())

    def download_results(self, recon=True):
        return {k: v.copy() for k, v in self.out.items()}

    def prepare_filter_mask(self, want_nz=True):
        self.nz, self.mask = np.zeros(self.mbs, np.int32), np.zeros(self.mbs, np.int32)
        self.st.prepare_filter_mask(self.MB, self.nz, self.parts, self.mask, self.W, self.H)

    def loop_filter(self):
        W, H = self.W, self.H
        f = [r.copy() for r in self.recon]
        self.st.loop_filter_frame(f[0], self.seg, self.mask, self.sd, W, H, 16)
        self.st.loop_filter_frame(f[1], self.seg, self.mask, self.sd, W // 2, H // 2, 8)
        self.st.loop_filter_frame(f[2], self.seg, self.mask, self.sd, W // 2, H // 2, 8)
        self.refs[0] = tuple(f)

    def download_last(self):
        return self.refs[0]

    def export_last(self):
        return np.concatenate([p.reshape(-1) for p in self.refs[0]]).copy()

    def import_last(self, t):
        a = np.asarray(t)
        n = self.W * self.H
        self.refs[0] = (a[:n].reshape(self.H, self.W).copy(), a[n:n + n // 4].reshape(self.H // 2, self.W // 2).copy(),
                        a[n + n // 4:].reshape(self.H // 2, self.W // 2).copy())

    def close(self):
        pass
