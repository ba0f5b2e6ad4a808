"""Stage-by-stage inter-frame pipeline over a `Stages` object (oracle restatement or reference kernels).

Mirrors the enqueue order of src/inter_part.h:96-384 and src/loop_filter.h so that the SAME
driver code can run the reference's own kernels (ref_*) and the restatement (vp8o_*) and the
outputs of every stage can be compared.  Test infrastructure.
"""
from __future__ import annotations

import ast
import json

import numpy as np


def load_meta(z) -> dict:
    """the one-line description stored with a golden fixture: JSON (gfx950 fixtures) or a dict literal (x86 fixtures)"""
    text = str(z["meta"])
    try:
        return json.loads(text)
    except ValueError:
        return ast.literal_eval(text)


def default_segments(qi=(12, 24, 36, 48), lf_levels=(6, 10, 14, 20), sharp=0, inter=True) -> np.ndarray:
    """A plausible segment_data[4] (the real one comes from the host mirror, vp8oclenc_amd.host)."""
    sd = np.zeros((4, 11), np.int32)
    for i in range(4):
        sd[i, 0] = qi[i]
        lvl = lf_levels[i]
        il = lvl
        if sharp:
            il >>= 2 if sharp > 4 else 1
            il = min(il, 9 - sharp)
        il = max(il, 1)
        sd[i, 6] = lvl
        sd[i, 9] = il
        sd[i, 7] = (lvl + 2) * 2 + il
        sd[i, 8] = lvl * 2 + il
        sd[i, 10] = 3 if lvl >= 40 else (2 if lvl >= 20 else (1 if lvl >= 15 else 0))
    sd[0, 1] = 15
    sd[0, 4] = -15 if inter else 0
    sd[0, 5] = -15 if inter else 0
    return sd


def pyramid(st, y: np.ndarray) -> list[np.ndarray]:
    levels = [np.ascontiguousarray(y)]
    for _ in range(4):
        s = levels[-1]
        d = np.zeros((s.shape[0] // 2, s.shape[1] // 2), np.uint8)
        st.downsample_x2(s, d, s.shape[1], s.shape[0])
        levels.append(d)
    return levels


def run_inter_frame(st, cur, refs, sd, use_golden, use_altref, ssim_target=-1.0, keep=None) -> dict:
    """cur = (Y,U,V); refs = [(Y,U,V)]*3 (LAST, GOLDEN, ALTREF).  Returns every stage output."""
    Y, U, V = cur
    H, W = Y.shape
    mbs = (W // 16) * (H // 16)
    b8 = mbs * 4
    net_width = (W // 16) * 2
    sd = np.ascontiguousarray(sd, np.int32).reshape(-1)
    use = [1, int(use_golden), int(use_altref)]
    out = {}
    cur_pyr = pyramid(st, Y)
    out["cur_pyr"] = cur_pyr
    nets1, bdiffs = [], []
    for r in range(3):
        net = [np.zeros((b8, 2), np.int16), np.zeros((b8, 2), np.int16)]
        bd = np.full(b8, 0x7FFFFFFF, np.int32)
        if use[r]:
            ref_pyr = pyramid(st, refs[r][0])
            out[f"ref{r}_pyr"] = ref_pyr
            src = 0
            for l in range(4, -1, -1):
                st.luma_search_1step(cur_pyr[l], ref_pyr[l], net[src], net[src ^ 1], net_width, W >> l, H >> l, 1 << l)
                out[f"net_r{r}_l{l}"] = net[src ^ 1].copy()
                src ^= 1
            st.luma_search_2step(Y, np.ascontiguousarray(refs[r][0]), net[1], net[0], bd, W, H)
        out[f"net1_r{r}"] = net[0].copy()
        out[f"bdiff_r{r}"] = bd.copy()
        nets1.append(net[0])
        bdiffs.append(bd)
    MB_ref = np.zeros(mbs, np.int32)
    MB_vec = np.zeros((mbs, 4, 2), np.int16)
    MB_parts = np.zeros(mbs, np.int32)
    MB_SSIM = np.zeros(mbs, np.float32)
    st.select_reference(nets1[0], nets1[1], nets1[2], bdiffs[0], bdiffs[1], bdiffs[2], MB_ref, MB_vec, W, H,
                        use[1], use[2])
    st.pack_8x8_into_16x16(MB_vec, MB_parts, MB_SSIM, mbs)
    planes = [(Y, W, H), (U, W // 2, H // 2), (V, W // 2, H // 2)]
    pred = [np.zeros_like(p[0]) for p in planes]
    resid = [np.zeros(p[0].shape, np.int16) for p in planes]
    recon = [np.zeros_like(p[0]) for p in planes]
    for r in range(3):
        if not use[r]:
            continue
        for p, (pl, w, h) in enumerate(planes):
            st.prepare_predictors_and_residual(np.ascontiguousarray(pl), np.ascontiguousarray(refs[r][p]), pred[p],
                                               resid[p], MB_ref, MB_vec, w, h, p, r)
    MB = keep["MB_coeffs"].copy() if keep else np.zeros((mbs, 25, 16), np.int16)
    MB_seg = keep["MB_segment_id"].copy() if keep else np.zeros(mbs, np.int32)
    metric = [np.zeros(mbs, np.float32) for _ in range(3)]
    for seg in range(3, -1, -1):
        for p, (pl, w, h) in enumerate(planes):
            st.dct4x4(resid[p], MB, MB_seg, MB_parts, MB_SSIM, w, h, sd, seg, ssim_target, p)
        st.wht4x4_iwht4x4(MB, MB_seg, MB_parts, sd, seg, mbs)
        for p, (pl, w, h) in enumerate(planes):
            st.idct4x4(recon[p], pred[p], MB, MB_seg, MB_parts, w, h, sd, seg, p)
        st.count_SSIM(np.ascontiguousarray(Y), recon[0], MB_seg, metric[0], W, H, seg, 16)
        st.count_SSIM(np.ascontiguousarray(U), recon[1], MB_seg, metric[1], W // 2, H // 2, seg, 8)
        st.count_SSIM(np.ascontiguousarray(V), recon[2], MB_seg, metric[2], W // 2, H // 2, seg, 8)
        st.gather_SSIM(metric[0], metric[1], metric[2], MB_SSIM, mbs)
    nz = np.zeros(mbs, np.int32)
    mask = np.zeros(mbs, np.int32)
    st.prepare_filter_mask(MB, nz, MB_parts, mask, W, H)
    filt = [r.copy() for r in recon]
    st.loop_filter_frame(filt[0], MB_seg, mask, sd, W, H, 16)
    st.loop_filter_frame(filt[1], MB_seg, mask, sd, W // 2, H // 2, 8)
    st.loop_filter_frame(filt[2], MB_seg, mask, sd, W // 2, H // 2, 8)
    out.update(MB_reference_frame=MB_ref, MB_vectors=MB_vec, MB_parts=MB_parts, MB_SSIM=MB_SSIM, MB_coeffs=MB,
               MB_segment_id=MB_seg, MB_non_zero_coeffs=nz, mb_mask=mask,
               pred_Y=pred[0], pred_U=pred[1], pred_V=pred[2], resid_Y=resid[0], resid_U=resid[1], resid_V=resid[2],
               prefilter_Y=recon[0], prefilter_U=recon[1], prefilter_V=recon[2],
               recon_Y=filt[0], recon_U=filt[1], recon_V=filt[2])
    return out
