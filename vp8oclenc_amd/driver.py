"""Frame loop of the reference (`main`, vp8enc.cpp:351-488) reduced to the inter-frame path.

The loop keeps the reference's sequencing -- frame-type state machine, prepare_segments_data per
frame, inter_transform, result read-back, check_SSIM's filter-parameter update, filter mask, loop
filter -- and calls a *backend* through the method names of the C ABI (`vp8oclenc_amd.api.Vp8Hip`;
the tests run the CPU oracle through the same loop).  What is NOT here, because it is outside the
path: intra coding of key frames, the per-macroblock intra fallback,
scene-change detection and the entropy coder.  A key frame is therefore "coded" by a stand-in that
hands the source planes over as its reconstruction (`vp8hip_upload_last`), which is all the inter
path needs from a key frame: a LAST/GOLDEN/ALTREF reference.
"""
from __future__ import annotations

import numpy as np

from . import api


class InterPathDriver:
    def __init__(self, backend, width: int, height: int, gop_size: int = 150, altref_range: int = 5,
                 qi_min: int = 0, qi_max: int = 48, ssim_target: float = -1.0, download: bool = True):
        self.be = backend
        self.W, self.H = width, height
        self.mbs = (width // 16) * (height // 16)
        self.gop = api.Gop(gop_size, altref_range)
        self.qi_min = min(qi_min, qi_max)
        self.lastqi, self.altrefqi = api.quantizer_ladders(qi_min, qi_max)
        self.ssim_target = ssim_target
        self.download = download
        self.inter_frames = 0
        self.key_frames = 0

    def segments_for(self, y: np.ndarray, is_key: bool, is_altref: bool, update_filter: bool = False) -> np.ndarray:
        reductor, sharp = api.loopfilter_strength(y)
        refqi = self.altrefqi if is_altref else self.lastqi
        return api.prepare_segments_data(is_key, refqi, self.qi_min, reductor, sharp, update_filter, 7)

    def encode_frame(self, y: np.ndarray, u: np.ndarray, v: np.ndarray, force_key: bool = False):
        """One iteration of the while-loop body.  Returns None for a key frame, else the frame's outputs."""
        g = self.gop.next()
        if g.current_is_key or force_key:
            # stand-in for intra_transform() (intra_part.h:1089-1128): out of scope, see module docstring
            self.gop.key_coded()
            self.be.upload_last(y, u, v)
            self.gop.frame_done()
            self.key_frames += 1
            return None
        self.be.upload_current(y, u, v)                                        # vp8enc.cpp:386-388
        sd = self.segments_for(y, False, bool(g.current_is_altref))             # vp8enc.cpp:419
        self.be.set_segments(sd)
        use_golden, use_altref = self.gop.inter_flags()                         # inter_part.h:103-104
        self.be.inter_transform(g.prev_is_golden, g.prev_is_altref, use_golden, use_altref)
        out = {"segments": sd, "use_golden": use_golden, "use_altref": use_altref}
        if self.download:
            res = self.be.download_results(recon=True)                          # vp8enc.cpp:422-440
            out.update(res)
            # check_SSIM, vp8enc.cpp:231-263: only its filter-parameter update is part of this path
            min1 = float(res["MB_SSIM"].min()) if self.mbs else 2.0
            out["new_SSIM"] = float(res["MB_SSIM"].astype(np.float32).sum(dtype=np.float32) / np.float32(self.mbs))
            if min1 > 0.95:
                sd = self.segments_for(y, False, bool(g.current_is_altref), update_filter=True)
                self.be.set_segments(sd)
                out["segments"] = sd
        self.be.prepare_filter_mask(want_nz=False)                             # vp8enc.cpp:472
        self.be.loop_filter()                                                   # vp8enc.cpp:473
        self.gop.frame_done()
        self.inter_frames += 1
        return out
