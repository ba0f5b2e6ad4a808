"""Frame loop of the reference (`main`, vp8enc.cpp:351-488) in Python, for the parity tests.

The loop keeps the reference's sequencing -- frame-type state machine, prepare_segments_data per frame, key
frames through intra_transform, inter_transform, check_SSIM with its intra fallback, its filter-parameter update
and its "redo as key frame" decision, filter mask, loop filter -- and calls a *backend* through the method names
of the C ABI (`vp8oclenc_amd.api.Vp8Hip`; the tests run the CPU oracle through the same loop).  The production
loop is the native one, csrc/vp8_driver.cpp (include/vp8hip_driver.h); the tests hold the two against each other.
Not here, because outside the path: scene-change detection input (the caller passes
force_key), the header/MV entropy coder and the container.
"""
from __future__ import annotations

import numpy as np

from . import api


class InterPathDriver:
    def __init__(self, backend, width: int, height: int, gop_size: int = 150, altref_range: int = 5,
                 qi_min: int = 0, qi_max: int = 48, ssim_target: float = -1.0, download: bool = True,
                 check_ssim: bool = True, device_intra: bool = True, ref_mask: int = 3):
        self.be = backend
        self.W, self.H = width, height
        self.mbs = (width // 16) * (height // 16)
        self.gop = api.Gop(gop_size, altref_range)
        self.qi_min = min(qi_min, qi_max)
        self.lastqi, self.altrefqi = api.quantizer_ladders(qi_min, qi_max)
        self.ssim_target = ssim_target
        self.download = download
        self.check = check_ssim
        self.device_intra = device_intra     # False: key frames through the old stand-in (source planes = reconstruction)
        self.ref_mask = ref_mask             # bit 0 GOLDEN, bit 1 ALTREF may be searched (vp8drv_config.ref_mask; 0 = LAST only, BASELINE configs[1])
        self.inter_frames = 0
        self.key_frames = 0
        self.redone_as_key = 0
        self.last_key = None

    def segments_for(self, y: np.ndarray, is_key: bool, is_altref: bool, update_filter: bool = False) -> np.ndarray:
        reductor, sharp = api.loopfilter_strength(y)
        refqi = self.altrefqi if is_altref else self.lastqi
        self.sharpness = 7 if update_filter else sharp      # video.loop_filter_sharpness, which the frame header carries
        return api.prepare_segments_data(is_key, refqi, self.qi_min, reductor, sharp, update_filter, 7)

    def _key_frame(self, y, u, v):
        """prepare_segments_data + intra_transform (vp8enc.cpp:379-383 / :411-414 / :446-450), then the common tail."""
        if not self.device_intra:
            self.gop.key_coded()
            self.be.upload_last(y, u, v)
            self.gop.frame_done()
            self.key_frames += 1
            return None
        sd = self.segments_for(y, True, True)
        self.be.set_segments(sd)
        self.be.intra_transform()
        self.gop.key_coded()                                                    # intra_part.h:1091-1098
        if self.download:
            self.last_key = self.be.download_results(recon=True)
            self.last_key["segments"] = sd
            self.last_key["modes"] = self.be.download_intra()[0]
            self.last_key["sharpness"] = self.sharpness
        self.be.prepare_filter_mask(want_nz=False)                              # vp8enc.cpp:472
        self.be.loop_filter()                                                   # vp8enc.cpp:473
        self.gop.frame_done()
        self.key_frames += 1
        return None

    def encode_frame(self, y: np.ndarray, u: np.ndarray, v: np.ndarray, force_key: bool = False):
        """One iteration of the while-loop body.  Returns None for a key frame, else the frame's outputs."""
        g = self.gop.next()
        self.be.upload_current(y, u, v)                                        # vp8enc.cpp:386-388
        if g.current_is_key or force_key:
            return self._key_frame(y, u, v)
        sd = self.segments_for(y, False, bool(g.current_is_altref))             # vp8enc.cpp:419
        self.be.set_segments(sd)
        use_golden, use_altref = self.gop.inter_flags()                         # inter_part.h:103-104
        use_golden, use_altref = use_golden & (self.ref_mask & 1), use_altref & ((self.ref_mask >> 1) & 1)
        self.be.inter_transform(g.prev_is_golden, g.prev_is_altref, use_golden, use_altref)
        out = {"segments": sd, "use_golden": use_golden, "use_altref": use_altref, "is_altref": int(g.current_is_altref)}
        if self.check:
            replaced, new_ssim, min1 = self.be.check_ssim()                     # vp8enc.cpp:442, 231-263
            out.update(replaced=replaced, new_SSIM=new_ssim, min_SSIM=min1)
            if min1 > np.float32(0.95):                                         # :260-261
                sd = self.segments_for(y, False, bool(g.current_is_altref), update_filter=True)
                self.be.set_segments(sd)
                out["segments"] = sd
            if replaced > self.mbs // 6 or new_ssim < np.float32(self.ssim_target):   # :443-453: redo as intra
                self.redone_as_key += 1
                return self._key_frame(y, u, v)
        if self.download:
            out.update(self.be.download_results(recon=True))                    # vp8enc.cpp:422-440, after the fallback
            if self.check:
                out["modes"], out["is_inter"] = self.be.download_intra()
        out["sharpness"] = self.sharpness
        self.be.prepare_filter_mask(want_nz=False)                             # vp8enc.cpp:472
        self.be.loop_filter()                                                   # vp8enc.cpp:473
        self.gop.frame_done()
        self.inter_frames += 1
        return out
