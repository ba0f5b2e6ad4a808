"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, where it was built, for the
reference's own kernels compiled to x86 (oracle/_ref/libvp8ref.so).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
Both libraries export the same per-stage functions (prefix vp8o_ / ref_), so `Stages` wraps either.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libvp8ref.so")
REF_HOST_SO = os.path.join(ORACLE_DIR, "_ref", "libvp8refhost.so")
REF_CL_SO = os.path.join(ORACLE_DIR, "_ref", "libvp8ref_cl.so")   # the reference's kernels on gfx950 through OpenCL (ref_cl_driver.c)

SD_INTS = 11

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
ci = C.c_int

_STAGES = {
    "downsample_x2": [u8p, u8p, ci, ci],
    "luma_search_1step": [u8p, u8p, i16p, i16p, ci, ci, ci, ci],
    "luma_search_2step": [u8p, u8p, i16p, i16p, i32p, ci, ci],
    "select_reference": [i16p, i16p, i16p, i32p, i32p, i32p, i32p, i16p, ci, ci, ci, ci],
    "pack_8x8_into_16x16": [i16p, i32p, f32p, ci],
    "prepare_predictors_and_residual": [u8p, u8p, u8p, i16p, i32p, i16p, ci, ci, ci, ci],
    "dct4x4": [i16p, i16p, i32p, i32p, f32p, ci, ci, i32p, ci, C.c_float, ci],
    "wht4x4_iwht4x4": [i16p, i32p, i32p, i32p, ci, ci],
    "idct4x4": [u8p, u8p, i16p, i32p, i32p, ci, ci, i32p, ci, ci],
    "count_SSIM": [u8p, u8p, i32p, f32p, ci, ci, ci, ci],
    "gather_SSIM": [f32p, f32p, f32p, f32p, ci],
    "prepare_filter_mask": [i16p, i32p, i32p, i32p, ci, ci],
    "loop_filter_frame": [u8p, i32p, i32p, i32p, ci, ci, ci],
    # coefficient entropy stage (src/CPU_kernels.cl:347-778), all partitions per call
    "count_probs": [i16p, i32p, i32p, u32p, u32p, u8p, ci, ci, ci],
    "num_div_denom": [u32p, u32p, ci],
    "encode_coefficients": [i16p, i32p, i32p, u8p, i32p, u8p, u32p, ci, ci, ci, ci],
}


def build_oracle(force: bool = False) -> str:
    """Compile oracle/liboracle.so (and oracle/_ref when /root/reference is present)."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("vp8_oracle.c", "vp8_entropy_oracle.c", "vp8_intra_oracle.c", "vp8_oracle.h")]
    stale = (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(f) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    drvs = [(REF_SO, os.path.join(ORACLE_DIR, "ref_driver.c")), (REF_HOST_SO, os.path.join(ORACLE_DIR, "ref_host_driver.cpp"))]
    if os.path.isdir("/root/reference/src") and (force or any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(d) for so, d in drvs)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
    # the reference's own main() against libvp8hip.so (oracle/ref_main/build.sh; needs the library: skipped where it is not built yet)
    main_sh = os.path.join(ORACLE_DIR, "ref_main", "build.sh")
    main_bin = os.path.join(ORACLE_DIR, "_ref", "vp8oclenc_hip")
    lib = os.path.join(os.path.dirname(ORACLE_DIR), "vp8oclenc_amd", "libvp8hip.so")
    deps = [main_sh, os.path.join(ORACLE_DIR, "ref_main", "vp8hip_drop_in.h"), lib]
    if os.path.isdir("/root/reference/src") and os.path.exists(lib) and (force or not os.path.exists(main_bin) or os.path.getmtime(main_bin) < max(os.path.getmtime(d) for d in deps)):
        subprocess.check_call(["sh", main_sh], stdout=subprocess.DEVNULL)
    return ORACLE_SO


class Stages:
    """Per-kernel entry points of one library (prefix 'vp8o_' = restatement, 'ref_' = reference kernels)."""

    def __init__(self, lib: C.CDLL, prefix: str):
        self.lib, self.prefix = lib, prefix
        for name, argtypes in _STAGES.items():
            fn = getattr(lib, prefix + name)
            fn.argtypes = argtypes
            fn.restype = None
            setattr(self, name, fn)


class Results(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "MB_parts", "MB_reference_frame", "MB_vectors", "MB_coeffs", "MB_segment_id", "MB_SSIM",
        "MB_non_zero_coeffs", "mb_mask", "recon_Y", "recon_U", "recon_V",
        "prefilter_Y", "prefilter_U", "prefilter_V")]


def alloc_results(W: int, H: int) -> dict:
    mbs = (W // 16) * (H // 16)
    return {
        "MB_parts": np.zeros(mbs, np.int32), "MB_reference_frame": np.zeros(mbs, np.int32),
        "MB_vectors": np.zeros((mbs, 4, 2), np.int16), "MB_coeffs": np.zeros((mbs, 25, 16), np.int16),
        "MB_segment_id": np.zeros(mbs, np.int32), "MB_SSIM": np.zeros(mbs, np.float32),
        "MB_non_zero_coeffs": np.zeros(mbs, np.int32), "mb_mask": np.zeros(mbs, np.int32),
        "recon_Y": np.zeros((H, W), np.uint8), "recon_U": np.zeros((H // 2, W // 2), np.uint8),
        "recon_V": np.zeros((H // 2, W // 2), np.uint8),
        "prefilter_Y": np.zeros((H, W), np.uint8), "prefilter_U": np.zeros((H // 2, W // 2), np.uint8),
        "prefilter_V": np.zeros((H // 2, W // 2), np.uint8),
    }


class Oracle:
    """Whole-frame driver of the restatement, with the method names of vp8oclenc_amd.api.Vp8Hip."""

    _lib = None

    @classmethod
    def lib(cls) -> C.CDLL:
        if cls._lib is None:
            build_oracle()
            # libgomp reads these once, when the library is loaded.  Tests use a modest team (the many
            # small parallel regions of the restatement crawl when 128 spinning threads share a box);
            # bench.py's cpu_baseline leg sets OMP_NUM_THREADS itself before calling in here.
            os.environ.setdefault("OMP_WAIT_POLICY", "passive")
            os.environ.setdefault("OMP_NUM_THREADS", str(min(8, len(os.sched_getaffinity(0)))))
            lib = C.CDLL(ORACLE_SO)
            lib.vp8o_create.restype = C.c_void_p
            lib.vp8o_create.argtypes = [ci, ci, C.c_float]
            lib.vp8o_destroy.argtypes = [C.c_void_p]
            lib.vp8o_upload_last.argtypes = [C.c_void_p, u8p, u8p, u8p]
            lib.vp8o_set_segments.argtypes = [C.c_void_p, i32p]
            lib.vp8o_inter_transform.argtypes = [C.c_void_p, u8p, u8p, u8p, ci, ci, ci, ci, C.POINTER(Results)]
            lib.vp8o_loop_filter.argtypes = [C.c_void_p, C.POINTER(Results)]
            lib.vp8o_upload_mb_data.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.vp8o_upload_recon.argtypes = [C.c_void_p, u8p, u8p, u8p]
            lib.vp8o_debug_net.restype = C.POINTER(C.c_int16)
            lib.vp8o_debug_net.argtypes = [C.c_void_p, ci, ci]
            lib.vp8o_debug_bdiff.restype = C.POINTER(C.c_int32)
            lib.vp8o_debug_bdiff.argtypes = [C.c_void_p, ci]
            lib.vp8o_debug_pyramid.restype = C.POINTER(C.c_uint8)
            lib.vp8o_debug_pyramid.argtypes = [C.c_void_p, ci, ci]
            lib.vp8o_num_threads.restype = ci
            lib.vp8o_weight.argtypes = [i32p]
            lib.vp8o_weight.restype = ci
            cls._lib = lib
        return cls._lib

    @classmethod
    def stages(cls) -> Stages:
        return Stages(cls.lib(), "vp8o_")

    def __init__(self, W: int, H: int, ssim_target: float = -1.0):
        self.W, self.H = W, H
        self.mbs = (W // 16) * (H // 16)
        self.ssim_target = float(ssim_target)
        self.h = self.lib().vp8o_create(W, H, ssim_target)
        if not self.h:
            raise ValueError("vp8o_create failed (size must be a multiple of 16)")

    def close(self):
        if self.h:
            self.lib().vp8o_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upload_last(self, y, u, v):
        self.lib().vp8o_upload_last(self.h, y, u, v)

    def set_segments(self, sd):
        self._sd = np.ascontiguousarray(sd, np.int32).reshape(-1).copy()
        self.lib().vp8o_set_segments(self.h, self._sd)

    # -- host intra path (vp8_intra_oracle.c), with the ABI's method names ------------------------
    def intra_transform(self):
        k = oracle_intra().intra_transform(self._cur, self._sd)
        self._out = alloc_results(self.W, self.H)
        for key in ("MB_coeffs", "MB_parts", "MB_segment_id"):
            self._out[key][...] = k[key]
        for p in "YUV":
            self._out["prefilter_" + p][...] = k["recon_" + p]
        self._intra = (k["modes"], np.zeros(self.mbs, np.int32))
        self.upload_recon(k["recon_Y"], k["recon_U"], k["recon_V"])
        self.upload_mb_data(k["MB_coeffs"], k["MB_parts"], k["MB_segment_id"])

    def check_ssim(self):
        o = self._out
        inter = {"recon_Y": o["prefilter_Y"], "recon_U": o["prefilter_U"], "recon_V": o["prefilter_V"], "MB_coeffs": o["MB_coeffs"],
                 "MB_parts": o["MB_parts"], "MB_segment_id": o["MB_segment_id"], "MB_SSIM": o["MB_SSIM"]}
        r = oracle_intra().check_ssim(self._cur, self._sd, self.ssim_target, inter)
        for key in ("MB_coeffs", "MB_parts", "MB_segment_id", "MB_SSIM"):
            o[key][...] = r[key]
        for p in "YUV":
            o["prefilter_" + p][...] = r["recon_" + p]
        self._intra = (r["modes"], r["is_inter"])
        if r["replaced"]:
            self.upload_recon(r["recon_Y"], r["recon_U"], r["recon_V"])
            self.upload_mb_data(r["MB_coeffs"], r["MB_parts"], r["MB_segment_id"])
        return r["replaced"], np.float32(r["new_SSIM"]), np.float32(r["min_SSIM"])

    def download_intra(self):
        return self._intra

    # -- same call sequence as the C ABI (include/vp8hip.h) ------------------------------------
    def upload_current(self, y, u, v):
        self._cur = (np.ascontiguousarray(y), np.ascontiguousarray(u), np.ascontiguousarray(v))

    def inter_transform(self, prev_is_golden, prev_is_altref, use_golden, use_altref):
        self._out = alloc_results(self.W, self.H)
        r = Results(**{k: a.ctypes.data for k, a in self._out.items()})
        y, u, v = self._cur
        self.lib().vp8o_inter_transform(self.h, y, u, v, int(prev_is_golden), int(prev_is_altref), int(use_golden),
                                        int(use_altref), C.byref(r))

    def download_results(self, recon: bool = True) -> dict:
        keys = ["MB_parts", "MB_reference_frame", "MB_vectors", "MB_coeffs", "MB_segment_id", "MB_SSIM"]
        if recon:
            keys += ["prefilter_Y", "prefilter_U", "prefilter_V"]
        return {k: self._out[k].copy() for k in keys}

    def upload_mb_data(self, coeffs=None, parts=None, seg=None):
        p = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data
        self._keep = (coeffs, parts, seg)
        self.lib().vp8o_upload_mb_data(self.h, p(coeffs), p(parts), p(seg))

    def upload_recon(self, y, u, v):
        self.lib().vp8o_upload_recon(self.h, np.ascontiguousarray(y), np.ascontiguousarray(u), np.ascontiguousarray(v))

    def prepare_filter_mask(self, want_nz: bool = True):
        # the restatement computes mask and counts inside vp8o_loop_filter (see filter_outputs)
        return None

    def loop_filter(self):
        res = alloc_results(self.W, self.H)
        r = Results(**{k: a.ctypes.data for k, a in res.items()})
        self.lib().vp8o_loop_filter(self.h, C.byref(r))
        self._lf = res

    def filter_outputs(self) -> dict:
        return {k: self._lf[k] for k in ("MB_non_zero_coeffs", "mb_mask", "recon_Y", "recon_U", "recon_V")}

    def download_last(self):
        return self._lf["recon_Y"], self._lf["recon_U"], self._lf["recon_V"]

    def synchronize(self):
        pass

    def net(self, ref: int, which: int) -> np.ndarray:
        p = self.lib().vp8o_debug_net(self.h, ref, which)
        return np.ctypeslib.as_array(p, shape=(self.mbs * 4, 2)).copy()

    def bdiff(self, ref: int) -> np.ndarray:
        p = self.lib().vp8o_debug_bdiff(self.h, ref)
        return np.ctypeslib.as_array(p, shape=(self.mbs * 4,)).copy()

    def pyramid(self, ref: int, level: int) -> np.ndarray:
        p = self.lib().vp8o_debug_pyramid(self.h, ref, level)
        return np.ctypeslib.as_array(p, shape=(self.H >> level, self.W >> level)).copy()


def ref_stages() -> Stages | None:
    """The reference's own kernels (x86 build) or None when oracle/_ref was not built."""
    if not os.path.exists(REF_SO):
        try:
            build_oracle()
        except Exception:
            return None
    if not os.path.exists(REF_SO):
        return None
    return Stages(C.CDLL(REF_SO), "ref_")


def ref_cl_stages() -> Stages | None:
    """The reference's own kernels compiled by AMD's OpenCL compiler for gfx950 and run on the GPU through the vendor's
    OpenCL runtime (oracle/ref_cl_driver.c + the code objects oracle/build_ref.sh leaves in oracle/_ref), or None where
    the build is absent or there is no OpenCL GPU device (this container)."""
    if not (os.path.exists(REF_CL_SO) and os.path.exists(os.path.join(ORACLE_DIR, "_ref", "ref_gpu_kernels_gfx950.co"))):
        return None
    try:
        lib = C.CDLL(REF_CL_SO)
    except OSError:
        return None
    lib.ref_cl_init.restype = ci
    if lib.ref_cl_init() != 0:
        return None
    lib.ref_cl_device_name.restype = C.c_char_p
    st = Stages(lib, "ref_")
    st.device_name = lib.ref_cl_device_name().decode()
    st.image_support = int(lib.ref_cl_image_support())   # 0 on MI355X: see oracle/ref_image_as_buffer.cl
    return st


class Intra:
    """The host intra path (key frames, check_SSIM's intra fallback) of one library: prefix 'vp8o_' = the restatement
    oracle/vp8_intra_oracle.c, 'ref_' = the reference's own code (oracle/_ref/libvp8refhost.so)."""

    def __init__(self, lib: C.CDLL, prefix: str):
        self.prefix = prefix
        self._key = getattr(lib, prefix + "intra_transform")
        self._key.argtypes = [ci, ci, u8p, u8p, u8p, i32p, u8p, u8p, u8p, i16p, i32p, i32p, i32p]
        self._key.restype = None
        self._chk = getattr(lib, prefix + "check_ssim")
        self._chk.argtypes = [ci, ci, C.c_float, u8p, u8p, u8p, i32p, u8p, u8p, u8p, i16p, i32p, i32p, f32p, i32p, i32p,
                              C.POINTER(ci), C.POINTER(C.c_float), C.c_void_p]
        self._chk.restype = None
        self._ssim = getattr(lib, prefix + ("count_ssim_16x16"))
        self._ssim.argtypes = [u8p, u8p, u8p, ci, u8p, u8p, u8p, ci]
        self._ssim.restype = C.c_float
        self._pick = getattr(lib, prefix + "pick_luma_predictor")
        self._pick.argtypes = [u8p, u8p, i16p, i16p, i16p, ci]
        self._pick.restype = ci

    def intra_transform(self, cur, sd):
        """Key frame.  cur = (Y, U, V) tight planes; returns dict(recon_Y/U/V, MB_coeffs, MB_parts, MB_segment_id, modes)."""
        y, u, v = (np.ascontiguousarray(p) for p in cur)
        H, W = y.shape
        mbs = (W // 16) * (H // 16)
        o = {"recon_Y": np.zeros_like(y), "recon_U": np.zeros_like(u), "recon_V": np.zeros_like(v),
             "MB_coeffs": np.zeros((mbs, 25, 16), np.int16), "MB_parts": np.zeros(mbs, np.int32),
             "MB_segment_id": np.zeros(mbs, np.int32), "modes": np.zeros((mbs, 16), np.int32)}
        self._key(W, H, y, u, v, np.ascontiguousarray(sd, np.int32).reshape(-1), o["recon_Y"], o["recon_U"], o["recon_V"],
                  o["MB_coeffs"], o["MB_parts"], o["MB_segment_id"], o["modes"])
        return o

    def check_ssim(self, cur, sd, ssim_target, inter):
        """check_SSIM on the results of an inter frame (dict with prefilter recon_Y/U/V, MB_coeffs, MB_parts,
        MB_segment_id, MB_SSIM); returns the updated copies + is_inter, modes, replaced, new_SSIM, filter_updated."""
        y, u, v = (np.ascontiguousarray(p) for p in cur)
        H, W = y.shape
        mbs = (W // 16) * (H // 16)
        o = {k: np.ascontiguousarray(inter[k]).copy() for k in ("recon_Y", "recon_U", "recon_V", "MB_coeffs", "MB_parts", "MB_segment_id", "MB_SSIM")}
        o["is_inter"] = np.zeros(mbs, np.int32)
        o["modes"] = np.zeros((mbs, 16), np.int32)
        repl, new = ci(0), C.c_float(0)
        if self.prefix == "ref_":
            third = ci(0)
            self._chk(W, H, ssim_target, y, u, v, np.ascontiguousarray(sd, np.int32).reshape(-1), o["recon_Y"], o["recon_U"], o["recon_V"],
                      o["MB_coeffs"], o["MB_parts"], o["MB_segment_id"], o["MB_SSIM"], o["is_inter"], o["modes"], C.byref(repl), C.byref(new),
                      C.cast(C.byref(third), C.c_void_p))
            o["filter_updated"] = int(third.value)
        else:
            third = C.c_float(0)
            self._chk(W, H, ssim_target, y, u, v, np.ascontiguousarray(sd, np.int32).reshape(-1), o["recon_Y"], o["recon_U"], o["recon_V"],
                      o["MB_coeffs"], o["MB_parts"], o["MB_segment_id"], o["MB_SSIM"], o["is_inter"], o["modes"], C.byref(repl), C.byref(new),
                      C.cast(C.byref(third), C.c_void_p))
            o["min_SSIM"] = float(third.value)
            o["filter_updated"] = int(third.value > np.float32(0.95))      # src/vp8enc.cpp:260
        o["replaced"], o["new_SSIM"] = int(repl.value), np.float32(new.value)
        return o

    def count_ssim_16x16(self, a, b):
        """a, b = (Y 16x16, U 8x8, V 8x8) blocks."""
        a = [np.ascontiguousarray(p) for p in a]
        b = [np.ascontiguousarray(p) for p in b]
        return np.float32(self._ssim(a[0], a[1], a[2], 16, b[0], b[1], b[2], 16))

    def pick_luma_predictor(self, orig, top, left, top_left):
        pred, resid = np.zeros(16, np.uint8), np.zeros(16, np.int16)
        m = self._pick(np.ascontiguousarray(orig, np.uint8).reshape(-1), pred, resid, np.ascontiguousarray(top, np.int16),
                       np.ascontiguousarray(left, np.int16), int(top_left))
        return int(m), pred, resid


def oracle_intra() -> Intra:
    return Intra(Oracle.lib(), "vp8o_")


def ref_intra() -> Intra | None:
    """The reference's own host intra code (x86 build) or None when oracle/_ref was not built."""
    if not os.path.exists(REF_HOST_SO):
        try:
            build_oracle()
        except Exception:
            return None
    if not os.path.exists(REF_HOST_SO):
        return None
    return Intra(C.CDLL(REF_HOST_SO), "ref_")
