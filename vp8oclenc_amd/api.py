"""ctypes binding of libvp8hip.so: the C ABI of include/vp8hip.h and include/vp8hip_host.h.

Python is plumbing here (tests, bench, multi-GPU launch); every computation of the hot path
happens behind the C ABI in hand-written HIP.  There is NO fallback: if the library is missing
or no gfx950 device is usable, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

K_NAMES = ["pack", "downsample", "search1_l4", "search1_l3", "search1_l2", "search1_l1", "search1_l0", "search2",
           "select", "mb", "filter_mask", "loop_filter", "border", "ent_count", "ent_encode", "intra", "hdr_encode"]
K_COUNT = len(K_NAMES)

DBG_NET1, DBG_NET2, DBG_BDIFF, DBG_PYRAMID, DBG_MB_MASK, DBG_MB_NZ, DBG_THIRD_CONTEXT, DBG_CURRENT_CHROMA = range(8)

# every symbol include/vp8hip.h and include/vp8hip_host.h declare
ABI_SYMBOLS = [
    "vp8hip_hw_queues", "vp8hip_profile_read_clock", "vp8hip_inter_search", "vp8hip_inter_finish", "vp8hip_export_search",
    "vp8hip_import_search", "vp8hip_export_last", "vp8hip_import_last", "vp8hip_group_rendezvous", "vp8hip_group_create", "vp8hip_group_destroy", "vp8hip_group_rank", "vp8hip_group_world", "vp8hip_group_count", "vp8hip_group_barrier", "vp8hip_group_max", "vp8hip_group_all_gather", "vp8hip_group_broadcast", "vp8hip_group_gather_bytes", "vp8hip_group_last_hip_error", "vp8hip_batch_create", "vp8hip_batch_destroy", "vp8hip_batch_set_current_device",
    "vp8hip_batch_auto_segments", "vp8hip_batch_inter_transform", "vp8hip_batch_loop_filter", "vp8drv_batch_create", "vp8drv_batch_destroy",
    "vp8drv_batch_encode_frame_device", "vp8hip_batch_encode_frame_begin", "vp8drv_batch_get_frame_begin", "vp8hip_profile_context_switches", "vp8hip_profile_read_search2_clock", "vp8hip_profile_search2_clock",
    "vp8hip_create", "vp8hip_destroy", "vp8hip_upload_current", "vp8hip_set_current_device", "vp8hip_upload_last",
    "vp8hip_set_last_device", "vp8hip_set_segments", "vp8hip_inter_transform", "vp8hip_download_results",
    "vp8hip_upload_mb_data", "vp8hip_upload_recon", "vp8hip_prepare_filter_mask", "vp8hip_loop_filter",
    "vp8hip_download_last", "vp8hip_synchronize", "vp8hip_stream", "vp8hip_last_hip_error", "vp8hip_status_string",
    "vp8hip_profile_enable", "vp8hip_profile_read", "vp8hip_debug_download", "vp8hip_count_probs", "vp8hip_encode_coefficients", "vp8hip_loopfilter_strength", "vp8hip_chroma_change", "vp8hip_chroma_change_async", "vp8hip_chroma_change_result", "vp8hip_auto_segments", "vp8hip_get_segments",
    "vp8hip_intra_transform", "vp8hip_check_ssim", "vp8hip_download_intra", "vp8hip_conformant_stream", "vp8hip_set_source_size", "vp8hip_abi_version", "vp8hip_experiments_compiled_in", "vp8hip_batch_prep_mode", "vp8hip_device_count", "vp8hip_device_alloc", "vp8hip_device_free", "vp8hip_device_upload", "vp8hip_device_download", "vp8hip_device_synchronize", "vp8hip_device_mem_info", "vp8hip_device_pci_bus_id", "vp8hip_runtime_version", "vp8hip_shard_unique_id", "vp8hip_shard_init", "vp8hip_shard_rank", "vp8hip_shard_world", "vp8hip_shard_share_search", "vp8hip_shard_share_last", "vp8hip_shard_max", "vp8hip_encode_header", "vp8hip_encode_frame",
    "vp8hip_encode_frame_begin", "vp8hip_encode_frame_end", "vp8hip_filter_overlap",
    "vp8host_quantizer_ladders", "vp8host_loopfilter_strength", "vp8host_prepare_segments_data", "vp8host_skip_prob",
    "vp8host_gop_init", "vp8host_gop_next", "vp8host_gop_key_coded", "vp8host_gop_inter_flags",
    "vp8host_gop_frame_done", "vp8host_scene_change", "vp8host_y4m_parse_header", "vp8host_y4m_frame_marker_ok",
    "vp8drv_default_config", "vp8drv_create", "vp8drv_destroy", "vp8drv_context", "vp8drv_encode_frame_device",
    "vp8drv_encode_frame_host", "vp8drv_get_stats", "vp8hip_reserve_frame_path", "vp8hip_reserve_frame_path_dense", "vp8drv_resolve", "vp8drv_ready", "vp8drv_batch_ready", "vp8drv_batches_encode_frame_device", "vp8drv_batches_encode_frames_device", "vp8drv_batches_encode_frames_host", "vp8drv_batch_encode_frame_host", "vp8hip_batch_upload_current", "vp8hip_batch_intra_transform", "vp8hip_batch_prefetch_current", "vp8hip_prefetch_current", "vp8drv_prefetch_frame_host", "vp8drv_stage_frame_host", "vp8drv_batch_prefetch_frame_host", "vp8hip_host_alloc", "vp8hip_host_free", "vp8drv_frame_check", "vp8drv_encode_video_device", "vp8hip_check_ssim_ready", "vp8hip_check_ssim_async", "vp8hip_check_ssim_result", "vp8hip_batch_check_ssim_async", "vp8drv_get_frame", "vp8drv_get_frame_begin", "vp8drv_get_frame_end",
    "vp8bs_default_probs", "vp8bs_encode_header", "vp8bs_gather_frame", "vp8bs_ivf_file_header", "vp8bs_ivf_frame_header",
]


class Vp8HipError(RuntimeError):
    pass


ABI_VERSION = 4008  # VP8HIP_ABI_VERSION, include/vp8hip.h
ERR_OVERFLOW = -7   # VP8HIP_ERR_OVERFLOW, include/vp8hip.h
ERR_FORMAT = -8     # VP8HIP_ERR_FORMAT
SHARPNESS_ON_DEVICE = -2 ** 31   # VP8HIP_SHARPNESS_ON_DEVICE


class _Results(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("MB_parts", "MB_reference_frame", "MB_vectors", "MB_coeffs",
                                          "MB_segment_id", "MB_SSIM", "recon_Y", "recon_U", "recon_V")]


class GopState(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "gop_size", "altref_range", "frame_number", "frames_until_key", "frames_until_altref",
        "golden_frame_number", "altref_frame_number", "current_is_key", "current_is_golden", "current_is_altref",
        "prev_is_key", "prev_is_golden", "prev_is_altref")]


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libvp8hip.so (building it first if the sources are newer).  Raises if impossible."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _build.LIB
    if path is None and _build.stale():
        try:
            _build.build()
        except Exception as e:  # no hipcc on this box: use the prebuilt file if there is one
            if not os.path.exists(p):
                raise Vp8HipError(f"libvp8hip.so is missing and cannot be built: {e}") from e
    if not os.path.exists(p):
        raise Vp8HipError(f"{p} not found: build it with `python -m vp8oclenc_amd.build`")
    lib = C.CDLL(p)
    # the struct layouts below (DrvConfig, DrvStats, ...) belong to one ABI: a library built from other sources is refused
    if not hasattr(lib, "vp8hip_abi_version") or lib.vp8hip_abi_version() != ABI_VERSION:
        raise Vp8HipError(f"{p} is not ABI {ABI_VERSION} (include/vp8hip.h): rebuild it with `python -m vp8oclenc_amd.build`")
    u8 = C.c_void_p
    vp = C.c_void_p
    lib.vp8hip_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_float, C.c_int]
    lib.vp8hip_destroy.argtypes = [vp]
    lib.vp8hip_destroy.restype = None
    for n in ("vp8hip_upload_current", "vp8hip_set_current_device", "vp8hip_upload_last", "vp8hip_set_last_device",
              "vp8hip_upload_recon", "vp8hip_download_last"):
        getattr(lib, n).argtypes = [vp, u8, u8, u8]
    lib.vp8hip_set_segments.argtypes = [vp, C.c_void_p]
    lib.vp8hip_inter_transform.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.vp8hip_download_results.argtypes = [vp, C.POINTER(_Results)]
    lib.vp8hip_upload_mb_data.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vp8hip_prepare_filter_mask.argtypes = [vp, C.c_void_p]
    lib.vp8hip_loop_filter.argtypes = [vp]
    lib.vp8hip_synchronize.argtypes = [vp]
    lib.vp8hip_stream.argtypes = [vp]
    lib.vp8hip_stream.restype = C.c_void_p
    lib.vp8hip_last_hip_error.argtypes = [vp]
    lib.vp8hip_status_string.argtypes = [C.c_int]
    lib.vp8hip_status_string.restype = C.c_char_p
    lib.vp8hip_profile_enable.argtypes = [vp, C.c_uint32]
    lib.vp8hip_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.vp8hip_debug_download.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.vp8hip_count_probs.argtypes = [vp, C.c_int, C.c_void_p, C.c_void_p]
    lib.vp8hip_loopfilter_strength.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.vp8hip_chroma_change.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.vp8hip_auto_segments.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.c_int]
    lib.vp8hip_get_segments.argtypes = [vp, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.vp8hip_encode_coefficients.argtypes = [vp, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.vp8hip_intra_transform.argtypes = [vp]
    lib.vp8hip_check_ssim.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.vp8hip_download_intra.argtypes = [vp, C.c_void_p, C.c_void_p]
    i32p = C.POINTER(C.c_int32)
    lib.vp8host_quantizer_ladders.argtypes = [C.c_int, C.c_int, i32p, i32p]
    lib.vp8host_quantizer_ladders.restype = None
    lib.vp8host_loopfilter_strength.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, i32p]
    lib.vp8host_loopfilter_strength.restype = None
    lib.vp8host_prepare_segments_data.argtypes = [C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
    lib.vp8host_prepare_segments_data.restype = None
    lib.vp8host_skip_prob.argtypes = [C.c_void_p, C.c_int]
    for n in ("vp8host_gop_next", "vp8host_gop_key_coded", "vp8host_gop_frame_done"):
        getattr(lib, n).argtypes = [C.POINTER(GopState)]
        getattr(lib, n).restype = None
    lib.vp8host_gop_init.argtypes = [C.POINTER(GopState), C.c_int, C.c_int]
    lib.vp8host_gop_init.restype = None
    lib.vp8host_gop_inter_flags.argtypes = [C.POINTER(GopState), i32p, i32p]
    lib.vp8host_gop_inter_flags.restype = None
    lib.vp8hip_device_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.vp8hip_device_free.argtypes = [C.c_int, C.c_void_p]
    lib.vp8hip_device_upload.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.vp8hip_device_download.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.vp8hip_device_synchronize.argtypes = [C.c_int]
    lib.vp8hip_device_mem_info.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.vp8hip_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    if path is None:
        _lib = lib
    return lib


class DeviceBuffer:
    """Device memory from the library's own runtime (vp8hip_device_alloc): what a host without a GPU framework of its own
    hands to vp8hip_set_current_device & co.  `data_ptr()` like a torch tensor, so call sites read the same."""

    def __init__(self, nbytes: int, device: int = 0):
        self.lib, self.device, self.nbytes = load_library(), device, int(nbytes)
        p = C.c_void_p()
        rc = self.lib.vp8hip_device_alloc(device, self.nbytes, C.byref(p))
        if rc != 0:
            raise Vp8HipError(f"vp8hip_device_alloc({self.nbytes}): {self.lib.vp8hip_status_string(rc).decode()}")
        self.ptr = p.value

    def data_ptr(self) -> int:
        return self.ptr

    def upload(self, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        rc = self.lib.vp8hip_device_upload(self.device, self.ptr + offset, a.ctypes.data, a.nbytes)
        if rc != 0:
            raise Vp8HipError(f"vp8hip_device_upload: {self.lib.vp8hip_status_string(rc).decode()}")
        return self

    def download(self, dtype=np.uint8, shape=None) -> np.ndarray:
        out = np.empty(self.nbytes, np.uint8)
        rc = self.lib.vp8hip_device_download(self.device, out.ctypes.data, self.ptr, self.nbytes)
        if rc != 0:
            raise Vp8HipError(f"vp8hip_device_download: {self.lib.vp8hip_status_string(rc).decode()}")
        out = out.view(dtype)
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.vp8hip_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostBuffer:
    """Page-locked host memory (vp8hip_host_alloc): planes handed to vp8hip_batch_upload_current from it are copied asynchronously."""

    def __init__(self, a: np.ndarray, device: int = 0):
        a = np.ascontiguousarray(a)
        self.lib, self.device, self.nbytes = load_library(), device, int(a.nbytes)
        p = C.c_void_p()
        self.lib.vp8hip_host_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        self.lib.vp8hip_host_free.argtypes = [C.c_int, C.c_void_p]
        rc = self.lib.vp8hip_host_alloc(device, self.nbytes, C.byref(p))
        if rc != 0:
            raise Vp8HipError(f"vp8hip_host_alloc({self.nbytes}): {self.lib.vp8hip_status_string(rc).decode()}")
        self.ptr = p.value
        C.memmove(self.ptr, a.ctypes.data, self.nbytes)

    def data_ptr(self) -> int:
        return self.ptr

    def free(self):
        if getattr(self, "ptr", None):
            self.lib.vp8hip_host_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def to_device(a: np.ndarray, device: int = 0) -> DeviceBuffer:
    """a copy of the array in the device's memory (blocking)"""
    a = np.ascontiguousarray(a)
    return DeviceBuffer(a.nbytes, device).upload(a)


def device_synchronize(device: int = 0) -> None:
    rc = load_library().vp8hip_device_synchronize(device)
    if rc != 0:
        raise Vp8HipError(f"vp8hip_device_synchronize: {load_library().vp8hip_status_string(rc).decode()}")


def device_count() -> int:
    return int(load_library().vp8hip_device_count())


def device_mem_info(device: int = 0):
    f, t = C.c_size_t(), C.c_size_t()
    rc = load_library().vp8hip_device_mem_info(device, C.byref(f), C.byref(t))
    if rc != 0:
        raise Vp8HipError("vp8hip_device_mem_info failed")
    return int(f.value), int(t.value)


def device_pci_bus_id(device: int = 0) -> str:
    buf = C.create_string_buffer(32)
    rc = load_library().vp8hip_device_pci_bus_id(device, buf, 32)
    if rc != 0:
        raise Vp8HipError("vp8hip_device_pci_bus_id failed")
    return buf.value.decode().lower()


SHARD_ID_BYTES = 128     # VP8HIP_SHARD_ID_BYTES


class Group:
    """vp8hip_group_* (include/vp8hip.h): the process group of a GOP-sharded run -- one process per GPU, RCCL inside the library,
    no GPU framework in the host.  Every method is collective and blocks.

    Group.from_env(device): rank / world from RANK / WORLD_SIZE (torchrun's and bench.py's own launcher's variables), the id through a
    file named by `key` (vp8hip_group_rendezvous)."""

    def __init__(self, device: int, unique_id: bytes, rank: int, world: int, key: str | None = None):
        self.lib = load_library()
        L = self.lib
        vp = C.c_void_p
        L.vp8hip_group_create.argtypes = [C.POINTER(vp), C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_char_p]
        L.vp8hip_group_destroy.argtypes = [vp]
        L.vp8hip_group_destroy.restype = None
        for n in ("vp8hip_group_rank", "vp8hip_group_world", "vp8hip_group_count", "vp8hip_group_barrier", "vp8hip_group_last_hip_error"):
            getattr(L, n).argtypes = [vp]
        L.vp8hip_group_max.argtypes = [vp, C.POINTER(C.c_double)]
        L.vp8hip_group_all_gather.argtypes = [vp, C.c_void_p, C.c_size_t, C.c_void_p]
        L.vp8hip_group_broadcast.argtypes = [vp, C.c_int, C.c_void_p, C.c_size_t]
        L.vp8hip_group_gather_bytes.argtypes = [vp, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        assert len(unique_id) == SHARD_ID_BYTES
        h = vp()
        rc = L.vp8hip_group_create(C.byref(h), device, unique_id, rank, world, key.encode() if key else None)
        if rc != 0:
            raise Vp8HipError(f"vp8hip_group_create(rank {rank} of {world}): {L.vp8hip_status_string(rc).decode()}")
        self.h, self.rank, self.world, self.device = h, rank, world, device

    @staticmethod
    def rendezvous(key: str, rank: int, timeout_s: float = 120.0) -> bytes:
        lib = load_library()
        lib.vp8hip_group_rendezvous.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_void_p]
        buf = (C.c_uint8 * SHARD_ID_BYTES)()
        rc = lib.vp8hip_group_rendezvous(key.encode(), rank, float(timeout_s), buf)
        if rc != 0:
            raise Vp8HipError(f"vp8hip_group_rendezvous({key!r}, rank {rank}): {lib.vp8hip_status_string(rc).decode()}")
        return bytes(buf)

    @classmethod
    def from_env(cls, device: int, key: str, timeout_s: float = 120.0) -> "Group":
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        return cls(device, cls.rendezvous(key, rank, timeout_s), rank, world, key)

    def _chk(self, rc, what):
        if rc != 0:
            raise Vp8HipError(f"vp8hip_group_{what}: {self.lib.vp8hip_status_string(rc).decode()} (hip error {self.lib.vp8hip_group_last_hip_error(self.h)})")

    def count(self) -> int:
        """ncclCommCount: the ranks RCCL itself counts"""
        return int(self.lib.vp8hip_group_count(self.h))

    def barrier(self):
        self._chk(self.lib.vp8hip_group_barrier(self.h), "barrier")

    def max(self, value: float) -> float:
        v = C.c_double(value)
        self._chk(self.lib.vp8hip_group_max(self.h, C.byref(v)), "max")
        return float(v.value)

    def all_gather(self, record: np.ndarray) -> np.ndarray:
        """a small fixed-size record (<= 4 KB) from every rank -> [world, ...] on every rank"""
        a = np.ascontiguousarray(record)
        out = np.empty((self.world,) + a.shape, a.dtype)
        self._chk(self.lib.vp8hip_group_all_gather(self.h, a.ctypes.data, a.nbytes, out.ctypes.data), "all_gather")
        return out

    def broadcast_bytes(self, data: bytes | None, nbytes: int, root: int = 0) -> bytes:
        buf = C.create_string_buffer(data if self.rank == root else b"", nbytes) if nbytes else None
        if nbytes:
            self._chk(self.lib.vp8hip_group_broadcast(self.h, root, buf, nbytes), "broadcast")
        return buf.raw[:nbytes] if nbytes else b""

    def gather_bytes(self, data, root: int = 0):
        """every rank's bytes (any length) end to end in rank order on `root`: (buffer, counts) there, (None, counts) elsewhere.
        `data`: bytes-like or a contiguous uint8 array."""
        a = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        counts = self.all_gather(np.array([a.nbytes], np.uint64)).reshape(-1).copy()
        total = int(counts.sum())
        dst = np.empty(max(total, 1), np.uint8) if self.rank == root else None
        self._chk(self.lib.vp8hip_group_gather_bytes(self.h, root, a.ctypes.data if a.nbytes else None, a.nbytes,
                                                     dst.ctypes.data if dst is not None else None, counts.ctypes.data), "gather_bytes")
        return (dst[:total] if dst is not None else None), counts

    def close(self):
        if getattr(self, "h", None):
            self.lib.vp8hip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def frame_check(h: int, frame: bytes) -> int:
    """vp8drv_frame_check (include/vp8hip_driver.h): the fold bench.py holds a run's frames against a second coding with"""
    lib = load_library()
    lib.vp8drv_frame_check.argtypes = [C.c_uint64, C.c_char_p, C.c_size_t]
    lib.vp8drv_frame_check.restype = C.c_uint64
    return int(lib.vp8drv_frame_check(h, frame, len(frame)))


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if isinstance(a, DeviceBuffer):
        return a.ptr
    return int(a)  # raw device/host address (e.g. torch tensor.data_ptr())


# ---- host-side mirror (no GPU needed) -----------------------------------------------------------
def quantizer_ladders(qi_min: int = 0, qi_max: int = 48):
    lib = load_library()
    a, b = (C.c_int32 * 4)(), (C.c_int32 * 4)()
    lib.vp8host_quantizer_ladders(qi_min, qi_max, a, b)
    return list(a), list(b)


def loopfilter_strength(y: np.ndarray):
    lib = load_library()
    r, s = C.c_int32(), C.c_int32()
    y = np.ascontiguousarray(y, np.uint8)
    lib.vp8host_loopfilter_strength(y.ctypes.data, y.shape[1], y.shape[0], C.byref(r), C.byref(s))
    return r.value, s.value


class SceneState(C.Structure):
    """vp8host_scene_state: the hold-over of scene_change() and frames.last_key_detect (vp8enc.cpp:265-311)."""
    _fields_ = [("holdover", C.c_int32), ("last_key_detect", C.c_int32)]


def scene_change(state: SceneState, Udiff: int, Vdiff: int, frame_number: int) -> bool:
    lib = load_library()
    lib.vp8host_scene_change.argtypes = [C.POINTER(SceneState), C.c_int, C.c_int, C.c_int]
    return bool(lib.vp8host_scene_change(C.byref(state), int(Udiff), int(Vdiff), int(frame_number)))


def prepare_segments_data(is_key: bool, refqi, qi_min: int, reductor: int, sharpness: int, update_filter: bool = False,
                          shrpnss: int = 0) -> np.ndarray:
    lib = load_library()
    q = (C.c_int32 * 4)(*[int(v) for v in refqi])
    sd = (C.c_int32 * 44)()
    lib.vp8host_prepare_segments_data(int(is_key), q, qi_min, reductor, sharpness, int(update_filter), shrpnss, sd)
    return np.array(list(sd), np.int32).reshape(4, 11)


def skip_prob(nz: np.ndarray) -> int:
    nz = np.ascontiguousarray(nz, np.int32)
    return load_library().vp8host_skip_prob(nz.ctypes.data, nz.size)


class Gop:
    """Frame-type state machine of the reference main loop (vp8enc.cpp:340-374)."""

    def __init__(self, gop_size: int = 150, altref_range: int = 5):
        self.lib = load_library()
        self.s = GopState()
        self.lib.vp8host_gop_init(C.byref(self.s), gop_size, altref_range)

    def next(self):
        self.lib.vp8host_gop_next(C.byref(self.s))
        return self.s

    def key_coded(self):
        self.lib.vp8host_gop_key_coded(C.byref(self.s))

    def inter_flags(self):
        g, a = C.c_int32(), C.c_int32()
        self.lib.vp8host_gop_inter_flags(C.byref(self.s), C.byref(g), C.byref(a))
        return g.value, a.value

    def frame_done(self):
        self.lib.vp8host_gop_frame_done(C.byref(self.s))


class DrvConfig(C.Structure):
    """vp8drv_config, include/vp8hip_driver.h"""
    _fields_ = [("gop_size", C.c_int32), ("altref_range", C.c_int32), ("qi_min", C.c_int32), ("qi_max", C.c_int32),
                ("ssim_target", C.c_float), ("device_params", C.c_int32), ("check_ssim", C.c_int32),
                ("num_partitions", C.c_int32), ("display_width", C.c_int32), ("display_height", C.c_int32),
                ("host_bitstream", C.c_int32), ("overlap_filter", C.c_int32), ("ref_mask", C.c_int32),
                ("conformant_stream", C.c_int32), ("scene_detect", C.c_int32),
                ("src_width", C.c_int32), ("src_height", C.c_int32)]


class DrvStats(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("frame_number", "inter_frames", "key_frames", "last_use_golden",
                                          "last_use_altref", "last_prev_is_golden", "last_prev_is_altref",
                                          "last_was_altref", "redone_as_key", "last_replaced")] + \
               [("last_new_ssim", C.c_float), ("last_min_ssim", C.c_float), ("scene_changes", C.c_int32), ("refs_searched", C.c_int32)]


class NativeDriver:
    """The reference's frame loop as native host code (vp8_driver.cpp, include/vp8hip_driver.h): one call per
    frame.  `.hip` is a view of its context for downloads and taps."""

    def __init__(self, width: int, height: int, device: int = 0, **cfg):
        self.lib = load_library()
        lib = self.lib
        lib.vp8drv_default_config.argtypes = [C.POINTER(DrvConfig)]
        lib.vp8drv_default_config.restype = None
        lib.vp8drv_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(DrvConfig)]
        lib.vp8drv_destroy.argtypes = [C.c_void_p]
        lib.vp8drv_destroy.restype = None
        lib.vp8drv_context.argtypes = [C.c_void_p]
        lib.vp8drv_context.restype = C.c_void_p
        lib.vp8drv_encode_frame_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.vp8drv_encode_frame_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.vp8drv_get_stats.argtypes = [C.c_void_p, C.POINTER(DrvStats)]
        lib.vp8drv_get_stats.restype = None
        self.cfg = DrvConfig()
        lib.vp8drv_default_config(C.byref(self.cfg))
        for k, v in cfg.items():
            setattr(self.cfg, k, v)
        h = C.c_void_p()
        rc = lib.vp8drv_create(C.byref(h), width, height, device, C.byref(self.cfg))
        if rc != 0:
            raise Vp8HipError(f"vp8drv_create({width}x{height}) failed: {lib.vp8hip_status_string(rc).decode()} ({rc})")
        self.h = h
        self.hip = Vp8Hip.borrowed(lib.vp8drv_context(h), width, height)

    def _ret(self, rc: int) -> bool:
        if rc < 0:
            raise Vp8HipError(f"vp8drv_encode_frame: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")
        return rc == 1

    def encode_frame_device(self, d_y: int, d_u: int, d_v: int, force_key: bool = False) -> bool:
        """True if the frame was coded as a key frame (vp8hip_intra_transform), False for an inter frame."""
        return self._ret(self.lib.vp8drv_encode_frame_device(self.h, d_y, d_u, d_v, int(force_key)))

    def encode_frame_host(self, y, u, v, force_key: bool = False) -> bool:
        y, u, v = (np.ascontiguousarray(p, np.uint8) for p in (y, u, v))
        return self._ret(self.lib.vp8drv_encode_frame_host(self.h, y.ctypes.data, u.ctypes.data, v.ctypes.data, int(force_key)))

    def encode_frame_host_ptr(self, y: int, u: int, v: int, force_key: bool = False) -> bool:
        """encode_frame_host on host ADDRESSES (page-locked buffers: HostBuffer.data_ptr())"""
        self.lib.vp8drv_encode_frame_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        return self._ret(self.lib.vp8drv_encode_frame_host(self.h, y, u, v, int(force_key)))

    def prefetch_frame_host_ptr(self, y: int, u: int, v: int) -> None:
        """the NEXT frame's planes started on their way (vp8drv_prefetch_frame_host); hand the same addresses to the next encode_frame_host_ptr"""
        self.lib.vp8drv_prefetch_frame_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = self.lib.vp8drv_prefetch_frame_host(self.h, y, u, v)
        if rc < 0:
            raise Vp8HipError(f"vp8drv_prefetch_frame_host: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")

    def resolve(self) -> bool:
        """check_SSIM's verdict on the frame just coded (vp8drv_resolve): True if the frame ended as a key frame"""
        self.lib.vp8drv_resolve.argtypes = [C.c_void_p]
        return self._ret(self.lib.vp8drv_resolve(self.h))

    def get_frame(self) -> bytes:
        """The frame just coded as the reference's entropy_encode() + gather_frame() emit it (vp8drv_get_frame)."""
        self.lib.vp8drv_get_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        n = C.c_size_t(0)
        while True:
            buf = self._frame_buffer()
            rc = self.lib.vp8drv_get_frame(self.h, buf.ctypes.data, len(buf), C.byref(n))
            if rc == ERR_OVERFLOW and len(buf) < self._frame_cap_max():   # the frame is still there: a larger buffer, same call
                self._frame_buf = np.zeros(min(2 * len(buf), self._frame_cap_max()), np.uint8)
                continue
            break
        if rc != 0:
            raise Vp8HipError(f"vp8drv_get_frame: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")
        return self._frame_buf[:n.value].tobytes()

    def _frame_cap_max(self) -> int:
        # the most a frame can be: 304 bools per 4x4 block at one byte each is far above it; the device scratch bound
        return self.hip.mbs * 25 * 304 // 4 + (1 << 20)

    def _frame_buffer(self):
        if getattr(self, "_frame_buf", None) is None:
            self._frame_buf = np.zeros(self.hip.mbs * 900 + 65536, np.uint8)
        return self._frame_buf

    def get_frame_begin(self) -> None:
        """Enqueue the entropy stage of the frame just coded and return (vp8drv_get_frame_begin); get_frame_end collects."""
        self.lib.vp8drv_get_frame_begin.argtypes = [C.c_void_p]
        rc = self.lib.vp8drv_get_frame_begin(self.h)
        if rc != 0:
            raise Vp8HipError(f"vp8drv_get_frame_begin: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")

    def get_frame_end(self) -> bytes:
        self.lib.vp8drv_get_frame_end.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        n = C.c_size_t(0)
        while True:
            buf = self._frame_buffer()
            rc = self.lib.vp8drv_get_frame_end(self.h, buf.ctypes.data, len(buf), C.byref(n))
            if rc == ERR_OVERFLOW and len(buf) < self._frame_cap_max():
                self._frame_buf = np.zeros(min(2 * len(buf), self._frame_cap_max()), np.uint8)
                continue
            break
        if rc != 0:
            raise Vp8HipError(f"vp8drv_get_frame_end: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")
        return self._frame_buf[:n.value].tobytes()

    def video_out_buffer(self, nframes: int) -> np.ndarray:
        """host memory for the frames of `nframes` calls' worth of video (twenty times a dense frame at the reference's default
        quantisers), every page touched: what a caller allocates ONCE, before its clock starts, and hands to encode_video_device"""
        out = np.empty(nframes * (self.hip.mbs * 64 + (1 << 16)), np.uint8)
        out.fill(0)
        return out

    def encode_video_device(self, nframes: int, frame_ptrs, start: int = 0, out: np.ndarray | None = None, views: bool = False):
        """vp8drv_encode_video_device: `nframes` frames of one video with the frames out, natively (frame t =
        frame_ptrs[(start + t) % len]); returns (list of frames, key-frame count).  The frames land back to back in `out`
        (video_out_buffer(); allocated here when None); views=True returns them as memoryviews into it instead of copies."""
        nd = len(frame_ptrs)
        F = ((C.c_void_p * 3) * nd)(*[(C.c_void_p * 3)(*p) for p in frame_ptrs])
        if out is None:
            out = np.empty(nframes * (self.hip.mbs * 64 + (1 << 16)), np.uint8)
        cap = int(out.size)
        sizes = (C.c_uint32 * max(nframes, 1))()
        keys = C.c_int(0)
        self.lib.vp8drv_encode_video_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                        C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        rc = self.lib.vp8drv_encode_video_device(self.h, int(nframes), C.cast(F, C.c_void_p), nd, int(start), out.ctypes.data, cap, sizes, C.byref(keys))
        if rc < 0:
            raise Vp8HipError(f"vp8drv_encode_video_device: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")
        frames, o = [], 0
        mv = memoryview(out)
        for t in range(nframes):
            n = int(sizes[t])
            frames.append(mv[o:o + n] if views else mv[o:o + n].tobytes())
            o += n
        return frames, int(keys.value)

    def encode_video_device_no_frames(self, nframes: int, frame_ptrs, start: int = 0):
        """vp8drv_encode_video_device with out = NULL: `nframes` frames of one video, no frames out, no interpreter in the loop"""
        nd = len(frame_ptrs)
        F = ((C.c_void_p * 3) * nd)(*[(C.c_void_p * 3)(*p) for p in frame_ptrs])
        self.lib.vp8drv_encode_video_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                        C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        rc = self.lib.vp8drv_encode_video_device(self.h, int(nframes), C.cast(F, C.c_void_p), nd, int(start), None, 0, None, None)
        if rc < 0:
            raise Vp8HipError(f"vp8drv_encode_video_device: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")

    def stats(self) -> DrvStats:
        s = DrvStats()
        self.lib.vp8drv_get_stats(self.h, C.byref(s))
        return s

    def close(self):
        if getattr(self, "h", None):
            self.hip.close()
            self.lib.vp8drv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeBatch:
    """Up to four NativeDrivers advanced one frame at a time with ONE launch per stage (vp8drv_batch_*, include/vp8hip_driver.h)."""

    def __init__(self, drivers):
        self.lib = load_library()
        self.drivers = list(drivers)
        n = len(self.drivers)
        self.lib.vp8drv_batch_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]
        self.lib.vp8drv_batch_destroy.argtypes = [C.c_void_p]
        self.lib.vp8drv_batch_destroy.restype = None
        self.lib.vp8drv_batch_encode_frame_device.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                              C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        arr = (C.c_void_p * n)(*[d.h for d in self.drivers])
        h = C.c_void_p()
        rc = self.lib.vp8drv_batch_create(C.byref(h), arr, n)
        if rc != 0:
            raise Vp8HipError(f"vp8drv_batch_create: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")
        self.h = h
        self.n = n
        self._ptrs = [(C.c_void_p * n)() for _ in range(3)]
        self._key = (C.c_int * n)()
        self._members = (C.c_int * n)()

    def encode_frame_device(self, planes, members=None, host=False):
        """planes[i] = (d_y, d_u, d_v) device pointers of member i's frame (None for a member that sits this call out, or
        members[i] false); returns the list of "was a key frame" flags.  host=True: the pointers are HOST addresses
        (vp8drv_batch_encode_frame_host; the planes stay unchanged until the next call has returned)."""
        for i, p in enumerate(planes):
            on = p is not None and (members is None or members[i])
            self._members[i] = 1 if on else 0
            if on:
                self._ptrs[0][i], self._ptrs[1][i], self._ptrs[2][i] = p
        fn = self.lib.vp8drv_batch_encode_frame_host if host else self.lib.vp8drv_batch_encode_frame_device
        fn.argtypes = self.lib.vp8drv_batch_encode_frame_device.argtypes
        rc = fn(self.h, self._members, self._ptrs[0], self._ptrs[1], self._ptrs[2], None, self._key)
        if rc < 0:
            raise Vp8HipError(f"vp8drv_batch_encode_frame_{'host' if host else 'device'}: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")
        return [bool(k) for k in self._key]

    def encode_frame_host(self, planes, members=None):
        return self.encode_frame_device(planes, members, host=True)

    def ready(self) -> bool:
        """no member's check_SSIM verdict is still on its way: the next encode_frame_device would not wait (vp8drv_batch_ready)"""
        self.lib.vp8drv_batch_ready.argtypes = [C.c_void_p]
        return bool(self.lib.vp8drv_batch_ready(self.h))

    def get_frames_begin(self, members=None) -> None:
        """The entropy stage of the members' frames in one set of launches (vp8drv_batch_get_frame_begin); every member's
        driver then takes its frame with get_frame_end()."""
        for i in range(self.n):
            self._members[i] = 1 if (members is None or members[i]) else 0
        self.lib.vp8drv_batch_get_frame_begin.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        rc = self.lib.vp8drv_batch_get_frame_begin(self.h, self._members)
        if rc != 0:
            raise Vp8HipError(f"vp8drv_batch_get_frame_begin: {self.lib.vp8hip_status_string(rc).decode()} ({rc})")

    def close(self):
        if getattr(self, "h", None):
            self.lib.vp8drv_batch_destroy(self.h)
            self.h = None

    @staticmethod
    def encode_frames_device_all(batches, nframes, frame_ptrs, starts, frames_out=False, host=False):
        """nframes frames on every batch, one native host thread per batch (vp8drv_batches_encode_frames_device): frame t of member i of
        batch k = frame_ptrs[(starts[k][i] + t) % len(frame_ptrs)], frame_ptrs = [(d_y, d_u, d_v)].  Returns key-frame counts [k][i].
        host=True: the frames are in HOST memory (HostBuffer) and cross the link on their way in (vp8drv_batches_encode_frames_host)."""
        lib = batches[0].lib
        n, nd = len(batches), len(frame_ptrs)
        F = ((C.c_void_p * 3) * nd)(*[(C.c_void_p * 3)(*p) for p in frame_ptrs])
        st = [(C.c_int * b.n)(*[int(x) for x in s]) for b, s in zip(batches, starts)]
        ko = [(C.c_int * b.n)() for b in batches]
        bo = [(C.c_uint64 * b.n)() for b in batches]
        co = [(C.c_uint64 * b.n)() for b in batches]       # vp8drv_frame_check folded over every delivered frame, per member
        IP, UP = C.POINTER(C.c_int), C.POINTER(C.c_uint64)
        fn = lib.vp8drv_batches_encode_frames_host if host else lib.vp8drv_batches_encode_frames_device
        fn.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(IP), C.POINTER(IP), C.POINTER(UP), C.POINTER(UP)]
        rc = fn((C.c_void_p * n)(*[b.h for b in batches]), n, int(nframes), C.cast(F, C.c_void_p), nd,
                                                      (IP * n)(*[C.cast(a, IP) for a in st]), (IP * n)(*[C.cast(a, IP) for a in ko]),
                                                      (UP * n)(*[C.cast(a, UP) for a in bo]) if frames_out else None,
                                                      (UP * n)(*[C.cast(a, UP) for a in co]) if frames_out else None)
        if rc < 0:
            raise Vp8HipError(f"vp8drv_batches_encode_frames_{'host' if host else 'device'}: {lib.vp8hip_status_string(rc).decode()} ({rc})")
        if frames_out == "check":
            return [list(a) for a in ko], [list(a) for a in bo], [list(a) for a in co]
        if frames_out:
            return [list(a) for a in ko], [list(a) for a in bo]
        return [list(a) for a in ko]

    @staticmethod
    def encode_frame_device_all(batches, planes):
        """one frame on every batch, each served as its verdicts come in (vp8drv_batches_encode_frame_device); planes[k][i] = member
        i of batch k's (d_y, d_u, d_v).  Returns the per-batch lists of "was a key frame" flags."""
        lib = batches[0].lib
        n = len(batches)
        for b, pl in zip(batches, planes):
            for i, p in enumerate(pl):
                b._ptrs[0][i], b._ptrs[1][i], b._ptrs[2][i] = p
        PP = C.POINTER(C.c_void_p)
        arr = lambda j: (PP * n)(*[C.cast(b._ptrs[j], PP) for b in batches])
        keys = (C.POINTER(C.c_int) * n)(*[C.cast(b._key, C.POINTER(C.c_int)) for b in batches])
        hs = (C.c_void_p * n)(*[b.h for b in batches])
        lib.vp8drv_batches_encode_frame_device.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(PP), C.POINTER(PP), C.POINTER(PP), C.POINTER(C.POINTER(C.c_int))]
        rc = lib.vp8drv_batches_encode_frame_device(hs, n, arr(0), arr(1), arr(2), keys)
        if rc < 0:
            raise Vp8HipError(f"vp8drv_batches_encode_frame_device: {lib.vp8hip_status_string(rc).decode()} ({rc})")
        return [[bool(k) for k in b._key] for b in batches]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- device path ------------------------------------------------------------------------------
class Vp8Hip:
    """One encoder context on one MI355X (vp8hip_create ... vp8hip_destroy)."""

    def __init__(self, width: int, height: int, ssim_target: float = -1.0, device: int = 0):
        self.lib = load_library()
        self.W, self.H = width, height
        self.mbs = (width // 16) * (height // 16)
        self.b8 = self.mbs * 4
        h = C.c_void_p()
        rc = self.lib.vp8hip_create(C.byref(h), width, height, ssim_target, device)
        if rc != 0:
            raise Vp8HipError(f"vp8hip_create({width}x{height}, device {device}) failed: "
                              f"{self.lib.vp8hip_status_string(rc).decode()} ({rc}); the HIP path has no CPU fallback")
        self.h = h

    def _chk(self, rc: int, what: str):
        if rc != 0:
            raise Vp8HipError(f"{what}: {self.lib.vp8hip_status_string(rc).decode()} ({rc}), hipError "
                              f"{self.lib.vp8hip_last_hip_error(self.h)}")

    @classmethod
    def borrowed(cls, handle, width: int, height: int):
        """View of a context owned by somebody else (the native driver): same methods, no destroy."""
        self = cls.__new__(cls)
        self.lib = load_library()
        self.W, self.H = width, height
        self.mbs = (width // 16) * (height // 16)
        self.b8 = self.mbs * 4
        self.h = C.c_void_p(handle)
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "_borrowed", False):
            self.h = None
            return
        if getattr(self, "h", None):
            self.lib.vp8hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_current(self, y, u, v):
        self._chk(self.lib.vp8hip_upload_current(self.h, _ptr(y), _ptr(u), _ptr(v)), "upload_current")

    def set_current_device(self, y, u, v):
        self._chk(self.lib.vp8hip_set_current_device(self.h, _ptr(y), _ptr(u), _ptr(v)), "set_current_device")

    def upload_last(self, y, u, v):
        self._chk(self.lib.vp8hip_upload_last(self.h, _ptr(y), _ptr(u), _ptr(v)), "upload_last")

    def set_last_device(self, y, u, v):
        self._chk(self.lib.vp8hip_set_last_device(self.h, _ptr(y), _ptr(u), _ptr(v)), "set_last_device")

    def upload_recon(self, y, u, v):
        self._chk(self.lib.vp8hip_upload_recon(self.h, _ptr(y), _ptr(u), _ptr(v)), "upload_recon")

    def set_segments(self, sd):
        sd = np.ascontiguousarray(sd, np.int32).reshape(-1)
        assert sd.size == 44
        self._chk(self.lib.vp8hip_set_segments(self.h, sd.ctypes.data), "set_segments")

    def inter_transform(self, prev_is_golden, prev_is_altref, use_golden, use_altref):
        self._chk(self.lib.vp8hip_inter_transform(self.h, int(prev_is_golden), int(prev_is_altref), int(use_golden),
                                                  int(use_altref)), "inter_transform")

    def download_results(self, recon: bool = True) -> dict:
        W, H, n = self.W, self.H, self.mbs
        out = {
            "MB_parts": np.zeros(n, np.int32), "MB_reference_frame": np.zeros(n, np.int32),
            "MB_vectors": np.zeros((n, 4, 2), np.int16), "MB_coeffs": np.zeros((n, 25, 16), np.int16),
            "MB_segment_id": np.zeros(n, np.int32), "MB_SSIM": np.zeros(n, np.float32),
        }
        if recon:
            out.update(recon_Y=np.zeros((H, W), np.uint8), recon_U=np.zeros((H // 2, W // 2), np.uint8),
                       recon_V=np.zeros((H // 2, W // 2), np.uint8))
        r = _Results(**{k: a.ctypes.data for k, a in out.items()})
        self._chk(self.lib.vp8hip_download_results(self.h, C.byref(r)), "download_results")
        if recon:  # before the loop filter these are the unfiltered planes
            out["prefilter_Y"], out["prefilter_U"], out["prefilter_V"] = out.pop("recon_Y"), out.pop("recon_U"), out.pop("recon_V")
        return out

    def intra_transform(self):
        """intra_transform (intra_part.h:1089-1109): the current frame as a key frame, on the device."""
        self._chk(self.lib.vp8hip_intra_transform(self.h), "intra_transform")

    def set_source_size(self, src_width: int, src_height: int):
        """copy_with_padding on the device: current frames come as tight planes of this size (include/vp8hip.h)"""
        self.lib.vp8hip_set_source_size.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._chk(self.lib.vp8hip_set_source_size(self.h, int(src_width), int(src_height)), "set_source_size")
        self.src = (int(src_width), int(src_height))

    def conformant_stream(self, on: bool = True):
        """NOT the reference: the stream decodes to the encoder's own reconstruction (include/vp8hip.h)"""
        self.lib.vp8hip_conformant_stream.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.lib.vp8hip_conformant_stream(self.h, int(on)), "conformant_stream")

    def check_ssim(self):
        """check_SSIM (vp8enc.cpp:231-263) on the device: (replaced, new_SSIM, min SSIM)."""
        repl, new, mn = C.c_int32(), C.c_float(), C.c_float()
        self._chk(self.lib.vp8hip_check_ssim(self.h, C.byref(repl), C.byref(new), C.byref(mn)), "check_ssim")
        return repl.value, np.float32(new.value), np.float32(mn.value)

    def reserve_frame_path(self):
        """the entropy stage's scratch and the frame buffer now instead of at the first frame (vp8hip_reserve_frame_path)"""
        self.lib.vp8hip_reserve_frame_path.argtypes = [C.c_void_p]
        self._chk(self.lib.vp8hip_reserve_frame_path(self.h), "reserve_frame_path")

    def reserve_frame_path_dense(self):
        """... for the densest frame there can be: a caller that starts frame n + 1 before taking frame n's bytes never loses a recode"""
        self.lib.vp8hip_reserve_frame_path_dense.argtypes = [C.c_void_p]
        self._chk(self.lib.vp8hip_reserve_frame_path_dense(self.h), "reserve_frame_path_dense")

    def check_ssim_async(self, refqi, qi_min: int):
        """check_SSIM enqueued, nobody waiting (vp8hip_check_ssim_async); check_ssim_result() collects the verdict"""
        q = (C.c_int32 * 4)(*[int(x) for x in refqi])
        self.lib.vp8hip_check_ssim_async.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        self._chk(self.lib.vp8hip_check_ssim_async(self.h, q, int(qi_min)), "check_ssim_async")

    def check_ssim_result(self):
        """(replaced, new_SSIM, min SSIM, filter updated) of the last check_ssim_async"""
        repl, new, mn, upd = C.c_int32(), C.c_float(), C.c_float(), C.c_int32()
        self.lib.vp8hip_check_ssim_result.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        self._chk(self.lib.vp8hip_check_ssim_result(self.h, C.byref(repl), C.byref(new), C.byref(mn), C.byref(upd)), "check_ssim_result")
        return repl.value, np.float32(new.value), np.float32(mn.value), bool(upd.value)

    def download_intra(self):
        """(modes[MBs][16], is_inter[MBs]) of the last intra_transform / check_ssim."""
        modes, is_inter = np.zeros((self.mbs, 16), np.int32), np.zeros(self.mbs, np.int32)
        self._chk(self.lib.vp8hip_download_intra(self.h, modes.ctypes.data, is_inter.ctypes.data), "download_intra")
        return modes, is_inter

    def encode_header(self, is_key, is_golden=0, is_altref=0, sharpness=SHARPNESS_ON_DEVICE, partitions_log2=0, use_intra_info=False,
                      loop_filter_type=0, width=0, height=0) -> np.ndarray:
        """encode_header (entropy_host.cpp:709-1256) on the device: the first partition with its frame tag."""
        class P(C.Structure):
            _fields_ = [(n, C.c_int32) for n in ("is_key", "is_golden", "is_altref", "loop_filter_type", "loop_filter_sharpness",
                                                  "partitions_log2", "width", "height", "use_intra_info")]
        p = P(int(is_key), int(is_golden), int(is_altref), loop_filter_type, int(sharpness), partitions_log2, width, height, int(use_intra_info))
        self.lib.vp8hip_encode_header.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        cap = self.mbs * 96 + 16384
        out = np.zeros(cap, np.uint8)
        n = C.c_size_t(0)
        self._chk(self.lib.vp8hip_encode_header(self.h, C.byref(p), out.ctypes.data, cap, C.byref(n)), "encode_header")
        return out[:n.value].copy()

    def _debug_set_ssim(self, ssim):
        s = np.ascontiguousarray(ssim, np.float32)
        assert s.size == self.mbs
        self.lib.vp8hip_debug_upload_ssim.argtypes = [C.c_void_p, C.c_void_p]
        self._chk(self.lib.vp8hip_debug_upload_ssim(self.h, s.ctypes.data), "debug_upload_ssim")

    def upload_mb_data(self, coeffs=None, parts=None, seg=None):
        self._chk(self.lib.vp8hip_upload_mb_data(self.h, _ptr(coeffs), _ptr(parts), _ptr(seg)), "upload_mb_data")

    def prepare_filter_mask(self, want_nz: bool = True):
        nz = np.zeros(self.mbs, np.int32) if want_nz else None
        self._chk(self.lib.vp8hip_prepare_filter_mask(self.h, _ptr(nz)), "prepare_filter_mask")
        return nz

    def loop_filter(self):
        self._chk(self.lib.vp8hip_loop_filter(self.h), "loop_filter")

    def filter_overlap(self, on: bool = True):
        """vp8hip_filter_overlap: one video coded frame after frame -- the loop filter on a stream of its own, the next frame's GOLDEN /
        ALTREF searches beside it, the search's coarse levels as one launch (what vp8drv_config.overlap_filter turns on)"""
        self.lib.vp8hip_filter_overlap.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.lib.vp8hip_filter_overlap(self.h, int(bool(on))), "filter_overlap")

    def loopfilter_strength(self):
        """get_loopfilter_strength (vp8enc.cpp:96-127) of the current frame, computed on the device."""
        red, sh = C.c_int32(), C.c_int32()
        self._chk(self.lib.vp8hip_loopfilter_strength(self.h, C.byref(red), C.byref(sh)), "loopfilter_strength")
        return red.value, sh.value

    def auto_segments(self, is_key: bool, refqi, qi_min: int):
        """get_loopfilter_strength + prepare_segments_data on the device, asynchronous (no host round trip)."""
        q = (C.c_int32 * 4)(*[int(x) for x in refqi])
        self._chk(self.lib.vp8hip_auto_segments(self.h, int(bool(is_key)), q, int(qi_min)), "auto_segments")

    def get_segments(self):
        sd = np.zeros(44, np.int32)
        r, s = C.c_int32(), C.c_int32()
        self._chk(self.lib.vp8hip_get_segments(self.h, sd.ctypes.data, C.byref(r), C.byref(s)), "get_segments")
        return sd.reshape(4, 11), r.value, s.value

    def chroma_change(self):
        """scene_change's Udiff, Vdiff (vp8enc.cpp:265-282) between this and the previous current frame."""
        ud, vd = C.c_int32(), C.c_int32()
        self._chk(self.lib.vp8hip_chroma_change(self.h, C.byref(ud), C.byref(vd)), "chroma_change")
        return ud.value, vd.value

    def chroma_change_async(self):
        """the same scan enqueued behind the current frame's pack, nobody waiting (vp8hip_chroma_change_async); chroma_change_result() collects it"""
        self._chk(self.lib.vp8hip_chroma_change_async(self.h), "chroma_change_async")

    def chroma_change_result(self):
        ud, vd = C.c_int32(), C.c_int32()
        self._chk(self.lib.vp8hip_chroma_change_result(self.h, C.byref(ud), C.byref(vd)), "chroma_change_result")
        return ud.value, vd.value

    def count_probs(self, num_partitions: int):
        """count_probs + num_div_denom (CPU_kernels.cl:536-778): (probs[1056], partition-0 denominators[1056])."""
        probs, denom = np.zeros(1056, np.uint32), np.zeros(1056, np.uint32)
        self._chk(self.lib.vp8hip_count_probs(self.h, num_partitions, probs.ctypes.data, denom.ctypes.data), "count_probs")
        return probs, denom

    def encode_coefficients(self, probs, num_partitions: int, partition_step: int = 0):
        """encode_coefficients (CPU_kernels.cl:347-414) on the device: list of partition byte strings."""
        probs = np.ascontiguousarray(probs, np.uint32)
        step = partition_step or (self.mbs * 25 * 16 * 3 // num_partitions + 4096)
        out = np.zeros(num_partitions * step, np.uint8)
        sizes = np.zeros(num_partitions, np.int32)
        self._chk(self.lib.vp8hip_encode_coefficients(self.h, probs.ctypes.data, num_partitions, step, out.ctypes.data,
                                                      sizes.ctypes.data), "encode_coefficients")
        return [out[p * step: p * step + sizes[p]].copy() for p in range(num_partitions)]

    def download_last(self):
        W, H = self.W, self.H
        y, u, v = np.zeros((H, W), np.uint8), np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)
        self._chk(self.lib.vp8hip_download_last(self.h, y.ctypes.data, u.ctypes.data, v.ctypes.data), "download_last")
        return y, u, v

    def synchronize(self):
        self._chk(self.lib.vp8hip_synchronize(self.h), "synchronize")

    @property
    def stream(self) -> int:
        return int(self.lib.vp8hip_stream(self.h) or 0)

    def profile_enable(self, kernels):
        mask = 0
        for k in kernels:
            mask |= 1 << (K_NAMES.index(k) if isinstance(k, str) else int(k))
        self._chk(self.lib.vp8hip_profile_enable(self.h, mask), "profile_enable")

    def profile_read(self) -> dict:
        ms = (C.c_double * K_COUNT)()
        n = (C.c_int64 * K_COUNT)()
        self._chk(self.lib.vp8hip_profile_read(self.h, ms, n), "profile_read")
        return {K_NAMES[i]: (ms[i], n[i]) for i in range(K_COUNT) if n[i]}

    def profile_read_clock(self):
        """(ms, launches, shader clock GHz) of the loop filter by the kernel's own clock since the last call (vp8hip_profile_read_clock)"""
        ms, n, ghz = C.c_double(0), C.c_int64(0), C.c_double(0)
        self.lib.vp8hip_profile_read_clock.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        self._chk(self.lib.vp8hip_profile_read_clock(self.h, C.byref(ms), C.byref(n), C.byref(ghz)), "profile_read_clock")
        return ms.value, n.value, ghz.value

    def profile_read_search2_clock(self):
        """(ms, launches) of k_search2 by the kernel's own clock since the last call (vp8hip_profile_read_search2_clock)"""
        ms, n = C.c_double(0), C.c_int64(0)
        self.lib.vp8hip_profile_read_search2_clock.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        self._chk(self.lib.vp8hip_profile_read_search2_clock(self.h, C.byref(ms), C.byref(n)), "profile_read_search2_clock")
        return ms.value, n.value

    def profile_search2_clock(self, on: bool) -> None:
        self.lib.vp8hip_profile_search2_clock.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.lib.vp8hip_profile_search2_clock(self.h, int(on)), "profile_search2_clock")

    def profile_context_switches(self) -> int:
        """among the launches of the last profile_read_clock(): how many had the loop filter's last wave change its hardware slot
        (vp8hip_profile_context_switches; 0 unless the process holds more queues than the part's scheduler keeps resident)"""
        self.lib.vp8hip_profile_context_switches.argtypes = [C.c_void_p]
        self.lib.vp8hip_profile_context_switches.restype = C.c_int64
        return int(self.lib.vp8hip_profile_context_switches(self.h))

    def debug(self, what: int, ref: int = 0, level: int = 0) -> np.ndarray:
        if what in (DBG_NET1, DBG_NET2):
            a = np.zeros((self.b8, 2), np.int16)
        elif what == DBG_BDIFF:
            a = np.zeros(self.b8, np.int32)
        elif what == DBG_PYRAMID:
            a = np.zeros((self.H >> level, self.W >> level), np.uint8)
        elif what == DBG_THIRD_CONTEXT:
            a = np.zeros((self.mbs, 25), np.uint8)
        elif what == DBG_CURRENT_CHROMA:
            a = np.zeros((self.H // 2, self.W // 2), np.uint8)
        else:
            a = np.zeros(self.mbs, np.int32)
        self._chk(self.lib.vp8hip_debug_download(self.h, what, ref, level, a.ctypes.data, a.nbytes), "debug_download")
        return a
