"""One GOP split BY REFERENCE over the ranks of a process group -- SURVEY.md section 8e(i), north_star's "RCCL broadcast
of reconstructed reference frames".

The reference runs the LAST / GOLDEN / ALTREF searches of a frame on three command queues (inter_part.h:122-135,
201-236): they share nothing but the current frame.  Here each of up to three ranks (one per GPU) searches the
references it owns (reference r belongs to rank r mod world), the quarter-pel vector nets and cost nets -- 8 bytes per
8x8 block and reference, 261 KB at 1080p -- meet in one all_gather (RCCL over xGMI on the GPUs), rank 0 finishes the frame
(select_reference ... SSIM, filter mask, loop filter) and broadcasts the filtered reconstruction (1.5 * W * H bytes,
3.1 MB at 1080p), which becomes every rank's LAST.  The frame-type state machine runs identically on every rank, so the
GOLDEN / ALTREF rotation needs no message.

This is the only data-path collective of the package; GOP-level sharding (gop_shard.py) needs none and scales better
-- a 3 MB exchange per 0.5 ms frame is latency-bound -- so this path is for ONE live stream that must be coded faster
than one GPU codes it.  The loop is InterPathDriver's (driver.py) without check_SSIM.

Ranks that are THREADS of one process (ThreadGroup below: several contexts driven from one host process, on one device or on
several) exchange device buffers: vp8hip_export_search / vp8hip_export_last on the owner, vp8hip_import_search / vp8hip_import_last
on the others -- the import calls land the data where the receiving side of the RCCL exchanges lands it, by the same code
(csrc/api_shard.hip: search_nets, receive_last_surface, adopt_last), so a three-context run on ONE GPU executes what a non-root rank does.

A backend is an encoder context with the C ABI's method names.  The MI355X backend (HipRefBackend over vp8oclenc_amd.api.Vp8Hip)
makes both exchanges INSIDE the library -- vp8hip_shard_share_search / vp8hip_shard_share_last: RCCL broadcasts on the context's
stream, in place in the context's own buffers, no host synchronisation per frame -- and this file keeps only the frame-type state
machine.  A backend without those calls (the CPU oracle, tests/refshard_cpu.py; several contexts as threads of one process) gets the
same exchanges made here over OBJECT collectives (ThreadGroup below; the multi-process CPU tests hand in an adapter of the same shape
over gloo, tests/torch_transport.py -- this package holds no second transport), through four methods of the backend:
    export_search(ref) -> (vectors, costs) of reference ref          import_search(ref, parcel)
    export_last()      -> the filtered reconstruction (Y, U, V)      import_last(parcel)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import api

SHARD_ID_BYTES = 128     # VP8HIP_SHARD_ID_BYTES


def shard_unique_id() -> bytes:
    """vp8hip_shard_unique_id: on ONE rank; hand the bytes to the others (a store, a broadcast of the launcher's)"""
    buf = (C.c_uint8 * SHARD_ID_BYTES)()
    rc = api.load_library().vp8hip_shard_unique_id(buf)
    if rc != 0:
        raise api.Vp8HipError(f"vp8hip_shard_unique_id: {rc}")
    return bytes(buf)


class ThreadGroup:
    """The ranks of a by-reference split as threads of one process: object collectives over a barrier (nothing is copied -- the
    objects are device buffers of this process).  member(r) is what RefShardDriver takes as `dist` on rank r's thread."""

    object_collectives = True

    def __init__(self, world: int):
        import threading
        self.world = world
        self.slots = [None] * world
        self.bar = threading.Barrier(world)

    class _Member:
        object_collectives = True

        def __init__(self, group, rank):
            self.g, self.rank = group, rank

        def is_initialized(self):
            return True

        def get_rank(self):
            return self.rank

        def get_world_size(self):
            return self.g.world

        def barrier(self):
            self.g.bar.wait()

        def all_gather_object(self, obj):
            self.g.slots[self.rank] = obj
            self.g.bar.wait()
            out = list(self.g.slots)
            self.g.bar.wait()          # (nobody overwrites a slot before everybody has read it)
            return out

        def broadcast_object(self, obj, src=0):
            return self.all_gather_object(obj if self.rank == src else None)[src]

    def member(self, rank: int):
        return ThreadGroup._Member(self, rank)

    def abort(self):
        self.bar.abort()


class HipRefBackend(api.Vp8Hip):
    """Vp8Hip + vp8hip_inter_search / vp8hip_inter_finish and the library's own exchanges (vp8hip_shard_*, include/vp8hip.h)."""

    def __init__(self, width: int, height: int, ssim_target: float = -1.0, device: int = 0):
        super().__init__(width, height, ssim_target, device)
        vp = C.c_void_p
        self.lib.vp8hip_inter_search.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        self.lib.vp8hip_inter_finish.argtypes = [vp, C.c_int, C.c_int]
        self.lib.vp8hip_shard_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        self.lib.vp8hip_shard_share_search.argtypes = [vp, C.c_int]
        self.lib.vp8hip_shard_share_last.argtypes = [vp, C.c_int]
        self.lib.vp8hip_shard_max.argtypes = [vp, C.POINTER(C.c_double)]
        self.lib.vp8hip_shard_rank.argtypes = [vp]
        self.lib.vp8hip_shard_world.argtypes = [vp]
        for n in ("vp8hip_export_search", "vp8hip_import_search"):
            getattr(self.lib, n).argtypes = [vp, C.c_int, vp, vp]
        for n in ("vp8hip_export_last", "vp8hip_import_last"):
            getattr(self.lib, n).argtypes = [vp, vp, vp, vp]
        self.native_shard = False
        self.device_ordinal = device
        self._search_parcel = {}
        self._last_parcel = None

    # ---- the exchanges by hand: device buffers out of / into this context (what the RCCL calls below do in place) ----
    def export_search(self, ref: int):
        """reference `ref`'s vector and cost nets as two device buffers of this context's (reused frame after frame); complete on return"""
        if ref not in self._search_parcel:
            self._search_parcel[ref] = (api.DeviceBuffer(self.b8 * 4, self.device_ordinal), api.DeviceBuffer(self.b8 * 4, self.device_ordinal))
        v, c = self._search_parcel[ref]
        self._chk(self.lib.vp8hip_export_search(self.h, ref, v.ptr, c.ptr), "export_search")
        self.synchronize()
        return v, c

    def import_search(self, ref: int, parcel):
        v, c = parcel
        self._chk(self.lib.vp8hip_import_search(self.h, ref, v.ptr, c.ptr), "import_search")
        self.synchronize()       # (the owner reuses the buffers for its next frame)

    def export_last(self):
        """the filtered reconstruction (LAST) as three tight planes in device buffers of this context's; complete on return"""
        if self._last_parcel is None:
            n = self.W * self.H
            self._last_parcel = tuple(api.DeviceBuffer(k, self.device_ordinal) for k in (n, n // 4, n // 4))
        y, u, v = self._last_parcel
        self._chk(self.lib.vp8hip_export_last(self.h, y.ptr, u.ptr, v.ptr), "export_last")
        self.synchronize()
        return y, u, v

    def import_last(self, parcel):
        """vp8hip_import_last: what a receiving rank of vp8hip_shard_share_last does with the planes"""
        y, u, v = parcel
        self._chk(self.lib.vp8hip_import_last(self.h, y.ptr, u.ptr, v.ptr), "import_last")
        self.synchronize()

    def shard_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == SHARD_ID_BYTES
        self._chk(self.lib.vp8hip_shard_init(self.h, unique_id, rank, world), "shard_init")
        self.native_shard = True

    def shard_rank(self) -> int:
        return int(self.lib.vp8hip_shard_rank(self.h))

    def shard_world(self) -> int:
        return int(self.lib.vp8hip_shard_world(self.h))

    def shard_share_search(self, used_mask: int):
        self._chk(self.lib.vp8hip_shard_share_search(self.h, int(used_mask)), "shard_share_search")

    def shard_share_last(self, root: int = 0):
        self._chk(self.lib.vp8hip_shard_share_last(self.h, int(root)), "shard_share_last")

    def shard_max(self, value: float) -> float:
        v = C.c_double(value)
        self._chk(self.lib.vp8hip_shard_max(self.h, C.byref(v)), "shard_max")
        return float(v.value)

    def inter_search(self, prev_is_golden, prev_is_altref, use_golden, use_altref, mask):
        self._chk(self.lib.vp8hip_inter_search(self.h, int(prev_is_golden), int(prev_is_altref), int(use_golden), int(use_altref), int(mask)), "inter_search")

    def inter_finish(self, use_golden, use_altref):
        self._chk(self.lib.vp8hip_inter_finish(self.h, int(use_golden), int(use_altref)), "inter_finish")


class RefShardDriver:
    """The reference's frame loop with a frame's reference searches spread over the ranks of `dist` (None = one process).

    encode_frame returns the frame's outputs on rank 0 ({"key": True} for a key frame) and None on the other ranks."""

    def __init__(self, backend, dist, width: int, height: int, gop_size: int = 150, altref_range: int = 5, qi_min: int = 0,
                 qi_max: int = 48, force_collective: bool = False, download: bool = True, loopback: bool = False, device_segments: bool = False):
        self.be, self.dist = backend, dist
        self.W, self.H = width, height
        self.gop = api.Gop(gop_size, altref_range)
        self.qi_min = min(qi_min, qi_max)
        self.lastqi, self.altrefqi = api.quantizer_ladders(qi_min, qi_max)
        self.native = bool(getattr(backend, "native_shard", False))    # the exchanges are the library's (vp8hip_shard_*)
        if self.native:
            self.collective = True
            self.rank, self.world = backend.shard_rank(), backend.shard_world()
        else:
            self.collective = dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective)
            self.rank = dist.get_rank() if self.collective else 0
            self.world = dist.get_world_size() if self.collective else 1
        self.download = download
        self.loopback = loopback     # also import what this rank itself exported (a one-rank run then walks every exchange)
        self.device_segments = device_segments   # segment data by vp8hip_auto_segments instead of the host scan (timing runs: the host
                                                 # scan of a 1080p frame takes longer than the frame)
        self.bytes_gathered = self.bytes_broadcast = 0

    def owner(self, ref: int) -> int:
        return ref % self.world

    def _set_segments(self, y, is_key: bool, is_altref: bool):
        if self.device_segments:
            self.be.auto_segments(is_key, self.altrefqi if is_altref else self.lastqi, self.qi_min)
            return None
        sd = self._segments(y, is_key, is_altref)
        self.be.set_segments(sd)
        return sd

    def _segments(self, y, is_key: bool, is_altref: bool):
        reductor, sharp = api.loopfilter_strength(y)
        return api.prepare_segments_data(is_key, self.altrefqi if is_altref else self.lastqi, self.qi_min, reductor, sharp)

    def _share_last(self):
        """rank 0's filtered reconstruction becomes every rank's LAST (broadcast; RCCL on the GPUs)"""
        if not self.collective:
            return
        if self.native:
            self.be.shard_share_last(0)
            self.bytes_broadcast += self.W * self.H * 3 // 2
            return
        if getattr(self.dist, "object_collectives", False):     # ranks are threads of this process: device buffers change hands
            parcel = self.dist.broadcast_object(self.be.export_last() if self.rank == 0 else None, src=0)
            self.bytes_broadcast += self.W * self.H * 3 // 2
            if self.rank != 0 or self.loopback:
                self.be.import_last(parcel)
            self.dist.barrier()                                  # (rank 0 writes its parcel again only after everybody has taken it)
            return
        raise TypeError("RefShardDriver: `dist` must offer object collectives (ref_shard.ThreadGroup.member(r), tests/torch_transport.py) or the backend its own exchanges (native_shard)")

    def _share_search(self, used):
        """every used reference's vectors and costs from its owner to every rank (one all_gather)"""
        if not self.collective:
            return
        if self.native:
            self.be.shard_share_search(sum(1 << r for r in used))
            self.bytes_gathered += len(used) * 8 * self.be.b8
            return
        if getattr(self.dist, "object_collectives", False):
            mine = {r: self.be.export_search(r) for r in used if self.owner(r) == self.rank}
            allv = self.dist.all_gather_object(mine)
            self.bytes_gathered += len(used) * 8 * self.be.b8
            for r in used:
                if self.owner(r) != self.rank or self.loopback:
                    self.be.import_search(r, allv[self.owner(r)][r])
            self.dist.barrier()
            return
        raise TypeError("RefShardDriver: `dist` must offer object collectives or the backend its own exchanges (native_shard)")

    def encode_frame(self, y: np.ndarray, u: np.ndarray, v: np.ndarray):
        g = self.gop.next()
        if isinstance(y, np.ndarray):
            self.be.upload_current(y, u, v)
        else:                                   # planes already in this device's memory (api.DeviceBuffer / addresses)
            self.be.set_current_device(y, u, v)
        if g.current_is_key:
            if self.rank == 0:       # key frames are one raster-order wavefront: one device codes them
                self._set_segments(y, True, True)
                self.be.intra_transform()
                self.be.prepare_filter_mask(want_nz=False)
                self.be.loop_filter()
            self.gop.key_coded()
            self._share_last()
            self.gop.frame_done()
            return {"key": True} if self.rank == 0 else None
        sd = self._set_segments(y, False, bool(g.current_is_altref))
        use_golden, use_altref = self.gop.inter_flags()
        used = [0] + ([1] if use_golden else []) + ([2] if use_altref else [])
        mask = sum(1 << r for r in used if self.owner(r) == self.rank)
        self.be.inter_search(g.prev_is_golden, g.prev_is_altref, use_golden, use_altref, mask)
        self._share_search(used)
        out = None
        if self.rank == 0:
            self.be.inter_finish(use_golden, use_altref)
            out = {"key": False, "segments": sd, "use_golden": use_golden, "use_altref": use_altref}
            if self.download:
                out.update(self.be.download_results(recon=True))
            self.be.prepare_filter_mask(want_nz=False)
            self.be.loop_filter()
        self._share_last()
        self.gop.frame_done()
        return out
